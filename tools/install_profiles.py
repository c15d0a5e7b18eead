"""tools/install_profiles.py <tag> : copy the summaries tools/refresh_profiles.sh left under gpurun_out/profiles_<tag>/
into profiles/ and rebuild profiles/traffic.json -- the PMC-measured HBM bytes per launch of every bench leg's dominant
kernel, keyed the way bench.py looks them up, each entry naming the profile file and the commit it was taken at."""
import csv, glob, json, os, shutil, subprocess, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"profiles_{tag}"), os.path.join(root, "profiles")
for f in os.listdir(src):
    if f.startswith(tag) and (f.endswith(".json") or f.endswith("kernel_stats.csv")):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "recsys_pytorch_amd"], capture_output=True, text=True).stdout.strip())
note = ("rocprofv3 --pmc passes (tools/refresh_profiles.sh, one pass per counter group); read side = TCC_EA0_RDREQ x 128 B (every "
        "request of these kernels is 128-byte; equals 2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md), write side = "
        "WRITE_SIZE KB (= TCC_EA0_WRREQ_64B x 64 B + atomics); mean per launch")
tpath = os.path.join(dst, "traffic.json")
t = json.load(open(tpath)) if os.path.exists(tpath) and "--fresh" not in sys.argv else {}
for meta_path in sorted(glob.glob(os.path.join(dst, f"{tag}_pmc_*.meta.json"))):
    meta = json.load(open(meta_path))
    prof = os.path.basename(meta_path).replace(".meta.json", ".json")
    try:
        d = json.load(open(os.path.join(dst, prof)))
    except OSError:
        print("no counters file for", prof); continue
    want = meta["kernel"].replace(" ", "")
    ks = [k for k in d if k.replace(" ", "") == want] or [k for k in d if k.replace(" ", "").startswith(want)]
    if not ks or "TCC_EA0_RDREQ_sum" not in d[ks[0]] or "WRITE_SIZE" not in d[ks[0]]:
        print("no counters for", prof, meta["kernel"], list(d)); continue
    v = d[ks[0]]
    if meta["key"].startswith("lightgcn"):
        # one propagation product is several kernels since round 6 (the plan's segments, the longest rows by scatter, their two edge
        # kernels, the clearing of split rows): the product's traffic is the SUM of their per-launch traffic (each runs once per product)
        v = dict(v)
        for kname, kv in d.items():
            if kname != ks[0] and "TCC_EA0_RDREQ_sum" in kv and "WRITE_SIZE" in kv:
                for c_ in ("TCC_EA0_RDREQ_sum", "WRITE_SIZE", "FETCH_SIZE", "TCC_EA0_ATOMIC_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"):
                    if c_ in kv:
                        v[c_] = v.get(c_, 0.0) + kv[c_]
    # a chunked step launches the kernel once per item range: the leg's figure is the STEP's (sum over its launches)
    per_step = int(meta.get("env", {}).get("CHUNKS", 1)) if "_c" in meta["key"].rsplit("_nb", 1)[-1] else 1
    v = {k_: (x * per_step if isinstance(x, (int, float)) and k_ not in ("launches",) else x) for k_, x in v.items()}
    rd = v["TCC_EA0_RDREQ_sum"] * 128.0
    # everything a STEP moves at the fabric side, not only its dominant kernel: the apply and the sampler's kernels too (each kernel's
    # bytes per launch x its launches per step; the loop samples two steps ahead, so the sampler has a couple of launches more)
    steps = int(meta.get("argv", [0, 0, 12])[2]) if len(meta.get("argv", [])) > 2 else 12
    step_total = 0.0
    for kname, kv in d.items():
        if "TCC_EA0_RDREQ_sum" in kv and "WRITE_SIZE" in kv and steps > 0:
            step_total += (kv["TCC_EA0_RDREQ_sum"] * 128.0 + kv["WRITE_SIZE"] * 1024.0) * max(1, round(kv.get("launches", steps) / steps))
    t[meta["key"]] = {"step_total_bytes": step_total if not meta["key"].startswith("lightgcn") else None,
                      "hbm_bytes_per_launch": rd + v["WRITE_SIZE"] * 1024.0, "read_bytes": rd, "write_bytes": v["WRITE_SIZE"] * 1024.0,
                      "FETCH_SIZE_KB": v.get("FETCH_SIZE"), "WRITE_SIZE_KB": v["WRITE_SIZE"], "ea_atomic_requests": v.get("TCC_EA0_ATOMIC_sum"),
                      "tcc_hit": v.get("TCC_HIT_sum"), "tcc_miss": v.get("TCC_MISS_sum"), "tcc_req": v.get("TCC_REQ_sum"),
                      "kernel": ks[0], "launches_per_step": per_step, "profile": prof, "commit": commit + ("+uncommitted" if dirty else ""),
                      "kernel_us_under_profiler": v.get("mean_us"), "command": meta, "note": note,
                      "sources_sha": meta.get("sources_sha")}      # hash of the kernel sources of the run that was profiled (bench.py: stale)
    print("%-16s %-46s %.3f GB/launch" % (prof.replace(f"{tag}_pmc_", "").replace(".json", ""), meta["key"], t[meta["key"]]["hbm_bytes_per_launch"] / 1e9))
json.dump(t, open(tpath, "w"), indent=1)
try:
    b = json.load(open(os.path.join(dst, f"{tag}_bench_default.json")))
    print("bench", b["value"], b["ms_per_step"], "kernel_ms", b["roofline"]["kernel_ms"])
    for r in list(csv.DictReader(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))))[:16]:
        print(r["Name"].replace("(anonymous namespace)::", "")[:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
except (OSError, ValueError, KeyError) as e:
    print("(no bench summary in this refresh:", e, ")")

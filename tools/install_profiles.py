"""tools/install_profiles.py <tag> : copy the summaries tools/refresh_profiles.sh left under gpurun_out/profiles_<tag>/
into profiles/ and rebuild profiles/traffic.json (PMC-measured HBM bytes per launch of each leg's step kernel)."""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"profiles_{tag}"), os.path.join(root, "profiles")
for f in os.listdir(src):
    if f.startswith(tag) and (f.endswith(".json") or f.endswith("kernel_stats.csv")):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
b = json.load(open(os.path.join(dst, f"{tag}_bench_default.json")))
note = ("rocprofv3 --pmc passes over tools/step_prof.py (tools/refresh_profiles.sh); read side = TCC_EA0_RDREQ x 128 B (all requests of these "
        "kernels are 128-byte; equals 2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md), write side = WRITE_SIZE "
        "(= TCC_EA0_WRREQ_64B x 64 B + atomics)")
t = {}
legs = {"B1M_blocked": (b["roofline"]["traffic_key"], "bpr_step_blocked_kernel<128, 3, unsigned int, true>"),
        "B65536_plain": (b["legs"]["base_batch_65536"]["roofline"]["traffic_key"], "bpr_step_kernel<128, 0, 3, unsigned int>"),
        "B1M_iid": (b["legs"]["independent_uniform_negatives"]["roofline"]["traffic_key"], "bpr_step_blocked_kernel<128, 3, unsigned int, false>")}
if "--pmc-only" in sys.argv:
    pass
for leg, (key, kname) in legs.items():
    d = json.load(open(os.path.join(dst, f"{tag}_pmc_step_{leg}.json")))
    ks = [k for k in d if k.replace(" ", "") == kname.replace(" ", "")]
    if not ks:
        print("no counters for", leg, list(d)); continue
    v = d[ks[0]]
    rd = v["TCC_EA0_RDREQ_sum"] * 128.0
    t[key] = {"hbm_bytes_per_launch": rd + v["WRITE_SIZE"] * 1024.0, "read_bytes": rd, "write_bytes": v["WRITE_SIZE"] * 1024.0,
              "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"], "ea_atomic_requests": v.get("TCC_EA0_ATOMIC_sum"),
              "kernel": kname, "profile": f"{tag}_pmc_step_{leg}.json", "kernel_us_under_profiler": v.get("mean_us"), "note": note}
    print(leg, key, "%.2f GB" % (t[key]["hbm_bytes_per_launch"] / 1e9))
json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("bench", b["value"], b["ms_per_step"], "kernel_ms", b["roofline"]["kernel_ms"])
for r in list(csv.DictReader(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))))[:16]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:80], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))

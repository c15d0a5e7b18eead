"""tools/install_profiles.py <tag> : copy the summaries tools/refresh_profiles.sh left under
gpurun_out/profiles_<tag>/ into profiles/ (condensed PMC file, traffic.json entry for the bench workload)"""
import json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"profiles_{tag}"), os.path.join(root, "profiles")
shutil.copy(os.path.join(src, f"{tag}_bench_default.json"), os.path.join(dst, f"{tag}_bench_default.json"))
shutil.copy(os.path.join(src, f"{tag}_bench_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
keep = ("bpr_step", "apply_item", "bucket_", "score_tile", "merge_cand", "topk_rows", "take_tau", "mask_seen",
        "fold_hot", "item_mass", "item_cdf", "build_signature", "bpr_sample")
out = {}
for part in ("fetch", "write", "mfma"):
    d = json.load(open(os.path.join(src, f"{tag}_{part}_counters.json")))
    out[part] = {k: v for k, v in d.items() if any(s in k for s in keep)}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w"), indent=1, sort_keys=True)
k = [k for k in out["fetch"] if "bpr_step_blocked" in k][0]
F, W = out["fetch"][k]["FETCH_SIZE"], out["write"][k]["WRITE_SIZE"]
tpath = os.path.join(dst, "traffic.json")
t = json.load(open(tpath))
key = "U1000000_I100000_d128_B1000000_zipf_nb8"
t[key].update({"hbm_bytes_per_launch": 2 * F * 1024 + W * 1024, "FETCH_SIZE_KB": F, "WRITE_SIZE_KB": W})
json.dump(t, open(tpath, "w"), indent=1)
b = json.load(open(os.path.join(dst, f"{tag}_bench_default.json")))
print("bench", b["value"], b["ms_per_step"], "kernel_ms", b["roofline"]["kernel_ms"], "traffic GB", t[key]["hbm_bytes_per_launch"] / 1e9)
print("scoring", b["scoring"]["value"], "small", b["small_batch"]["value"], "cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["sample"])
import csv
for r in list(csv.DictReader(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))))[:14]:
    print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))

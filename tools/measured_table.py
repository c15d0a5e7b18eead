"""tools/measured_table.py [tag] : print the 'Measured' rows of DESIGN.md section 7 from profiles/<tag>_bench_default.json"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
b = json.load(open(os.path.join(root, "profiles", f"{tag}_bench_default.json")))
r = b["roofline"]
f = lambda x, n=2: "n/a" if x is None else f"{x:.{n}f}"
print(f"headline: {b['value']:.3e} triplets/s, {b['ms_per_step']*1e3:.0f} us/step, kernel {r['kernel_ms']*1e3:.0f} us x {r['kernel_launches_timed']}; "
      f"traffic {r['traffic']/1e9 if r['traffic'] else 0:.3f} GB -> achieved {f(r['achieved'],0)} GB/s, frac {f(r['frac'],3)}; frac_compulsory {f(r['frac_compulsory'],3)}; "
      f"traffic/compulsory {f(r['traffic_over_compulsory'],2)}; algorithmic_rate_over_peak {f(r['algorithmic_rate_over_peak'],2)}; frac_end_to_end {f(b.get('frac_end_to_end'),3)}")
for k, v in b.get("legs", {}).items():
    if isinstance(v, dict) and "roofline" in v:
        q = v["roofline"]
        print(f"{k}: {v.get('value', 0):.3e} {v.get('unit','')}, {(v.get('ms_per_step') or v.get('train_step_ms') or 0)*1e3:.0f} us/step, kernel {q['kernel_ms']*1e3:.1f} us, "
              f"traffic {q['traffic']/1e9 if q.get('traffic') else 0:.3f} GB, frac {f(q.get('frac'),3)}, algorithmic/peak {f(q.get('algorithmic_rate_over_peak'),2)}, "
              f"frac_end_to_end {f(v.get('frac_end_to_end'),3)}")
    elif isinstance(v, list):
        for e in v:
            print(f"sweep B={e['batch_per_gpu']}: {e['value']:.3e}, {e['ms_per_step']*1e3:.0f} us/step, kernel {e['kernel_ms']*1e3:.1f} us, frac {f(e.get('frac'),3)}, algorithmic/peak {f(e.get('algorithmic_rate_over_peak'),2)}")
s = b.get("scoring")
if s:
    print(f"scoring: {s['value']:.3e} scores/s, frac {s['roofline']['frac']:.3f} ({s['roofline']['achieved']:.1f} TF)")
c = b.get("cpu_baseline")
if c:
    print(f"cpu: {c['cpu_model']}, {c['cores']} threads of {c.get('host_logical_cpus')}: " + ", ".join(f"{k} {v['value']:.3e}" for k, v in c["legs"].items()))

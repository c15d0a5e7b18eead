"""tools/pmc_groups.py <out.json> <kernel substrings, comma separated> -- <program and its args>
One rocprofv3 pass per counter group (counters only, with --kernel-trace; never combined with other trace domains),
mean per dispatch of every counter (summed over its dimensions) for the kernels whose name contains one of the
substrings.  Groups respect the per-block slot limits of MI355X_MICROARCH.md (TCC 4; FETCH_SIZE / WRITE_SIZE alone)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tools"))
out_path, pats = os.path.abspath(sys.argv[1]), sys.argv[2].split(",")
prog = [os.path.join(root, a) if os.path.exists(os.path.join(root, a)) and not os.path.isabs(a) and "/" in a else a
        for a in sys.argv[sys.argv.index("--") + 1:]]
GROUPS = os.environ.get("PMC_GROUPS")
groups = [g.split() for g in GROUPS.split(";")] if GROUPS else [
    ["FETCH_SIZE"], ["WRITE_SIZE"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"],
    ["TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_DRAM_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_ATOMIC_sum"],
    ["TCC_EA0_ATOMIC_sum", "TCC_READ_sum", "TCC_WRITE_sum", "TCC_TAG_STALL_sum"],
    ["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TOTAL_ACCESSES_sum"],
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS",
     "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"],
]
env = dict(os.environ, TMPDIR="/tmp")
res = {}
for gi, grp in enumerate(groups):
    d = f"/tmp/pmcg_{os.getpid()}_{gi}"
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *grp, "--output-format", "csv", "-d", d, "--", *prog]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=600)
    subprocess.run([sys.executable, os.path.join(root, "tools", "prof_summarize.py"), d, d + "_o"], capture_output=True)
    try:
        c = json.load(open(d + "_o_counters.json"))
    except Exception as e:      # noqa: BLE001
        print("group failed:", grp, r.stderr[-300:])
        continue
    dur = json.load(open(d + "_o_durations.json")) if os.path.exists(d + "_o_durations.json") else {}
    for k, v in c.items():
        if any(p in k for p in pats):
            name = k.replace("(anonymous namespace)::", "").replace("void ", "")
            key = name.split("(")[0].strip()              # kernel name with its template arguments, without the parameter list
            res.setdefault(key, {}).update({a: round(b, 1) for a, b in v.items()})
            if k in dur:
                res[key].setdefault("mean_us", []).append(round(dur[k]["mean_ns"] / 1e3, 1))
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
for k, v in res.items():
    print(k, json.dumps(v))

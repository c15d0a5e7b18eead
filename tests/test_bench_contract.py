"""bench.py prints ONE JSON line with the driver's contract fields (+ roofline, legs, cpu_baseline, scoring)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(cmd, env=None):
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT, env={**os.environ, **(env or {})})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_json_contract_small_shape():
    d = run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--users", "80000",
                   "--items", "30000", "--batch", "80000", "--score-tiles", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "scoring", "legs"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "bpr_step_blocked_kernel"
    # `achieved` / `frac` are PHYSICAL (PMC traffic of this leg / kernel time): this small shape was never profiled, so
    # they are null with a reason; the full-size legs quote profiles/traffic.json (test_roofline_is_physical below)
    assert r["frac"] is None and r["achieved"] is None and r["traffic"] is None and "frac_null_reason" in r
    assert r["kernel_ms"] > 0 and r["kernel_launches_timed"] == 3          # every step kernel of the timed region
    assert 0 < r["frac_compulsory"] <= 1.0 and r["compulsory_bytes"] < r["algorithmic_bytes_per_launch"]
    assert r["algorithmic_rate_over_peak"] == r["algorithmic_GBs"] / r["peak"]
    # SURVEY 8d's formula is flagged where it is not a fraction (item sums on chip), with the reason
    assert r["algorithmic_valid"] == (r["algorithmic_rate_over_peak"] <= 1.0) and (r["algorithmic_valid"] or "algorithmic_invalid_reason" in r)
    # the independent-negatives leg (the one where 24 d bytes per triplet ARE moved) as flat scalars of the object the driver keeps
    for k in ("iid_value", "iid_ms_per_step", "iid_kernel_ms", "iid_frac", "iid_algorithmic_frac", "iid_algorithmic_frac_e2e"):
        assert k in r and not isinstance(r[k], (dict, list)), k
    assert 0 < r["iid_algorithmic_frac"] and r["iid_value"] > 0
    assert d["config"]["sampler"].endswith(d["legs"]["independent_uniform_negatives"]["sampler"]) or "CSC walk" in d["config"]["sampler"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    legs = d["legs"]
    assert 2 <= legs["base_batch_65536"]["neg_block"] <= 8 and legs["base_batch_65536"]["roofline"]["kernel_ms"] > 0     # 65536 >= 2 * 30000
    assert legs["independent_uniform_negatives"]["neg_block"] == 0
    assert legs["independent_uniform_negatives"]["roofline"]["kernel"] == "bpr_step_blocked_kernel<TILE=false>"   # 80000 >= 2 * 30000: ordered batch
    assert [s["batch_per_gpu"] for s in legs["batch_sweep"]] == [256, 4096, 16384] and legs["uniform_item_popularity"]["value"] > 0
    assert legs["config1_d64"]["value"] > 0 and "config3_slice_1.25Mx1M" not in legs       # only beside the default shape
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["cpu_model"]
    assert {"sgd_B80000", "sgd_B65536", "adam_B80000", "score_tile_1024xI", "top50_numpy_argpartition",
            "top50_cxx_partial_sort_1_thread"} <= set(c["legs"])
    s = d["scoring"]["roofline"]
    assert s["bound"] == "mfma" and s["peak"] == 157.3
    # the headline is the MEDIAN of three back-to-back timed regions
    tr = d["timed_regions"]
    assert tr["count"] == 3 and tr["min"] <= d["ms_per_step"] <= tr["max"] and sorted(tr["ms_per_step_each"])[1] == d["ms_per_step"]
    # every config's figure inside the object the driver keeps (`roofline`), compact
    cf = r["configs"]
    assert {"base_batch_65536", "config1_d64", "scoring"} <= set(cf) and len(json.dumps(cf)) <= 700
    for v in cf.values():
        assert set(v) == {"value", "ms_per_step", "kernel_ms", "frac", "frac_end_to_end"} and v["value"] > 0
    # ... and flat: the driver's record keeps scalars of `roofline` / `config` only
    for k in ("base65536_value", "base65536_frac", "base65536_frac_e2e", "d64_value", "d64_frac", "d64_frac_e2e", "scoring_value",
              "scoring_frac", "traffic_stale", "traffic_commit", "frac_e2e", "hbm_utilisation_e2e", "copy_rate_GBs", "achieved_over_copy_rate",
              "hbm_utilisation_e2e_over_copy_rate"):
        assert k in r and not isinstance(r[k], (dict, list)), k
        assert d["config"]["roofline_" + k] == r[k]
    assert r["base65536_value"] == cf["base_batch_65536"]["value"] and r["scoring_frac"] == cf["scoring"]["frac"]
    # the guide's measured float4-copy rate is a second reference point; the roofline's `peak` stays the 8 TB/s spec
    assert r["copy_rate_GBs"] == 6290.0 and r["achieved_over_copy_rate"] is None      # (no PMC profile of this small shape)


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("two_pass,exchange,chunks", [("1", "allreduce", None), ("0", "allreduce", "3"), ("1", "direct", None), ("1", "scatter_gather", "0")]
                         # (every run is three fresh processes: the committed suite keeps bench.py's own N > 1 default, the finer pipeline, the
                         #  library's mesh and the two-pass scatter-gather; RSX_FULL_SUITE=1 adds the other schedules)
                         + ([("0", "allreduce", "0"), ("1", "allreduce", "0"), ("0", "direct", "0")] if os.environ.get("RSX_FULL_SUITE") == "1" else []))
def test_bench_two_ranks_on_one_gpu(two_pass, exchange, chunks):
    """the N > 1 code path of bench.py (torch.distributed.run, native loop with the exchange callbacks) with two
    ranks sharing the box's GPU and gloo moving G: not a performance number, a does-it-run-and-agree check.  chunks = None:
    bench.py's own N > 1 DEFAULT -- two item ranges, the exchange range by range (here through the exchange_range callback; on
    a multi-GPU node through the library's RCCL)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() + 31 + int(two_pass) + 7 * len(exchange) + 3 * int(chunks or 2)) % 2000), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "4", "--warmup", "1", "--users", "120000", "--batch", "120000", "--items", "30000",
           "--score-tiles", "1" if chunks is None else "0", "--no-legs", *(["--chunks", chunks] if chunks is not None else [])]
    d = run_bench(cmd, env={"RSX_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "RSX_TWO_PASS": two_pass,
                                "RSX_EXCHANGE": exchange, "RSX_CHUNKS": "-1"})
    assert ("reduce-scatter" in d["config"]["parallelism"]) == (exchange in ("scatter_gather", "direct"))
    assert d["config"]["exchange"] == exchange and ("rsx_mesh" in d["config"]["exchange_issued_by"]) == (exchange == "direct")
    if exchange == "direct":       # every step's exchange (x item ranges) went through the library's own mesh: warm-up + 3 timed regions
        # (+ 3: bench.py's mesh_selfcheck -- three whole-table exchanges checked against a torch.distributed all-reduce before timing)
        assert d["config"]["mesh_exchanges"] == (1 + 3 * 4) * max(1, int(chunks) if chunks is not None else 2) + 3
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 240000 and d["config"]["item_replicas_identical"] is True
    want_chunks = int(chunks) if chunks is not None else 2
    assert d["config"]["item_chunks"] == want_chunks
    assert ("negatives_with_item_ranges" in d["config"]) == (want_chunks > 1)
    assert ("range by range" in d["config"]["exchange_issued_by"]) == (want_chunks > 1)
    assert ("two-pass" in d["config"]["parallelism"]) == (two_pass == "1" and want_chunks == 0 and exchange != "direct")
    assert d["roofline"]["kernel_launches_timed"] == 4
    assert abs(d["value"] - 240000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    if chunks is None:           # scoring: every rank its own users, the job's rate = all ranks' scores / the slowest rank's time
        sc = d["scoring"]
        assert sc["n_gpus"] == 2 and sc["value"] > 0 and sc["roofline"]["peak"] == 2 * 157.3 and 0 < sc["roofline"]["frac"] < 1


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_plain_command_starts_its_own_ranks():
    """the driver's command for the scaling run is the PLAIN `python bench.py --gpus N ...` (no torch.distributed.run around
    it): bench.py starts its N ranks itself before it touches the GPU and relays rank 0's one JSON line.  Two ranks on the
    box's one GPU, gloo moving G: a does-it-run-and-agree check of bench.py's own N > 1 default (two item ranges)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
                          "--users", "120000", "--batch", "120000", "--items", "30000", "--score-tiles", "1", "--no-legs"],
                         capture_output=True, text=True, timeout=800, cwd=ROOT,
                         env={**env, "RSX_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "RSX_CHUNKS": "-1"})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["config"]["global_batch"] == 240000
    assert d["config"]["item_replicas_identical"] is True and d["config"]["item_chunks"] == 2
    assert d["config"]["launched_by"] == "bench.py" and d["config"]["dist_backend"] == "gloo"
    assert d["config"]["rccl_world"] is None            # gloo: the callbacks moved G; over "nccl" this is RCCL's own world size
    assert d["scoring"]["n_gpus"] == 2
    # ... and the launcher's second, short job: the same headline over the library's own mesh (RSX_EXCHANGE=direct), merged into the line
    mesh = d["legs"]["exchange_direct_mesh"]
    assert mesh["value"] > 0 and mesh["item_replicas_identical"] is True and mesh["item_chunks"] == 2, mesh
    assert "rsx_mesh" in mesh["exchange_issued_by"] and mesh["mesh_exchanges"] == (1 + 3 * 5) * 2 + 3
    assert d["config"]["mesh_value"] == mesh["value"] and "replicas identical" in d["config"]["mesh_note"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_plain_command_with_eight_ranks_on_one_gpu():
    """world size 8 before an 8-GPU node ever sees it: the plain command `python bench.py --gpus 8` at a reduced shape, eight
    processes on the box's one GPU, gloo moving G -- launcher, rendezvous, the ranks' agreement on the relabelled item ranges, the
    replica check, scoring on every rank; then the second job over the library's own mesh (slices of 1/8 with a remainder:
    2 x 6 001 rows per range).  Not a performance number."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                          "--users", "30000", "--batch", "30000", "--items", "12001", "--score-tiles", "1", "--no-legs"],
                         capture_output=True, text=True, timeout=560, cwd=ROOT,
                         env={**env, "RSX_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "RSX_CHUNKS": "-1"})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    if os.environ.get("RSX_SAVE_8RANK_LINE"):             # tools/r06_*.sh keep the line under profiles/
        with open(os.environ["RSX_SAVE_8RANK_LINE"], "w") as f:
            f.write(lines[0] + "\n")
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["config"]["global_batch"] == 240000
    assert d["config"]["item_replicas_identical"] is True and d["config"]["item_chunks"] == 2
    assert d["config"]["launched_by"] == "bench.py" and d["scoring"]["n_gpus"] == 8
    mesh = d["legs"]["exchange_direct_mesh"]
    assert mesh["value"] > 0 and mesh["item_replicas_identical"] is True and mesh["item_chunks"] == 2, mesh
    assert mesh["mesh_exchanges"] == (1 + 3 * 3) * 2 + 3


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_mesh_leg_that_fails_leaves_the_first_jobs_line_alone():
    """the launcher's second job (the library's own mesh) is cut off after one second: the line of the FIRST job comes out whole, exit
    status 0, and the leg says what happened"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--users", "120000", "--batch", "120000", "--items", "30000", "--score-tiles", "0", "--no-legs"],
                         capture_output=True, text=True, timeout=800, cwd=ROOT,
                         env={**env, "RSX_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "RSX_MESH_LEG_LIMIT_S": "1"})
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["item_replicas_identical"] is True
    leg = d["legs"]["exchange_direct_mesh"]
    assert leg["value"] is None and "still running" in leg["error"] and d["config"]["mesh_value"] is None


def test_bench_launcher_reports_a_failed_rank():
    """a rank that cannot start (here: no GPU in this container; on a GPU box: a launcher limit of zero seconds) must give a
    non-zero status and ONE JSON error line, not silence"""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.is_available():
        env["RSX_LAUNCH_LIMIT_S"] = "0"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--users", "50000", "--batch", "50000", "--items", "20000", "--score-tiles", "0", "--no-legs"],
                         capture_output=True, text=True, timeout=280, cwd=ROOT, env={**env, "RSX_DIST_BACKEND": "gloo"})
    assert out.returncode != 0
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 2 and "error" in lines[0]


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_watchdog_turns_a_hang_into_an_error_line_and_a_nonzero_exit():
    """N > 1: a stuck collective must not look like a slow run.  RSX_WATCHDOG_S = 0: the watchdog fires at once -- rank 0's
    stdout carries a JSON line with "error" and "value": null, the exit status is not 0"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() + 97) % 2000), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "4", "--warmup", "1", "--users", "120000", "--batch", "120000", "--items", "30000", "--score-tiles", "0", "--no-legs"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT,
                         env={**os.environ, "RSX_DIST_BACKEND": "gloo", "HSA_ENABLE_IPC_MODE_LEGACY": "0", "RSX_WATCHDOG_S": "0"})
    assert out.returncode != 0
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines and all(l["value"] is None and "watchdog" in l["error"] for l in lines)


def test_roofline_is_physical():
    """bench.roofline(): `frac` = PMC-measured HBM bytes of the launch / kernel time / peak -- a fraction, never the
    algorithmic 24 d bytes per triplet (which exceeds the peak where item sums stay on chip); null with a reason for a
    leg that was not profiled; the source of the traffic figure is named"""
    import json
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    keys = [k for k in t if k.startswith("U1000000_I100000_d128_B1000000_zipf_nb") and not k.endswith("nb0") and "_c" not in k]
    assert len(keys) == 1, keys                             # the headline leg, whatever item block it runs with
    key = keys[0]
    r = bench.roofline("bpr_step_blocked_kernel", 0.33, 50, 1_000_000, 100_000, 128, key)
    assert r["traffic"] == t[key]["hbm_bytes_per_launch"]
    assert abs(r["achieved"] - r["traffic"] / 0.33e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert 0.3 < r["frac"] < 1.0 and r["traffic_source"]["file"].startswith("profiles/")
    # stale: the kernel sources hash differently from the ones the profile was taken on (None: a profile older than the record)
    assert r["traffic_source"]["stale"] in (None, False, True)
    if t[key].get("sources_sha"):
        assert r["traffic_source"]["stale"] == (t[key]["sources_sha"] != bench.sources_sha("step"))
    assert r["algorithmic_rate_over_peak"] > 1.0            # the contract figure: not a fraction, and named accordingly
    assert 0.9 < r["traffic_over_compulsory"] < 1.5
    n = bench.roofline("bpr_step_kernel", 0.1, 10, 4321, 100_000, 128, "no_such_leg")
    assert n["frac"] is None and n["achieved"] is None and "no PMC profile" in n["frac_null_reason"]
    assert not any(k.startswith("frac") and isinstance(v, float) and v > 1.0 for k, v in {**r, **n}.items())

"""bench.py prints ONE JSON line with the driver's contract fields (+ roofline, cpu_baseline, scoring)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_json_contract_small_shape():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--users", "60000",
           "--items", "40000", "--batch", "60000", "--score-tiles", "1", "--cpu-batch", "4096", "--small-batch", "8192"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "scoring"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["small_batch"]["batch"] == 8192 and d["small_batch"]["value"] > 0
    s = d["scoring"]["roofline"]
    assert s["bound"] == "mfma" and s["peak"] == 157.3

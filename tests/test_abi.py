"""The C-ABI library: builds for gfx950, loads, and exports every symbol that
include/rsx.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rsx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsx_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def libpath():
    from recsys_pytorch_amd import build
    return build.build()


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("rsx_bpr_step", "rsx_apply_item_grad", "rsx_bpr_sample", "rsx_score", "rsx_topk",
                 "rsx_score_topk", "rsx_version", "rsx_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol(libpath):
    L = ctypes.CDLL(libpath)
    for name in declared_functions():
        assert hasattr(L, name), f"{name} declared in include/rsx.h but not exported"
    L.rsx_version.restype = ctypes.c_int
    assert L.rsx_version() == 8


def test_library_exports_nothing_the_header_does_not_declare(libpath):
    """the converse: no undeclared entry point (in particular no rsx_debug_* work-skipping hooks)"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and not l.split()[-1].startswith("_"))
    assert exported == declared_functions()
    assert not any("debug" in n or "ablat" in n for n in exported)


def test_options_are_validated(libpath):
    from recsys_pytorch_amd import rsx
    rsx.set_option("score_lanes", 2)
    rsx.set_option("sample_sort_cap", 0)
    for name, bad in (("score_lanes", 0), ("score_lanes", 5), ("sample_sort_cap", 4096), ("skip_atomics", 1)):
        with pytest.raises(rsx.RsxError):
            rsx.set_option(name, bad)


def test_binding_covers_every_declared_symbol(libpath):
    from recsys_pytorch_amd import rsx
    assert sorted(rsx.SIGNATURES) == declared_functions()
    assert rsx.version() == 8


def test_no_cpu_fallback():
    """host tensors are refused loudly, never silently computed on the CPU"""
    import torch
    from recsys_pytorch_amd import rsx
    P = torch.zeros(4, 32)
    with pytest.raises(rsx.RsxError):
        rsx.apply_item_grad(P, P.clone(), 0.1)


def test_argument_errors_use_the_error_channel(libpath):
    from recsys_pytorch_amd import rsx
    L = rsx.lib()
    assert L.rsx_bpr_step_workspace(10, 10, 48) < 0          # unsupported d
    rc = L.rsx_apply_item_grad(None, None, 10, 32, 0.1, None, None, 0, None)
    assert rc == -1 and b"null" in L.rsx_last_error()
    # sampler sizing calls run on the host: shapes are validated, sizes grow with the batch
    assert L.rsx_bpr_sample_workspace(-1, 10) < 0 and L.rsx_bpr_sample_workspace(10, 0) < 0
    assert L.rsx_bpr_sample_workspace(0, 10) == 0
    small, big = L.rsx_bpr_sample_workspace(1000, 100_000), L.rsx_bpr_sample_workspace(1_000_000, 100_000)
    assert 0 < small < big < (1 << 30)
    assert L.rsx_bpr_item_cdf_workspace(0) < 0 and L.rsx_bpr_item_cdf_workspace(1000) >= 8000
    assert L.rsx_score_topk_workspace(-1, 10) < 0 and L.rsx_score_topk_workspace(1024, 100_000) >= 1024 * 100_000 * 4
    rc = L.rsx_bpr_sample(None, None, 10, 10, 5, 1, 0, 0, 0, 0, 0, None, 0, None, None, None, None, None, None)
    assert rc == -1 and b"null" in L.rsx_last_error()
    rc = L.rsx_bpr_build_item_cdf(None, None, 10, 10, None, None, 0, None)
    assert rc == -1


_NULL_SWEEP = """
import ctypes as C, sys
sys.path.insert(0, %r)
from recsys_pytorch_amd import rsx
L = rsx.lib()
sizing = {"rsx_bpr_item_cdf_workspace", "rsx_bpr_sample_workspace", "rsx_bpr_step_det_workspace", "rsx_bpr_step_workspace", "rsx_chunk_rows",
          "rsx_score_topk_workspace",
          # (queries / no-ops by contract: a counter, and free(NULL))
          "rsx_mesh_alloc_refused", "rsx_mesh_free"}
bad = []
for name, (res, args) in sorted(rsx.SIGNATURES.items()):
    if name in ("rsx_last_error", "rsx_version"):
        continue
    for pat in (0, -1, 7):
        a = [None if t in (C.c_void_p, C.c_char_p) else 0.0 if t in (C.c_float, C.c_double) else (abs(pat) if t in (C.c_uint, C.c_uint64) else pat)
             for t in args]
        print("CALL", name, pat, flush=True)
        rc = getattr(L, name)(*a)
        if res is C.c_int and rc >= 0 and name not in sizing:
            bad.append((name, pat, rc))
print("DONE", bad)
"""


def test_every_entry_point_refuses_null_pointers_and_garbage_sizes(libpath):
    """every function of include/rsx.h called with ALL pointers null and every size 0, -1 and 7 (in a child process: a crash would be
    reported, not fatal): an error code through the error channel -- only the sizing functions may answer -- and no crash.  Runs
    without a GPU: the argument checks come before anything touches the device"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", _NULL_SWEEP % ROOT], capture_output=True, text=True, timeout=300)
    last = [ln for ln in r.stdout.splitlines() if ln.startswith("CALL")][-1:] or ["(none)"]
    assert r.returncode == 0, f"crashed in / after {last[0]}: {r.stderr[-1500:]}"
    done = [ln for ln in r.stdout.splitlines() if ln.startswith("DONE")]
    assert done and done[0] == "DONE []", done


def test_trainer_config_binding_mirrors_the_header_struct(libpath, tmp_path):
    """recsys_pytorch_amd/rsx.py:TrainerConfig against include/rsx.h:rsx_bpr_trainer_config: same fields in the same
    order, and the same size and offsets as the C compiler lays them out (a probe compiled from the header)"""
    import subprocess
    from recsys_pytorch_amd import rsx
    text = open(os.path.join(ROOT, "include", "rsx.h")).read()
    body = re.search(r"typedef struct rsx_bpr_trainer_config \{(.*?)\} rsx_bpr_trainer_config;", text, flags=re.S).group(1)
    fields = [re.search(r"(\w+);", line).group(1) for line in body.splitlines() if ";" in line]
    assert fields == [f[0] for f in rsx.TrainerConfig._fields_]
    probe = tmp_path / "probe.c"
    probe.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rsx.h"\nint main(void) {\n'
                     + 'printf("%zu\\n", sizeof(rsx_bpr_trainer_config));\n'
                     + "".join(f'printf("%zu\\n", offsetof(rsx_bpr_trainer_config, {f}));\n' for f in fields)
                     + "return 0; }\n")
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)])
    nums = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert nums[0] == ctypes.sizeof(rsx.TrainerConfig)
    assert nums[1:] == [getattr(rsx.TrainerConfig, f).offset for f in fields]

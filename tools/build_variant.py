"""tools/build_variant.py <name> [file.hip ...] [-Dflag ...] : build recsys_pytorch_amd/build/variants/librsx_<name>.so
from the current csrc/ with the given replacement sources (matched by basename) and extra compiler
flags -- for same-box A/B runs:  RSX_LIB=recsys_pytorch_amd/build/variants/librsx_<name>.so python tools/step_time.py"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from recsys_pytorch_amd import build as B
name, rest = sys.argv[1], sys.argv[2:]
repl = {os.path.basename(f): f for f in rest if not f.startswith("-")}
flags = [f for f in rest if f.startswith("-")]
vdir = os.path.join(B.HERE, "build", "variants", name)
shutil.rmtree(vdir, ignore_errors=True)
os.makedirs(vdir)
for f in os.listdir(B.CSRC):
    src = repl.get(f, os.path.join(B.CSRC, f))
    text = open(src).read().replace('"../../include/rsx.h"', '"%s"' % os.path.join(root, "include", "rsx.h"))
    open(os.path.join(vdir, f), "w").write(text)
procs, objs = [], []
for f in sorted(os.listdir(vdir)):
    if f.endswith(".hip"):
        o = os.path.join(vdir, f + ".o")
        procs.append(subprocess.Popen([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), f"--offload-arch={B.ARCH}", *B.FLAGS, *flags,
                                       "-c", os.path.join(vdir, f), "-o", o]))
        objs.append(o)
assert all(p.wait() == 0 for p in procs)
out = os.path.join(B.HERE, "build", "variants", f"librsx_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", out, *objs])
print(out)

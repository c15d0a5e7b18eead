"""tools/asm_mix.py <file.s> <kernel substring> : instruction mix of a kernel and of its loops (backward branches) in hipcc's -S output.
How: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only -o /tmp/k.s recsys_pytorch_amd/csrc/rsx_bpr.hip"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
want = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(want) + r'\S*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M)
print(m.group(1))
lines = [l.split(';')[0].strip() for l in m.group(2).split('\n')]
lines = [l for l in lines if l and (l.endswith(':') or not l.startswith('.'))]


def cls(l):
    op = l.split()[0]
    if op.endswith(':'): return 'label'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    return 'other'


print(len(lines), 'lines', dict(Counter(cls(l) for l in lines)))
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
for i, l in enumerate(lines):
    mm = re.match(r's_c?branch\w*\s+(\S+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        a = labels[mm.group(1)]
        c = Counter(cls(x) for x in lines[a:i + 1])
        print('loop lines', a, '-', i, dict(c))
        if '--ops' in sys.argv:
            print(Counter(x.split()[0] for x in lines[a:i + 1] if cls(x) == 'valu').most_common(40))

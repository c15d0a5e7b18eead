"""tools/gemm_ref.py : the library fp32 GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the scoring shape
[rows x d] @ [d x items], for comparison with score_tile_kernel.  Output written, nothing fused."""
import sys, torch
rows, I, d = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 100_000, 128
dev = torch.device("cuda")
P = torch.randn(rows, d, device=dev) * 0.1
Q = torch.randn(I, d, device=dev) * 0.1
out = torch.empty(rows, I, device=dev)
for name, f in (("P @ Q.T (NT)", lambda: torch.matmul(P, Q.t(), out=out)),
                ("P @ Qt (NN, item table pre-transposed)", None)):
    if f is None:
        Qt = Q.t().contiguous()
        f = lambda: torch.matmul(P, Qt, out=out)
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        f()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(f"{name}: {ms*1e3:.0f} us per {rows} x {I} x {d}: {2*rows*I*d/ms/1e9:.1f} TFLOP/s fp32 ({rows*I/ms/1e6:.1f} G scores/s), "
          f"output write {rows*I*4/ms/1e6:.0f} GB/s")

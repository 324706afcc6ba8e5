"""Trailing-update kernel alone (no panel stream beside it): TFLOP/s against K and matrix size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import py_schedule as multigpu

eng = multigpu.HipEngine(0)
DATA = sys.argv[1] if len(sys.argv) > 1 else "random"
SIZES = (8192, 16384, 32768) if DATA == "random" else (32768,)
if os.environ.get("GEMM_SIZES"):
    SIZES = tuple(int(v) for v in os.environ["GEMM_SIZES"].split(","))
print("operand data:", DATA)
for Np in SIZES:
    ld = Np + 32
    C = torch.zeros(Np * ld, dtype=torch.float64, device="cuda")
    for K in (128, 256, 512, 1024, 2048):
        if DATA == "random":
            P = torch.randn(Np * K, dtype=torch.float64, device="cuda") * 1e-3
        elif DATA == "zeros":
            P = torch.zeros(Np * K, dtype=torch.float64, device="cuda")
        else:
            P = torch.full((Np * K,), 1e-3, dtype=torch.float64, device="cuda")
        eng.update_block(P, Np, 0, K, C, ld, Np, 0, Np)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            eng.update_block(P, Np, 0, K, C, ld, Np, 0, Np)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        mt = Np // 128
        flops = mt * (mt + 1) / 2 * 2 * 128 * 128 * K
        print(f"Np={Np} K={K}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s", flush=True)
        del P
    del C
# narrow updates (the next block column of the panel chain: Wc = 512 columns, all rows below): latency of one partial wave of
# workgroups rather than throughput.  GEMM_NARROW=1
if os.environ.get("GEMM_NARROW"):
    for Np in (4096, 8192, 16384, 32768):
        ld = Np + 32
        Cn = torch.zeros(512 * ld, dtype=torch.float64, device="cuda")
        for K in (128, 512):
            P = torch.randn(Np * K, dtype=torch.float64, device="cuda") * 1e-3
            eng.update_block(P, Np, 0, K, Cn, ld, Np, 0, 512)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record()
            for _ in range(reps):
                eng.update_block(P, Np, 0, K, Cn, ld, Np, 0, 512)
            e1.record()
            torch.cuda.synchronize()
            print(f"narrow Np={Np} Wc=512 K={K}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us", flush=True)

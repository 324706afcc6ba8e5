"""The trailing-update kernel under a SUSTAINED load (0.6 s of back-to-back launches, like a factorisation) instead of
a 30 ms burst: TFLOP/s per group of launches over time, with rocm-smi power / clock samples beside it.
  python tools/sustained_gemm.py [K] ; GPAK_GEMM=lds python tools/sustained_gemm.py"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import py_schedule as multigpu

K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Np = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
eng = multigpu.HipEngine(0)
ld = Np + 32
C = torch.zeros(Np * ld, dtype=torch.float64, device="cuda")
P = torch.randn(Np * K, dtype=torch.float64, device="cuda") * 1e-3
samples, stop = [], False


def smi():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-P", "-c", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=5).stdout.decode()
            d = json.loads(out)
            c0 = d.get("card0", {})
            samples.append((time.perf_counter(), {k: v for k, v in c0.items() if "ower" in k or "sclk" in k}))
        except Exception as e:  # noqa
            samples.append((time.perf_counter(), {"error": str(e)[:80]}))
        time.sleep(0.05)


th = threading.Thread(target=smi, daemon=True)
th.start()
time.sleep(0.3)
mt = Np // 128
flops = mt * (mt + 1) / 2 * 2 * 128 * 128 * K
eng.update_block(P, Np, 0, K, C, ld, Np, 0, Np)
torch.cuda.synchronize()
groups, per = int(os.environ.get("SUS_GROUPS", "12")), 8
evs = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
t0 = time.perf_counter()
evs[0].record()
for g in range(groups):
    for _ in range(per):
        eng.update_block(P, Np, 0, K, C, ld, Np, 0, Np)
    evs[g + 1].record()
torch.cuda.synchronize()
t1 = time.perf_counter()
stop = True
rates = [round(per * flops / (evs[g].elapsed_time(evs[g + 1]) * 1e-3) / 1e12, 1) for g in range(groups)]
print(f"kernel={os.environ.get('GPAK_GEMM', 'rs')} Np={Np} K={K}: TFLOP/s per group of {per} launches over {t1 - t0:.2f} s: {rates}")
busy = [s for t, s in samples if t0 <= t <= t1]
print("rocm-smi during the load:", busy[:12])
print("rocm-smi before:", [s for t, s in samples if t < t0][:2])

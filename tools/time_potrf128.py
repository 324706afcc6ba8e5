"""Micro-timing of the diagonal-block kernel alone (diagnostic, GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import py_schedule as multigpu
eng = multigpu.HipEngine(0)
rng = np.random.default_rng(0)
G = rng.normal(size=(128, 128)); A = G @ G.T + 128 * np.eye(128)
ld = 128
blk0 = eng.from_numpy(np.asfortranarray(A).T.ravel().copy())
inv = eng.empty(2 * 128 * 128)
info = eng.zeros(4, dtype=torch.int32); info.fill_(0x7fffffff)
blks = [blk0.clone() for _ in range(200)]
torch.cuda.synchronize()
for b in blks[:20]:
    eng.factor_panel(b, ld, 128, 0, 128, inv, info)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for b in blks[20:]:
    eng.factor_panel(b, ld, 128, 0, 128, inv, info)
e1.record(); torch.cuda.synchronize()
print("potrf128 avg us:", e0.elapsed_time(e1) * 1e3 / 180)
L = blks[50].cpu().numpy().reshape(128, 128).T
print("residual", np.abs(np.tril(L) @ np.tril(L).T - A).max(), "inv err", np.abs(inv.cpu().numpy()[:128*128].reshape(128,128).T @ np.tril(L) - np.eye(128)).max())

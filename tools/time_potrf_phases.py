import sys, os, ctypes as C
sys.path.insert(0, "/root/repo") if os.path.exists("/root/repo") else None
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gp_ss_ak_amd import multigpu, _lib
eng = multigpu.HipEngine(0)
lib = _lib.load()
rng = np.random.default_rng(0)
G = rng.normal(size=(128, 128)); A = G @ G.T + 128 * np.eye(128)
blk0 = eng.from_numpy(np.asfortranarray(A).T.ravel().copy())
inv = eng.empty(2 * 128 * 128)
info = eng.zeros(4, dtype=torch.int32); info.fill_(0x7fffffff)
for rep in range(3):
    b = blk0.clone()
    eng.factor_panel(b, 128, 128, 0, 128, inv, info)
    torch.cuda.synchronize()
out = (C.c_longlong * 64)()
lib.gpak_dev_potrf_timing(out)
t = list(out)
print("load", t[1]-t[0], "diag0", t[2]-t[1])
for kb in range(7):
    print("kb", kb, "panel-phase(incl barrier)", t[3+4*kb]-(t[2] if kb==0 else t[6+4*(kb-1)]), "tile0", t[4+4*kb]-t[3+4*kb], "diag", t[5+4*kb]-t[4+4*kb], "wait-others", t[6+4*kb]-t[5+4*kb])
print("store", t[33]-t[32], "inverse", t[34]-t[33], "total", t[34]-t[0])
print("kb0 helpers rel. to slot start", "w1 start", t[44]-t[3], "trailing done", t[40]-t[3], "phase_b done", t[41]-t[3], "| w4 stores", t[42]-t[3], t[43]-t[3], "| w0 diag done", t[5]-t[3], "barrier", t[6]-t[3])
b = t[6+4*2]
print("kb3 panel phase, wave 1 rel. to the barrier before it (t==0 stamp): enter", t[45]-b, "operands in", t[46]-b, "mfma done", t[47]-b, "stored", t[48]-b, "after barrier", t[49]-b, "wave0 after barrier", t[3+4*3]-b)
print("diag(3): tile-0 update", t[50]-t[4+4*2], "column load", t[51]-t[50], "pivots 0-7", t[52]-t[51], "pivots 8-15", t[53]-t[52], "write", t[54]-t[53])

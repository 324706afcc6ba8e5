import sys, os, ctypes as C
sys.path.insert(0, "/root/repo") if os.path.exists("/root/repo") else None
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import py_schedule as multigpu
from gp_ss_ak_amd import _lib
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
    print("kb", kb, "panel-phase(incl barrier)", t[3+4*kb]-(t[2] if kb==0 else t[6+4*(kb-1)]), "next diagonal block (wave 0)", t[5+4*kb]-t[3+4*kb], "wait-others", t[6+4*kb]-t[5+4*kb])
print("total", t[34]-t[0], "cycles")
print("last row of the inverse + final stores", t[34]-t[33])
print("kb0 helper wave 1 rel. to its slot's start: start", t[44]-t[3], "trailing tiles done", t[40]-t[3], "phase B done", t[41]-t[3], "| wave 0's block done", t[5]-t[3], "barrier", t[6]-t[3])
print("block 3 (wave 0): tile-0 update", t[50]-t[4+4*2], "column read", t[51]-t[50], "pivots 0-7", t[52]-t[51], "pivots 8-15", t[53]-t[52], "write", t[54]-t[53])

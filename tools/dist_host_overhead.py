"""How far ahead of the GPU does the Python host run in the distributed schedule (one rank, pipelined)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import py_schedule as multigpu
from gp_ss_ak_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
eng = multigpu.HipEngine(0)
X, y = synth.drillholes(N)
gp = multigpu.DistGP(eng, X, y, nb=512, pipeline=True)
gp.set_params(synth.DEFAULT_EXPANS, synth.DEFAULT_BIAS, synth.DEFAULT_SN2, 1)
for it in range(3):
    gp.fill(); torch.cuda.synchronize()
    t0 = time.perf_counter(); bad = None
    # factor() ends with a device->host read of info, so time the enqueue part separately
    gp.info.fill_(multigpu.INT_MAX)
    t0 = time.perf_counter()
    gp._factor_pipelined_enqueue = True
    v = gp.factor()
    t1 = time.perf_counter()
    rhs = gp.y / gp.params[2]
    torch.cuda.synchronize(); t2 = time.perf_counter()
    x = gp.solve(rhs)
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"factor total {1e3*(t1-t0):.1f} ms | solve enqueue {1e3*(t3-t2):.1f} ms, solve total {1e3*(t4-t2):.1f} ms")
import cProfile, pstats
gp.fill(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); gp.factor(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)

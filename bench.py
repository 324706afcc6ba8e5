#!/usr/bin/env python3
"""bench.py -- GP train step/sec (Gram + Cholesky + logML) at N=32768 fp64 on MI355X.

One "step" = one hot-path evaluation at a NEW hyper-parameter vector, i.e. what every
ObjVal() of the reference's optimiser costs (GP_utils::set_GP_Pars + logLikelihood,
GP_Utils.cpp:130-157, 1138-1162): fused Gram/B fill, blocked Cholesky, two triangular solves,
f = K*alpha, log-det and the nlZ reductions.  X and y are resident in HBM before the timed
region; parameters (10 doubles) are the only per-step host input.

Prints ONE JSON line (see the driver contract): metric/value/unit + `roofline` for the
dominant kernel (the MFMA trailing update of the Cholesky) + `cpu_baseline` (the oracle's
reference-sequence on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# fp64 matrix peak of MI355X: 78.6 TFLOP/s (AMD CDNA4 data sheet; = 256 CU x 4 SIMD x 2.4 GHz x
# 32 flop/clk, i.e. one v_mfma_f64_16x16x4_f64 per 64 clk per SIMD).  MI355X_MICROARCH.md lists no
# fp64 row, so the calibrated issue-rate microbenchmark is reported next to it.
PEAK_F64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def params_for_step(i):
    """Reference defaults (Kernel.cpp:763-773, :317-320, GP_Utils.cpp:43), nudged per step so that
    every step is a genuinely new parameter vector (nothing can be memoised)."""
    from gp_ss_ak_amd import synth
    e = np.array(synth.DEFAULT_EXPANS, dtype=np.float64)
    e[1] += 1e-3 * (i % 7)
    e[3] += 5e-4 * (i % 5)
    return e, synth.DEFAULT_BIAS, synth.DEFAULT_SN2 * (1.0 + 1e-3 * (i % 3))


def cpu_baseline(n_full, sample_n):
    """Reference operation sequence on the host cores through the CPU oracle (kind: port)."""
    from gp_ss_ak_amd import synth
    from oracle import oracle as orc
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    X, y = synth.drillholes(sample_n)
    e, bias, sn2 = params_for_step(0)
    t0 = time.perf_counter()
    K = orc.gram(X, X, e, bias, orc.DIST_EXPANSION)
    t_gram = time.perf_counter() - t0
    # give the CPU its best shot: OpenBLAS on all visible hardware threads is often slower than on the
    # physical cores, and a container may see many more cores than its CPU share (256 visible, 16 granted
    # on the GPU box) -- a small probe picks the thread count, the timed run uses it
    lapack, best = False, (None, cores)
    for thr in sorted({t for t in (cores, cores // 2, cores // 4, 32, 16, 8) if 1 <= t <= cores}, reverse=True):
        lapack = orc.use_lapack(thr)
        if not lapack:
            break
        Kp = K[:4096, :4096].copy(order="F")
        t0 = time.perf_counter()
        orc.nlz_lean(Kp, y[:4096], sn2, want_L=False)
        dt = time.perf_counter() - t0
        if best[0] is None or dt < best[0]:
            best = (dt, thr)
    cores = best[1]
    if lapack:
        orc.use_lapack(cores)
    orc.lib().orc_set_threads(min(cores, 64))      # the oracle's own loops (Gram fill, GEMV fallbacks)
    t0 = time.perf_counter()
    K = orc.gram(X, X, e, bias, orc.DIST_EXPANSION)   # timed again with the chosen team
    t_gram = time.perf_counter() - t0
    t0 = time.perf_counter()
    info, _, _ = orc.nlz_refseq(K, y, sn2, want_L=False)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    info2, _, _ = orc.nlz_lean(K, y, sn2, want_L=False)
    t_lean = time.perf_counter() - t0
    scale = (sample_n / float(n_full)) ** 3
    step_ref = t_gram + t_ref
    step_lean = t_gram + t_lean
    return {
        "value": scale / step_ref,
        "unit": "steps/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"N={sample_n} reference sequence (expansion Gram + IRLS/Brent: {info.n_chol} Cholesky, "
                   f"{info.n_gemv} GEMV) measured {step_ref:.2f} s/step, "
                   f"{'SciPy-OpenBLAS dpotrf/dtrsm/dgemv' if lapack else 'in-repo blocked C'} on {cores} threads; "
                   + (f"scaled to N={n_full} by (N_s/N)^3" if sample_n != n_full else "full size, no scaling")),
        "sample_n": sample_n,
        "sample_s_per_step": step_ref,
        "lean_value": scale / step_lean,
        "lean_sample_s_per_step": step_lean,
        "nlz_sample": info.nlz,
    }


def _timed_steps(g, mode, steps, warmup):
    """`steps` timed train steps after `warmup` untimed ones; returns (wall seconds, last nlZ, summed phases, launches)."""
    def step(i):
        e, bias, sn2 = params_for_step(i)
        g.set_params(e, bias, sn2, mode)
        return g.logLikelihood()

    for i in range(warmup):
        step(i)
    keys = ("gram_ms", "factor_ms", "solve_ms", "nlz_ms", "kmatvec_ms", "trailing_ms", "trailing_flops", "trailing_bytes")
    phases = {k: 0.0 for k in keys}
    launches = 0
    t0 = time.perf_counter()
    nlz = None
    for i in range(steps):
        nlz = step(warmup + i)  # synchronous: returns after the device finished
        t = g.timing()
        for k in keys:
            phases[k] += t[k]
        launches += t["trailing_launches"]
    return time.perf_counter() - t0, nlz, phases, launches


def _sources_sha16():
    """Fingerprint of the kernel / schedule sources the dominant kernel's behaviour depends on: the PMC summary
    records it when it is made (tools/pmc_summary.py), bench.py recomputes it -- a profile of other code is flagged."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gemm.hip", "potrf.hip", "api.hip", "gpak_internal.h"):
        with open(os.path.join(ROOT, "gp_ss_ak_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc(tag, N):
    """HBM-side traffic and in-situ clock of the dominant kernel from the committed rocprofv3 --pmc passes of THIS
    command (profiles/<tag>_pmc_summary_N<N>.json, written by tools/profile_round.sh <tag>): per launch,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane reads."""
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary_N{N}.json")
    if not os.path.exists(path):
        return None
    z = json.load(open(path))
    out = {"source": os.path.relpath(path, ROOT), "profile_sources_sha16": z.get("sources_sha16"),
           "current_sources_sha16": _sources_sha16()}
    out["stale"] = out["profile_sources_sha16"] != out["current_sources_sha16"]

    def pick(section):   # the bulk trailing update: gpak_gemm_nt_f64_rs<4, 2, true[, false]>
        for name, v in z.get(section, {}).items():
            if name.startswith("void gpak_gemm_nt_f64_rs<4, 2, true"):
                return v
        raise KeyError(section)
    try:
        f, w = pick("FETCH_SIZE")["FETCH_SIZE"], pick("WRITE_SIZE")["WRITE_SIZE"]
        out["traffic"] = (2.0 * f["sum"] / f["dispatches"] + w["sum"] / w["dispatches"]) * 1024.0
    except KeyError:
        out["traffic"] = None
    try:
        sq = pick("SQ")
        out["mfma_busy_cycles_per_launch"] = sq["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / sq["SQ_VALU_MFMA_BUSY_CYCLES"]["dispatches"]
        out["grbm_gui_active_per_launch"] = sq["GRBM_GUI_ACTIVE"]["sum"] / sq["GRBM_GUI_ACTIVE"]["dispatches"]
    except KeyError:
        pass
    return out


def config3(iters, N=32768):
    """BASELINE.json configs[2]: the full L-BFGS hyper-parameter loop through the C++ CLI (per-evaluation Gram rebuild
    + Cholesky; gradient evaluations add the B^-1 build).  Returns ms per iteration and evaluations per iteration."""
    import subprocess
    import tempfile
    from gp_ss_ak_amd import synth
    exe = os.path.join(ROOT, "gp_ss_ak_amd", "host", "gp_ss_ak")
    if not os.path.exists(exe):
        return {"error": "gp_ss_ak_amd/host/gp_ss_ak is not built"}
    with tempfile.TemporaryDirectory() as d:
        Xr, yr = synth.drillholes_raw(N)
        with open(os.path.join(d, "train.txt"), "w") as f:
            for r, v in zip(Xr, yr):
                f.write("\t".join(f"{t:.17g}" for t in list(r) + [v]) + "\n")
        env = dict(os.environ, GPAK_MAX_ITERS=str(iters), GPAK_OPT_TRACE=os.path.join(d, "trace.txt"))
        t0 = time.perf_counter()
        out = subprocess.run([exe, "-v", "1", "-np", "--timing", os.path.join(d, "timing.json"), "train", "-k", "ExpAns",
                              "-kn", "1", "-o", "LBFGS", os.path.join(d, "train.txt"), os.path.join(d, "model")],
                             env=env, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.perf_counter() - t0
        if out.returncode != 0:
            return {"error": out.stderr.decode()[-500:]}
        rows = [[float(v) for v in line.split()] for line in open(os.path.join(d, "trace.txt"))]
        tim = json.load(open(os.path.join(d, "timing.json")))
    evals = int(rows[-1][2])
    acc = tim["accumulated"]
    dev_ms = acc["gram_ms"] + acc["factor_ms"] + acc["solve_ms"] + acc["nlz_ms"]
    return {"N": N, "iterations": len(rows), "evaluations": evals, "evaluations_per_iteration": evals / len(rows),
            "kept_objective": [r[1] for r in rows], "train_verb_wall_s": wall,
            "hot_path_evaluations": tim["evaluations"], "hot_path_ms_per_evaluation": dev_ms / max(tim["evaluations"], 1),
            "last_grad_ms": tim["last"]["grad_ms"],
            "note": "train verb end to end (csv read, L-BFGS, model file, Calc_Out on the N training points); every "
                    "evaluation is a new parameter vector: fill + Cholesky + solves + nlZ, gradient evaluations add "
                    "B^-1 (2N^3/3) + the fused pair pass"}


def vendor_potrf(sizes, limit_s=420):
    """The vendor stack's Cholesky of the SAME matrices on the same device: torch.linalg.cholesky, fp64 (ROCm: hipSOLVER /
    rocSOLVER dpotrf), in a process of its own (tools/vendor_potrf.py --json: torch brings its own HIP runtime, which stays
    out of the measured process).  A reference point, like cpu_baseline: not the thing measured, not part of the product."""
    import subprocess
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "vendor_potrf.py"), "--json",
           *[str(n) for n in sizes]]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
        rows = [json.loads(ln) for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        if not rows:
            return {"error": (r.stderr.decode() or "no output")[-300:]}
        return {"what": "torch.linalg.cholesky, fp64 (hipSOLVER / rocSOLVER dpotrf) of the same B = I + K/sn2 on the same device, "
                        "own process; library_factor_ms measured in that process too", "sizes": rows}
    except Exception as ex:   # never let a reference point take the bench line down
        return {"error": repr(ex)[:300]}


def run_single(args):
    from gp_ss_ak_amd import gpak, synth
    N = args.n
    X, y = synth.drillholes(N)
    g = gpak.Gpak(int(os.environ.get("LOCAL_RANK", "0")))
    g.set_option(gpak.OPT_PROFILE, 1)
    if args.nb_outer:
        g.set_option(gpak.OPT_NB_OUTER, args.nb_outer)
    g.set_train(X, y)  # inputs resident in HBM from here on
    mode = gpak.DIST_DIRECT if args.dist == "direct" else gpak.DIST_EXPANSION
    wall, nlz, phases, launches = _timed_steps(g, mode, args.steps, args.warmup)
    ms_per_step = wall / args.steps * 1e3
    achieved = phases["trailing_flops"] / (phases["trailing_ms"] * 1e-3) / 1e12 if phases["trailing_ms"] > 0 else None
    tim = g.timing()
    Np = tim["n_padded"]
    fill_gbs = tim["gram_bytes"] / (phases["gram_ms"] / args.steps * 1e-3) / 1e9
    # fp64 vector rate: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz = 39.3e12 lane-instructions/s; one kernel evaluation
    # of the default composition is 32 fp64 instruction slots (6 distance, 13 sqrt incl. v_rsq_f64 at quarter rate,
    # 12 exp, 1 accumulate: csrc/gram.hip gpak_k1)
    # From 32 macro blocks of 512 points on (N > 15872) f = K alpha runs on the SYMMETRIC kernel: one workgroup per pair
    # of macro blocks, every value used for both of its entries -- nbm (nbm + 1) / 2 squares of 512^2 evaluations at
    # ~34.7 slots each (the row-side accumulate and the wave transpose-reduction on top of the 32)
    kmv_s = phases["kmatvec_ms"] / args.steps * 1e-3
    nbm = (N + 511) // 512
    kmv_sym = nbm >= 32 and nbm <= 64 and args.dist in ("direct", "expansion") and os.environ.get("GPAK_KMV_SYM", "1") != "0"
    kmv_evals = nbm * (nbm + 1) / 2.0 * 512.0 * 512.0 if kmv_sym else float(N) * N
    kmv_slots = 34.7 if kmv_sym else 32.0
    kmv_rate = kmv_evals * kmv_slots / kmv_s if kmv_s > 0 else None
    out = {
        "metric": f"GP train step/sec (Gram+Cholesky+logML) at N={N} fp64",
        "value": args.steps / wall,
        "unit": "steps/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"N={N} fp64 ExpAns+Bias Gram + blocked Cholesky + 2 trsv + logML, 3-D synthetic "
                               f"drill-holes, new hyper-parameters every step", "N": N, "dist_mode": args.dist,
                   "nb_outer": args.nb_outer or 512, "parallelism": "1 GPU"},
        "phases_ms_per_step": {k: phases[k] / args.steps for k in ("gram_ms", "factor_ms", "solve_ms", "nlz_ms", "kmatvec_ms")},
        "nlz": nlz,
        "roofline": {
            "kernel": "gpak_gemm_nt_f64_rs (Cholesky trailing update, v_mfma_f64_16x16x4_f64, register-streamed operands)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": PEAK_F64_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": (achieved / PEAK_F64_MFMA_TFLOPS) if achieved else None,
            "traffic": None,
            # every lower C tile read once and written once per launch, whatever the launch's panel width K
            "algorithmic_bytes_per_launch": phases["trailing_bytes"] / max(launches, 1),
            "avg_launch_ms": phases["trailing_ms"] / max(launches, 1),
            "launches_per_step": launches / args.steps,
            "flops_per_step": phases["trailing_flops"] / args.steps,
            "whole_factor_frac": ((Np ** 3 / 3.0) / (phases["factor_ms"] / args.steps * 1e-3) / 1e12
                                  / PEAK_F64_MFMA_TFLOPS),
            "fill": {"kernel": "gpak_fill1_f64 (fused Gram/B fill, lower 128x64 tiles)", "bound": "hbm",
                     "achieved": fill_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": fill_gbs / PEAK_HBM_GBS,
                     "algorithmic_bytes": tim["gram_bytes"], "ms": phases["gram_ms"] / args.steps},
            # back substitution (the forward one rides along with the factorisation): L's lower triangle streamed once
            "solve": {"kernel": "gpak_bwd_step_f64 (back substitution, one launch per 512 columns: 64 at N=32768)",
                      "bound": "hbm", "algorithmic_bytes": 8.0 * Np * (Np + 512) / 2.0,
                      "achieved": 8.0 * Np * (Np + 512) / 2.0 / (phases["solve_ms"] / args.steps * 1e-3) / 1e9,
                      "peak": PEAK_HBM_GBS, "unit": "GB/s",
                      "frac": 8.0 * Np * (Np + 512) / 2.0 / (phases["solve_ms"] / args.steps * 1e-3) / 1e9 / PEAK_HBM_GBS,
                      "ms": phases["solve_ms"] / args.steps},
            "kmatvec": {"kernel": ("gpak_kmatvec1_sym_f64 (f = K alpha, K recomputed, each value used for both entries)" if kmv_sym
                                   else "gpak_kmatvec1_part_f64 (f = K alpha, K recomputed, nothing stored)"),
                        "bound": "valu_f64",
                        "achieved": kmv_rate / 1e12 if kmv_rate else None, "peak": 39.3, "unit": "T lane-instr/s",
                        "frac": kmv_rate / 39.3e12 if kmv_rate else None, "evaluations": kmv_evals,
                        "fp64_instructions_per_evaluation": kmv_slots, "ms": kmv_s * 1e3,
                        # the same time against the N^2 x 32 slots of the kernel that does not use the symmetry
                        "frac_of_unsymmetric_work": (float(N) * N * 32.0 / kmv_s / 39.3e12) if kmv_s > 0 else None},
        },
    }
    tag = args.profile_tag
    out["profile_tag"] = tag
    pm = _pmc(tag, N)
    if pm is not None and pm["stale"]:
        # the committed profile was taken on other kernel / schedule sources than the ones running now: its counters are
        # not this code's -- reported as absent, with the reason, rather than quoted
        out["roofline"]["traffic"] = None
        out["roofline"]["traffic_note"] = (f"{pm['source']} was recorded on sources {pm['profile_sources_sha16']}, this run "
                                           f"has {pm['current_sources_sha16']}: re-run tools/profile_round.sh {tag}")
    elif pm is not None:
        out["roofline"]["traffic"] = pm.get("traffic")
        out["roofline"]["traffic_source"] = (pm["source"] + " (2*FETCH_SIZE + WRITE_SIZE per launch; separate --pmc passes of "
                                             "this command on the same sources, sha " + str(pm["current_sources_sha16"]) + ")")
        if "grbm_gui_active_per_launch" in pm and out["roofline"]["avg_launch_ms"]:
            # GRBM_GUI_ACTIVE counts per XCD; the collected value is the sum over the 8 XCDs
            ghz = pm["grbm_gui_active_per_launch"] / 8.0 / (out["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
            out["roofline"]["in_situ_clock_ghz"] = ghz
            out["roofline"]["peak_at_in_situ_clock"] = PEAK_F64_MFMA_TFLOPS * ghz / 2.4
            out["roofline"]["mfma_busy_frac"] = (pm["mfma_busy_cycles_per_launch"] / (256 * 4) /
                                                 (pm["grbm_gui_active_per_launch"] / 8.0))
    if args.grad:
        # config 3 of BASELINE.json: what one Grad_Values() of the L-BFGS loop costs on top of ObjVal()
        t0 = time.perf_counter()
        for i in range(args.grad):
            e, bias, sn2 = params_for_step(100 + i)
            g.set_params(e, bias, sn2, mode)
            gv = g.GradLL()
        gwall = (time.perf_counter() - t0) / args.grad
        tg = g.timing()
        out["grad_step"] = {"ms_per_grad_eval": gwall * 1e3, "grad_ms": tg["grad_ms"],
                            "algorithmic_flops": 2.0 * Np ** 3 / 3.0,
                            "tflops": 2.0 * Np ** 3 / 3.0 / (tg["grad_ms"] * 1e-3) / 1e12,
                            "g": [float(v) for v in gv]}
    if args.calibrate:
        tf, gbs = g.calibrate()
        out["roofline"]["calibrated_mfma_f64_tflops"] = tf
        out["roofline"]["calibrated_hbm_write_gbs"] = gbs
        out["roofline"]["fill"]["calibrated_store_gbs"] = gbs
        out["roofline"]["fill"]["frac_of_calibrated"] = fill_gbs / gbs
    if N != 65536 and not args.no_n65536:
        # north_star's scaling curve is quoted at N=65536 (BASELINE.json configs[3]): the 1-GPU point (34 GB matrix)
        X6, y6 = synth.drillholes(65536)
        g.set_train(X6, y6)
        w6, nlz6, ph6, _ = _timed_steps(g, mode, 2, 1)
        out["n65536"] = {"N": 65536, "steps_per_s": 2 / w6, "ms_per_step": w6 / 2 * 1e3, "nlz": nlz6,
                         "factor_ms": ph6["factor_ms"] / 2,
                         "whole_factor_frac": (65536 ** 3 / 3.0) / (ph6["factor_ms"] / 2 * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS}
    if args.config2 and N == 32768:
        # BASELINE.json configs[1]: N=8192 Gram + Cholesky on one GPU against the CPU path -- small enough for the CPU
        # leg to run at FULL size (no (N_s/N)^3 scaling), which cpu_baseline() does below
        X2, y2 = synth.drillholes(args.config2)
        g.set_train(X2, y2)
        w2, nlz2, ph2, _ = _timed_steps(g, mode, 20, 3)
        e0, b0, s0 = params_for_step(0)          # the parameters the CPU leg uses: same inputs, comparable nlZ
        g.set_params(e0, b0, s0, mode)
        out["config2"] = {"N": args.config2, "steps_per_s": 20 / w2, "ms_per_step": w2 / 20 * 1e3,
                          "nlz_step0_params": g.logLikelihood(),
                          "factor_ms": ph2["factor_ms"] / 20, "gram_ms": ph2["gram_ms"] / 20}
    if args.config5 and N == 32768:
        # BASELINE.json configs[4]: N=32768, M=1e6 block-model points, fp32 prediction (GPAK_F32 context: fp64 training
        # step; fp32 cross-kernel / forward substitution -- fp32 MFMA chunks of K=128 summed in fp64 -- / variance sums),
        # with the fp64 context on the first 65536 points as the accuracy reference of the same run
        M5 = args.config5
        X5, y5 = synth.drillholes(N)
        Xt5 = synth.test_points(M5)
        e0, b0, s0 = params_for_step(0)
        g32 = gpak.Gpak(int(os.environ.get("LOCAL_RANK", "0")), gpak.F32)
        g32.set_train(X5, y5)
        g32.set_params(e0, b0, s0, mode)
        g32.logLikelihood()
        g32.posteriorMeanVar(Xt5[:min(M5, 65536)])          # one full batch: buffers + the fp32 image of the factor
        t0 = time.perf_counter()
        m32, v32 = g32.posteriorMeanVar(Xt5)
        w32 = time.perf_counter() - t0
        g32.close()
        g.set_train(X5, y5)
        g.set_params(e0, b0, s0, mode)
        Mref = min(M5, 65536)
        t0 = time.perf_counter()
        m64, v64 = g.posteriorMeanVar(Xt5[:Mref])
        w64 = time.perf_counter() - t0
        out["config5"] = {"N": N, "M": M5, "precision": "fp64 fill / factorisation / alpha / mean; fp32 MFMA for the "
                          "M-proportional variance work, accumulated in fp64 across K=128 chunks",
                          "wall_s": w32, "points_per_s": M5 / w32,
                          "variance_tflops": float(N) * N * M5 / w32 / 1e12,
                          "frac_of_fp32_mfma_peak": float(N) * N * M5 / w32 / 1e12 / 157.3,
                          "fp64_reference": {"M": Mref, "wall_s": w64, "variance_tflops": float(N) * N * Mref / w64 / 1e12,
                                             "variance_max_rel_diff": float(np.abs(v32[:Mref] - v64).max() / v64.max()),
                                             "mean_max_rel_diff": float(np.abs(m32[:Mref] - m64).max() / np.abs(m64).max())}}
    g.close()
    if args.vendor and N == 32768:
        out["vendor_potrf"] = vendor_potrf([N] + ([args.config2] if args.config2 else []))
    if args.config3 and N == 32768:
        out["config3"] = config3(args.config3, N)
    if not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(N, args.cpu_n)
        if "config2" in out:
            c2 = cpu_baseline(args.config2, args.config2)
            out["config2"]["cpu"] = {"s_per_step": c2["sample_s_per_step"], "lean_s_per_step": c2["lean_sample_s_per_step"],
                                     "cores": c2["cores"], "kind": c2["kind"], "sample": c2["sample"],
                                     "nlz_step0_params": c2["nlz_sample"]}
            out["config2"]["nlz_rel_diff_gpu_cpu"] = abs(out["config2"]["nlz_step0_params"] - c2["nlz_sample"]) / abs(c2["nlz_sample"])
            out["config2"]["gpu_over_cpu"] = c2["sample_s_per_step"] / (out["config2"]["ms_per_step"] * 1e-3)
    return out


def run_inproc(args):
    """ONE process driving all N GPUs (gpak_create_multi: one host thread per device, RCCL communicators made by one
    ncclCommInitAll, or the in-process peer-copy transport).  Same step, same timing contract."""
    from gp_ss_ak_amd import gpak, synth
    P = args.gpus
    devs = [int(v) for v in os.environ["GPAK_MULTI_DEVICES"].split(",")] if os.environ.get("GPAK_MULTI_DEVICES") else list(range(P))
    if len(devs) != P:
        raise SystemExit(f"GPAK_MULTI_DEVICES names {len(devs)} devices, --gpus is {P}")
    g = gpak.Gpak(devices=devs)
    mode = gpak.DIST_DIRECT if args.dist == "direct" else gpak.DIST_EXPANSION

    def one_size(N, steps, warmup):
        X, y = synth.drillholes(N)
        g.set_train(X, y)
        wall, nlz, _, _ = _timed_steps(g, mode, steps, warmup)
        per_rank = []
        for r in range(P):
            st = g.rank_stats(r)
            per_rank.append({k: round(v, 3) for k, v in st.items()
                             if k.endswith("_ms") or k in ("bulk_flops", "bytes_broadcast", "rank")})
        Np = g.timing()["n_padded"]
        flops = Np ** 3 / 3.0
        per_step = wall / steps
        bulk_tf = [p["bulk_flops"] / (p["bulk_ms"] * 1e-3) / 1e12 if p["bulk_ms"] > 0 else None for p in per_rank]
        return {"N": N, "steps_per_s": steps / wall, "ms_per_step": per_step * 1e3, "nlz": nlz,
                "whole_step_tflops_per_gpu": flops / per_step / 1e12 / P,
                "whole_step_frac_of_mfma_peak": flops / per_step / 1e12 / P / PEAK_F64_MFMA_TFLOPS,
                "bulk_update_tflops_per_rank": bulk_tf, "bytes_broadcast_per_step": per_rank[0]["bytes_broadcast"],
                "phases_ms_per_rank_last_step": per_rank, "stream_flags": g.rank_stats(0)["flags"]}

    main = one_size(args.n, args.steps, args.warmup)
    transport = g.transport()
    tfs = [v for v in main["bulk_update_tflops_per_rank"] if v]
    out = {
        "metric": f"GP train step/sec (Gram+Cholesky+logML) at N={args.n} fp64",
        "value": main["steps_per_s"], "unit": "steps/s", "n_gpus": P, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"N={args.n} fp64 ExpAns+Bias Gram + block-column-cyclic Cholesky + solves + logML, C++ "
                               f"schedule (csrc/dist.hip) driven by ONE process with a host thread per GPU "
                               f"(gpak_create_multi), sub-panel broadcast over {transport}",
                   "N": args.n, "dist_mode": args.dist, "nb_outer": 512, "parallelism": f"block-column-cyclic x{P}",
                   "host": "one process, one thread per GPU", "transport": transport, "devices": devs},
        "nlz": main["nlz"],
        "roofline": {"kernel": "gpak_gemm_nt_f64_rs<4,2,true> (bulk trailing update of the owned block columns)",
                     "bound": "mfma", "achieved": min(tfs) if tfs else None, "peak": PEAK_F64_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": (min(tfs) / PEAK_F64_MFMA_TFLOPS) if tfs else None, "traffic": None,
                     "traffic_note": "no PMC pass exists for a multi-GPU run (the builder's boxes have one GPU)",
                     "note": "slowest rank's bulk launches of the LAST step: algorithmic flops / summed launch durations (hipEvents)"},
        "bytes_broadcast_per_step": main["bytes_broadcast_per_step"],
        "phases_ms_per_rank": main["phases_ms_per_rank_last_step"],
        "whole_step_frac_of_mfma_peak_per_gpu": main["whole_step_frac_of_mfma_peak"],
        "stream_flags": main["stream_flags"],
    }
    if args.n != 65536 and not args.no_n65536:
        out["n65536"] = one_size(65536, max(1, min(args.steps, 3)), 1)
    g.close()
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_child(cmd, env, limit_s):
    """Runs one child (its own process group) to completion or to `limit_s`; returns (rc or None on time-out, last JSON
    object printed on stdout or None, tail of stderr)."""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, cwd=ROOT)
    try:
        so, se = p.communicate(timeout=limit_s)
        rc = p.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)     # exactly the group this call started
        except ProcessLookupError:
            pass
        so, se = p.communicate()
        rc = None
    line = None
    for ln in so.decode(errors="replace").splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                line = json.loads(ln)
            except ValueError:
                pass
    return rc, line, se.decode(errors="replace")[-1500:]


def launch_multi(args, argv):
    """`python3 bench.py --gpus N` typed as is (no launcher, WORLD_SIZE unset): this process has not touched a GPU; it
    starts the run as a CHILD and relays the child's JSON line.  Order: (1) one process per GPU under
    torch.distributed.run (the contract's shape: RCCL inside the library, gloo control plane); (2) if that fails or
    does not finish: ONE process with a host thread per GPU (gpak_create_multi).  Whatever ran is named in the line
    (`launcher`, `config.host`, `config.transport`); attempts that failed are listed with their reason."""
    N = args.gpus
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    limit = float(os.environ.get("GPAK_BENCH_CHILD_TIMEOUT_S", "300"))
    attempts = []
    order = [m for m in os.environ.get("GPAK_BENCH_MULTI_ORDER", "torchrun,inproc").split(",") if m]
    t_all = time.perf_counter()
    for how in order:
        if how == "torchrun":
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={N}", "--master-addr",
                   "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + argv + ["--no-cpu"]
        else:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + argv + ["--inproc", "--no-cpu"]
        t0 = time.perf_counter()
        rc, line, err = _run_child(cmd, env, limit)
        took = time.perf_counter() - t0
        if rc == 0 and line is not None and line.get("value"):
            line["launcher"] = {"how": ("bench.py started `python -m torch.distributed.run --nproc-per-node N bench.py` as a child"
                                        if how == "torchrun" else "bench.py started `bench.py --inproc` (gpak_create_multi) as a child"),
                                "child_wall_s": took, "failed_attempts": attempts}
            if not args.no_cpu:
                line["cpu_baseline"] = cpu_baseline(args.n, args.cpu_n)
            return line
        attempts.append({"how": how, "rc": rc, "timed_out": rc is None, "wall_s": took, "stderr_tail": err[-600:]})
    raise SystemExit("bench.py --gpus %d: every multi-GPU start failed after %.0f s: %s" %
                     (N, time.perf_counter() - t_all, json.dumps(attempts)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    # --size rather than only --n: torch.distributed.run abbreviation-matches "--n" against its own options
    ap.add_argument("--n", "--size", dest="n", type=int, default=32768)
    ap.add_argument("--dist", choices=["direct", "expansion"], default="direct")
    ap.add_argument("--nb-outer", type=int, default=0)
    ap.add_argument("--cpu-n", type=int, default=16384, help="sample size of the CPU baseline (~15 s of CPU work)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--grad", type=int, default=0, help="also time this many GradLL evaluations (config 3)")
    ap.add_argument("--calibrate", action="store_true", default=True)
    ap.add_argument("--no-n65536", action="store_true", help="skip the N=65536 sub-run (north_star's scaling size)")
    ap.add_argument("--vendor", type=int, default=1, help="1: time torch.linalg.cholesky (hipSOLVER / rocSOLVER) on the same matrix as a "
                    "reference point (N=32768 and the config-2 size)")
    ap.add_argument("--config2", type=int, default=8192, help="size of the configs[1] sub-run (GPU step and the CPU "
                                                               "reference sequence at full size; 0 = skip)")
    ap.add_argument("--config3", type=int, default=3, help="L-BFGS iterations of the configs[2] sub-run through the CLI "
                                                            "(N=32768 only; 0 = skip)")
    ap.add_argument("--config5", type=int, default=1000000, help="test points of the configs[4] sub-run (N=32768 fp32 "
                                                                  "prediction; 0 = skip)")
    ap.add_argument("--grid", default="auto", help="N > 1: also time the row-block x column-block layouts: PrxPc, 'auto' "
                                                    "(every Pr in 2, 4, 8 dividing N) or 'none'")
    ap.add_argument("--inproc", action="store_true", help="N > 1: one process, one host thread per GPU (gpak_create_multi)")
    ap.add_argument("--profile-tag", default=os.environ.get("GPAK_PROFILE_TAG", "r03"),
                    help="profiles/<tag>_pmc_summary_N<N>.json supplies `traffic` and the in-situ clock")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.inproc and args.gpus > 1:
        print(json.dumps(run_inproc(args)))
        return
    if world > 1 or os.environ.get("GPAK_FORCE_DIST"):
        # one rank per process under torch.distributed.run: the C++ schedule of csrc/dist.hip over RCCL
        from gp_ss_ak_amd import dist as gdist
        out = gdist.bench(args)
        if out is not None:
            if not args.no_cpu:
                out["cpu_baseline"] = cpu_baseline(args.n, args.cpu_n)
            print(json.dumps(out))
        return
    if args.gpus > 1:
        print(json.dumps(launch_multi(args, sys.argv[1:])))
        return
    out = run_single(args)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

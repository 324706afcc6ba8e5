#!/bin/bash
# Round 3, VERDICT item 6: HBM-side traffic, launch time and in-situ clock of the bulk trailing update for other
# super-tile shapes (GPAK_SUPER_LR: 2 = 4 x 16 tiles per XCD, 3 = 8 x 8 (default), 4 = 16 x 4).  Separate --pmc passes
# as MI355X_MICROARCH.md prescribes.  Output: gpurun_out/supertile/ + one summary text.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/supertile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu --no-n65536 --config3 0 --config2 0 --config5 0"
for LR in 3 2 4; do
  export GPAK_SUPER_LR=$LR
  python3 $ROOT/bench.py --steps 10 --warmup 3 $ARGS > $OUT/bench_lr$LR.json 2> $OUT/bench_lr$LR.err
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_lr$LR -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> $OUT/fetch_lr$LR.log
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/write_lr$LR -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> $OUT/write_lr$LR.log
  rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq_lr$LR -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> $OUT/sq_lr$LR.log
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_lr$LR.json FETCH_SIZE=$OUT/fetch_lr$LR WRITE_SIZE=$OUT/write_lr$LR SQ=$OUT/sq_lr$LR
  echo "lr=$LR done"
done
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import json
out = []
for lr in (3, 2, 4):
    b = json.load(open("$OUT/bench_lr%d.json" % lr))
    z = json.load(open("$OUT/pmc_lr%d.json" % lr))
    def pick(sec):
        for k, v in z[sec].items():
            if k.startswith("void gpak_gemm_nt_f64_rs<4, 2, true"):
                return v
    f, w, sq = pick("FETCH_SIZE")["FETCH_SIZE"], pick("WRITE_SIZE")["WRITE_SIZE"], pick("SQ")
    fetch = f["sum"] / f["dispatches"] * 1024 * 2      # KiB, doubled for 16-B-per-lane reads (the guide's correction)
    write = w["sum"] / w["dispatches"] * 1024
    r = b["roofline"]
    busy = sq["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / sq["SQ_VALU_MFMA_BUSY_CYCLES"]["dispatches"]
    gui = sq["GRBM_GUI_ACTIVE"]["sum"] / sq["GRBM_GUI_ACTIVE"]["dispatches"]
    out.append("super-tile 2^%d x 2^%d tiles: step %.2f ms, factor %.2f ms, bulk launches avg %.3f ms = %.2f TFLOP/s (frac %.3f); "
               "per launch: 2*FETCH %.2f GB + WRITE %.2f GB = %.2f GB against %.2f GB algorithmic C traffic; MFMA busy %.3f of SIMD cycles "
               "(under the profiler)" % (lr, 6 - lr, b["ms_per_step"], b["phases_ms_per_step"]["factor_ms"], r["avg_launch_ms"],
                                         r["achieved"], r["frac"], fetch / 1e9, write / 1e9, (fetch + write) / 1e9,
                                         r["algorithmic_bytes_per_launch"] / 1e9, busy / 1024 / (gui / 8)))
open("$OUT/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY

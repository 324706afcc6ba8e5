#!/bin/bash
# Counters of the fp32 prediction's dominant kernel (gpak_gemm_nt_f32_rsw) at M = 65536: bash tools/predict_pmc.sh (gpurun)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/predict_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/sq -o p -- python3 $ROOT/tools/time_predict.py pmc 65536 > $OUT/sq.log 2>&1
echo "SQ done"
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/tcc -o p -- python3 $ROOT/tools/time_predict.py pmc 65536 > $OUT/tcc.log 2>&1
echo "TCC done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p -- python3 $ROOT/tools/time_predict.py pmc 65536 > $OUT/fetch.log 2>&1
echo "FETCH done"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq", "tcc", "fetch"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "f32" in k:
            print(sub, k, {c: f"{v:.4g}" for c, v in d.items()}, "dispatches", max(n[(k, c)] for c in d))
PY
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete

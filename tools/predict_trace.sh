#!/bin/bash
# Kernel statistics of the prediction path (configs[4]) at M = 131072: bash tools/predict_trace.sh   (through gpurun)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/predict_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 $ROOT/tools/time_predict.py trace 131072 > $OUT/run.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -16 $OUT/kernel_stats.csv | cut -c1-170
tail -25 $OUT/run.log

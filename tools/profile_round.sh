#!/bin/bash
# Profiles of the default bench command on the GPU box (run through gpurun from the repo root):
#   kernel-trace statistics, then separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ), as
#   /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Output: gpurun_out/prof_$TAG/ + summary JSON.
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu --no-n65536 --config3 0 --config2 0 --config5 0 --vendor 0 > $OUT/bench_kt.json 2> $OUT/kt.log
echo "kernel-trace done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-n65536 --config3 0 --config2 0 --config5 0 --vendor 0 > $OUT/bench_fetch.json 2> $OUT/fetch.log
echo "FETCH_SIZE done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-n65536 --config3 0 --config2 0 --config5 0 --vendor 0 > $OUT/bench_write.json 2> $OUT/write.log
echo "WRITE_SIZE done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-n65536 --config3 0 --config2 0 --config5 0 --vendor 0 > $OUT/bench_sq.json 2> $OUT/sq.log
echo "SQ done"
python3 $ROOT/tools/pmc_summary.py $OUT/pmc_summary.json FETCH_SIZE=$OUT/pmc_fetch WRITE_SIZE=$OUT/pmc_write SQ=$OUT/pmc_sq
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the raw per-dispatch CSVs are large; keep the summaries
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT

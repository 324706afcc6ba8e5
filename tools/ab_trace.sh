#!/bin/bash
# per-kernel statistics of the N = 8192 step under two builds of the library: bash tools/ab_trace.sh libA.so libB.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in $1 $2; do
  cp $ROOT/$L $ROOT/gp_ss_ak_amd/libgpak_hip.so
  OUT=$ROOT/gpurun_out/ab_trace/$(basename $L .so)
  mkdir -p $OUT
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT -o kt -- python3 $ROOT/tools/time_sizes.py 8192 > $OUT/run.log 2>&1
  echo "== $L"; grep "^N=" $OUT/run.log
  find $OUT -name "*kernel_stats.csv" -exec head -12 {} \; | cut -c1-60,150-260
  find $OUT -name "*kernel_trace.csv" -delete
done

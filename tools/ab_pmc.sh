#!/bin/bash
# L2 hit / miss and fabric fetch counters of the trailing-update kernel alone under two builds: bash tools/ab_pmc.sh A.so B.so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  cp $ROOT/$L $ROOT/gp_ss_ak_amd/libgpak_hip.so
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    OUT=$ROOT/gpurun_out/ab_pmc/$(basename $L .so)_$(echo $set | cut -c1-8)
    rm -rf $OUT; mkdir -p $OUT
    GEMM_SIZES=32768 rocprofv3 --output-format csv --kernel-trace --pmc $set -d $OUT -o p -- python3 $ROOT/tools/time_gemm.py random > $OUT/run.log 2>&1
    python3 - $OUT "$L" <<'PY'
import csv, glob, sys, collections
out, lib = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_f64_rs<" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
# dispatches come in groups of 4 per K (1 warm-up + 3 timed): print the mean per group
ids = sorted(rows)
for g in range(0, len(ids), 4):
    grp = ids[g:g + 4]
    agg = collections.defaultdict(float)
    for i in grp:
        for c, v in rows[i].items():
            agg[c] += sum(v) / len(grp)
    print(lib, "K-group", g // 4, {c: f"{v:.4g}" for c, v in agg.items()})
PY
    find $OUT -name "*.csv" -size +4M -delete
  done
done

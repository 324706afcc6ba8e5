#!/bin/bash
# Kernel trace of a few steps at a small N, where the panel chain is the critical path: which kernels make up the
# chain and how long each one is.  bash tools/chain_trace.sh [N]   (through gpurun, from the repo root)
set -e
N=${1:-8192}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/chain_$N
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/kt -o kt -- python3 $ROOT/tools/time_sizes.py $N > $OUT/run.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/kt -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: everything after the last gram fill
idx = max(i for i, r in enumerate(rows) if "fill" in r["Kernel_Name"])
last = rows[idx:]
t0 = int(last[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in last)
print("last step: %.3f ms, %d launches" % ((t1 - t0) / 1e6, len(last)))
agg = collections.OrderedDict()
for r in last:
    k = r["Kernel_Name"][:70]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%7.1f us total %5d x %7.2f us  %s" % (us, n, us / n, k))
# timeline of one 512-wide outer panel in the middle
mid = [r for r in last if "potrf128" in r["Kernel_Name"]]
a = int(mid[len(mid) // 2]["Start_Timestamp"]); b = int(mid[len(mid) // 2 + 4]["Start_Timestamp"])
print("--- timeline of four 128-column steps in the middle (start us, dur us, stream/queue, kernel)")
for r in last:
    s = int(r["Start_Timestamp"])
    if a <= s < b:
        print("%8.1f %7.1f  q%s  %s" % ((s - a) / 1e3, (int(r["End_Timestamp"]) - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
PY

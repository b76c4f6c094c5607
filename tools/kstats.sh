#!/bin/bash
# per-kernel average durations of the fused step under rocprofv3 (run on the GPU box):  tools/kstats.sh <tag> [env assignments for trace_step.py]
# -> gpurun_out/kstats_<tag>.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kstats_$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
case "${QLIB:-}" in ""|/*) ;; *) export QLIB=$ROOT/$QLIB;; esac
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/trace_step.py > $OUT/run.log 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$F" > $ROOT/gpurun_out/kstats_$TAG.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r["Name"]
    short = name[:110]
    print(f'{float(r["AverageNs"]) / 1e3:9.2f} us avg  {float(r["MinNs"]) / 1e3:9.2f} min  x{r["Calls"]:>6}  {short}')
PY
find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
cat $ROOT/gpurun_out/kstats_$TAG.txt

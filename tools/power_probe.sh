#!/bin/bash
# Sample socket power and clocks (rocm-smi, read-only) while bench.py loops on the hot path.
#   tools/power_probe.sh [bench.py flags...]   -> gpurun_out/power_probe.log
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
LOG=$OUT/power_probe.log
: > $LOG
echo "== idle ==" >> $LOG
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" >> $LOG
python3 $ROOT/bench.py --steps 40000 --warmup 50 --no-extras --no-cpu-baseline --no-live-traffic "$@" > $OUT/power_probe_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  echo "== under load, sample $i ==" >> $LOG
  rocm-smi --showpower --showclocks 2>&1 | grep -i -E "power|sclk|mclk|fclk" >> $LOG
  sleep 2
done
wait $BP
echo "== bench line ==" >> $LOG
cat $OUT/power_probe_bench.json >> $LOG

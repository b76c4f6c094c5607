#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root).
#   tools/profile_bench.sh <tag> [bench.py flags...]   -> gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,pmc_sq,pmc_sq2}/...
# Counters are collected in their own passes (never mixed with sys/hip traces), as the pool requires.
set -u
TAG=${1:-run}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# the kernel sources the counters belong to (tools/summarize_profile.py stamps traffic.json with it; bench.py csrc_sha16 recomputes it)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.csrc_sha16())" > $OUT/csrc_sha16.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
STEPS=${PROF_STEPS:-20}
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-extras --no-live-traffic $*"
# PROF_SCRIPT=1: profile tools/trace_step.py instead (MODE / SHAPE / CAUSAL / PREC from the environment: the token-wise and 16-bit
# paths, which bench.py does not drive); its STEPS stand in for --steps
if [ "${PROF_SCRIPT:-0}" = "1" ]; then
  export STEPS=${PROF_STEPS:-20}
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/trace_step.py > $OUT/stats.log 2>&1
  export STEPS=3
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/trace_step.py > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/trace_step.py > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/trace_step.py > $OUT/pmc_sq.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/tools/trace_step.py > $OUT/pmc_sq2.log 2>&1
  find $OUT -name "*.db" -delete 2>/dev/null; find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
  du -sh $OUT; exit 0
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH --steps $STEPS --warmup 5 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/pmc_sq2 -- $BENCH --steps 2 --warmup 1 > $OUT/pmc_sq2.log 2>&1
grep -h "^{" $OUT/stats.log > $OUT/bench_under_profiler.json 2>/dev/null
# drop the big per-dispatch traces we do not need (keep stats + counter collections)
find $OUT -name "*.db" -delete 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $OUT

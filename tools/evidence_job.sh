#!/bin/bash
# The job that produced a round's final evidence (run on the GPU box: `cp tools/evidence_job.sh tools/_job.sh; bash tools/gpurun_retry.sh <log> 5400`):
# GPU suite, smoke, bench line with the driver's arguments, rocprofv3 + PMC for C2 / C3 / C5, randomised sweeps.  Afterwards, here:
#   R=r06; for c in "c2 4 32 4096 128" "c3 4 32 4096 128 1" "c5 4 40 16384 128 1"; do set -- $c; n=$1; shift; python tools/summarize_profile.py gpurun_out/prof_${R}_$n profiles/${R}_$n "$@"; done
#   cp profiles/${R}_c2/traffic.json profiles/traffic.json; python tools/pmc_table.py profiles/${R}_c2 profiles/${R}_c3 profiles/${R}_c5 > profiles/$R/pmc_table_c2_c3_c5.md
R=${ROUND_TAG:-r06}
mkdir -p gpurun_out/$R
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/$R/pytest_gpu.log 2>&1
tail -3 gpurun_out/$R/pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/$R/smoke.log 2>&1
tail -1 gpurun_out/$R/smoke.log
python bench.py --steps 20 --warmup 5 > gpurun_out/$R/bench_steps20_warmup5.json 2> gpurun_out/$R/bench_steps20_warmup5.err
rm -rf gpurun_out/prof_${R}_c2 gpurun_out/prof_${R}_c3 gpurun_out/prof_${R}_c5
bash tools/profile_bench.sh ${R}_c2 > gpurun_out/$R/prof_c2.log 2>&1
bash tools/profile_bench.sh ${R}_c3 --causal > gpurun_out/$R/prof_c3.log 2>&1
PROF_STEPS=5 bash tools/profile_bench.sh ${R}_c5 --batch 4 --heads 40 --seq 16384 --causal --fp8 e5m2 > gpurun_out/$R/prof_c5.log 2>&1
timeout 1200 python tools/fuzz_launch.py 300 171 > gpurun_out/$R/fuzz_launch_seed171_n300.log 2>&1; tail -1 gpurun_out/$R/fuzz_launch_seed171_n300.log
timeout 1500 python tools/fuzz_parity.py 300 172 adv > gpurun_out/$R/fuzz_parity_adv_seed172_n300.log 2>&1; grep "cases," gpurun_out/$R/fuzz_parity_adv_seed172_n300.log
timeout 1500 python tools/fuzz_parity.py 300 173 > gpurun_out/$R/fuzz_parity_seed173_n300.log 2>&1; grep "cases," gpurun_out/$R/fuzz_parity_seed173_n300.log
timeout 600 python tools/soak.py 6 > gpurun_out/$R/soak_final.log 2>&1; tail -2 gpurun_out/$R/soak_final.log

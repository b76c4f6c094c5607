mkdir -p gpurun_out/r03r
P=tools/bin/libqattn_prev.so
N=quantumattention_amd/libqattn_hip.so
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r03r/pytest.log
python tools/ab.py new=$N prev=$P --paths fused,attn --rounds 9 > gpurun_out/r03r/ab_c2.log 2>&1
python tools/ab.py new=$N prev=$P --causal --paths fused,attn --rounds 9 > gpurun_out/r03r/ab_c3.log 2>&1
python tools/ab.py new=$N prev=$P --causal --paths fused --shape 4,40,16384,128 --fp8 e5m2 --rounds 3 --calls 5 > gpurun_out/r03r/ab_c5.log 2>&1
python tools/ab.py new=$N prev=$P --paths attn --shape 16,16,8192,128 --rounds 3 --calls 5 > gpurun_out/r03r/ab_big.log 2>&1
tail -n 9 gpurun_out/r03r/*.log

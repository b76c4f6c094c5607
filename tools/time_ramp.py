"""Development: ms/step of the fused step in consecutive windows from a cold start (DPM ramp / power-cap settling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import quantumattention_amd as qa
B, H, S, D = 4, 32, 4096, 128
prec = os.environ.get("PREC", "fast")
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
win = int(os.environ.get("WIN", "100"))
total = float(os.environ.get("SECS", "20"))
out = []
with qa.config.patch({"attention.precision": prec}):
    qa.fp8_attn_func(q, k, v); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < total:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(win): qa.fp8_attn_func(q, k, v)
        e1.record(); torch.cuda.synchronize()
        out.append((time.time() - t0, e0.elapsed_time(e1) / win))
idx = [0, 1, 2, 3, 5, 8, 12, 20, 30, 50, 80, 120, 160, 200, 250, 300, len(out) - 1]
print("prec", prec, "windows of", win, "steps:", " ".join("%.1fs:%.4f" % out[i] for i in idx if i < len(out)))

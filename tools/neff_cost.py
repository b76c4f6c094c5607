"""What the effective-key-count statistic costs: the C2 attention launch with precision auto, with and without the N_eff
MFMA (dev library, QATTN_ABL_NONEFF=1: the checked kernel without that product -- results not graded), and fast; time per launch, socket
power while the launch alone keeps the queue full (rocm-smi, read-only), joules per launch."""
import os, re, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native
_native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_libs", "libqattn_dev.so")

def power_w():
    out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True).stdout
    m = re.search(r"Power \(W\):\s*([0-9.]+)", out)
    return float(m.group(1)) if m else float("nan")

B, H, S, D = 4, 32, 4096, 128
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
def launch(prec): return _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False, precision=prec)
rows = []
for rnd in range(3):
    for name, prec, env in (("auto", "auto", {}), ("auto without the N_eff product", "auto", {"QATTN_ABL_NONEFF": "1"}), ("fast", "fast", {})):
        for k_, v_ in env.items(): os.environ[k_] = v_
        for _ in range(200): launch(prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, watts = 0, []
        e0.record()
        t_end = time.time() + 4.0
        while time.time() < t_end:
            for _ in range(400): launch(prec)
            n += 400
            if len(watts) < 4: watts.append(power_w())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        w = sum(watts[1:]) / max(1, len(watts) - 1)
        rows.append((rnd, name, ms, w, ms * 1e-3 * w))
        for k_ in env: os.environ.pop(k_)
for rnd, name, ms, w, j in rows:
    print(f"round {rnd}  {name:32s} {ms:.4f} ms/launch  {w:6.0f} W  {j:.3f} J/launch")

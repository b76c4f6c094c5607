"""Experiment: overlap the HBM-bound quant pre-pass of batch slice b+1 with the MFMA-bound attention of slice b
(two streams inside ONE fp8 attention call).  Development aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quantumattention_amd import _native

B, H, S, D = 4, 32, 4096, 128
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4   # groups along the batch dimension
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def plain():
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
    return _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False)


def piped():
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur)
    sb.wait_stream(cur)
    outs = []
    n = B // G
    for g in range(G):
        sl = slice(g * n, (g + 1) * n)
        with torch.cuda.stream(sa):
            pack = _native.quant_qkv_fp8(q[sl], k[sl], v[sl])
            ev = torch.cuda.Event()
            ev.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(ev)
            q8, kf, vf, sq, sk, sv = pack
            outs.append(_native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=False))
    cur.wait_stream(sb)
    return outs


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = plain()
got = torch.cat(piped(), 0)
torch.cuda.synchronize()
print("equal:", torch.equal(ref, got))
print("plain  %.4f ms" % timeit(plain))
print("piped  %.4f ms (G=%d)" % (timeit(piped), G))

#!/usr/bin/env python3
"""Generate golden input/output vectors from the *reference* Python package.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU box).
The reference is imported read-only from /root/reference/src with `quantum_attn.inductor`
stubbed out (its Inductor lowering does not import on torch 2.10); everything
stored here comes from the reference's own functions:

  * quantum_attn.nn._dynamically_quantize_fp8          (eager numerics,    nn.py:14-19)
  * quantum_attn.dynamically_quantize_fp8              (compiled numerics, nn.py:22-42)
  * quantum_attn.ops._fp8_attention_forward            (semantic definition of the op, ops.py:64-95)
  * quantum_attn.ops._attention_forward                (16-bit sibling op, ops.py:17-29)
  * quantum_attn.*_with_fallback on CPU tensors        (interface.py:62-98,134-176,209-248)

Besides the reference's own outputs the files hold two fp64-evaluated oracles on the SAME quantised inputs (SURVEY.md
section 8c), computed with torch.float64 ops on the reference's q8 / k8 / scales:
  o2_*  = softmax((sq q8)(sk k8)^T / sqrt(D) [+ causal mask]) v            (v = the 16-bit tensor, exact in fp64)
  o3_*  = the same with v replaced by sv * v8, (v8, sv) = quantum_attn.dynamically_quantize_fp8(v, [2,3])
and, for the e5m2 format the reference does not have, the torch restatement of its compiled quantiser numerics with
torch.float8_e5m2 (`*_e5m2` keys: scale = amax * (1/57344) in fp32, payload = e5m2(clamp(round_to_input_dtype(fp32(t) / scale)))).

Outputs are data only (npz with raw bit patterns / fp32 arrays); no reference source is copied.
Usage:  python tests/golden/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    sys.path.insert(0, REF)
    sys.modules["quantum_attn.inductor"] = types.ModuleType("quantum_attn.inductor")
    import quantum_attn  # noqa: F401

    return quantum_attn


def bits16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def bits8(t):
    return t.contiguous().view(torch.uint8).numpy().copy()


def sdpa64(q8, k8, v64, sq, sk, causal, token):
    """fp64 SDPA on de-quantised q8, k8 (scales applied in fp64): the definition ops.py:64-95 evaluates in 16 bit."""
    q = q8.to(torch.float64) * (sq.double()[..., None] if token else sq.double()[..., None, None])
    k = k8.to(torch.float64) * (sk.double()[..., None] if token else sk.double()[..., None, None])
    s = q @ k.transpose(-1, -2) / (q.shape[-1] ** 0.5)
    if causal:
        s = s.masked_fill(~torch.ones(s.shape[-2:], dtype=torch.bool).tril(), float("-inf"))
    return (torch.softmax(s, -1) @ v64).to(torch.float32).numpy()


def quant_e5m2_compiled(t, rdim):
    """torch restatement of the compiled quantiser numerics (nn.py:14-19 as Inductor evaluates it) for float8_e5m2."""
    qmax = 57344.0
    scale = t.float().abs().amax(rdim, keepdim=True).mul(1.0 / qmax).clamp_min(torch.finfo(torch.float32).eps)
    payload = (t.float() / scale).to(t.dtype).clamp(-qmax, qmax).to(torch.float8_e5m2)
    return payload, scale.squeeze(rdim)


def main():
    qa = import_reference()
    from quantum_attn import nn as qnn, ops as qops

    torch.set_num_threads(4)
    cases = [
        # name, B, H, Sq, Skv, D, dtype, seed, with_eager
        ("c1_b1h2s128d64_bf16_s0", 1, 2, 128, 128, 64, torch.bfloat16, 0, True),
        ("c1_b1h2s128d64_bf16_s1", 1, 2, 128, 128, 64, torch.bfloat16, 1, False),
        ("c1_b1h2s128d64_fp16_s0", 1, 2, 128, 128, 64, torch.float16, 0, True),
        ("b1h2s256d128_bf16_s0", 1, 2, 256, 256, 128, torch.bfloat16, 0, False),
        ("ragged_b1h2s200d128_bf16_s1", 1, 2, 200, 200, 128, torch.bfloat16, 1, False),
        ("cross_b1h2sq96skv160d128_bf16_s2", 1, 2, 96, 160, 128, torch.bfloat16, 2, False),
    ]
    for name, B, H, Sq, Skv, D, dtype, seed, with_eager in cases:
        torch.manual_seed(seed)
        # same construction as the reference's tests (tests/test_interface.py:40-43)
        q = torch.randn(B, H, Sq, D, dtype=dtype)
        k = torch.randn(B, H, Skv, D, dtype=dtype)
        v = torch.randn(B, H, Skv, D, dtype=dtype)
        out = {
            "meta": np.array([B, H, Sq, Skv, D, 1 if dtype == torch.bfloat16 else 0, seed], dtype=np.int64),
            "q": bits16(q), "k": bits16(k), "v": bits16(v),
        }
        for method, rdim in (("head", [2, 3]), ("token", 3)):
            # compiled numerics == what the reference's GPU path runs (nn.py:521-539)
            q8, sq = qa.dynamically_quantize_fp8(q, reduction_dim=rdim)
            k8, sk = qa.dynamically_quantize_fp8(k, reduction_dim=rdim)
            out[f"q8_{method}_compiled"] = bits8(q8)
            out[f"k8_{method}_compiled"] = bits8(k8)
            out[f"sq_{method}_compiled"] = sq.numpy().copy()
            out[f"sk_{method}_compiled"] = sk.numpy().copy()
            if with_eager:
                q8e, sqe = qnn._dynamically_quantize_fp8(q, reduction_dim=rdim)
                k8e, ske = qnn._dynamically_quantize_fp8(k, reduction_dim=rdim)
                out[f"q8_{method}_eager"] = bits8(q8e)
                out[f"k8_{method}_eager"] = bits8(k8e)
                out[f"sq_{method}_eager"] = sqe.numpy().copy()
                out[f"sk_{method}_eager"] = ske.numpy().copy()
            if method == "head":
                v8, sv = qa.dynamically_quantize_fp8(v, reduction_dim=[2, 3])   # the build also quantises V, head-wise
                out["v8_head_compiled"] = bits8(v8)
                out["sv_head_compiled"] = sv.numpy().copy()
                v3 = v8.to(torch.float64) * sv.double()[..., None, None]
            q5, sq5 = quant_e5m2_compiled(q, rdim)
            k5, sk5 = quant_e5m2_compiled(k, rdim)
            out[f"q8_{method}_e5m2"], out[f"sq_{method}_e5m2"] = bits8(q5), sq5.numpy().copy()
            out[f"k8_{method}_e5m2"], out[f"sk_{method}_e5m2"] = bits8(k5), sk5.numpy().copy()
            for causal in (False, True):
                if causal and Sq != Skv:
                    continue  # reference tests skip this (tests/test_interface.py:32-33)
                tag = "causal" if causal else "full"
                o1 = qops._fp8_attention_forward(q8, k8, v, sq, sk, is_causal=causal)
                out[f"o1_{method}_{tag}"] = bits16(o1)
                # fp64 oracles, kept small: every ROW_STEP-th query row (all of them for the short shapes), token-wise only there
                step = 1 if Sq <= 128 else 4
                if method == "head" or step == 1:
                    out[f"o2_{method}_{tag}"] = sdpa64(q8, k8, v.double(), sq, sk, causal, method == "token")[:, :, ::step].copy()
                if method == "head":
                    out[f"o3_{method}_{tag}"] = sdpa64(q8, k8, v3, sq, sk, causal, False)[:, :, ::step].copy()
                out["o23_row_step"] = np.array([step], dtype=np.int64)
        for causal in (False, True):
            if causal and Sq != Skv:
                continue
            o16 = qops._attention_forward(q, k, v, is_causal=causal)
            out[f"o16_{'causal' if causal else 'full'}"] = bits16(o16)
        # CPU plumbing (BASELINE config 1): *_with_fallback degenerate to F.sdpa on CPU tensors
        fb = qa.attn_func_with_fallback(q, k, v)
        fb8 = qa.fp8_attn_func_with_fallback(q, k, v)
        fbt = qa.fp8_token_wise_attn_func_with_fallback(q, k, v)
        assert torch.equal(fb, fb8) and torch.equal(fb, fbt)
        out["fallback_full"] = bits16(fb)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, {k_: v_.shape for k_, v_ in out.items() if k_ != "meta"})

    # can_use_attention reason strings on CPU (interface contract, nn.py:209-211,282-307)
    q = torch.randn(1, 2, 128, 64, dtype=torch.bfloat16)
    ok, reason = qnn.can_use_attention(q, q, q)
    with open(os.path.join(HERE, "can_use_attention_cpu.txt"), "w") as f:
        f.write(f"{ok}\n{reason}\n")
    print(ok, reason)


if __name__ == "__main__":
    main()

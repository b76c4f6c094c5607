"""-m gpu: the long end of the size range -- the rows of include/qattn.h's PATH TABLE for Skv > 16384 (more than 256 chunks of 64
keys per head: the fused entry keeps ONE fp32 V scale per head there instead of one power-of-two scale per chunk), sequences up to
131072, and tensors whose byte offsets do not fit 32 bits (element index > 2^31, 16-bit tensors > 2^32 bytes, fp8 images > 2^31
bytes).  The reference itself has no upper size (its launcher takes any N: tk/attention.py:355-437) and tests up to S = 8192
(tests/test_interface.py:62-102); BASELINE's largest configuration is S = 16384.

Graded like every fused call (tests/gpu_utils.py): per row against THE oracle of the path the kernel reports, bound per element.  The
oracle runs on row bands (first rows, a mid-sequence band, the last rows) so that it finishes in seconds."""
import numpy as np
import pytest
import torch

import oracle
import quantumattention_amd as qa
from quantumattention_amd import _native
from tests.gpu_utils import (FMT, PATH_ONE_TERM, PATH_V16, TDT, assert_within_bound, bits16, check_path_structure, fmt16, fused_call,
                             fused_step_uses_block_v, oracle_for_fp8_path, out_to_f32)

pytestmark = pytest.mark.gpu


def _bands(Sq, n=128):
    """first rows, a band in the middle that straddles a 256-row block boundary, the last rows (ragged tail included)"""
    mid = (Sq // 2) // 256 * 256 - n // 2
    out = [(0, min(Sq, 2 * n))]
    if Sq > 6 * n:
        out += [(mid, mid + n), (Sq - n, Sq)]
    return out


LONG = [
    # Hq, Hkv, Sq, Skv, D, causal, fp8, scaling, dtype
    (2, 2, 16448, 16448, 128, False, "e4m3", "head-wise", torch.bfloat16),   # 257 chunks: the first length with a per-head V scale
    (2, 1, 16384, 16384, 128, True, "e4m3", "head-wise", torch.bfloat16),    # 256 chunks: the last length with block-scaled V (GQA)
    (2, 2, 20000, 20000, 128, True, "e4m3", "head-wise", torch.bfloat16),    # ragged, causal, hand-scheduled kernel
    (1, 1, 32768, 32768, 128, False, "e5m2", "head-wise", torch.float16),
    (2, 1, 32768, 32768, 64, True, "e4m3", "head-wise", torch.bfloat16),     # templated kernel, GQA
    (1, 1, 16500, 16500, 256, False, "e4m3", "head-wise", torch.bfloat16),
    (2, 2, 20000, 20000, 128, True, "e4m3", "token-wise", torch.bfloat16),   # token-wise scales (templated kernel, D = 128)
    (1, 1, 24000, 24000, 64, False, "e5m2", "token-wise", torch.float16),
    (2, 2, 300, 40000, 128, False, "e4m3", "head-wise", torch.bfloat16),     # a short query block over a long, ragged key range
    (1, 1, 131072, 131072, 128, True, "e4m3", "head-wise", torch.bfloat16),  # 2048 chunks per head
    (1, 1, 65536, 65536, 128, False, "e4m3", "head-wise", torch.bfloat16),
]


@pytest.mark.parametrize("case", LONG, ids=lambda c: "Hq{}Hkv{}Sq{}Skv{}D{}{}_{}_{}_{}".format(
    c[0], c[1], c[2], c[3], c[4], "c" if c[5] else "f", c[6], c[7][:4], "bf16" if c[8] == torch.bfloat16 else "fp16"))
def test_fused_entry_on_long_sequences(case):
    Hq, Hkv, Sq, Skv, D, causal, fp8, scaling, dtype = case
    g = torch.Generator(device="cuda").manual_seed(Sq + D)
    q = torch.randn(1, Hq, Sq, D, device="cuda", generator=g).to(dtype)
    k = torch.randn(1, Hkv, Skv, D, device="cuda", generator=g).to(dtype)
    v = torch.randn(1, Hkv, Skv, D, device="cuda", generator=g).to(dtype)
    fn = qa.fp8_attn_func if scaling == "head-wise" else qa.fp8_token_wise_attn_func
    with qa.config.patch({"attention.fp8_format": fp8}):
        out = fn(q, k, v, is_causal=causal)
    assert torch.isfinite(out).all()
    got, path, lse = fused_call(q, k, v, causal=causal, fp8=fp8, scaling=scaling, return_lse=True)
    path_lse = path
    d128_head = D == 128 and scaling == "head-wise"
    if d128_head:
        assert np.array_equal(got, out_to_f32(out)), "row_path / lse must not change the output"
    else:   # (an LSE request selects the templated kernel's exact-exponential sweep: same bound, other bits)
        assert_lse_call = (got, path)
        got, path = fused_call(q, k, v, causal=causal, fp8=fp8, scaling=scaling)
        assert np.array_equal(got, out_to_f32(out))
    check_path_structure(path, Sq, Skv, causal, "auto", d128_head)
    m = "head" if scaling == "head-wise" else "token"
    vb = fused_step_uses_block_v(D, scaling, dtype, Skv)
    assert vb == (scaling == "head-wise" and Skv <= 16384)
    qc, kc, vc = bits16(q), bits16(k), bits16(v)
    q8, sq = oracle.quantize_fp8(qc, fmt16(dtype), m, FMT[fp8])
    k8, sk = oracle.quantize_fp8(kc, fmt16(dtype), m, FMT[fp8])
    for r0, r1 in _bands(Sq):
        sq_b = sq if m == "head" else sq[:, :, r0:r1]
        # every key of the head goes in (a per-head V scale is the abs-max of the WHOLE head); causal: q_offset masks the keys beyond the row
        ref, ref_lse = oracle_for_fp8_path(q8[:, :, r0:r1], k8, vc, sq_b, sk, fp8=fp8, v_dtype=dtype, scaling=m, causal=causal, v_block=vb,
                                           fused=True, q_offset=r0 + (Skv - Sq if causal else 0), return_lse=True)
        mx, rmse = assert_within_bound(got[:, :, r0:r1], ref, path[:, :, r0:r1], what=(case, r0))
        assert rmse < 2e-3 * max(1.0, float(np.abs(ref.fp8v).max())), (r0, mx, rmse)
        if not d128_head:   # the exact-exponential call that produced the LSE meets the same bound
            assert_within_bound(assert_lse_call[0][:, :, r0:r1], ref, path_lse[:, :, r0:r1], what=(case, r0, "lse call"))
        p = path_lse[:, :, r0:r1]
        tol = np.where(p == PATH_ONE_TERM, 2e-2 if d128_head else 2e-3, np.where(p == PATH_V16, 4e-3, 2e-3))
        err = np.abs(lse[:, :, r0:r1] - ref_lse)
        assert (err < tol).all(), (r0, float(err.max()))


def test_byte_offsets_beyond_32_bits_B33_H32_S16384_D128_causal():
    """33 x 32 heads of 16384 x 128: 2.2e9 elements per tensor (> 2^31), 4.4 GB per 16-bit tensor (> 2^32 bytes), 2.2 GB per fp8 image
    (> 2^31 bytes) -- 23 GB in all, a twelfth of the 288 GB the device has.  Head-wise scales make every head independent of the others
    (test_config5_at_its_stated_size...: batch-shard equivalence), so the heads at the far end of the tensors must equal, bit for bit,
    the same heads attended in a call of their own; the last head is also held against the oracle."""
    B, H, S, D = 33, 32, 16384, 128
    g = torch.Generator(device="cuda").manual_seed(33)
    q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
    assert q.numel() > 2 ** 31 and q.numel() * 2 > 2 ** 32
    out, lse, path = _native.fp8_quant_attention_forward(q, k, v, is_causal=True, return_lse=True, return_path=True)
    assert torch.equal(out, qa.fp8_attn_func(q, k, v, is_causal=True))
    for b, h in ((B - 1, H - 1), (B - 1, 0), (16, 5), (0, 0)):      # (16, 5): straddles the 2^32-byte line of the 16-bit tensors
        sl = lambda t: t[b:b + 1, h:h + 1].contiguous()
        o1, l1, p1 = _native.fp8_quant_attention_forward(sl(q), sl(k), sl(v), is_causal=True, return_lse=True, return_path=True)
        assert torch.equal(out[b, h], o1[0, 0]), (b, h)
        assert torch.equal(lse[b, h], l1[0, 0]) and torch.equal(path[b, h], p1[0, 0]), (b, h)
    assert torch.isfinite(out[::8]).all()
    b, h = B - 1, H - 1
    qs, ks, vs = (bits16(t[b:b + 1, h:h + 1]) for t in (q, k, v))
    q8, sq = oracle.quantize_fp8(qs, oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    k8, sk = oracle.quantize_fp8(ks, oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    pth = path[b:b + 1, h:h + 1].cpu().numpy()
    check_path_structure(pth, S, S, True, "auto", True)
    for r0, r1 in _bands(S, 256):
        ref = oracle_for_fp8_path(q8[:, :, r0:r1], k8[:, :, :r1], vs[:, :, :r1], sq, sk, causal=True, v_block=True, q_offset=r0)
        assert_within_bound(out_to_f32(out[b:b + 1, h:h + 1, r0:r1]), ref, pth[:, :, r0:r1], what=(b, h, r0))


def test_byte_offsets_beyond_32_bits_on_the_templated_and_16bit_kernels():
    """The same statement for the other kernels, at sizes that keep the test short: D = 64 token-wise (templated kernel), D = 256
    head-wise, and the 16-bit sibling `attn_func` -- each with > 2^31 elements per tensor, the far heads against calls of their own."""
    for D, H, S, fn, kw in ((64, 64, 8192, qa.fp8_token_wise_attn_func, {}), (256, 16, 8192, qa.fp8_attn_func, {}), (128, 32, 8192, qa.attn_func, {})):
        B = 2 ** 31 // (H * S * D) + 1
        g = torch.Generator(device="cuda").manual_seed(D)
        q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, generator=g) for _ in range(3))
        assert q.numel() > 2 ** 31
        out = fn(q, k, v, is_causal=True, **kw)
        for b, h in ((B - 1, H - 1), (B - 1, 0), (B // 2, 3), (0, 0)):
            sl = lambda t: t[b:b + 1, h:h + 1].contiguous()
            assert torch.equal(out[b, h], fn(sl(q), sl(k), sl(v), is_causal=True, **kw)[0, 0]), (D, b, h)
        assert torch.isfinite(out[::8]).all()
        del q, k, v, out
        torch.cuda.empty_cache()

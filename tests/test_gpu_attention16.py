"""-m gpu: the 16-bit sibling path (quantum_attn::attention_forward, ops.py:17-45) -- bf16/fp16 MFMA kernel and the
K16FRAG / V16FRAG re-layout -- against the oracle (fp64 SDPA on the same 16-bit inputs) and the reference's golden
`o16_*` outputs.  Tolerances: the exact-exponential kernel (the default; 16-bit P as the reference, tk/attention.py:286) max-abs
< 2^-7 (scaled by |O|max/2 above 2: one bf16 output ulp there is 2^-6), rmse < 2e-3; the opt-in fast exponential
(`fast_exp`, config.attention.fast_exp16) 2^-6 on flat rows -- and it is asserted to break that on peaked ones."""
import os

import numpy as np
import pytest
import torch

import oracle
import quantumattention_amd as qa
from quantumattention_amd import _native
from tests.conftest import GOLDEN, golden_files
from tests.gpu_utils import bits16, err_stats, fmt16, from_bits16, out_to_f32

pytestmark = pytest.mark.gpu

TOL = 2.0 ** -6
TOL_EXACT = 2.0 ** -7


def tol_for(ref, tol=TOL):
    return tol * max(1.0, float(np.abs(ref).max()) / 2.0)


def unpack16(buf: np.ndarray, layout: int, B: int, H: int, S: int, D: int) -> np.ndarray:
    """Invert QATTN_LAYOUT_K16FRAG / _V16FRAG (include/qattn.h) -> row-major uint16 [B,H,Sp,D]."""
    Sp = (S + 63) // 64 * 64
    x = np.asarray(buf).view(np.uint16).reshape(B, H, Sp // 64, 64 * D)
    key = np.arange(64)[:, None]
    d = np.arange(D)[None, :]
    if layout == _native.LAYOUT_K16FRAG:
        t, kl, s, hh, j = key >> 5, key & 31, d >> 4, (d >> 3) & 1, d & 7
        off = ((((t * (D // 16) + s) * 2 + hh) * 32 + kl) * 8) + j
    else:
        t, s, jh, hh, jl = key >> 5, (key >> 4) & 1, (key >> 3) & 1, (key >> 2) & 1, key & 3
        m, dl = d >> 5, d & 31
        off = ((((((t * (D // 32) + m) * 2 + s) * 2 + hh) * 32 + dl) * 8) + 4 * jh + jl)
    return x[..., off].reshape(B, H, Sp, D)


@pytest.mark.parametrize("D", [64, 128, 256])
@pytest.mark.parametrize("S", [64, 200, 1])
def test_pack16_layouts_bit_exact(D, S):
    torch.manual_seed(1)
    x = torch.randn(2, 3, S, D, dtype=torch.bfloat16)
    xb = bits16(x)
    for layout in (_native.LAYOUT_K16FRAG, _native.LAYOUT_V16FRAG):
        buf = _native.pack16(x.cuda(), layout).cpu().numpy()
        assert buf.size == _native.lib().qattn_16bit_tensor_bytes(layout, 2, 3, S, D)
        back = unpack16(buf, layout, 2, 3, S, D)
        np.testing.assert_array_equal(back[:, :, :S], xb)
        assert not back[:, :, S:].any()  # zero padding to a multiple of 64 keys


CASES = [
    # B, Hq, Hkv, Sq, Skv, D, causal, dtype
    (1, 2, 2, 128, 128, 64, False, torch.bfloat16),      # BASELINE config 1 shape
    (1, 2, 2, 128, 128, 128, True, torch.bfloat16),
    (2, 4, 4, 512, 512, 128, False, torch.bfloat16),
    (2, 4, 4, 512, 512, 128, True, torch.float16),
    (1, 2, 2, 1000, 1000, 128, False, torch.bfloat16),   # ragged
    (1, 2, 2, 1000, 1000, 128, True, torch.bfloat16),
    (1, 2, 2, 1024, 1000, 64, False, torch.float16),     # Sq != Skv
    (1, 2, 2, 333, 1024, 128, False, torch.bfloat16),
    (1, 8, 2, 512, 512, 128, True, torch.bfloat16),       # GQA
    (1, 1, 1, 1, 1, 128, False, torch.bfloat16),
    (1, 1, 1, 3, 70, 64, False, torch.bfloat16),
    (3, 5, 5, 300, 300, 128, True, torch.bfloat16),       # B*H not a multiple of 8
    (1, 8, 8, 2048, 2048, 128, False, torch.bfloat16),
    (1, 2, 2, 1100, 1100, 256, True, torch.bfloat16),     # D = 256: exact + fast launches, ragged
    (1, 4, 2, 1500, 1200, 256, False, torch.float16),     # GQA, Sq != Skv
    (2, 2, 2, 100, 100, 256, False, torch.bfloat16),
    (1, 2, 2, 999, 999, 128, False, torch.bfloat16),     # odd lengths of the reference's grid (tests/test_interface.py:62-73)
    (1, 2, 2, 999, 999, 128, True, torch.float16),
    (2, 8, 8, 1024, 999, 64, False, torch.bfloat16),
    (1, 2, 2, 999, 1024, 256, False, torch.float16),
    (1, 2, 2, 999, 999, 64, True, torch.bfloat16),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B{}Hq{}Hkv{}Sq{}Skv{}D{}{}_{}".format(
    c[0], c[1], c[2], c[3], c[4], c[5], "c" if c[6] else "f", "bf16" if c[7] == torch.bfloat16 else "fp16"))
def test_16bit_kernel_vs_oracle(case):
    B, Hq, Hkv, Sq, Skv, D, causal, dtype = case
    torch.manual_seed(0)
    q = torch.randn(B, Hq, Sq, D, dtype=dtype)
    k = torch.randn(B, Hkv, Skv, D, dtype=dtype)
    v = torch.randn(B, Hkv, Skv, D, dtype=dtype)
    f = fmt16(dtype)
    ref, ref_lse = oracle.attention_forward(bits16(q), bits16(k), bits16(v), f, f, f, causal=causal, return_lse=True)
    out = torch.ops.quantumattention_amd.attention_forward(q.cuda(), k.cuda(), v.cuda(), None, 0.0, causal)
    assert out.dtype == dtype and out.shape == q.shape and out.is_contiguous()
    got = out_to_f32(out)
    assert np.isfinite(got).all()
    mx, rmse = err_stats(got, ref)
    assert mx < tol_for(ref, TOL_EXACT), (mx, rmse)
    assert rmse < 2e-3 * max(1.0, float(np.abs(ref).max())), (mx, rmse)
    kf = _native.pack16(k.cuda(), _native.LAYOUT_K16FRAG)
    vf = _native.pack16(v.cuda(), _native.LAYOUT_V16FRAG)
    out2, lse = _native.attention_forward_16(q.cuda(), kf, vf, Hkv=Hkv, Skv=Skv, is_causal=causal, return_lse=True)
    mx2, _ = err_stats(out_to_f32(out2), ref)  # asking for the LSE selects the exact-exponential instantiation
    assert mx2 < tol_for(ref, TOL_EXACT), mx2
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse, rtol=0, atol=2e-3)
    # the opt-in fast exponential (C argument fast_exp; used where a row sees >= 1024 keys): flat N(0,1) rows meet 2^-6
    out3 = _native.attention_forward_16(q.cuda(), kf, vf, Hkv=Hkv, Skv=Skv, is_causal=causal, fast_exp=True)
    mx3, rmse3 = err_stats(out_to_f32(out3), ref)
    assert mx3 < tol_for(ref), (mx3, rmse3)
    if Skv >= 1024 and not causal:
        assert not torch.equal(out3, out2)     # ... and it really is another instantiation on these shapes


@pytest.mark.parametrize("D", [64, 128, 256])
def test_16bit_fast_exponential_is_for_flat_rows_only(D):
    """config.attention.fast_exp16: 1.8 % rms error per weight averages out only over rows whose weight is spread over many
    keys (docs/DESIGN_history_rounds_1_to_4.md 4.4).  On q x 4 rows (a few keys carry each row) it must break 2^-6 while the exact default keeps 2^-7; the
    flag reaches the kernel through the op (qa.attn_func)."""
    torch.manual_seed(2)
    S = 2048
    q, k, v = (torch.randn(1, 2, S, D) for _ in range(3))
    q = (q * 4.0).to(torch.bfloat16); k = k.to(torch.bfloat16); v = v.to(torch.bfloat16)
    f = oracle.FMT_BF16
    ref = oracle.attention_forward(bits16(q), bits16(k), bits16(v), f, f, f, causal=False)
    exact = out_to_f32(qa.attn_func(q.cuda(), k.cuda(), v.cuda()))
    with qa.config.patch({"attention.fast_exp16": True}):
        fast = out_to_f32(qa.attn_func(q.cuda(), k.cuda(), v.cuda()))
    assert err_stats(exact, ref)[0] < tol_for(ref, TOL_EXACT), err_stats(exact, ref)
    assert err_stats(fast, ref)[0] > TOL, ("the fast exponential was expected to break the bound on peaked rows", err_stats(fast, ref))


@pytest.mark.parametrize("name", golden_files())
def test_attn_func_vs_reference_golden_16bit_output(name):
    """attn_func on the golden q/k/v vs the reference op's own 16-bit output `o16_*` (ops.py:17-29 eager impl)."""
    z = np.load(os.path.join(GOLDEN, name))
    dtype = torch.bfloat16 if int(z["meta"][5]) else torch.float16
    q, k, v = (from_bits16(z[n], dtype).cuda() for n in ("q", "k", "v"))
    to_f32 = oracle.bf16_bits_to_f32 if dtype == torch.bfloat16 else oracle.fp16_bits_to_f32
    for causal in (False, True):
        key = "o16_causal" if causal else "o16_full"
        if key not in z:
            continue
        out = qa.attn_func(q, k, v, is_causal=causal)
        ref = to_f32(z[key])
        mx, rmse = err_stats(out_to_f32(out), ref)
        assert mx < tol_for(ref), (key, mx, rmse)  # = 2 x the exact kernel's bound: BOTH sides carry a 16-bit output rounding + 16-bit P rounding
        assert rmse < 2e-3 * max(1.0, float(np.abs(ref).max())), (key, mx, rmse)
        assert torch.equal(qa.attn_func_with_fallback(q, k, v, is_causal=causal), out)  # supported -> same kernel


def test_attn_func_rejects_what_the_reference_rejects():
    q = torch.randn(1, 2, 128, 128, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        qa.attn_func(q, q, q, dropout_p=0.1)
    with pytest.raises(ValueError):
        qa.attn_func(q, q, q, scale=0.5)
    with pytest.raises(ValueError):
        qa.attn_func(q, q, q.half())
    with pytest.raises(ValueError):
        qa.attn_func(q.float(), q.float(), q.float())
    q96 = torch.randn(1, 2, 128, 96, dtype=torch.bfloat16, device="cuda")   # head_dim not in {64,128,256} (nn.py:45-49)
    with pytest.raises(ValueError):
        qa.attn_func(q96, q96, q96)
    # ... and the with_fallback op routes the same input to aten SDPA instead (interface.py:62-98)
    out = qa.attn_func_with_fallback(q96, q96, q96)
    ref = torch.nn.functional.scaled_dot_product_attention(q96, q96, q96)
    assert torch.equal(out, ref)


def test_16bit_properties_at_full_size():
    """BASELINE config-2 shape on the 16-bit kernel: batch independence, V-linearity, key-permutation invariance."""
    torch.manual_seed(3)
    B, H, S, D = 2, 32, 4096, 128
    q = torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda")
    k = torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda")
    v = torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda")
    out = qa.attn_func(q, k, v)
    assert torch.isfinite(out).all()
    assert torch.equal(out, qa.attn_func(q, k, v))
    assert torch.equal(out[1:2], qa.attn_func(q[1:2], k[1:2], v[1:2]))
    assert torch.equal(qa.attn_func(q[:1], k[:1], v[:1] * 2), out[:1] * 2)   # exact: power-of-two scaling of V
    perm = torch.randperm(S, device="cuda")
    outp = qa.attn_func(q[:1], k[:1, :, perm], v[:1, :, perm])
    assert (outp.float() - out[:1].float()).abs().max().item() < TOL
    ref = torch.nn.functional.scaled_dot_product_attention(q[:1, :4], k[:1, :4], v[:1, :4])
    assert (out[:1, :4].float() - ref.float()).abs().max().item() < TOL

"""-m gpu: the reference kernel's own P.V numerics -- FP8 QK^T, 16-bit P, the ORIGINAL 16-bit V (src/quantum_attn/tk/attention.py:72,286,318) --
as built in csrc/qattn_pv16.h:

  * qattn_fp8_attention_forward(v_fmt = QATTN_FMT_BF16 / _FP16): whole tensors through the 16-bit-V pass, against the fp64 oracle on the
    same quantised q, k and the 16-bit V (the reference's semantics: fixture O2 of SURVEY 8c) and against the committed O2 / O1 fixtures;
  * the fused step `fp8_attn_func(q, k, v)`, every head dim: query blocks whose first row sees fewer than 1024 keys (early causal rows,
    short sequences) run the same pass (inside the D = 128 bf16 kernel, else a launch of its own); the other rows keep both GEMMs on
    FP8 MFMA.

Tolerance: P carries 8 mantissa bits and the output is rounded to 16 bit: max-abs < 2^-7 max(1, |O|max) against the 16-bit-V oracle
(the 16-bit sibling path's bar, tests/test_gpu_attention16.py)."""
import os

import numpy as np
import pytest
import torch

import oracle
import quantumattention_amd as qa
from quantumattention_amd import _native
from tests.conftest import GOLDEN, golden_files
from tests.gpu_utils import (FMT, PATH_ONE_TERM, PATH_V16, TDT, assert_within_bound, bits16, bits8, check_path_structure, err_stats, fmt16, from_bits16, fused_call,
                             oracle_for_fp8_path, out_to_f32)

pytestmark = pytest.mark.gpu
TOL16 = 2.0 ** -7


def _tol(ref):
    return TOL16 * max(1.0, float(np.abs(ref).max()))


def _run_v16(q, k, v, *, causal, fp8="e4m3", scaling="head-wise", return_lse=False):
    """quantise q, k with the library, attend the 16-bit V through the C ABI; returns (out, q8 bits, k8 bits, sq, sk)"""
    B, Hkv, Skv, D = k.shape
    q8, sq = _native.quant_fp8(q, scaling=scaling, fp8_dtype=TDT[fp8])
    kf, sk = _native.quant_fp8(k, scaling=scaling, fp8_dtype=TDT[fp8], layout=_native.LAYOUT_KFRAG)
    k8, _ = _native.quant_fp8(k, scaling=scaling, fp8_dtype=TDT[fp8])
    out = _native.fp8_attention_forward(q8, kf, v, sq, sk, None, Hkv=Hkv, Skv=Skv, out_dtype=v.dtype, is_causal=causal, scaling=scaling,
                                        return_lse=return_lse)
    return out, bits8(q8), bits8(k8), sq.cpu().numpy(), sk.cpu().numpy()


CASES = [
    # B, Hq, Hkv, Sq, Skv, causal, fp8, dtype, scaling
    (1, 2, 2, 256, 256, False, "e4m3", torch.bfloat16, "head-wise"),
    (1, 2, 2, 256, 256, True, "e4m3", torch.bfloat16, "head-wise"),
    (2, 4, 2, 1000, 1000, True, "e4m3", torch.bfloat16, "head-wise"),      # ragged, GQA
    (1, 3, 1, 96, 160, False, "e4m3", torch.bfloat16, "head-wise"),        # Sq != Skv, partial blocks
    (1, 2, 2, 1, 777, False, "e4m3", torch.bfloat16, "head-wise"),         # a single query row
    (2, 2, 2, 2048, 2048, True, "e4m3", torch.bfloat16, "head-wise"),
    (1, 4, 4, 1536, 2050, False, "e5m2", torch.bfloat16, "head-wise"),
    (1, 2, 2, 640, 640, True, "e4m3", torch.float16, "head-wise"),
    (1, 2, 1, 1100, 1100, True, "e4m3", torch.bfloat16, "token-wise"),
    (1, 2, 2, 513, 700, False, "e5m2", torch.float16, "token-wise"),
    # fewer chunks than ring stages: the loops' prologues and the last trips (D = 128: the two-group loop's four-stage ring)
    (1, 2, 2, 40, 40, False, "e4m3", torch.bfloat16, "head-wise"),
    (1, 2, 1, 70, 70, True, "e4m3", torch.bfloat16, "head-wise"),
    (2, 2, 2, 129, 129, True, "e4m3", torch.float16, "token-wise"),
    (1, 3, 3, 300, 65, False, "e5m2", torch.bfloat16, "head-wise"),
]


@pytest.mark.parametrize("D", [128, 64, 256])
@pytest.mark.parametrize("B,Hq,Hkv,Sq,Skv,causal,fp8,dtype,scaling", CASES, ids=lambda x: str(x).replace("torch.", ""))
def test_16bit_v_mode_against_the_oracle(B, Hq, Hkv, Sq, Skv, causal, fp8, dtype, scaling, D):
    torch.manual_seed(Sq + Skv + Hq)
    q = torch.randn(B, Hq, Sq, D, device="cuda").to(dtype)
    k = torch.randn(B, Hkv, Skv, D, device="cuda").to(dtype)
    v = torch.randn(B, Hkv, Skv, D, device="cuda").to(dtype)
    (out, lse), q8, k8, sq, sk = _run_v16(q, k, v, causal=causal, fp8=fp8, scaling=scaling, return_lse=True)
    assert out.dtype == dtype and out.is_contiguous()
    m = "head" if scaling == "head-wise" else "token"
    ref, ref_lse = oracle.attention_forward(q8, k8, bits16(v), FMT[fp8], FMT[fp8], fmt16(dtype), sq, sk, None, scale_mode=m, causal=causal,
                                            return_lse=True)
    got = out_to_f32(out)
    assert np.isfinite(got).all()
    mx, rmse = err_stats(got, ref)
    assert mx < _tol(ref), (mx, rmse)
    np.testing.assert_allclose(lse.float().cpu().numpy(), ref_lse, rtol=0, atol=2e-3)


@pytest.mark.parametrize("name", golden_files())
def test_16bit_v_mode_against_the_committed_reference_fixtures(name):
    """The reference's own q8 / k8 / scales (compiled numerics) and 16-bit V through the op with pv_precision = "16bit": against O2 (fp64
    SDPA of exactly these inputs, 16-bit V -- what the reference's kernel computes up to its 16-bit P) and O1, the reference's literal
    eager output.  With V and P in 16 bit the distance to O1 is that of two 16-bit roundings, not of an fp8 V: the bar is 2^-7 instead
    of the fp8 path's rmse < 1e-2 (tests/test_interface.py:57-59)."""
    z = np.load(os.path.join(GOLDEN, name))
    dtype = torch.bfloat16 if int(z["meta"][5]) else torch.float16
    step = int(z["o23_row_step"][0])
    v = from_bits16(z["v"], dtype).cuda()
    for method in ("head", "token"):
        q8 = torch.from_numpy(z[f"q8_{method}_compiled"].copy()).view(torch.float8_e4m3fn).cuda()
        k8 = torch.from_numpy(z[f"k8_{method}_compiled"].copy()).view(torch.float8_e4m3fn).cuda()
        sq = torch.from_numpy(z[f"sq_{method}_compiled"].copy()).cuda()
        sk = torch.from_numpy(z[f"sk_{method}_compiled"].copy()).cuda()
        for causal in (False, True):
            tag = "causal" if causal else "full"
            if f"o1_{method}_{tag}" not in z:
                continue
            with qa.config.patch({"attention.pv_precision": "16bit"}):
                out = torch.ops.quantumattention_amd.fp8_attention_forward(q8, k8, v, sq, sk, None, 0.0, causal)
            got = out_to_f32(out)
            if f"o2_{method}_{tag}" in z:
                o2 = z[f"o2_{method}_{tag}"]
                mx2, _ = err_stats(got[:, :, ::step], o2)
                assert mx2 < _tol(o2), (method, tag, mx2)
            o1 = oracle.bf16_bits_to_f32(z[f"o1_{method}_{tag}"]) if dtype == torch.bfloat16 else oracle.fp16_bits_to_f32(z[f"o1_{method}_{tag}"])
            mx1, rm1 = err_stats(got, o1)
            assert mx1 < 2 * _tol(o1) and rm1 < 2.0 ** -8, (method, tag, mx1, rm1)


@pytest.mark.parametrize("D", [128, 64, 256])
@pytest.mark.parametrize("S", [300, 1024, 2304])
def test_fused_step_attends_the_16bit_v_on_rows_that_see_few_keys(S, D):
    """Causal row 0 of every head IS V[0]: with an fp8 V its error is V's fp8 rounding (max-abs 0.1 at C3, VERDICT r3 Missing-1).  The fused
    step now runs the query blocks whose first row sees < 1024 keys through the 16-bit-V pass: those rows meet 2^-7 against fp64 SDPA with
    the ORIGINAL V, row 0 reproduces V[0] to the output rounding; the later blocks keep FP8 MFMA for both GEMMs and their 2^-6 bound
    against the block-scaled V (the mixed oracle of tests/gpu_utils.py)."""
    torch.manual_seed(S)
    B, H = 1, 3
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q8, sq = oracle.quantize_fp8(bits16(q), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    k8, sk = oracle.quantize_fp8(bits16(k), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    got, path = fused_call(q, k, v, causal=True)
    np.testing.assert_array_equal(got, out_to_f32(qa.fp8_attn_func(q, k, v, is_causal=True)))   # (the public call: row_path = NULL)
    np.testing.assert_array_equal(got[:, :, 0], v[:, :, 0].float().cpu().numpy())   # softmax over one key: the output IS that value row
    ref16 = oracle.attention_forward(q8, k8, bits16(v), oracle.FMT_E4M3, oracle.FMT_E4M3, oracle.FMT_BF16, sq, sk, None, causal=True)
    n_early = min(S, 1024)   # blocks 0..3: first rows 0, 256, 512, 768 see < 1024 keys
    assert (path[:, :, :n_early] == PATH_V16).all()
    check_path_structure(path, S, S, True, "auto", D == 128)
    mx, rmse = err_stats(got[:, :, :n_early], ref16[:, :, :n_early])
    assert mx < _tol(ref16), (mx, rmse)
    both = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, causal=True, v_block=True)
    assert_within_bound(got, both, path)    # every row against the oracle of ITS path
    # non-causal with fewer than 1024 keys: every block takes the pass
    if S < 1024:
        got_nc, path_nc = fused_call(q, k, v, causal=False)
        assert (path_nc == PATH_V16).all()
        ref_nc = oracle.attention_forward(q8, k8, bits16(v), oracle.FMT_E4M3, oracle.FMT_E4M3, oracle.FMT_BF16, sq, sk, None, causal=False)
        mx_nc, _ = err_stats(got_nc, ref_nc)
        assert mx_nc < _tol(ref_nc), mx_nc


@pytest.mark.parametrize("D", [128, 64, 256])
def test_causal_call_with_more_rows_than_keys_and_few_keys_runs_every_block_on_the_16bit_v(D):
    """causal Sq > Skv with Skv < 1024 -- no block sees 1024 keys, so EVERY block is early (pv16_early_blocks), on every head
    dim: the templated kernel's launcher used to clamp its early rows by the key count and left the later rows on one-term fp8 P."""
    torch.manual_seed(D)
    B, H, Sq, Skv = 1, 2, 1500, 600
    q = torch.randn(B, H, Sq, D, dtype=torch.bfloat16, device="cuda")
    k, v = (torch.randn(B, H, Skv, D, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    q8, sq = oracle.quantize_fp8(bits16(q), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    k8, sk = oracle.quantize_fp8(bits16(k), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    got, path = fused_call(q, k, v, causal=True)
    assert (path == PATH_V16).all()
    ref16 = oracle.attention_forward(q8, k8, bits16(v), oracle.FMT_E4M3, oracle.FMT_E4M3, oracle.FMT_BF16, sq, sk, None, causal=True)
    mx, _ = err_stats(got, ref16)
    assert mx < _tol(ref16), mx


def test_16bit_v_mode_argument_errors():
    q, k, v = (torch.randn(1, 2, 128, 64, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q8, sq = _native.quant_fp8(q)
    kf, sk = _native.quant_fp8(k, layout=_native.LAYOUT_KFRAG)
    with pytest.raises(ValueError):                                           # the output takes the value tensor's dtype
        _native.fp8_attention_forward(q8, kf, v, sq, sk, None, Hkv=2, Skv=128, out_dtype=torch.float16, is_causal=False)
    with qa.config.patch({"attention.pv_precision": "int8"}):
        with pytest.raises(ValueError, match="pv_precision"):
            torch.ops.quantumattention_amd.fp8_attention_forward(q8, _native.quant_fp8(k)[0], v, sq, sk, None, 0.0, False)

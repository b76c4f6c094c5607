"""-m gpu: the HIP fused attention against the oracle (fp64 SDPA on the same fp8-quantised q, k, v) and the
reference's golden vectors.  Tolerances (stated per BASELINE.json north_star):

  per ELEMENT |got - oracle| < 2^-6 * max(1, |oracle| / 2): north_star's flat 2^-6 = 0.015625 wherever |O| <= 2 (one bf16 output
  ulp there is 2^-7..2^-8); above that the bound follows the bf16 ulp of THAT element.  A fused call is graded row by row against
  the oracle of the path the kernel reports for the row (tests/gpu_utils.py: row_path, PathRef, grade) -- never "the closer of two".
"""
import os

import numpy as np
import pytest
import torch

import oracle
import quantumattention_amd as qa
from quantumattention_amd import _native
from tests.conftest import GOLDEN, golden_files
from tests.gpu_utils import (FMT, PATH_ONE_TERM, PATH_TWO_TERM, PATH_V16, TDT, assert_within_bound, bits16, bits8, check_path_structure, early_rows, err_stats,
                             fmt16, from_bits16, fused_call, fused_step_uses_block_v, grade, oracle_for_fp8_path, out_to_f32)

pytestmark = pytest.mark.gpu

TOL = 2.0 ** -6


@pytest.mark.parametrize("name", golden_files())
@pytest.mark.parametrize("method", ["head", "token"])
def test_op_on_reference_quantised_inputs_vs_oracle_and_golden(name, method):
    """Feed the reference's own q8/k8/scales through the op boundary (ops.py:98-110 contract)."""
    z = np.load(os.path.join(GOLDEN, name))
    dtype = torch.bfloat16 if int(z["meta"][5]) else torch.float16
    q8 = torch.from_numpy(z[f"q8_{method}_compiled"].copy()).view(torch.float8_e4m3fn).cuda()
    k8 = torch.from_numpy(z[f"k8_{method}_compiled"].copy()).view(torch.float8_e4m3fn).cuda()
    sq = torch.from_numpy(z[f"sq_{method}_compiled"].copy()).cuda()
    sk = torch.from_numpy(z[f"sk_{method}_compiled"].copy()).cuda()
    v = from_bits16(z["v"], dtype).cuda()
    for causal in (False, True):
        key = f"o1_{method}_{'causal' if causal else 'full'}"
        if key not in z:
            continue
        out = torch.ops.quantumattention_amd.fp8_attention_forward(q8, k8, v, sq, sk, None, 0.0, causal)
        assert out.dtype == dtype and out.is_contiguous()
        got = out_to_f32(out)
        ref = oracle_for_fp8_path(z[f"q8_{method}_compiled"], z[f"k8_{method}_compiled"], z["v"], z[f"sq_{method}_compiled"],
                                  z[f"sk_{method}_compiled"], v_dtype=dtype, scaling=method, causal=causal)
        assert_within_bound(got, ref, what=key)
        if method == "head":   # the committed fp64 fixture O3 (reference q8 / k8 / v8 / scales), no oracle code in the loop
            step = int(z["o23_row_step"][0])
            o3 = z[f"o3_head_{'causal' if causal else 'full'}"]
            assert_within_bound(got[:, :, ::step], o3, what=key + " vs fixture O3")
        # distance to the reference's literal eager output O1 (its V is NOT quantised): the reference's own bar
        o1 = oracle.bf16_bits_to_f32(z[key]) if dtype == torch.bfloat16 else oracle.fp16_bits_to_f32(z[key])
        rm1 = float(np.sqrt(np.mean((got - o1) ** 2)))
        assert rm1 < 1e-2, (key, rm1)  # tests/test_interface.py:57-59


@pytest.mark.parametrize("name", golden_files())
def test_one_call_rowmajor_entry_equals_the_three_call_sequence_and_the_op(name):
    """qattn_fp8_attention_forward_rowmajor has the pybind function's contract -- attention_forward(q, k, v, scale_q,
    scale_k, causal) on row-major tensors (tk/attention.py:357-360, 419-437).  Fed the reference's golden q8 / k8 / scales + v through
    ctypes it must equal, bit for bit, (i) qattn_pack_fp8 + qattn_quant_fp8 + qattn_fp8_attention_forward and (ii) the op; the LSE of the
    same call matches too; pv_16bit selects the reference kernel's own 16-bit P.V."""
    import ctypes

    z = np.load(os.path.join(GOLDEN, name))
    dtype = torch.bfloat16 if int(z["meta"][5]) else torch.float16
    v = from_bits16(z["v"], dtype).cuda()
    L = _native.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for method in ("head", "token"):
        q8 = torch.from_numpy(z[f"q8_{method}_compiled"].copy()).view(torch.float8_e4m3fn).cuda()
        k8 = torch.from_numpy(z[f"k8_{method}_compiled"].copy()).view(torch.float8_e4m3fn).cuda()
        sq = torch.from_numpy(z[f"sq_{method}_compiled"].copy()).cuda()
        sk = torch.from_numpy(z[f"sk_{method}_compiled"].copy()).cuda()
        B, Hq, Sq, D = q8.shape
        Hkv, Skv = k8.shape[1], k8.shape[2]
        scaling = "head-wise" if method == "head" else "token-wise"
        for causal in (False, True):
            if causal and Sq != Skv:
                continue
            # the raw ctypes call, as INTEGRATION.md section 2 shows it
            out = torch.empty(B, Hq, Sq, D, dtype=dtype, device="cuda")
            need = L.qattn_fp8_attention_rowmajor_workspace_bytes(B, Hq, Hkv, Sq, Skv, D)
            ws = torch.empty(need, dtype=torch.uint8, device="cuda")
            rc = L.qattn_fp8_attention_forward_rowmajor(P(q8), P(k8), P(v), P(out), None, P(sq), P(sk), B, Hq, Hkv, Sq, Skv, D, 0, _native.fmt_of(dtype), 0,
                                                        0 if method == "head" else 1, int(causal), ctypes.c_float(0.0), 0, 0, P(ws), ctypes.c_size_t(need),
                                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
            assert L.qattn_fp8_attention_forward_rowmajor(P(q8), P(k8), P(v), P(out), None, P(sq), P(sk), B, Hq, Hkv, Sq, Skv, D, 0, _native.fmt_of(dtype), 0,
                                                          0, int(causal), ctypes.c_float(0.0), 0, 0, P(ws), ctypes.c_size_t(need - 1), None) == -4
            kf = _native.pack_fp8(k8, _native.LAYOUT_KFRAG)
            vf, sv = _native.quant_fp8(v, scaling="head-wise", layout=_native.LAYOUT_VFRAG)
            three, lse3 = _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=Hkv, Skv=Skv, out_dtype=dtype, is_causal=causal, scaling=scaling,
                                                        return_lse=True)
            plain = _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=Hkv, Skv=Skv, out_dtype=dtype, is_causal=causal, scaling=scaling)
            assert torch.equal(out, plain), (method, causal)
            assert torch.equal(out, torch.ops.quantumattention_amd.fp8_attention_forward(q8, k8, v, sq, sk, None, 0.0, causal))
            one, lse1 = _native.fp8_attention_forward_rowmajor(q8, k8, v, sq, sk, is_causal=causal, return_lse=True)
            assert torch.equal(one, three) and torch.equal(lse1, lse3)
            v16 = _native.fp8_attention_forward(q8, kf, v, sq, sk, None, Hkv=Hkv, Skv=Skv, out_dtype=dtype, is_causal=causal, scaling=scaling)
            assert torch.equal(_native.fp8_attention_forward_rowmajor(q8, k8, v, sq, sk, is_causal=causal, pv_16bit=True), v16)
    assert L.qattn_fp8_attention_forward_rowmajor(None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 128, 0, 2, 0, 0, 0, ctypes.c_float(0.0), 0, 0,
                                                  None, 0, None) == -1


CASES = [
    # B, Hq, Hkv, Sq, Skv, D, causal, fp8, scaling, dtype
    (1, 2, 2, 128, 128, 128, False, "e4m3", "head-wise", torch.bfloat16),
    (1, 2, 2, 128, 128, 128, True, "e4m3", "head-wise", torch.bfloat16),
    (2, 4, 4, 512, 512, 128, False, "e4m3", "head-wise", torch.bfloat16),
    (2, 4, 4, 512, 512, 128, True, "e4m3", "head-wise", torch.bfloat16),
    (1, 2, 2, 1000, 1000, 128, False, "e4m3", "head-wise", torch.bfloat16),   # ragged (tests/test_interface.py:78)
    (1, 2, 2, 1000, 1000, 128, True, "e4m3", "head-wise", torch.float16),
    (1, 2, 2, 1024, 1000, 128, False, "e4m3", "head-wise", torch.bfloat16),   # Sq != Skv
    (1, 2, 2, 333, 1024, 128, False, "e4m3", "head-wise", torch.bfloat16),
    (1, 2, 2, 1024, 1024, 64, False, "e4m3", "head-wise", torch.bfloat16),
    (1, 2, 2, 1000, 1000, 64, True, "e4m3", "head-wise", torch.bfloat16),
    (1, 2, 2, 1024, 1024, 256, False, "e4m3", "head-wise", torch.bfloat16),
    (1, 2, 2, 1000, 1000, 256, True, "e4m3", "head-wise", torch.float16),
    (1, 8, 2, 512, 512, 128, True, "e4m3", "head-wise", torch.bfloat16),       # GQA
    (1, 2, 2, 2048, 2048, 128, True, "e5m2", "head-wise", torch.bfloat16),     # e5m2 (BASELINE config 5 format)
    (1, 2, 2, 1000, 1000, 128, False, "e5m2", "head-wise", torch.bfloat16),
    (1, 2, 2, 1024, 1024, 128, False, "e4m3", "token-wise", torch.bfloat16),
    (1, 2, 2, 1000, 1000, 128, True, "e4m3", "token-wise", torch.bfloat16),
    (1, 1, 1, 1, 1, 128, False, "e4m3", "head-wise", torch.bfloat16),           # single token
    # fp16 inputs on the fused D = 128 path (round 5: in-kernel Q quantisation, block-scaled V and the 16-bit-V passes for both input types)
    (1, 2, 2, 2304, 2304, 128, False, "e4m3", "head-wise", torch.float16),
    (1, 2, 2, 2304, 2304, 128, True, "e4m3", "head-wise", torch.float16),
    (1, 4, 2, 1500, 1500, 128, True, "e5m2", "head-wise", torch.float16),      # GQA, e5m2, ragged
    (1, 2, 2, 700, 700, 128, False, "e4m3", "head-wise", torch.float16),       # fewer than 1024 keys: every block on the 16-bit V
    (1, 1, 1, 3, 70, 64, False, "e4m3", "head-wise", torch.bfloat16),
    (3, 5, 5, 300, 300, 128, True, "e4m3", "head-wise", torch.bfloat16),        # B*H not a multiple of 8
    (1, 2, 2, 1100, 1100, 64, True, "e4m3", "token-wise", torch.bfloat16),      # D = 64 / 256: every mode of the templated kernel
    (1, 2, 2, 1024, 1024, 256, False, "e5m2", "token-wise", torch.bfloat16),
    (1, 4, 2, 2100, 2100, 256, True, "e4m3", "head-wise", torch.bfloat16),      # two-term + byte launches, GQA
    (2, 4, 4, 1500, 1200, 64, False, "e5m2", "head-wise", torch.float16),
    (1, 8, 2, 1100, 1100, 128, False, "e4m3", "head-wise", torch.bfloat16),     # GQA on the hand-scheduled kernel (K/V head, moments and rescue
    (1, 6, 3, 1300, 1300, 128, True, "e4m3", "head-wise", torch.bfloat16),      #   index the kv head), with and without the XCD remap (B*Hq % 8)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "B{}Hq{}Hkv{}Sq{}Skv{}D{}{}_{}_{}_{}".format(
    c[0], c[1], c[2], c[3], c[4], c[5], "c" if c[6] else "f", c[7], c[8][:4], "bf16" if c[9] == torch.bfloat16 else "fp16"))
def test_fused_path_from_16bit_inputs(case):
    """quant pre-pass -> fragment layouts -> attention, vs oracle quantiser + oracle attention on the same seed."""
    B, Hq, Hkv, Sq, Skv, D, causal, fp8, scaling, dtype = case
    torch.manual_seed(0)
    q = torch.randn(B, Hq, Sq, D, dtype=dtype)
    k = torch.randn(B, Hkv, Skv, D, dtype=dtype)
    v = torch.randn(B, Hkv, Skv, D, dtype=dtype)
    m = "head" if scaling == "head-wise" else "token"
    q8, sq = oracle.quantize_fp8(bits16(q), fmt16(dtype), m, FMT[fp8])
    k8, sk = oracle.quantize_fp8(bits16(k), fmt16(dtype), m, FMT[fp8])
    ref, ref_lse = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, fp8=fp8, v_dtype=dtype, scaling=m, causal=causal,
                                       return_lse=True)       # V with one scale per head: the separate C calls below (the LSE does not depend on V)
    vb = fused_step_uses_block_v(D, scaling, dtype, Skv)      # the fused step quantises V per 64-key chunk there
    ref_fused = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, fp8=fp8, v_dtype=dtype, scaling=m, causal=causal, v_block=vb, fused=True)
    out = torch.ops.quantumattention_amd.fp8_quant_attention_forward(q.cuda(), k.cuda(), v.cuda(), causal, scaling, fp8)
    assert torch.isfinite(out).all()
    # the same C call with its per-row path and the LSE vector of the SAME launch (ABI 7): the output bits do not depend on either
    got, path, lse_f = fused_call(q, k, v, causal=causal, fp8=fp8, scaling=scaling, return_lse=True)
    d128_head = D == 128 and scaling == "head-wise"
    if d128_head:   # (D = 64 / 256 / token-wise: an LSE request selects the exact-exponential one-term sweep -- same bound, other bits)
        assert np.array_equal(got, out_to_f32(out)), "row_path / lse must not change the output"
    else:
        got, path = fused_call(q, k, v, causal=causal, fp8=fp8, scaling=scaling)
        assert np.array_equal(got, out_to_f32(out)), "row_path must not change the output"
    check_path_structure(path, Sq, Skv, causal, "auto", d128_head)
    mx, rmse = assert_within_bound(got, ref_fused, path)
    assert rmse < 2e-3 * max(1.0, float(np.abs(ref_fused.fp8v).max())), (mx, rmse)
    # LSE from the fused call, every row (the per-row vector the reference defines, tk/attention.py:333-346, 439-446).  Stated tolerance
    # (include/qattn.h): rows of the FP8-MFMA sweep carry the sum of the e4m3-ROUNDED weights the second GEMM consumed (their mean offset,
    # +0.01353, is subtracted by the epilogue: csrc/qattn_attn.h kByteLseBias): 2e-2; every other row (exact exponentials, 16-bit or fp32
    # sums) 2e-3, 16-bit-V rows 4e-3 (sums of the ROUNDED 16-bit P: bf16's 2^-8 undiluted on a row carried by one key).
    lse_tol = np.where(path == PATH_ONE_TERM, 2e-2 if d128_head else 2e-3, np.where(path == PATH_V16, 4e-3, 2e-3))
    assert (np.abs(lse_f - ref_lse) < lse_tol).all(), float(np.abs(lse_f - ref_lse).max())
    # optional LSE output of the separate calls (exact-exponential path)
    qg8, sqg = _native.quant_fp8(q.cuda(), scaling=scaling, fp8_dtype=TDT[fp8])
    kf, skg = _native.quant_fp8(k.cuda(), scaling=scaling, fp8_dtype=TDT[fp8], layout=_native.LAYOUT_KFRAG)
    vf, svg = _native.quant_fp8(v.cuda(), scaling="head-wise", fp8_dtype=TDT[fp8], layout=_native.LAYOUT_VFRAG)
    out2 = _native.fp8_attention_forward(qg8, kf, vf, sqg, skg, svg, Hkv=Hkv, Skv=Skv, out_dtype=dtype,
                                         is_causal=causal, scaling=scaling)
    if vb:   # different V formats: each against its own oracle
        mx2, rmse2 = assert_within_bound(out_to_f32(out2), ref)
        assert rmse2 < 2e-3 * max(1.0, float(np.abs(ref).max())), (mx2, rmse2)
    else:
        # the op and the direct C-ABI sequence agree bit for bit -- outside the fused step's early rows (16-bit V there,
        # csrc/qattn_pv16.h; the separate calls have only the fp8 V)
        n_early = int(early_rows(Sq, Skv, causal).sum())
        assert torch.equal(out2[:, :, n_early:], out[:, :, n_early:]), "the op and the direct C-ABI sequence must agree bit for bit"
    # asking the SEPARATE attention call for the LSE selects its exact-exponential path
    out3, lse = _native.fp8_attention_forward(qg8, kf, vf, sqg, skg, svg, Hkv=Hkv, Skv=Skv, out_dtype=dtype,
                                              is_causal=causal, scaling=scaling, return_lse=True)
    assert_within_bound(out_to_f32(out3), ref)
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse, rtol=0, atol=2e-3)


def test_interface_matches_reference_test_bar():
    """The reference's own accuracy test (tests/test_interface.py:31-59): RMSE < 1e-2 vs 16-bit SDPA on the
    unquantised N(0,1) inputs, through the public fp8_attn_func."""
    torch.manual_seed(0)
    for (B, H, S, D, causal) in [(1, 8, 1024, 128, False), (2, 8, 1000, 128, True), (1, 8, 1024, 64, True)]:
        q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        out = qa.fp8_attn_func(q, k, v, is_causal=causal)
        ref = torch.nn.functional.scaled_dot_product_attention(q.float().cpu(), k.float().cpu(), v.float().cpu(), is_causal=causal)
        rmse = torch.sqrt(torch.nn.functional.mse_loss(out.float().cpu(), ref))
        assert rmse < 1e-2, rmse
        out_fb = qa.fp8_attn_func_with_fallback(q, k, v, is_causal=causal)
        assert torch.equal(out, out_fb)
        out_tw = qa.fp8_token_wise_attn_func(q, k, v, is_causal=causal)
        assert torch.sqrt(torch.nn.functional.mse_loss(out_tw.float().cpu(), ref)) < 1e-2


def test_unsupported_inputs_raise_or_fall_back_like_the_reference():
    q = torch.randn(1, 2, 128, 96, dtype=torch.bfloat16, device="cuda")  # head dim 96 unsupported (nn.py:45-49)
    with pytest.raises(ValueError, match="Unsupported head dimension"):
        qa.fp8_attn_func(q, q, q)
    ref = torch.nn.functional.scaled_dot_product_attention(q, q, q)
    assert torch.equal(qa.fp8_attn_func_with_fallback(q, q, q), ref)
    q = torch.randn(1, 2, 128, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError, match="dropout_p"):
        qa.fp8_attn_func(q, q, q, dropout_p=0.1)
    with pytest.raises(ValueError, match="scale must be None"):
        qa.fp8_attn_func(q, q, q, scale=0.5)
    with pytest.raises(ValueError, match="attn_mask"):
        qa.fp8_attn_func(q, q, q, attn_mask=torch.ones(128, 128, dtype=torch.bool, device="cuda"))
    assert _native.lib().qattn_fp8_attention_forward(None, None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 128, 0, 0, 2, 0, 0, 0.0, 1, 0, None, 0, None) == -1
    assert _native.lib().qattn_check_device() == 0


def test_rescale_branch_forced_by_a_spiked_key():
    """§5.4 rule 26: force the deferred-max rescale at a chosen late chunk (one key far above the running max)."""
    torch.manual_seed(1)
    B, H, S, D = 1, 2, 1024, 128
    q = torch.randn(B, H, S, D, dtype=torch.bfloat16)
    k = torch.randn(B, H, S, D, dtype=torch.bfloat16)
    v = torch.randn(B, H, S, D, dtype=torch.bfloat16)
    k[:, :, 777, :] = q[:, :, 100, :] * 1.5   # row 100's score for key 777 is ~ |q|^2 * 1.5 / sqrt(D) >> others
    k[:, :, 900, :] = q[:, :, 33, :] * 2.0
    for causal in (False, True):
        q8, sq = oracle.quantize_fp8(bits16(q), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
        k8, sk = oracle.quantize_fp8(bits16(k), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
        ref = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, causal=causal, v_block=True)   # fp8_attn_func = the fused step
        out = qa.fp8_attn_func(q.cuda(), k.cuda(), v.cuda(), is_causal=causal)
        got, path = fused_call(q, k, v, causal=causal)
        assert np.array_equal(got, out_to_f32(out)), "fp8_attn_func is the fused entry with row_path = NULL"
        check_path_structure(path, S, S, causal, "auto", True)
        assert_within_bound(got, ref, path, what=causal)


@pytest.mark.parametrize("causal", [False, True])
def test_full_size_properties_B4_H32_S4096_D128(causal):
    """BASELINE configs 2/3 at full size: oracle on a slice of heads/rows + size-independent properties."""
    torch.manual_seed(0)
    B, H, S, D = 4, 32, 4096, 128
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    out = qa.fp8_attn_func(q, k, v, is_causal=causal)
    assert torch.isfinite(out).all()
    # (1) determinism -- of the output AND of the per-row path the kernel reports (the grader's input below)
    out_p, path_t = _native.fp8_quant_attention_forward(q, k, v, is_causal=causal, return_path=True)
    assert torch.equal(out, out_p)
    path = path_t.cpu().numpy()
    check_path_structure(path, S, S, causal, "auto", True)
    # N(0,1) data: nearly every row outside the early blocks is swept once with one-term fp8 P on the fp8 V (both GEMMs on FP8 MFMA)
    rest = path[..., ~early_rows(S, S, causal)]
    assert (rest == PATH_ONE_TERM).mean() > 0.99, (rest == PATH_ONE_TERM).mean()
    # (2) batch-shard equivalence: a shard computed alone is bit-identical (the multi-GPU decomposition)
    assert torch.equal(out[1:2], qa.fp8_attn_func(q[1:2], k[1:2], v[1:2], is_causal=causal))
    # (3) exact power-of-two linearity in V (scale_v doubles, payload unchanged)
    assert torch.equal(qa.fp8_attn_func(q[:1], k[:1], v[:1] * 2, is_causal=causal), out[:1] * 2)
    # (4) oracle on a slice: heads (0,0) and (3,31), all rows for the first, a row band for the second
    for (b, h, rows) in [(0, 0, slice(0, S)), (3, 31, slice(S - 512, S))]:
        qs, ks, vs = q[b:b + 1, h:h + 1].cpu(), k[b:b + 1, h:h + 1].cpu(), v[b:b + 1, h:h + 1].cpu()
        q8, sq = oracle.quantize_fp8(bits16(qs), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
        k8, sk = oracle.quantize_fp8(bits16(ks), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
        ref = oracle_for_fp8_path(q8, k8, bits16(vs), sq, sk, causal=causal, v_block=True)[0, 0, rows]
        got = out_to_f32(out[b, h, rows])
        assert_within_bound(got, ref, path[b, h, rows], what=(b, h))
    # (5) non-causal only: permuting the keys (K and V rows together) leaves the output unchanged to rounding
    if not causal:
        perm = torch.randperm(S, device="cuda")
        outp = qa.fp8_attn_func(q[:1], k[:1, :, perm], v[:1, :, perm])
        # both runs round P to fp8 in different chunk groupings: each is within TOL/2 of the oracle here
        assert (outp.float() - out[:1].float()).abs().max() < TOL


def test_config4_global_batch_32_on_one_gpu_equals_its_eight_rank_shards():
    """BASELINE configs[3]: B = 32 H = 32 S = 4096 D = 128 non-causal, batch-sharded over 8 GPUs (4 per GPU = configs[1] per device).  The
    8-GPU scaling curve cannot be measured on this pool (one GPU per box: UNMEASURED, stated in DESIGN.md section 6); the strongest
    single-GPU statement is made here: the shard every rank r of 8 draws through the product's own helper
    (quantumattention_amd.utils.shard.synthetic_qkv, what bench.py does on every rank) IS the slice of the global batch, and attending
    it alone gives, bit for bit, the slice of the B = 32 call -- no rank's result depends on what the other ranks hold (no collective on
    the data path).  4 GiB of tensors on one device."""
    from quantumattention_amd.utils.shard import batch_shard, synthetic_qkv

    B, H, S, D, world = 32, 32, 4096, 128, 8
    q, k, v = synthetic_qkv(batch_shard(B, 0, 1), H, S, D, device="cuda", seed=6)
    assert q.shape == (B, H, S, D)
    whole = qa.fp8_attn_func(q, k, v)
    assert torch.isfinite(whole).all()
    for r in range(world):
        shard = batch_shard(B, r, world)
        qs, ks, vs = synthetic_qkv(shard, H, S, D, device="cuda", seed=6)
        assert torch.equal(qs, q[shard.start:shard.stop]) and torch.equal(ks, k[shard.start:shard.stop]) and torch.equal(vs, v[shard.start:shard.stop]), r
        assert torch.equal(qa.fp8_attn_func(qs, ks, vs), whole[shard.start:shard.stop]), r
        del qs, ks, vs
    # one oracle slice of the B = 32 call (a head of the last rank's shard)
    b, h = 31, 7
    q8, sq = oracle.quantize_fp8(bits16(q[b:b + 1, h:h + 1]), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    k8, sk = oracle.quantize_fp8(bits16(k[b:b + 1, h:h + 1]), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    rows = slice(S - 512, S)
    ref = oracle_for_fp8_path(q8[:, :, rows], k8, bits16(v[b:b + 1, h:h + 1]), sq, sk, v_block=True)
    _, path = _native.fp8_quant_attention_forward(q[b:b + 1], k[b:b + 1], v[b:b + 1], is_causal=False, return_path=True)
    assert_within_bound(out_to_f32(whole[b, h, rows]), ref[0, 0], path[0, h, rows].cpu().numpy())


@pytest.mark.parametrize("precision", ["auto", "fast"])
def test_persistent_launch_equals_one_workgroup_per_block(precision):
    """Non-causal launches with more blocks than CUs run persistent workgroups (csrc/qattn_attn_v2.hip); smaller ones one
    workgroup per block.  B = 2 x 8 heads x 32 blocks = 512 blocks is the former, each batch element alone (256 blocks) the
    latter: same bits.  One head is scaled so that rescues and two-term blocks occur in both forms."""
    torch.manual_seed(21)
    B, H, S, D = 2, 8, 8192, 128
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q[:, 3] *= 1.25
    q[:, 5] *= 2.0
    with qa.config.patch({"attention.precision": precision}):
        both = qa.fp8_attn_func(q, k, v)
        for b in range(B):
            assert torch.equal(both[b:b + 1], qa.fp8_attn_func(q[b:b + 1], k[b:b + 1], v[b:b + 1])), b
    assert torch.isfinite(both).all()


@pytest.mark.parametrize("precision", ["auto", "fast"])
def test_dynamic_hand_out_of_a_large_non_causal_launch_equals_static_shares(precision):
    """Non-causal launches with at least kDynMinRounds = 24 query blocks per workgroup draw their blocks from the per-XCD counters
    (csrc/qattn_attn_v2.hip launch_attn_v2_chk); smaller ones take equal static shares.  B = 8 x 32 heads x 24 blocks = 6144
    blocks is the former, every pair of batch elements (1536 blocks = 6 rounds) the latter: same bits, every row written."""
    torch.manual_seed(33)
    B, H, S, D = 8, 32, 6144, 128
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q[:, 3] *= 1.25
    q[:, 5] *= 2.0
    with qa.config.patch({"attention.precision": precision}):
        junk = torch.full_like(q, float("nan"))
        del junk   # (the caching allocator hands this block to the next empty_like(q): a block nobody wrote would stay NaN)
        whole = qa.fp8_attn_func(q, k, v)
        assert torch.isfinite(whole).all()
        for b in range(0, B, 2):
            assert torch.equal(whole[b:b + 2], qa.fp8_attn_func(q[b:b + 2], k[b:b + 2], v[b:b + 2])), b


@pytest.mark.parametrize("B,H,S", [(4, 8, 4096), (2, 4, 2304), (8, 8, 1280), (1, 8, 768), (16, 16, 2048)])
def test_causal_block_order_visits_every_block_once(B, H, S):
    """Causal AUTO launches take a head's query blocks longest first, except that the blocks right above the two-term line go
    before everything else (qattn_attn.h causal_order), persistent with a dynamic hand-out when there are more blocks than
    CUs.  Whatever the order, every block is computed exactly once: no row stays unwritten, and the result agrees with the
    ACCURATE launch (since round 5: every block on the 16-bit V with 16-bit P, no rescues) within the one-term budget plus what the fp8
    V of the one-term rows costs against the 16-bit V."""
    torch.manual_seed(S)
    q, k, v = (torch.randn(B, H, S, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    with qa.config.patch({"attention.precision": "accurate"}):
        ref = qa.fp8_attn_func(q, k, v, is_causal=True)
    for precision in ("auto", "fast"):
        with qa.config.patch({"attention.precision": precision}):
            junk = torch.full_like(q, float("nan"))
            del junk
            out = qa.fp8_attn_func(q, k, v, is_causal=True)
        assert torch.isfinite(out).all(), precision
        # (FAST has no budget on rows that see few keys; AUTO keeps every row within it)
        bound = 0.04 if precision == "auto" else 0.25
        assert (out.float() - ref.float()).abs().max().item() < bound, precision


def test_full_size_properties_config5_shape_S16384_H40_e5m2_causal():
    """BASELINE config 5's shape and format (float8_e5m2, causal, S = 16384, 40 heads, B = 1) at full size."""
    torch.manual_seed(5)
    B, H, S, D = 1, 40, 16384, 128
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    with qa.config.patch({"attention.fp8_format": "e5m2"}):
        out = qa.fp8_attn_func(q, k, v, is_causal=True)
        assert torch.isfinite(out).all()
        out_p, path_t = _native.fp8_quant_attention_forward(q, k, v, is_causal=True, fp8_dtype=torch.float8_e5m2, return_path=True)
        assert torch.equal(out, out_p)                                                                # determinism (and row_path changes nothing)
        path = path_t.cpu().numpy()
        check_path_structure(path, S, S, True, "auto", True)
        assert torch.equal(out[:, 8:16], qa.fp8_attn_func(q[:, 8:16], k[:, 8:16], v[:, 8:16], is_causal=True))  # head-shard equivalence
        assert torch.equal(qa.fp8_attn_func(q[:, :8], k[:, :8], v[:, :8] * 2, is_causal=True), out[:, :8] * 2)  # exact V-linearity
    # oracle on one head: the first 1.5 k rows (two-term rows and the switch to the byte path) and the last 256 rows
    h = 17
    qs, ks, vs = q[:, h:h + 1].cpu(), k[:, h:h + 1].cpu(), v[:, h:h + 1].cpu()
    q8, sq = oracle.quantize_fp8(bits16(qs), oracle.FMT_BF16, "head", oracle.FMT_E5M2)
    k8, sk = oracle.quantize_fp8(bits16(ks), oracle.FMT_BF16, "head", oracle.FMT_E5M2)
    top = 1536   # top-left causal alignment: query rows [0, top) against the full K / V (V quantised with the head's scale)
    ref_top = oracle_for_fp8_path(q8[:, :, :top], k8, bits16(vs), sq, sk, fp8="e5m2", causal=True, v_block=True)
    mx, rmse = assert_within_bound(out_to_f32(out[0, h, :top]), ref_top[0, 0], path[0, h, :top])
    assert rmse < 3e-3, (mx, rmse)
    # last rows see every key: non-causal oracle rows == causal rows for the final row only; use Sq != Skv non-causal instead
    tail = slice(S - 256, S)
    ref_tail = oracle_for_fp8_path(q8[:, :, tail], k8, bits16(vs), sq, sk, fp8="e5m2", causal=False)
    with qa.config.patch({"attention.fp8_format": "e5m2"}):
        out_tail = torch.ops.quantumattention_amd.fp8_attention_forward(
            torch.from_numpy(q8[:, :, tail].copy()).view(torch.float8_e5m2).cuda(), torch.from_numpy(k8).view(torch.float8_e5m2).cuda(),
            v[:, h:h + 1], torch.from_numpy(sq).cuda(), torch.from_numpy(sk).cuda(), None, 0.0, False)
    assert_within_bound(out_to_f32(out_tail[0, 0]), ref_tail[0, 0])


@pytest.mark.parametrize("B,H,S", [(2, 8, 1024), (4, 8, 4096)])
def test_hip_graph_capture_of_the_whole_step(B, H, S):
    """The C ABI promises "no host synchronisation, no allocation, graph-capture safe" (include/qattn.h): capture
    quant pre-pass + attention (and the 16-bit path) in a HIP graph, replay it on new input data, compare bit-exactly
    with the eager launches.  The larger causal shape (512 blocks > CUs) runs the persistent launch whose block hand-out
    counters are cleared by a KERNEL of the same capture (the quantise pass in the fused step, zero_words_kernel otherwise: as a
    memset node the 32-byte fill faulted on the second replay)."""
    torch.manual_seed(11)
    D = 128
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):  # warm-up outside the capture (lazy module load, hipFuncSetAttribute)
            qa.fp8_attn_func(q, k, v, is_causal=True)
            qa.attn_func(q, k, v)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out8 = qa.fp8_attn_func(q, k, v, is_causal=True)
        out16 = qa.attn_func(q, k, v)
    for seed in (1, 2):
        torch.manual_seed(seed)
        q.copy_(torch.randn_like(q)); k.copy_(torch.randn_like(k)); v.copy_(torch.randn_like(v))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out8, qa.fp8_attn_func(q, k, v, is_causal=True))
        assert torch.equal(out16, qa.attn_func(q, k, v))


@pytest.mark.parametrize("D,causal", [(128, True), (128, False), (64, True)])
def test_hip_graph_capture_with_the_lse_and_row_path_outputs(D, causal):
    """ABI 7: the fused entry's optional outputs -- the LSE vector of the same launch and the per-row path codes (pre-filled by a small
    kernel, marked by another on the templated kernel: kernel nodes, no memset node) -- are graph-capture safe like the rest of the call:
    captured once, replayed on new data, all three outputs equal the eager call's bit for bit.  q x 1.3 so that rescues happen."""
    torch.manual_seed(D + causal)
    B, H, S = 2, 4, 2304
    q = (torch.randn(B, H, S, D, device="cuda") * 1.3).to(torch.bfloat16)
    k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    call = lambda: _native.fp8_quant_attention_forward(q, k, v, is_causal=causal, return_lse=True, return_path=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        call()   # warm-up outside the capture (lazy module load, hipFuncSetAttribute, the side stream of the templated kernel)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out, lse, path = call()
    for seed in (1, 2):
        torch.manual_seed(seed)
        q.copy_((torch.randn(B, H, S, D, device="cuda") * 1.3).to(torch.bfloat16)); k.copy_(torch.randn_like(k)); v.copy_(torch.randn_like(v))
        g.replay()
        torch.cuda.synchronize()
        want_out, want_lse, want_path = call()
        assert torch.equal(out, want_out) and torch.equal(lse, want_lse) and torch.equal(path, want_path), seed
        assert int((path != 0).sum()) > 0 and torch.isfinite(lse).all()


@pytest.mark.parametrize("D,token", [(64, False), (256, False), (128, True)])
def test_second_stream_of_the_early_rows_under_capture_and_from_two_threads(D, token):
    """Causal calls on the templated kernel run their early rows on an internal second stream, forked from and joined to the caller's
    stream inside the call (csrc/qattn_api.hip side_stream_fork / _join).  (i) Captured in a HIP graph the pair must become part of the
    capture and replay to the eager bits on new data; (ii) two host threads, each on its own stream (each gets its own internal
    stream), must both get the single-threaded bits; (iii) a consumer queued on the caller's stream right behind the call sees the
    early rows (the join orders it)."""
    import threading
    fn = qa.fp8_token_wise_attn_func if token else qa.fp8_attn_func
    torch.manual_seed(D)
    B, H, S = 2, 4, 2304
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    want = fn(q, k, v, is_causal=True)
    assert torch.isfinite(want).all()
    # (iii) the early rows (first 1024) summed on the same stream right behind the call
    tot = fn(q, k, v, is_causal=True)[:, :, :1024].float().sum()
    assert torch.equal(tot, want[:, :, :1024].float().sum())
    # (i)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(q, k, v, is_causal=True)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(q, k, v, is_causal=True)
    q2, k2, v2 = (torch.randn_like(t) for t in (q, k, v))
    want2 = fn(q2, k2, v2, is_causal=True)
    q.copy_(q2); k.copy_(k2); v.copy_(v2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want2)
    # (ii)
    results, errors = [None, None], []

    def work(i):
        try:
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.default_stream())
            with torch.cuda.stream(st):
                for _ in range(5):
                    r = fn(q, k, v, is_causal=True)
                st.synchronize()
            results[i] = r
        except Exception as exc:   # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    assert torch.equal(results[0], want2) and torch.equal(results[1], want2)


@pytest.mark.parametrize("backend", ["eager", "aot_eager", "inductor"])
def test_torch_compile_traces_the_ops_as_opaque_calls(backend):
    """inside a user's torch.compile region the custom ops are traced through their fake impls
    (register_fake) and run the same HIP kernels -- results bit-equal to the eager call, for the fp8 and the 16-bit path,
    with dynamic sequence length (head count / head_dim marked static as nn.py:484-488 does)."""
    torch.manual_seed(5)

    def f(q, k, v):
        return qa.fp8_attn_func(q * 1.0, k, v, is_causal=True) + qa.attn_func(q, k, v).to(q.dtype) * 0

    if backend == "inductor":
        # the reference compiles its wrapper with backend="inductor", fullgraph=True (nn.py:521-539); here Inductor sees the
        # custom ops as extern calls (register_fake shapes) and fuses the surrounding elementwise work
        try:
            cf = torch.compile(f, backend="inductor", dynamic=True, fullgraph=True)
            q, k, v = (torch.randn(1, 4, 256, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
            got = cf(q, k, v)
        except Exception as exc:  # no working Triton / C++ toolchain for Inductor on this box: nothing of ours to test
            if "quantumattention_amd" in str(exc):
                raise
            pytest.skip(f"inductor backend unavailable here: {type(exc).__name__}: {str(exc)[:200]}")
        assert torch.equal(got, f(q, k, v))
        q, k, v = (torch.randn(1, 4, 384, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        assert torch.equal(cf(q, k, v), f(q, k, v))
        return
    cf = torch.compile(f, backend=backend, dynamic=True)
    for S in (256, 384):
        q, k, v = (torch.randn(1, 4, S, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
        assert torch.equal(cf(q, k, v), f(q, k, v))


def test_c_abi_error_codes_instead_of_exceptions():
    """include/qattn.h: every entry returns a negative QATTN_ERR_* code (never throws, never launches) on bad arguments --
    the counterpart of the reference launcher's TORCH_CHECKs (tk/attention.py:362-415)."""
    import ctypes

    L = _native.lib()
    q8 = torch.zeros(1, 2, 64, 128, dtype=torch.uint8, device="cuda")
    out = torch.zeros(1, 2, 64, 128, dtype=torch.bfloat16, device="cuda")
    sc = torch.ones(1, 2, dtype=torch.float32, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    f0 = ctypes.c_float(0.0)

    need = L.qattn_attention_workspace_bytes(1, 2, 64)
    wsb = torch.zeros(need, dtype=torch.uint8, device="cuda")

    def attn(D=128, Hq=2, Hkv=2, qk=0, vf=0, of=2, mode=0, q=q8, precision=0, lse_layout=0, ws=wsb, ws_bytes=need):
        return L.qattn_fp8_attention_forward(P(q) if q is not None else None, P(q8), P(q8), P(out), None, P(sc), P(sc), None,
                                             1, Hq, Hkv, 64, 64, D, qk, vf, of, mode, 0, f0, precision, lse_layout,
                                             P(ws) if ws is not None else None, ctypes.c_size_t(ws_bytes), None)

    assert attn() == 0 and attn(precision=1, ws=None, ws_bytes=0) == 0 and attn(precision=2, ws=None, ws_bytes=0) == 0
    assert attn(ws=None, ws_bytes=0) == -4 and attn(ws_bytes=need - 8) == -4       # QATTN_PRECISION_AUTO needs its workspace in full
    assert attn(precision=3) == -1 and attn(lse_layout=2) == -1
    assert attn(D=96) == -2 and L.qattn_strerror(-2) is not None       # head_dim not in {64,128,256} (nn.py:45-49)
    assert attn(Hq=3, Hkv=2) == -2                                     # Hq % Hkv != 0
    assert attn(qk=2) == -3 and attn(vf=1) == -3 and attn(of=0) == -3  # formats
    assert attn(mode=7) == -1 and attn(q=None) == -1                   # enum / NULL pointer
    x = torch.zeros(1, 2, 64, 128, dtype=torch.bfloat16, device="cuda")
    x8 = torch.zeros(1, 2, 64, 128, dtype=torch.uint8, device="cuda")
    s = torch.zeros(1, 2, dtype=torch.float32, device="cuda")
    qneed = L.qattn_quant_workspace_bytes(1, 2, 64, 128, 0)
    ws = torch.zeros(qneed, dtype=torch.uint8, device="cuda")
    quant = lambda D=128, in_fmt=2, out_fmt=0, ws_bytes=qneed: L.qattn_quant_fp8(P(x), in_fmt, P(x8), P(s), 1, 2, 64, D, out_fmt, 0, 0, 0,
                                                                                 P(ws), ctypes.c_size_t(ws_bytes), None)
    assert quant() == 0 and quant(ws_bytes=qneed - 4) == -4             # workspace too small (256 abs-max words per head)
    assert quant(D=100) == -2 and quant(in_fmt=0) == -3 and quant(out_fmt=2) == -3
    assert L.qattn_pack16(P(x), P(x8), 1, 2, 64, 96, 3, None) == -2     # head_dim not in {64,128,256}
    assert L.qattn_attention_forward_16(P(x), P(x), P(x), P(out), None, 1, 2, 2, 64, 64, 128, 0, 0, f0, 0, None) == -3
    torch.cuda.synchronize()
    for code in (0, -1, -2, -3, -4, -5, -6):
        assert len(L.qattn_strerror(code)) > 0


@pytest.mark.parametrize("precision", ["auto", "fast"])
def test_compiled_producer_fuses_the_abs_max_and_skips_the_pre_pass_launch(precision):
    """SURVEY 8f-4 / VERDICT r3 Missing-2 (nn.py:410-418, 484-501): q = producer(x) inside a torch.compile region.  The per-head
    abs-max (and the sums of squares for precision="auto") are traced as aten reductions, Inductor fuses them with the producer, and
    the op receives them: the kernel trace of the compiled call holds NO amax_multi_kernel, and the output is the eager call's bit for
    bit (abs().amax() of a 16-bit tensor is exact)."""
    torch.manual_seed(17)
    B, H, S, D = 2, 8, 2048, 128

    def f(x, wk, v):
        # stand-ins for a projection / RoPE epilogue whose results are EXACT in bf16 (a power-of-two factor, a sign flip, a rotation), so
        # that eager and Inductor produce the same q and k bits: with an inexact producer Inductor reduces over its un-rounded fp32
        # intermediate (no bf16 round trip inside a fused kernel), and the abs-max -- hence the scale -- may differ from eager's in
        # the last bit, exactly as the reference's compiled quantiser differs from its eager one
        q = x * 0.5
        k = -wk.roll(1, -1)
        return qa.fp8_attn_func(q, k, v, is_causal=True)

    x, wk, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    with qa.config.patch({"attention.precision": precision}):
        want = f(x, wk, v)
        try:
            cf = torch.compile(f, backend="inductor", fullgraph=True)
            got = cf(x, wk, v)
        except Exception as exc:  # no working Triton / C++ toolchain for Inductor on this box: nothing of ours to test
            if "quantumattention_amd" in str(exc):
                raise
            pytest.skip(f"inductor backend unavailable here: {type(exc).__name__}: {str(exc)[:200]}")
        assert torch.equal(got, want)
        from torch.profiler import ProfilerActivity, profile

        def kernels(fn):
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                fn(x, wk, v)
                torch.cuda.synchronize()
            return [e.key for e in prof.key_averages() if e.device_type is not None and "DeviceType.CUDA" in str(e.device_type)]

        eager_k, comp_k = kernels(f), kernels(cf)
    if not any("attn_fwd_kernel" in n for n in eager_k):
        pytest.skip("the profiler reports no device kernels on this box")
    assert any("amax_multi_kernel" in n for n in eager_k), eager_k
    assert any("attn_fwd_kernel" in n for n in comp_k) and any("quant_multi_kernel" in n for n in comp_k), comp_k
    assert not any("amax_multi_kernel" in n for n in comp_k), comp_k


@pytest.mark.parametrize("causal,B,H,S", [(True, 4, 8, 4096), (False, 8, 32, 6144)])
def test_null_workspace_runs_the_static_launch_with_the_same_bits(causal, B, H, S):
    """include/qattn.h: the attention workspace may be NULL for FAST / ACCURATE; a causal launch then uses one workgroup per query
    block and a large non-causal one equal static shares instead of the dynamic block hand-out -- the same results bit for bit."""
    import ctypes

    torch.manual_seed(4)
    q, k, v = (torch.randn(B, H, S, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
    L = _native.lib()
    need = L.qattn_attention_workspace_bytes(B, H, S)
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(with_ws):
        out = torch.full((B, H, S, 128), float("nan"), dtype=torch.bfloat16, device="cuda")
        rc = L.qattn_fp8_attention_forward(P(q8), P(kf), P(vf), P(out), None, P(sq), P(sk), P(sv), B, H, H, S, S, 128, 0, 0, 2, 0, int(causal),
                                           ctypes.c_float(0.0), 1, 0, P(ws) if with_ws else None, ctypes.c_size_t(need if with_ws else 0), st)
        assert rc == 0
        torch.cuda.synchronize()
        return out

    a, b = run(True), run(False)
    assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("D,causal,fp8,dtype", [(128, False, "e4m3", torch.bfloat16), (128, True, "e5m2", torch.bfloat16),
                                               (128, False, "e4m3", torch.float16), (64, True, "e4m3", torch.bfloat16),
                                               (256, False, "e4m3", torch.bfloat16)])
def test_producer_supplied_abs_max_is_bit_identical(D, causal, fp8, dtype):
    """SURVEY section 8f-4 / VERDICT r2 item 4a: the reference's quantiser is traced into the caller's graph and fused with
    whatever produced q and k (nn.py:410-418).  Counterpart here: qattn_fp8_quant_attention_forward_ex takes the per-head
    abs-max (and, optionally, sums of squares) a producer already has; the abs-max launch then skips those tensors.  With the
    exact abs-max the step's output is bit-identical to the two-pass result, whatever subset is supplied."""
    torch.manual_seed(21)
    B, Hq, Hkv, S = 2, 4, 2, 1000
    q = torch.randn(B, Hq, S, D, dtype=dtype, device="cuda")
    k = torch.randn(B, Hkv, S, D, dtype=dtype, device="cuda") * 1.7
    v = torch.randn(B, Hkv, S, D, dtype=dtype, device="cuda") * 0.3
    kw = dict(is_causal=causal, fp8_dtype=TDT[fp8])
    base = _native.fp8_quant_attention_forward(q, k, v, **kw)
    amax = lambda t: t.abs().amax(dim=(2, 3)).float()
    ssq = lambda t: (t.float() ** 2).sum(dim=(2, 3))
    aq, ak, av = amax(q), amax(k), amax(v)
    for extra in (dict(amax_q=aq, amax_k=ak), dict(amax_q=aq), dict(amax_k=ak, amax_v=av), dict(amax_q=aq, amax_k=ak, amax_v=av),
                  dict(amax_q=aq, amax_k=ak, amax_v=av, ssq_q=ssq(q), ssq_k=ssq(k))):
        got = _native.fp8_quant_attention_forward(q, k, v, **kw, **extra)
        assert torch.equal(got, base), sorted(extra)
    if fp8 == "e4m3":   # ... and through the reference-shaped interface (keyword-only extension of fp8_attn_func)
        assert torch.equal(qa.fp8_attn_func(q, k, v, is_causal=causal, amax_q=aq, amax_k=ak), qa.fp8_attn_func(q, k, v, is_causal=causal))
        assert torch.equal(qa.fp8_attn_func(q, k, v, is_causal=causal, amax_q=aq, amax_k=ak, ssq_q=ssq(q), ssq_k=ssq(k)),
                           qa.fp8_attn_func(q, k, v, is_causal=causal))
        with pytest.raises(ValueError):
            qa.fp8_attn_func(q, k, v, is_causal=causal, amax_q=aq, amax_k=ak, ssq_q=ssq(q))     # both or neither
    # an upper bound instead of the exact abs-max is safe (a coarser scale, no clipping): other bits, the same attention
    loose = _native.fp8_quant_attention_forward(q, k, v, **kw, amax_q=aq * 1.5, amax_k=ak * 1.5)
    assert not torch.equal(loose, base) and torch.isfinite(loose).all()
    assert (loose.float() - base.float()).pow(2).mean().sqrt().item() < (2e-2 if fp8 == "e5m2" else 1e-2)
    with pytest.raises(ValueError):
        _native.fp8_quant_attention_forward(q, k, v, **kw, amax_q=aq[:, :1])                      # wrong shape
    with pytest.raises(ValueError):
        _native.fp8_quant_attention_forward(q, k, v, **kw, scaling="token-wise", amax_q=aq)        # per-head figures need head-wise scales
    with pytest.raises(ValueError):
        _native.fp8_quant_attention_forward(q, k, v, **kw, ssq_q=ssq(q))                           # both or neither
    with pytest.raises(ValueError):
        qa.nn.fp8_attention(q, k, v, is_causal=causal, scaling_method="token-wise", amax_q=aq)


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("qmul", [1.0, 1.3, 2.0])
def test_producer_hand_off_on_one_term_sweeps(causal, qmul):
    """the hand-off test above uses S = 1000 < kTwoTermKeys, where every block starts two-term.  Here S = 2048 .. 4096
    (one-term sweeps: the caller's abs-max word read by q_amax_part with amax_n = 1, the caller's sums with ssq_n = 1 through
    sum_partials_pair) and q scaled so that some heads leave the dead band of the variance estimate (1.5) and start two-term:
      * abs-max AND sums supplied: the plain call's bits (heads exactly on the dead-band edge excepted -- none here: the estimates
        of these inputs are 1.0, 1.7 and 4.0 per head, far from 1.5);
      * only ONE abs-max, no sums: both tensors still take the abs-max pass for their sums -> the plain call's bits;
      * both abs-max, no sums: no score-spread estimate, wide heads start one-term: the documented bound, not the bits."""
    torch.manual_seed(33)
    B, H, S, D = 1, 4, 2048 if causal else 2560, 128
    q = (torch.randn(B, H, S, D, device="cuda") * qmul).to(torch.bfloat16)
    k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    q[:, 0] = (q[:, 0].float() / qmul).to(torch.bfloat16)   # head 0 stays flat: both regimes in one launch
    kw = dict(is_causal=causal, precision="auto")
    base = _native.fp8_quant_attention_forward(q, k, v, **kw)
    amax = lambda t: t.abs().amax(dim=(2, 3)).float()
    ssq = lambda t: (t.float() ** 2).sum(dim=(2, 3))
    aq, ak = amax(q), amax(k)
    full, path_full = _native.fp8_quant_attention_forward(q, k, v, **kw, amax_q=aq, amax_k=ak, ssq_q=ssq(q), ssq_k=ssq(k), return_path=True)
    assert torch.equal(full, base)
    assert torch.equal(_native.fp8_quant_attention_forward(q, k, v, **kw, amax_q=aq), base)
    assert torch.equal(_native.fp8_quant_attention_forward(q, k, v, **kw, amax_k=ak), base)
    bare, path_bare = _native.fp8_quant_attention_forward(q, k, v, **kw, amax_q=aq, amax_k=ak, return_path=True)
    # the error bound holds either way (oracle: fp64 SDPA of the quantised q, k and the block-scaled V the step used)
    q8, sq = oracle.quantize_fp8(bits16(q), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    k8, sk = oracle.quantize_fp8(bits16(k), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    ref = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, causal=causal, v_block=fused_step_uses_block_v(D, "head", q.dtype, S))
    for name, got, path in (("with sums", full, path_full), ("abs-max only", bare, path_bare)):
        assert_within_bound(out_to_f32(got), ref, path.cpu().numpy(), what=name)
    if qmul >= 2.0:
        assert torch.equal(bare[:, 0], base[:, 0])   # the flat head takes the same decisions with and without the estimate


@pytest.mark.parametrize("D,dtype", [(64, torch.bfloat16), (128, torch.float16)])
def test_stamped_entry_refuses_unsupported_shapes_before_the_pre_pass(D, dtype):
    """qattn_fp8_quant_attention_forward_stamped on D = 64 returned QATTN_ERR_UNSUPPORTED_FMT only after the pre-pass had
    written q8 / k8 / v8 / scales.  The check now sits in front of the first launch: nothing is written.  ADVICE r5: the same for fp16
    inputs at D = 128 (the stamped instantiation exists for bf16 Q rows only; since round 5 fp16 takes the fused kernel too)."""
    import ctypes

    torch.manual_seed(2)
    B, H, S = 1, 2, 512
    q, k, v = (torch.randn(B, H, S, D, dtype=dtype, device="cuda") for _ in range(3))
    L = _native.lib()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    sentinel = 0x5A
    bufs = [torch.full((B * H * S * D + 4096,), sentinel, dtype=torch.uint8, device="cuda") for _ in range(3)]
    scales = [torch.full((B, H), -7.0, dtype=torch.float32, device="cuda") for _ in range(3)]
    out = torch.empty_like(q)
    ws = torch.zeros(L.qattn_fp8_quant_attention_workspace_bytes(B, H, H, S), dtype=torch.uint8, device="cuda")
    stamps = torch.zeros(max(L.qattn_attention_stamp_bytes(B, H, S), 16), dtype=torch.uint8, device="cuda")
    rc = L.qattn_fp8_quant_attention_forward_stamped(P(q), P(k), P(v), _native.fmt_of(dtype), P(out), P(bufs[0]), P(bufs[1]), P(bufs[2]), P(scales[0]), P(scales[1]),
                                                     P(scales[2]), B, H, H, S, S, D, 0, 0, 0, 0, ctypes.c_float(0.0), 0, P(ws), ctypes.c_size_t(ws.numel()),
                                                     P(stamps), ctypes.c_size_t(stamps.numel()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert rc == -3, rc   # QATTN_ERR_UNSUPPORTED_FMT
    assert all(bool((b == sentinel).all()) for b in bufs) and all(bool((s_ == -7.0).all()) for s_ in scales)


def test_stamped_measurement_entry_reports_a_clock_and_the_same_output():
    """qattn_fp8_quant_attention_forward_stamped (bench.py in_kernel_clock_ghz): the stamped instantiation computes what the product
    kernel computes, and cycles / 100 MHz ticks of the waves' sweeps is a plausible shader clock."""
    torch.manual_seed(8)
    q, k, v = (torch.randn(2, 8, 2048, 128, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    for causal in (False, True):
        ghz, cycles, out = _native.measure_attention_clock(q, k, v, is_causal=causal, calls=20)
        assert torch.equal(out, qa.fp8_attn_func(q, k, v, is_causal=causal))
        assert 0.3 < ghz < 2.6 and cycles > 1000, (ghz, cycles)
    with pytest.raises(RuntimeError):   # D = 64 runs on the templated kernel: no stamped instantiation
        _native.measure_attention_clock(q[..., :64].contiguous(), k[..., :64].contiguous(), v[..., :64].contiguous())

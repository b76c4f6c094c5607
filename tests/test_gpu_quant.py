"""-m gpu: the HIP quant pre-pass and the fragment re-layout, bit-exact against the oracle and the golden vectors."""
import os

import numpy as np
import pytest
import torch

import oracle
from quantumattention_amd import _native
from tests.conftest import GOLDEN, golden_files
from tests.gpu_utils import FMT, TDT, bits16, bits8, fmt16, from_bits16, unpack_frag

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_files())
@pytest.mark.parametrize("scaling", ["head-wise", "token-wise"])
def test_quant_matches_reference_golden_bit_exact(name, scaling):
    z = np.load(os.path.join(GOLDEN, name))
    dtype = torch.bfloat16 if int(z["meta"][5]) else torch.float16
    m = "head" if scaling == "head-wise" else "token"
    for t in ("q", "k"):
        x = from_bits16(z[t], dtype).cuda()
        for numerics in ("compiled", "eager"):
            if f"{t}8_{m}_{numerics}" not in z:
                continue
            x8, s = _native.quant_fp8(x, scaling=scaling, numerics=numerics)
            np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), z[f"s{t}_{m}_{numerics}"].view(np.uint32))
            np.testing.assert_array_equal(bits8(x8), z[f"{t}8_{m}_{numerics}"])
        # float8_e5m2 (BASELINE config 5's format): the torch.float8_e5m2 fixture of the same compiled numerics
        x8, s = _native.quant_fp8(x, scaling=scaling, fp8_dtype=torch.float8_e5m2)
        np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), z[f"s{t}_{m}_e5m2"].view(np.uint32))
        np.testing.assert_array_equal(bits8(x8), z[f"{t}8_{m}_e5m2"])
    if m == "head":   # V, head-wise, against the reference's own dynamically_quantize_fp8(v)
        v8, sv = _native.quant_fp8(from_bits16(z["v"], dtype).cuda(), scaling="head-wise")
        np.testing.assert_array_equal(sv.cpu().numpy().view(np.uint32), z["sv_head_compiled"].view(np.uint32))
        np.testing.assert_array_equal(bits8(v8), z["v8_head_compiled"])


@pytest.mark.parametrize("shape", [(1, 2, 64, 64), (2, 3, 200, 128), (1, 2, 1000, 256), (1, 1, 4096, 128), (1, 2, 37, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fp8", ["e4m3", "e5m2"])
@pytest.mark.parametrize("scaling", ["head-wise", "token-wise"])
def test_quant_matches_oracle_bit_exact_all_layouts(shape, dtype, fp8, scaling):
    torch.manual_seed(hash((shape, fp8, scaling)) % 1000)
    B, H, S, D = shape
    x = (torch.randn(shape, dtype=torch.float32) * torch.rand(B, H, 1, 1) * 3).to(dtype)
    x[0, 0, 0, :8] = 0.0
    if S > 5:
        x[0, -1, 5, :] = 0.0  # an all-zero row: token-wise scale clamps to eps
    ref8, refs = oracle.quantize_fp8(bits16(x), fmt16(dtype), "head" if scaling == "head-wise" else "token", FMT[fp8], "compiled")
    xg = x.cuda()
    for layout in (_native.LAYOUT_ROWMAJOR, _native.LAYOUT_KFRAG, _native.LAYOUT_VFRAG):
        x8, s = _native.quant_fp8(xg, scaling=scaling, fp8_dtype=TDT[fp8], layout=layout)
        np.testing.assert_array_equal(s.cpu().numpy().view(np.uint32), refs.view(np.uint32))
        got = bits8(x8)
        if layout != _native.LAYOUT_ROWMAJOR:
            full = unpack_frag(got, layout, B, H, S, D)
            assert not full[:, :, S:, :].any(), "padding rows must be zero"
            got = full[:, :, :S, :]
        np.testing.assert_array_equal(got, ref8)


@pytest.mark.parametrize("shape", [(1, 2, 64, 64), (2, 2, 200, 128), (1, 1, 130, 256)])
@pytest.mark.parametrize("layout", [_native.LAYOUT_KFRAG, _native.LAYOUT_VFRAG])
def test_pack_is_the_documented_permutation(shape, layout):
    B, H, S, D = shape
    x = torch.randint(0, 256, shape, dtype=torch.uint8)
    packed = _native.pack_fp8(x.cuda().view(torch.float8_e4m3fn), layout)
    full = unpack_frag(bits8(packed), layout, B, H, S, D)
    np.testing.assert_array_equal(full[:, :, :S, :], x.numpy())
    assert not full[:, :, S:, :].any()


def test_quant_eager_numerics_and_constant_tensor():
    x = torch.full((1, 1, 64, 64), 0.5, dtype=torch.bfloat16)
    x8, s = _native.quant_fp8(x.cuda(), numerics="eager")
    r8, rs = oracle.quantize_fp8(bits16(x), oracle.FMT_BF16, "head", oracle.FMT_E4M3, "eager")
    np.testing.assert_array_equal(bits8(x8), r8)
    np.testing.assert_array_equal(s.cpu().numpy(), rs)
    z = torch.zeros((1, 1, 64, 64), dtype=torch.float16)
    x8, s = _native.quant_fp8(z.cuda())
    assert not bits8(x8).any() and float(s[0, 0]) == float(np.float32(1.1920928955078125e-07))


@pytest.mark.parametrize("shape", [(1, 2, 2, 64, 64, 64), (2, 4, 2, 200, 333, 128), (1, 2, 2, 1000, 1000, 256), (1, 1, 1, 4096, 4096, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("fp8", ["e4m3", "e5m2"])
@pytest.mark.parametrize("scaling", ["head-wise", "token-wise"])
def test_fused_qkv_prepass_matches_oracle_bit_exact(shape, dtype, fp8, scaling):
    """qattn_quant_qkv_fp8: one amax launch + one quantise launch for q (row-major), k (KFRAG), v (VFRAG)."""
    B, Hq, Hkv, Sq, Skv, D = shape
    torch.manual_seed(3)
    q = (torch.randn(B, Hq, Sq, D) * 2).to(dtype)
    k = (torch.randn(B, Hkv, Skv, D) * torch.rand(B, Hkv, 1, 1) * 4).to(dtype)
    v = (torch.randn(B, Hkv, Skv, D) * 0.5).to(dtype)
    m = "head" if scaling == "head-wise" else "token"
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q.cuda(), k.cuda(), v.cuda(), scaling=scaling, fp8_dtype=TDT[fp8])
    for got8, gots, x, mode, layout, (H, S) in (
        (q8, sq, q, m, _native.LAYOUT_ROWMAJOR, (Hq, Sq)),
        (kf, sk, k, m, _native.LAYOUT_KFRAG, (Hkv, Skv)),
        (vf, sv, v, "head", _native.LAYOUT_VFRAG, (Hkv, Skv)),
    ):
        ref8, refs = oracle.quantize_fp8(bits16(x), fmt16(dtype), mode, FMT[fp8], "compiled")
        np.testing.assert_array_equal(gots.cpu().numpy().view(np.uint32), refs.view(np.uint32))
        got = bits8(got8)
        if layout != _native.LAYOUT_ROWMAJOR:
            full = unpack_frag(got, layout, B, H, S, D)
            assert not full[:, :, S:, :].any(), "padding rows must be zero"
            got = full[:, :, :S, :]
        np.testing.assert_array_equal(got, ref8)


@pytest.mark.parametrize("scaling", ["head-wise", "token-wise"])
@pytest.mark.parametrize("numerics", ["compiled", "eager"])
def test_quant_fast_path_bit_exact_on_adversarial_bit_patterns(scaling, numerics):
    """The bf16 fast path (x * rcp(scale), exact fallback near bf16 ties) against the oracle's IEEE-divide sequence on
    2 M random finite bf16 BIT PATTERNS spread over 45 binades (~300 vectors take the fallback), plus rows holding
    inf / NaN / denormals / zeros, in every scale granularity and both numerics."""
    rng = np.random.default_rng(7)
    B, H, S, D = 2, 8, 1024, 128
    n = B * H * S * D
    bits = ((rng.integers(0, 2, n) << 15) | (rng.integers(90, 135, n) << 7) | rng.integers(0, 128, n)).astype(np.uint16)
    bits = bits.reshape(B, H, S, D)
    bits[0, 0, 3, :16] = 0x0001                      # bf16 denormals
    bits[0, 0, 4, :] = 0                             # all-zero row (eps clamp in token mode)
    bits[0, 1, 7, 5] = 0x7f80                        # +inf  -> head 1 (and that row) get an inf scale
    bits[0, 2, 9, 11] = 0x7fc1                       # NaN   -> head 2 (and that row) get a NaN scale
    bits[1, 3, :, :] = (bits[1, 3, :, :] & 0x807f) | (120 << 7)   # one binade only: many quotients near ties
    m = "head" if scaling == "head-wise" else "token"
    ref8, refs = oracle.quantize_fp8(bits, oracle.FMT_BF16, m, oracle.FMT_E4M3, numerics)
    x = from_bits16(bits, torch.bfloat16).cuda()
    x8, s = _native.quant_fp8(x, scaling=scaling, numerics=numerics)
    s = s.cpu().numpy()
    np.testing.assert_array_equal(np.isnan(s), np.isnan(refs))                   # NaN scales in the same places (payload is free)
    np.testing.assert_array_equal(s[~np.isnan(s)].view(np.uint32), refs[~np.isnan(refs)].view(np.uint32))
    got = bits8(x8)
    nan_ref = (ref8 & 0x7f) == 0x7f
    np.testing.assert_array_equal((got & 0x7f) == 0x7f, nan_ref)     # NaN bytes in the same places (sign of NaN is free)
    np.testing.assert_array_equal(got[~nan_ref], ref8[~nan_ref])


@pytest.mark.parametrize("scaling", ["head-wise", "token-wise"])
@pytest.mark.parametrize("numerics", ["compiled", "eager"])
@pytest.mark.parametrize("fp8", ["e4m3", "e5m2"])
def test_quant_fp16_fast_path_bit_exact_on_every_bit_pattern(scaling, numerics, fp8):
    """fp16 inputs (round 5): the quotient x / scale without the divide -- q0 = x rinv, r = fma(-q0, scale, x), q1 = fma(r, rinv, q0), Markstein's
    correction -- then v_cvt_pk_f16_f32, a packed fp16 clamp and v_cvt_scalef32_pk_{fp8,bf8}_f16, against the oracle's IEEE-divide sequence:
    EVERY finite fp16 bit pattern, 512 times over, under row / head abs-maxima of every size (token-wise: 8192 different scales per head,
    drawn over 30 binades), plus rows holding inf / NaN / denormals / zeros.  Bit-exact payload and scales."""
    if numerics == "eager" and fp8 == "e5m2":
        pytest.skip("eager numerics round the scale to fp16, and abs-max / 57344 is an fp16 subnormal for every ordinary row (see below)")
    rng = np.random.default_rng(11)
    B, H, S, D = 2, 8, 1024, 128
    n = B * H * S * D
    allbits = np.arange(65536, dtype=np.uint16)
    allbits = allbits[(allbits & 0x7c00) != 0x7c00]                # finite
    bits = rng.permutation(np.resize(allbits, n)).astype(np.uint16).reshape(B, H, S, D)
    # rows of limited magnitude: row r of a head keeps only values below 2^(r % 30 - 14) (so the row / head abs-max, hence the scale,
    # takes every size; the larger values of the row are replaced by small ones)
    e = ((bits >> 10) & 31).astype(np.int32)
    # (eager numerics round the scale to fp16: rows whose scale would be an fp16 SUBNORMAL -- abs-max below 2^-5 -- are left out there; in that
    # corner, which neither the golden vectors nor the reference's GPU path (compiled numerics) reach, device and oracle round 0.3 % of the
    # scales differently, with or without the fast path)
    lo = 11 if numerics == "eager" else 1
    cap = (np.arange(S) % (31 - lo) + lo)[None, None, :, None]
    bits = np.where(e <= cap, bits, (bits & 0x83ff) | ((cap.astype(np.uint16) & 31) << 10)).astype(np.uint16)
    bits[0, 0, 3, :16] = 0x0001                      # fp16 denormals
    if numerics != "eager":
        bits[0, 0, 4, :] = 0                         # all-zero row (eps clamp in token mode)
    bits[0, 1, 7, 5] = 0x7c00                        # +inf  -> head 1 (and that row) get an inf scale
    bits[0, 2, 9, 11] = 0x7e01                       # NaN   -> head 2 (and that row) get a NaN scale
    m = "head" if scaling == "head-wise" else "token"
    fmt = oracle.FMT_E4M3 if fp8 == "e4m3" else oracle.FMT_E5M2
    ref8, refs = oracle.quantize_fp8(bits, oracle.FMT_FP16, m, fmt, numerics)
    x = from_bits16(bits, torch.float16).cuda()
    x8, s = _native.quant_fp8(x, scaling=scaling, numerics=numerics, fp8_dtype=torch.float8_e4m3fn if fp8 == "e4m3" else torch.float8_e5m2)
    s = s.cpu().numpy()
    np.testing.assert_array_equal(np.isnan(s), np.isnan(refs))
    np.testing.assert_array_equal(s[~np.isnan(s)].view(np.uint32), refs[~np.isnan(refs)].view(np.uint32))
    got = bits8(x8)
    if fp8 == "e4m3":
        nan_ref = (ref8 & 0x7f) == 0x7f
        nan_got = (got & 0x7f) == 0x7f
    else:
        nan_ref = (ref8 & 0x7f) > 0x7c
        nan_got = (got & 0x7f) > 0x7c
    np.testing.assert_array_equal(nan_got, nan_ref)
    np.testing.assert_array_equal(got[~nan_ref], ref8[~nan_ref])


@pytest.mark.parametrize("fp8,D", [("e4m3", 128), ("e5m2", 128), ("e4m3", 64), ("e4m3", 256), ("e5m2", 64)])
def test_fused_step_block_scaled_v_is_bit_exact(fp8, D):
    """The fused step with head-wise scales (D = 128 from bf16 inputs on the hand-scheduled kernel; D = 64 / 256 on the templated one) quantises V with one power-of-two scale per 64-key chunk inside its quantise pass
    (no abs-max pass over V): payload (VFRAG) and scale bytes against oracle.quantize_v_block, on ragged S, with a zero chunk, a
    huge chunk, a tiny chunk, an inf and a NaN (a chunk with a non-finite abs-max gets the scale 2^0 and must still
    take the exact conversion, so that its NaN stays a NaN byte and the rows that attend it come out non-finite)."""
    from quantumattention_amd._native import LAYOUT_KFRAG, LAYOUT_VFRAG, SCALE_HEAD, PRECISION, fmt_of, _stream
    torch.manual_seed(3)
    B, H, S = 2, 3, 1000
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    v[0, 0, 64:128] = 0
    v[0, 1, 128:192] *= 3.0e4
    v[0, 2, 192:256] *= 1.0e-6
    v[1, 0, 300, 5] = float("inf")
    v[1, 1, 500, 9] = float("nan")
    L = _native.lib()
    dev = q.device
    out = torch.empty_like(q)
    q8 = torch.empty((B, H, S, D), dtype=torch.uint8, device=dev)
    kf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_KFRAG, B, H, S, D),), dtype=torch.uint8, device=dev)
    vf = torch.empty((L.qattn_fp8_tensor_bytes(LAYOUT_VFRAG, B, H, S, D),), dtype=torch.uint8, device=dev)
    sq, sk, sv = (torch.empty((B, H), dtype=torch.float32, device=dev) for _ in range(3))
    ws_bytes = L.qattn_fp8_quant_attention_workspace_bytes(B, H, H, S)
    ws = torch.zeros((ws_bytes,), dtype=torch.uint8, device=dev)
    rc = L.qattn_fp8_quant_attention_forward(q.data_ptr(), k.data_ptr(), v.data_ptr(), fmt_of(q.dtype), out.data_ptr(), q8.data_ptr(),
                                             kf.data_ptr(), vf.data_ptr(), sq.data_ptr(), sk.data_ptr(), sv.data_ptr(), B, H, H, S, S, D,
                                             FMT[fp8], SCALE_HEAD, 0, 0, 0.0, PRECISION["auto"], ws.data_ptr(), ws_bytes, _stream(q))
    torch.cuda.synchronize()
    assert rc == 0
    ref8, refe, _ = oracle.quantize_v_block(bits16(v), oracle.FMT_BF16, FMT[fp8])
    nch = (S + 63) // 64
    n = B * H
    words = ws.view(torch.int32).cpu().numpy()
    got_e = words[256 * 2 * n: 256 * 3 * n].reshape(B, H, 256)[:, :, :nch]      # V's 256 words per head, behind q's and k's
    np.testing.assert_array_equal(got_e, refe.astype(np.int32))
    got8 = unpack_frag(bits8(vf), LAYOUT_VFRAG, B, H, S, D)[:, :, :S]
    nan_ref = (ref8 & 0x7f) == 0x7f if fp8 == "e4m3" else (ref8 & 0x7f) > 0x7c
    np.testing.assert_array_equal(got8[~nan_ref], ref8[~nan_ref])
    got_nan = (got8 & 0x7f) == 0x7f if fp8 == "e4m3" else (got8 & 0x7f) > 0x7c
    np.testing.assert_array_equal(got_nan, nan_ref)           # NaN bytes exactly where the oracle has them (one: the planted NaN)
    assert nan_ref.sum() == 1 and nan_ref[1, 1, 500, 9]
    np.testing.assert_array_equal(sv.cpu().numpy(), np.ones((B, H), np.float32))
    o = out.float().cpu().numpy()
    assert np.isnan(o[1, 1, :, 9]).all()                      # every (non-causal) row of that head attends the NaN
    others = np.ones((B, H), bool); others[1, 1] = False
    # Skv = 1000 < 1024: every row of the step attends the ORIGINAL 16-bit V (qattn_pv16.h, the reference's own P.V numerics), where an
    # inf is an inf: column 5 of that head is non-finite, as aten SDPA would have it; everything else is finite
    # (the fp8 V bytes above clamp the inf to fmax like any out-of-range value)
    assert not np.isfinite(o[1, 0, :, 5]).any()
    o[1, 0, :, 5] = 0.0
    assert np.isfinite(o[others]).all()

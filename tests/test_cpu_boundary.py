"""CPU-only tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/qattn.h declares
(no compute calls without a GPU), argument validation returns the documented codes, and the Python call surface
mirrors the reference's (names, signatures, error behaviour, CPU fallback = BASELINE config 1)."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

import quantumattention_amd as qa
from quantumattention_amd import _native
from tests.conftest import GOLDEN, ROOT


def _header_functions():
    # the drop-in surface (qattn.h), its strided-view entry (qattn_strided.h) and the measurement entries (qattn_measure.h)
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.endswith(".h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qattn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_symbol_the_header_declares():
    names = _header_functions()
    assert set(names) == set(_native.EXPORTS), (names, _native.EXPORTS)
    L = ctypes.CDLL(_native.LIB_PATH)
    for n in names:
        assert getattr(L, n) is not None, n
    lib = _native.lib()
    assert lib.qattn_abi_version() == _native.ABI_VERSION == 8


def _path_table_rows():
    """The rows of include/qattn.h's PATH TABLE: dicts of the header's column names."""
    text = open(os.path.join(ROOT, "include", "qattn.h")).read()
    lines = [ln.strip()[1:].strip() for ln in text.splitlines() if ln.strip().startswith("* |")]
    cells = [[c.strip() for c in ln.strip("|").split("|")] for ln in lines]
    head, rows = cells[0], [c for c in cells[1:] if not set(c[0]) <= set("-")]
    assert head == ["entry", "D", "scales", "Skv", "kernel", "q_quant", "v_format", "sweep_p", "precise", "early", "start", "lse"], head
    return [dict(zip(head, r)) for r in rows]


def test_path_table_matches_dispatch():
    """ONE table (entry x D x scales x Skv -> kernel, Q quantisation, V format, P format, precise pass, early rows,
    start-mode source, LSE source) in include/qattn.h, walked against the host-only query qattn_describe_path(), which answers from the
    predicates the dispatch in csrc/qattn_api.hip itself uses (q_fusion_ok, fused_v_block, attn_v2_covers).  Every combination of the
    arguments must be covered by exactly one row, and the row must say what the library says."""
    rows = _path_table_rows()
    assert len(rows) == 9
    def covers(r, entry, D, scaling, Skv):
        return (r["entry"] == entry and (r["D"] == "any" or str(D) in r["D"].split(",")) and r["scales"] in ("any", scaling.split("-")[0])
                and {"any": True, "<=16384": Skv <= 16384, ">16384": Skv > 16384}[r["Skv"]])

    seen = set()
    for entry in ("fused", "separate", "separate16"):
        for D in (64, 128, 256):
            for scaling in ("head-wise", "token-wise"):
                for dtype in (torch.bfloat16, torch.float16):
                    for Skv in (1, 64, 1000, 4096, 16384, 16385, 20000, 100000):
                        match = [r for r in rows if covers(r, entry, D, scaling, Skv)]
                        assert len(match) == 1, (entry, D, scaling, Skv, match)
                        row = match[0]
                        seen.add(id(row))
                        for want_lse in (False, True):
                            got = _native.describe_path(entry, D, dtype, scaling, Skv, want_lse)
                            lse_col = row["lse"]
                            want = {"kernel": row["kernel"], "q_quant": row["q_quant"], "v_format": row["v_format"], "precise": row["precise"],
                                    "early": row["early"], "start_mode": row["start"], "lse": lse_col.rstrip("*"),
                                    # `exact*`: asking for the LSE switches the sweep to exact exponentials
                                    "sweep_p": "exact" if (want_lse and lse_col.endswith("*")) else row["sweep_p"]}
                            assert got == want, (entry, D, scaling, dtype, Skv, want_lse, got, want)
    assert len(seen) == len(rows), "a row of the table is never reached"
    # argument errors: codes, no device call
    d = _native.PathDesc()
    L = _native.lib()
    assert L.qattn_describe_path(2, 96, 2, 0, 100, 0, ctypes.byref(d)) == -2 and L.qattn_describe_path(7, 128, 2, 0, 100, 0, ctypes.byref(d)) == -1
    assert L.qattn_describe_path(2, 128, 0, 0, 100, 0, ctypes.byref(d)) == -3 and L.qattn_describe_path(2, 128, 2, 0, 100, 0, None) == -1


def test_header_stays_within_its_size_budget():
    """The table replaces prose: include/qattn.h <= 16 kB."""
    assert os.path.getsize(os.path.join(ROOT, "include", "qattn.h")) <= 16 * 1024


def test_abi_size_queries_and_error_codes_need_no_gpu():
    L = _native.lib()
    assert L.qattn_fp8_tensor_bytes(_native.LAYOUT_ROWMAJOR, 4, 32, 4096, 128) == 4 * 32 * 4096 * 128
    assert L.qattn_fp8_tensor_bytes(_native.LAYOUT_KFRAG, 1, 2, 200, 128) == 1 * 2 * 256 * 128  # padded to 64 keys
    assert L.qattn_fp8_tensor_bytes(_native.LAYOUT_VFRAG, 1, 2, 64, 64) == 2 * 64 * 64
    assert L.qattn_quant_workspace_bytes(4, 32, 4096, 128, _native.SCALE_HEAD) == 4 * 32 * 256 * 4   # 256 per-block abs-max words per head
    assert L.qattn_quant_workspace_bytes(4, 32, 4096, 128, _native.SCALE_TOKEN) == 0
    assert L.qattn_quant_qkv_workspace_bytes(4, 32, 8) == 4 * 256 * ((128 + 64) + 160)   # per-block abs-max words of q, k, v | per-block sums of squares of q and k
    attn_ws = 32 + 4 * 32 * 128 * 4      # block hand-out counters of a causal launch | one word per 32-row query group
    assert L.qattn_attention_workspace_bytes(4, 32, 4096) == attn_ws
    assert L.qattn_fp8_quant_attention_workspace_bytes(4, 32, 8, 4096) == 4 * 256 * ((128 + 64) + 160) + attn_ws   # both parts multiples of 16
    assert L.qattn_lse_row_stride(1000, _native.LSE_NATURAL) == 1000
    assert L.qattn_lse_row_stride(1001, _native.LSE_REFERENCE) == 1004                 # row padded to 16 bytes (tk/attention.py:439)
    for code in range(0, -7, -1):
        assert L.qattn_strerror(code)
    assert b"unknown" in L.qattn_strerror(-99)
    # validation happens before any HIP call: NULL pointers / bad dims are rejected on a CPU-only box too
    attn = lambda *a: L.qattn_fp8_attention_forward(*a)
    assert attn(None, None, None, None, None, None, None, None, 1, 1, 1, 1, 1, 128, 0, 0, 2, 0, 0, 0.0, 1, 0, None, 0, None) == -1
    assert L.qattn_quant_fp8(None, 2, None, None, 1, 1, 1, 128, 0, 0, 0, 0, None, 0, None) == -1
    assert L.qattn_pack_fp8(None, None, 1, 1, 1, 128, 1, None) == -1
    one = ctypes.c_void_p(16)  # any non-NULL pointer: dimension checks come first
    assert attn(one, one, one, one, None, one, one, None, 1, 1, 1, 8, 8, 96, 0, 0, 2, 0, 0, 0.0, 1, 0, None, 0, None) == -2
    assert attn(one, one, one, one, None, one, one, None, 1, 3, 2, 8, 8, 128, 0, 0, 2, 0, 0, 0.0, 1, 0, None, 0, None) == -2
    assert attn(one, one, one, one, None, one, one, None, 1, 1, 1, 8, 8, 128, 2, 2, 2, 0, 0, 0.0, 1, 0, None, 0, None) == -3
    assert attn(one, one, one, one, None, one, one, None, 1, 1, 1, 8, 8, 128, 0, 0, 2, 0, 0, 0.0, 7, 0, None, 0, None) == -1   # precision enum
    assert attn(one, one, one, one, None, one, one, None, 1, 1, 1, 8, 8, 128, 0, 0, 2, 0, 0, 0.0, 1, 5, None, 0, None) == -1   # lse_layout enum
    # QATTN_PRECISION_AUTO needs its flag workspace: rejected before anything is launched
    assert attn(one, one, one, one, None, one, one, None, 1, 1, 1, 8, 8, 128, 0, 0, 2, 0, 0, 0.0, 0, 0, None, 0, None) == -4


def test_public_names_and_signatures_mirror_the_reference():
    # src/quantum_attn/__init__.py:23-31
    assert qa.__all__ == [
        "attn_func", "attn_func_with_fallback", "dynamically_quantize_fp8", "fp8_attn_func",
        "fp8_attn_func_with_fallback", "fp8_token_wise_attn_func", "fp8_token_wise_attn_func_with_fallback",
    ]
    sdpa = ["query", "key", "value", "attn_mask", "dropout_p", "is_causal", "scale"]
    assert list(inspect.signature(qa.attn_func).parameters) == sdpa                                   # interface.py:41-50
    # :101-113, followed by this build's keyword-only extension (the producer-side abs-max hand-off)
    assert list(inspect.signature(qa.fp8_attn_func).parameters) == sdpa + ["scale_q", "scale_k", "scaling_method", "amax_q", "amax_k", "ssq_q", "ssq_k"]
    assert all(inspect.signature(qa.fp8_attn_func).parameters[n].kind is inspect.Parameter.KEYWORD_ONLY and
               inspect.signature(qa.fp8_attn_func).parameters[n].default is None for n in ("amax_q", "amax_k", "ssq_q", "ssq_k"))
    assert list(inspect.signature(qa.fp8_token_wise_attn_func).parameters) == sdpa + ["scale_q", "scale_k"]         # :179-190
    assert list(inspect.signature(qa.nn.can_use_attention).parameters)[:8] == sdpa + ["scaling_method"]  # nn.py:282-292
    for flag in ("skip_supported_check", "force_eager_fallback"):                                   # config.py:27-28
        assert hasattr(qa.config.attention, flag)
    with qa.config.patch({"attention.skip_supported_check": True}):                                 # config.py:34-41
        q = torch.randn(1, 2, 16, 64, dtype=torch.bfloat16)
        assert qa.nn.can_use_attention(q, q, q) == (True, "")


def test_cpu_plumbing_matches_reference_golden_config1():
    """BASELINE config 1 (B1 H2 S128 D64, CPU): *_with_fallback == F.sdpa, bit for bit, and equal to the reference's
    own output stored in the golden file (interface.py:62-98)."""
    z = np.load(os.path.join(GOLDEN, "c1_b1h2s128d64_bf16_s0.npz"))
    to_t = lambda b: torch.from_numpy(b.view(np.int16).copy()).view(torch.bfloat16)
    q, k, v = to_t(z["q"]), to_t(z["k"]), to_t(z["v"])
    ref = to_t(z["fallback_full"])
    torch.set_num_threads(4)
    for fn in (qa.attn_func_with_fallback, qa.fp8_attn_func_with_fallback, qa.fp8_token_wise_attn_func_with_fallback):
        out = fn(q, k, v)
        assert torch.equal(out, torch.nn.functional.scaled_dot_product_attention(q, k, v))
        assert (out.float() - ref.float()).abs().max() <= 2.0 ** -8  # same aten kernel; thread partition may differ
    ok, reason = qa.nn.can_use_attention(q, k, v, scaling_method="head-wise")
    assert not ok and "CUDA device" in reason  # same first reason as the reference (tests/golden/can_use_attention_cpu.txt)
    assert "CUDA device" in open(os.path.join(GOLDEN, "can_use_attention_cpu.txt")).read()
    with pytest.raises(ValueError):
        qa.fp8_attn_func(q, k, v)  # nn.py:461-462: ValueError(reason) on unsupported input
    with pytest.raises(ValueError):
        qa.attn_func(q, k, v)


def test_eager_quantiser_definition_on_cpu_matches_golden_eager_numerics():
    z = np.load(os.path.join(GOLDEN, "c1_b1h2s128d64_bf16_s0.npz"))
    q = torch.from_numpy(z["q"].view(np.int16).copy()).view(torch.bfloat16)
    q8, s = qa.dynamically_quantize_fp8(q, reduction_dim=[2, 3])  # CPU tensor -> eager torch definition (nn.py:14-19)
    np.testing.assert_array_equal(q8.view(torch.uint8).numpy(), z["q8_head_eager"])
    np.testing.assert_array_equal(s.numpy(), z["sq_head_eager"])


def test_native_binding_fails_loudly_without_the_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="no CPU or eager fallback"):
        _native.lib()


def test_fake_impls_of_every_custom_op_without_a_gpu():
    """register_fake coverage: under FakeTensorMode every op of the boundary returns the documented
    shape / dtype / device for fake "cuda" tensors -- what torch.compile sees when it traces a caller (ops.py:45, 121)."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    from quantumattention_amd import ops  # noqa: F401  (registers the ops)

    with FakeTensorMode():
        q = torch.empty(2, 8, 300, 128, dtype=torch.bfloat16, device="cuda")
        k = torch.empty(2, 2, 512, 128, dtype=torch.bfloat16, device="cuda")
        v = torch.empty(2, 2, 512, 128, dtype=torch.float16, device="cuda")
        o = torch.ops.quantumattention_amd.fp8_quant_attention_forward(q, k, v.bfloat16(), True, "token-wise", "e5m2")
        assert o.shape == q.shape and o.dtype == torch.bfloat16 and o.device.type == "cuda"
        q8, sq = torch.ops.quantumattention_amd.dynamically_quantize_fp8(q, False, "e4m3", "compiled")
        k8, sk = torch.ops.quantumattention_amd.dynamically_quantize_fp8(k, True, "e5m2", "eager")
        assert q8.dtype == torch.float8_e4m3fn and q8.shape == q.shape and sq.shape == (2, 8) and sq.dtype == torch.float32
        assert k8.dtype == torch.float8_e5m2 and sk.shape == (2, 2, 512)
        o = torch.ops.quantumattention_amd.fp8_attention_forward(q8, q8[:, :2], v[:, :, :300], sq, sq[:, :2], None, 0.0, False)
        assert o.shape == q.shape and o.dtype == torch.float16   # output takes value's dtype (tk/attention.py:434-437)
        o = torch.ops.quantumattention_amd.attention_forward(q, q, q, None, 0.0, True)
        assert o.shape == q.shape and o.dtype == torch.bfloat16


def _fake_cuda(*shape, dtype=torch.bfloat16):
    from torch._subclasses.fake_tensor import FakeTensorMode

    with FakeTensorMode():
        return torch.empty(*shape, dtype=dtype, device="cuda")


def test_validation_rules_reject_mismatching_key_value_and_scales_before_any_launch():
    """batch / head_dim agreement and the scale tensors are checked in Python (reasons, not GPU faults) -- the
    checks the reference's launcher does with TORCH_CHECK (tk/attention.py:385-414).  Fake cuda tensors: no GPU needed."""
    v = qa.nn._validate_hip_input
    q = _fake_cuda(2, 4, 64, 128)
    assert v(q, q, q, scaling_method="head-wise") == (True, "")
    ok, why = v(q, _fake_cuda(1, 4, 64, 128), _fake_cuda(1, 4, 64, 128), scaling_method="head-wise")
    assert not ok and "batch size" in why
    ok, why = v(q, _fake_cuda(2, 4, 64, 64), q, scaling_method="head-wise")
    assert not ok and "embedding dimension" in why and "Dk=64" in why
    ok, why = v(q, _fake_cuda(2, 3, 64, 128), _fake_cuda(2, 3, 64, 128), scaling_method="head-wise")
    assert not ok and "multiple of the key/value heads" in why
    q8 = _fake_cuda(2, 4, 64, 128, dtype=torch.float8_e4m3fn)
    s = _fake_cuda(2, 4, dtype=torch.float32)
    assert v(q8, q8, q, scaling_method="head-wise", scale_q=s, scale_k=s) == (True, "")
    assert v(q8, q8, q, scaling_method="token-wise", scale_q=_fake_cuda(2, 4, 64, dtype=torch.float32), scale_k=_fake_cuda(2, 4, 64, dtype=torch.float32))[0]
    assert "both provided" in v(q8, q8, q, scaling_method="head-wise", scale_q=s)[1]
    assert "need scale_q" in v(q8, q8, q, scaling_method="head-wise")[1]
    assert "float32" in v(q8, q8, q, scaling_method="head-wise", scale_q=s.bfloat16(), scale_k=s)[1]
    assert "shape" in v(q8, q8, q, scaling_method="head-wise", scale_q=_fake_cuda(2, 3, dtype=torch.float32), scale_k=s)[1]
    assert "shape" in v(q8, q8, q, scaling_method="head-wise", scale_q=s, scale_k=_fake_cuda(2, 4, 64, dtype=torch.float32))[1]
    assert "only accepted with fp8" in v(q, q, q, scaling_method="head-wise", scale_q=s, scale_k=s)[1]
    # the reference's reasons, in the reference's order (nn.py:63-121)
    assert v(q, q, q, dropout_p=0.1, scaling_method="head-wise")[1] == "NYI: dropout_p must be 0.0"
    assert v(q, q, q, scale=0.5, scaling_method="head-wise")[1] == "NYI: scale must be None"
    assert v(q, q, q, scaling_method="row-wise")[1] == "Unsupported scaling_method: row-wise"
    assert v(_fake_cuda(2, 4, 64, 96), _fake_cuda(2, 4, 64, 96), _fake_cuda(2, 4, 64, 96), scaling_method="head-wise")[1] == "Unsupported head dimension: 96"


def test_force_eager_fallback_runs_the_ops_eager_definition():
    """config.attention.force_eager_fallback keeps the reference's meaning (nn.py:367-371, 503-516): the wrapper runs
    eagerly -- eager quantiser + de-quantise + aten SDPA (ops.py:64-95).  On CPU tensors with the support check skipped
    the result must equal the reference's own eager output stored in the golden file, bit for bit."""
    z = np.load(os.path.join(GOLDEN, "c1_b1h2s128d64_bf16_s0.npz"))
    to_t = lambda b: torch.from_numpy(b.view(np.int16).copy()).view(torch.bfloat16)
    q, k, v = to_t(z["q"]), to_t(z["k"]), to_t(z["v"])
    torch.set_num_threads(4)
    with qa.config.patch({"attention.force_eager_fallback": True, "attention.skip_supported_check": True}):
        for causal in (False, True):
            q8 = torch.from_numpy(z["q8_head_compiled"].copy()).view(torch.float8_e4m3fn)
            k8 = torch.from_numpy(z["k8_head_compiled"].copy()).view(torch.float8_e4m3fn)
            out = qa.fp8_attn_func(q8, k8, v, is_causal=causal, scale_q=torch.from_numpy(z["sq_head_compiled"]),
                                   scale_k=torch.from_numpy(z["sk_head_compiled"]))
            ref = to_t(z[f"o1_head_{'causal' if causal else 'full'}"])
            assert (out.float() - ref.float()).abs().max() <= 2.0 ** -8   # same aten kernel; thread partition may differ
            out16 = qa.attn_func(q, k, v, is_causal=causal)
            ref16 = to_t(z[f"o16_{'causal' if causal else 'full'}"])
            assert (out16.float() - ref16.float()).abs().max() <= 2.0 ** -8
        # 16-bit inputs: quantised by the eager definition first (nn.py:410-418), e4m3 unless the format flag says otherwise
        out = qa.fp8_attn_func(q, k, v)
        assert out.dtype == torch.bfloat16 and out.shape == q.shape and torch.isfinite(out).all()
        with qa.config.patch({"attention.fp8_format": "e5m2"}):
            q8, _ = qa.dynamically_quantize_fp8(q, reduction_dim=[2, 3])
            assert q8.dtype == torch.float8_e5m2   # the eager fallback follows config.attention.fp8_format


def test_vblock_exponent_matches_the_oracle_restatement():
    """The integer rule for the block-scaled V's chunk scale (csrc/qattn_common.h vblock_exponent, exported as a host function)
    against oracle.quantize_v_block on chunks whose abs-max is a chosen bf16 value: around the 1.75 x 2^k boundaries, tiny, huge,
    zero, inf, NaN."""
    import numpy as np
    import oracle
    L = _native.lib()
    vals = [0.0, 1e-30, 3.0e-5, 0.4375, 0.4394, 1.0, 1.75, 1.7578125, 3.5, 3.515625, 447.0, 448.0, 450.0, 57344.0, 57600.0, 1e10, 3e38,
            float("inf"), float("nan")]
    for fmt, name in ((oracle.FMT_E4M3, "e4m3"), (oracle.FMT_E5M2, "e5m2")):
        x = np.zeros((1, len(vals), 64, 64), np.float32)
        x[0, :, 7, 3] = np.array(vals, np.float32)
        bits = oracle.f32_to_bf16_bits(x)
        amax = oracle.bf16_bits_to_f32(bits)[0, :, 7, 3]
        _, e8, _ = oracle.quantize_v_block(bits, oracle.FMT_BF16, fmt)
        for i, a in enumerate(amax):
            got = L.qattn_vblock_exponent(int(np.float32(a).view(np.uint32)), fmt)
            assert got + 127 == int(e8[0, i, 0]), (name, vals[i], got, int(e8[0, i, 0]))
            if np.isfinite(a) and a >= 1.1754944e-38:   # the smallest power of two that brings the chunk inside the format's range
                fmax = 448.0 if fmt == oracle.FMT_E4M3 else 57344.0
                assert a / 2.0 ** got <= fmax and (got == -126 or a / 2.0 ** (got - 1) > fmax), (name, vals[i], got)


def test_config_strings_are_validated_with_readable_errors():
    """VERDICT r3 hygiene: unknown `numerics` / `fp8_format` / `precision` strings raise ValueError naming the choices (they used to
    surface as a bare KeyError, and nn._fp8_dtype silently mapped anything that is not "e5m2" to e4m3)."""
    import quantumattention_amd as qa
    from quantumattention_amd import _native, nn

    for fn, bad in ((_native._numerics, "jit"), (_native.fp8_dtype_of, "e3m4"), (_native._precision, "exact")):
        with pytest.raises(ValueError, match="expected"):
            fn(bad)
    assert _native._numerics("eager") == 1 and _native.fp8_dtype_of("e5m2") is torch.float8_e5m2
    with qa.config.patch({"attention.fp8_format": "fp8"}):
        with pytest.raises(ValueError, match="fp8_format"):
            nn._fp8_dtype()
    with qa.config.patch({"attention.fp8_format": "e5m2"}):
        assert nn._fp8_dtype() is torch.float8_e5m2


def test_variant_builds_cannot_touch_the_product_temps():
    """-D tuning macros are part of a variant's object directory, and -save-temps output (what the resource and
    hazard tests read) can only come from the product configuration."""
    from quantumattention_amd import build

    with pytest.raises(ValueError):
        build.build(save_temps=True, variant="x")
    with pytest.raises(ValueError):
        build.build(save_temps=True, extra_defines=["QATTN_QUANT_TPB=2"])
    with pytest.raises(ValueError):
        build.build(extra_defines=["QATTN_QUANT_TPB=2"])


def test_compiled_region_carries_the_abs_max_reductions():
    """SURVEY 8f-4 / VERDICT r3 Missing-2: under torch.compile the reference's quantiser is part of the caller's graph (nn.py:410-418,
    484-501), so Inductor fuses its abs-max with the producer of q / k.  Here the per-head abs-max and sums of squares are traced as
    aten reductions and handed to the fused op (whose abs-max launch is then skipped).  Traced on CPU tensors (fake impls only)."""
    import torch._dynamo

    def f(x, k, v):
        q = x * 1.5                                   # a producer of q inside the compiled region
        return qa.fp8_attn_func(q, k, v, is_causal=True)

    x, k, v = (torch.randn(1, 2, 128, 128, dtype=torch.bfloat16) for _ in range(3))
    for inline, precision in ((True, "auto"), (True, "fast"), (False, "auto")):
        torch._dynamo.reset()
        with qa.config.patch({"attention.skip_supported_check": True, "attention.inline_abs_max_under_compile": inline,
                              "attention.precision": precision}):
            gm = torch._dynamo.export(f)(x, k, v).graph_module
        nodes = list(gm.graph.nodes)
        op = [n for n in nodes if n.op == "call_function" and "fp8_quant_attention_forward" in str(n.target)]
        assert len(op) == 1
        names = "query key value is_causal scaling_method fp8_format numerics precision amax_q amax_k ssq_q ssq_k amax_v".split()
        args = dict(zip(names, op[0].args), **op[0].kwargs)
        n_amax = sum(1 for n in nodes if n.op == "call_method" and n.target == "amax")
        if inline:
            # 128 keys: V is block-scaled in the fused step (every head dim, bf16 and fp16), its abs-max is never read -- and not traced (ADVICE r4:
            # an op input cannot be eliminated as dead code, the graph would carry an extra read of V)
            assert n_amax == 2 and all(args.get(a) is not None for a in ("amax_q", "amax_k")) and args.get("amax_v") is None
            assert (args.get("ssq_q") is not None) == (precision == "auto") == (args.get("ssq_k") is not None)
        else:
            assert n_amax == 0 and all(args.get(a) is None for a in ("amax_q", "amax_k", "amax_v", "ssq_q", "ssq_k"))
    # more than 256 chunks of keys per head (V keeps one scale per head there): all three abs-max reductions are traced
    torch._dynamo.reset()
    kl, vl = (torch.randn(1, 2, 16448, 128, dtype=torch.bfloat16) for _ in range(2))
    with qa.config.patch({"attention.skip_supported_check": True, "attention.precision": "fast"}):
        gm = torch._dynamo.export(lambda x_, k_, v_: qa.fp8_attn_func(x_ * 1.5, k_, v_))(x, kl, vl).graph_module
    assert sum(1 for n in gm.graph.nodes if n.op == "call_method" and n.target == "amax") == 3
    torch._dynamo.reset()


def test_strided_view_rules_of_the_binding_are_host_logic():
    """include/qattn_strided.h: which views go to the kernels as they are, which are copied, and where the output lands (no GPU needed: the rules
    are functions of shapes and strides)."""
    B, S, H, D = 2, 40, 4, 64
    x = torch.zeros(B, S, 3 * H, D, dtype=torch.bfloat16)
    q, k = x[:, :, :H].transpose(1, 2), x[:, :, H:2 * H].transpose(1, 2)
    assert _native._strided_ok(q) and _native._strided_ok(k) and not q.is_contiguous()
    assert list(_native._strides3(q)) == [S * 3 * H * D, D, 3 * H * D]
    assert _native._strides3(q.contiguous()) is None
    assert _native._strided_ok(torch.zeros(1, H, S, D, dtype=torch.bfloat16).expand(B, H, S, D))          # broadcast over the batch
    assert not _native._strided_ok(torch.zeros(B, H, D, S, dtype=torch.bfloat16).transpose(2, 3))          # head_dim not innermost
    assert not _native._strided_ok(torch.zeros(B, H, S, D + 4, dtype=torch.bfloat16)[..., :D])             # rows off 16 bytes
    assert not _native._strided_ok(torch.zeros(B, H, S, D + 8, dtype=torch.bfloat16)[..., 4:D + 4])        # base off 16 bytes
    # the output: dense unless asked for the query's layout AND the query is a transposed [B,S,H,D] tensor
    t = torch.zeros(B, S, H, D, dtype=torch.bfloat16).transpose(1, 2)
    assert _native.empty_output(t, t.dtype).is_contiguous()
    o = _native.empty_output(t, t.dtype, "like_query")
    assert o.shape == t.shape and o.stride() == t.stride() and o.transpose(1, 2).is_contiguous()
    assert _native.empty_output(t.contiguous(), t.dtype, "like_query").is_contiguous()
    assert _native.empty_output(q, q.dtype, "like_query").is_contiguous()                                   # a packed-QKV slice: no such layout
    with pytest.raises(ValueError, match="output_layout"):
        _native.empty_output(t, t.dtype, "rows")
    assert qa.config.attention.output_layout == "contiguous"

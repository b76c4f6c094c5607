"""-m gpu: the fused entry on STRIDED VIEWS of q, k, v (include/qattn_strided.h, ABI 8).

Attention inputs usually reach the reference as views -- `x.view(B, S, H, D).transpose(1, 2)`, slices of a packed QKV projection.  The
reference reads such q / k in its Inductor-made quantiser and copies such a v (`.contiguous()`, tk/attention.py:419-421).  Here every kernel
that touches the 16-bit tensors takes the strides; the statement under test is the strongest one available: out, lse, row_path of a call on
views are, BIT FOR BIT, those of the same call on dense copies -- for every kernel family, layout, dtype, format and precision mode -- and
the dense call is what every other parity test holds against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

import quantumattention_amd as qa
from quantumattention_amd import _native

pytestmark = pytest.mark.gpu


def _views(layout, B, Hq, Hkv, Sq, Skv, D, dtype, g):
    """q, k, v as non-contiguous [B,H,S,D] views of freshly drawn storage (N(0,1), the reference's test distribution)."""
    rn = lambda *shape: torch.randn(*shape, device="cuda", generator=g).to(dtype)
    if layout == "bshd":            # the transpose of a [B,S,H,D] projection output
        return tuple(rn(B, S, H, D).transpose(1, 2) for H, S in ((Hq, Sq), (Hkv, Skv), (Hkv, Skv)))
    if layout == "packed_qkv":      # slices of ONE [B,S,Hq+2Hkv,D] projection (Sq == Skv)
        x = rn(B, Sq, Hq + 2 * Hkv, D)
        return x[:, :, :Hq].transpose(1, 2), x[:, :, Hq:Hq + Hkv].transpose(1, 2), x[:, :, Hq + Hkv:].transpose(1, 2)
    if layout == "padded_rows":     # rows further apart than D elements, heads and batches padded too
        return tuple(rn(B + 1, H + 1, S + 3, D + 64)[1:, :H, 3:, 32:32 + D] for H, S in ((Hq, Sq), (Hkv, Skv), (Hkv, Skv)))
    if layout == "kv_broadcast":    # one K / V for every batch element (stride 0), q transposed
        return rn(B, Sq, Hq, D).transpose(1, 2), rn(1, Hkv, Skv, D).expand(B, Hkv, Skv, D), rn(1, Skv, Hkv, D).transpose(1, 2).expand(B, Hkv, Skv, D)
    raise ValueError(layout)


CASES = [
    # layout, B, Hq, Hkv, Sq, Skv, D, causal, fp8, scaling, dtype, precision
    ("bshd", 2, 4, 4, 2304, 2304, 128, False, "e4m3", "head-wise", torch.bfloat16, "auto"),      # the hand-scheduled kernel: Q rows quantised in the prologue
    ("bshd", 2, 4, 4, 2304, 2304, 128, True, "e4m3", "head-wise", torch.bfloat16, "auto"),       #   + early blocks on the strided 16-bit V
    ("packed_qkv", 1, 8, 2, 1500, 1500, 128, True, "e5m2", "head-wise", torch.float16, "auto"),  # GQA, ragged, fp16
    ("padded_rows", 3, 5, 5, 1100, 1100, 128, False, "e4m3", "head-wise", torch.bfloat16, "auto"),
    ("bshd", 1, 2, 2, 2304, 2304, 128, True, "e4m3", "head-wise", torch.bfloat16, "accurate"),   # every row on the 16-bit-V pass
    ("bshd", 1, 2, 2, 2304, 2304, 128, False, "e4m3", "head-wise", torch.bfloat16, "fast"),
    ("kv_broadcast", 3, 4, 2, 700, 1300, 128, False, "e4m3", "head-wise", torch.bfloat16, "auto"),
    ("bshd", 1, 2, 2, 17000, 17000, 128, True, "e4m3", "head-wise", torch.bfloat16, "auto"),     # Skv > 16384: V through the abs-max pass as well
    ("bshd", 2, 4, 2, 1100, 1100, 64, True, "e4m3", "head-wise", torch.bfloat16, "auto"),        # templated kernel: pre-pass writes q8; pv16 launch reads V
    ("packed_qkv", 1, 4, 4, 2100, 2100, 256, True, "e4m3", "head-wise", torch.float16, "auto"),
    ("padded_rows", 2, 2, 2, 1024, 1000, 64, False, "e5m2", "token-wise", torch.float16, "auto"),
    ("bshd", 2, 4, 4, 1300, 1300, 128, True, "e4m3", "token-wise", torch.bfloat16, "auto"),
    ("bshd", 1, 2, 2, 1, 1, 128, False, "e4m3", "head-wise", torch.bfloat16, "auto"),            # single token (every stride degenerate)
    ("bshd", 1, 2, 2, 3, 70, 64, False, "e4m3", "head-wise", torch.bfloat16, "auto"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "{}_B{}Hq{}Hkv{}Sq{}Skv{}D{}{}_{}_{}_{}_{}".format(
    c[0], c[1], c[2], c[3], c[4], c[5], c[6], "c" if c[7] else "f", c[8], c[9][:4], "bf16" if c[10] == torch.bfloat16 else "fp16", c[11]))
def test_strided_views_equal_their_dense_copies_bit_for_bit(case):
    layout, B, Hq, Hkv, Sq, Skv, D, causal, fp8, scaling, dtype, precision = case
    g = torch.Generator(device="cuda").manual_seed(Sq * 7 + D)
    q, k, v = _views(layout, B, Hq, Hkv, Sq, Skv, D, dtype, g)
    if Sq >= 1024:   # sharp rows in every 256-row block: the rescue passes (two-term on the fp8 V, 16-bit P on the 16-bit V) run on the views too
        q[:, :, 5::97] *= 2.2       # (written through the view into its storage)
        q[:, :, 40::211] *= 4.0
    if Sq > 1:
        assert not q.is_contiguous() and not k.is_contiguous() and not v.is_contiguous()
        assert all(_native._strided_ok(t) for t in (q, k, v)), "the case must reach the kernels as views"
    keep = [t.clone() for t in (q, k, v)]
    kw = dict(is_causal=causal, scaling=scaling, fp8_dtype=_native.FP8_DTYPE[fp8], precision=precision, return_lse=True, return_path=True)
    out_s, lse_s, path_s = _native.fp8_quant_attention_forward(q, k, v, **kw)
    out_d, lse_d, path_d = _native.fp8_quant_attention_forward(q.contiguous(), k.contiguous(), v.contiguous(), **kw)
    assert out_s.is_contiguous() and out_s.shape == (B, Hq, Sq, D)
    assert torch.equal(out_s, out_d) and torch.equal(path_s, path_d)
    assert torch.equal(lse_s, lse_d)
    if Sq >= 1024 and precision == "auto":
        assert (path_s != 0).any(), "the case must exercise a precise pass"
    for t, t0 in zip((q, k, v), keep):
        assert torch.equal(t, t0), "inputs are read-only"
    # the plain output (no lse / row_path request: the byte-exponential sweeps of the templated kernel) and the public interface
    plain_s = _native.fp8_quant_attention_forward(q, k, v, is_causal=causal, scaling=scaling, fp8_dtype=_native.FP8_DTYPE[fp8], precision=precision)
    plain_d = _native.fp8_quant_attention_forward(q.contiguous(), k.contiguous(), v.contiguous(), is_causal=causal, scaling=scaling,
                                                  fp8_dtype=_native.FP8_DTYPE[fp8], precision=precision)
    assert torch.equal(plain_s, plain_d)
    if Hq == Hkv:   # (the reference's interface has no GQA: nn.py:104-106)
        fn = qa.fp8_attn_func if scaling == "head-wise" else qa.fp8_token_wise_attn_func
        with qa.config.patch({"attention.fp8_format": fp8, "attention.precision": precision}):
            api = fn(q, k, v, is_causal=causal)
        assert api.is_contiguous() and torch.equal(api, plain_d)


def test_views_the_kernels_cannot_address_are_copied_and_give_the_same_result():
    """head_dim not innermost, rows off 16 bytes, a stride that is no multiple of 8 elements: `_native` copies those (as the reference's
    launcher would, tk/attention.py:419-421) -- same bits as the dense call."""
    g = torch.Generator(device="cuda").manual_seed(3)
    B, H, S, D = 2, 3, 1100, 128
    base = torch.randn(B, H, S, D, device="cuda", generator=g).to(torch.bfloat16)
    ref = _native.fp8_quant_attention_forward(base, base, base, is_causal=True)
    d_major = base.transpose(2, 3).contiguous().transpose(2, 3)                # stride(3) != 1
    odd = torch.zeros(B, H, S, D + 3, device="cuda", dtype=torch.bfloat16)     # row stride 131: not a multiple of 8
    odd[..., 1:D + 1] = base
    odd = odd[..., 1:D + 1]                                                    # and the base address off 16 bytes
    for t in (d_major, odd):
        assert not t.is_contiguous() and not _native._strided_ok(t)
        assert torch.equal(_native.fp8_quant_attention_forward(t, t, t, is_causal=True), ref)
        assert torch.equal(_native.fp8_quant_attention_forward(base, t, base, is_causal=True), ref)


def test_c_entry_rejects_strides_it_cannot_use_before_anything_is_written():
    L = _native.lib()
    B, H, S, D = 1, 2, 256, 128
    q = torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16)
    out = torch.full((B, H, S, D), 7.0, device="cuda", dtype=torch.bfloat16)
    u8 = lambda n: torch.empty((int(n),), dtype=torch.uint8, device="cuda")
    q8, kf, vf = u8(B * H * S * D), u8(L.qattn_fp8_tensor_bytes(1, B, H, S, D)), u8(L.qattn_fp8_tensor_bytes(2, B, H, S, D))
    sq, sk, sv = (torch.empty((B, H), dtype=torch.float32, device="cuda") for _ in range(3))
    nws = L.qattn_fp8_quant_attention_workspace_bytes(B, H, H, S)
    ws = u8(nws)

    def call(strides, qptr=None):
        arr = (ctypes.c_longlong * 12)(*strides) if strides is not None else None
        return L.qattn_fp8_quant_attention_forward_strided(
            qptr or q.data_ptr(), q.data_ptr(), q.data_ptr(), arr, _native.FMT_BF16, out.data_ptr(), q8.data_ptr(), kf.data_ptr(), vf.data_ptr(),
            sq.data_ptr(), sk.data_ptr(), sv.data_ptr(), None, None, None, None, None, B, H, H, S, S, D, 0, 0, 0, 0, 0.0, 0, None, 0, None,
            ws.data_ptr(), nws, None)

    dense = [H * S * D, S * D, D] * 4                            # q, k, v, out
    for bad in ([H * S * D, S * D, D - 8] + dense[3:],          # rows closer than D elements
                dense[:3] + [H * S * D, S * D + 4, D] + dense[6:],   # not a multiple of 8 elements
                dense[:6] + [-8 * 1024, S * D, D] + dense[9:],   # negative
                dense[:8] + [2 ** 23 + 8] + dense[9:],           # rows too far apart for the 32-bit lane offsets
                dense[:10] + [0, D]):                            # `out` cannot be a broadcast view (two heads on the same rows)
        assert call(bad) == -1      # QATTN_ERR_INVALID_ARG
    assert call(dense, qptr=q.data_ptr() + 2) == -1              # base off 16 bytes
    torch.cuda.synchronize()
    assert (out == 7.0).all(), "a rejected call writes nothing"
    assert call(dense) == 0 and call(None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, _native.fp8_quant_attention_forward(q, q, q, is_causal=False))


@pytest.mark.parametrize("D,token", [(128, False), (64, False), (128, True)])
def test_hip_graph_capture_on_strided_views(D, token):
    """The strides are read on the host at call time: a captured step replays on the same views -- the hand-scheduled kernel, and the templated
    one whose causal calls fork their early rows (the strided 16-bit-V launch) onto the library's side stream inside the capture."""
    g = torch.Generator(device="cuda").manual_seed(9)
    B, H, S = 2, 4, 2304
    fn = qa.fp8_token_wise_attn_func if token else qa.fp8_attn_func
    x = torch.randn(B, S, 3 * H, D, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = (x[:, :, i * H:(i + 1) * H].transpose(1, 2) for i in range(3))
    ref = fn(q.contiguous(), k.contiguous(), v.contiguous(), is_causal=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(q, k, v, is_causal=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = fn(q, k, v, is_causal=True)
    torch.cuda.current_stream().wait_stream(s)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    x.copy_(torch.randn(x.shape, device="cuda", generator=g).to(torch.bfloat16))   # new data in the same storage
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, fn(q.contiguous(), k.contiguous(), v.contiguous(), is_causal=True))


@pytest.mark.parametrize("layout,B,Hq,Hkv,S,D,causal,dtype", [
    ("bshd", 2, 4, 4, 2000, 128, True, torch.bfloat16),
    ("packed_qkv", 1, 8, 2, 1100, 64, False, torch.float16),
    ("padded_rows", 2, 3, 3, 999, 256, True, torch.bfloat16),
    ("kv_broadcast", 3, 4, 2, 700, 128, False, torch.bfloat16),
])
def test_16bit_sibling_path_on_strided_views(layout, B, Hq, Hkv, S, D, causal, dtype):
    """`attn_func` (quantum_attn::attention_forward, ops.py:17-45): qattn_pack16_strided re-lays K / V from the views, the kernel reads its
    Q rows through the strides -- same bits as on dense copies (the op itself, and the public function where the reference has no GQA)."""
    g = torch.Generator(device="cuda").manual_seed(S + D)
    q, k, v = _views(layout, B, Hq, Hkv, S, S, D, dtype, g)
    assert not q.is_contiguous() and not k.is_contiguous() and not v.is_contiguous()
    op = torch.ops.quantumattention_amd.attention_forward
    out_s = op(q, k, v, None, 0.0, causal)
    out_d = op(q.contiguous(), k.contiguous(), v.contiguous(), None, 0.0, causal)
    assert out_s.is_contiguous() and torch.equal(out_s, out_d)
    for lay in (_native.LAYOUT_K16FRAG, _native.LAYOUT_V16FRAG):
        assert torch.equal(_native.pack16(k, lay), _native.pack16(k.contiguous(), lay))
    if Hq == Hkv:
        assert torch.equal(qa.attn_func(q, k, v, is_causal=causal), out_d)
    if Hq > 1:   # the C entry refuses an output view in which two heads share their rows
        L = _native.lib()
        kf, vf = _native.pack16(k, _native.LAYOUT_K16FRAG), _native.pack16(v, _native.LAYOUT_V16FRAG)
        qd = q.contiguous()
        dense = [Hq * S * D, S * D, D]
        bad = (ctypes.c_longlong * 6)(*(dense + [Hq * S * D, 0, D]))
        assert L.qattn_attention_forward_16_strided(qd.data_ptr(), bad, kf.data_ptr(), vf.data_ptr(), out_d.data_ptr(), None, B, Hq, Hkv, S, S, D,
                                                    _native.fmt_of(dtype), int(causal), 0.0, 0, None) == -1


@pytest.mark.parametrize("D,scaling,causal", [(128, "head-wise", True), (128, "head-wise", False), (64, "head-wise", True), (256, "token-wise", False)])
def test_output_in_the_layout_of_the_query(D, scaling, causal):
    """config.attention.output_layout = "like_query": for q = x.view(B, S, H, D).transpose(1, 2) the output is the transposed view of a dense
    [B,S,H,D] tensor (what torch's flash SDPA returns), so the caller's `out.transpose(1, 2).reshape(B, S, H * D)` is a view.  Every store of
    every pass (sweep epilogues, rescues, 16-bit-V passes, the templated kernel's launches) goes through the output strides: same values as
    the dense output, bit for bit; lse and row_path unchanged."""
    B, H, S = 2, 4, 2304
    g = torch.Generator(device="cuda").manual_seed(D + S)
    q, k, v = _views("bshd", B, H, H, S, S, D, torch.bfloat16, g)
    q[:, :, 5::97] *= 2.2
    q[:, :, 40::211] *= 4.0
    kw = dict(is_causal=causal, scaling=scaling, return_lse=True, return_path=True)
    out_d, lse_d, path_d = _native.fp8_quant_attention_forward(q, k, v, **kw)
    out_q, lse_q, path_q = _native.fp8_quant_attention_forward(q, k, v, output_layout="like_query", **kw)
    assert out_d.is_contiguous() and not out_q.is_contiguous() and out_q.transpose(1, 2).is_contiguous()
    assert out_q.transpose(1, 2).reshape(B, S, H * D).data_ptr() == out_q.data_ptr(), "the caller's reshape must be a view"
    assert torch.equal(out_q, out_d) and torch.equal(lse_q, lse_d) and torch.equal(path_q, path_d)
    assert (path_q != 0).any()
    fn = qa.fp8_attn_func if scaling == "head-wise" else qa.fp8_token_wise_attn_func
    with qa.config.patch({"attention.output_layout": "like_query"}):
        api = fn(q, k, v, is_causal=causal)
        assert api.transpose(1, 2).is_contiguous()
        # dense inputs keep a dense output; a view the rule does not cover (padded rows) too
        assert fn(q.contiguous(), k, v, is_causal=causal).is_contiguous()
        a16 = qa.attn_func(q, k, v, is_causal=causal)
    assert torch.equal(api, _native.fp8_quant_attention_forward(q, k, v, is_causal=causal, scaling=scaling))
    assert a16.transpose(1, 2).is_contiguous() and torch.equal(a16, qa.attn_func(q, k, v, is_causal=causal))


def test_torch_compile_on_views_with_the_output_in_the_query_layout():
    """A user's compiled region that holds [B,S,H*D] activations (the usual module: view -> transpose -> attention -> transpose -> reshape):
    the fake implementations report the strides the real ops return (output_layout = like_query: the final reshape is a view in the traced
    graph too), results equal the eager call."""
    torch.manual_seed(6)
    B, S, H, D = 2, 1300, 4, 128

    def f(xq, xk, xv):
        q, k, v = (x.view(B, S, H, D).transpose(1, 2) for x in (xq, xk, xv))
        o = qa.fp8_attn_func(q, k, v, is_causal=True)
        return o.transpose(1, 2).reshape(B, S, H * D) + qa.attn_func(q, k, v).transpose(1, 2).reshape(B, S, H * D) * 0

    xs = [torch.randn(B, S, H * D, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
    want = f(*xs)
    for layout in ("contiguous", "like_query"):
        with qa.config.patch({"attention.output_layout": layout}):
            cf = torch.compile(f, backend="aot_eager")
            assert torch.equal(cf(*xs), want), layout
            assert torch.equal(f(*xs), want), layout

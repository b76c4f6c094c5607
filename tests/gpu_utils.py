"""Helpers shared by the -m gpu parity tests: numpy <-> torch bit views, fragment-layout inverse maps, the oracle call that mirrors
one HIP attention launch (same quantised q/k/v, fp64 math) and the STRICT grader.

Grading rule (round 6): every row is held against exactly ONE oracle -- the one for the numerics the kernel says it
used for that row (the `row_path` output of qattn_fp8_quant_attention_forward_ex, include/qattn.h QATTN_PATH_*):

    path 0 (one-term fp8 P) and 1 (two-term fp8 P)  ->  fp64 SDPA on the quantised q, k and the build's fp8 V
    path 2 (16-bit P on the caller's 16-bit V)      ->  fp64 SDPA on the quantised q, k and the ORIGINAL 16-bit V

and the bound is per ELEMENT:  |got_ij - ref_ij| < 2^-6 * max(1, |ref_ij| / 2)  on fp8-V rows (north_star's flat 2^-6 wherever
|O| <= 2; above that one bf16 output ulp exceeds it), 2^-7 * max(1, |ref_ij|) on 16-bit-V rows.  Nothing here imports a rule from the
product: which V format / which rows are "early" is restated below as literal tables of include/qattn.h's path table."""
import numpy as np
import torch

import oracle
from quantumattention_amd import _native

FMT = {"e4m3": oracle.FMT_E4M3, "e5m2": oracle.FMT_E5M2}
TDT = {"e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}
TOL = 2.0 ** -6        # fp8-V rows
TOL_V16 = 2.0 ** -7    # 16-bit-V rows
PATH_ONE_TERM, PATH_TWO_TERM, PATH_V16 = 0, 1, 2   # include/qattn.h QATTN_PATH_* (literal: the header is the contract)


def bits16(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits8(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.uint8).numpy()


def from_bits16(b: np.ndarray, dtype) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(b).view(np.int16).copy()).view(dtype)


def fmt16(dtype) -> int:
    return oracle.FMT_BF16 if dtype == torch.bfloat16 else oracle.FMT_FP16


def out_to_f32(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def unpack_frag(buf: np.ndarray, layout: int, B: int, H: int, S: int, D: int) -> np.ndarray:
    """Invert QATTN_LAYOUT_KFRAG / _VFRAG (byte maps: csrc/qattn_common.h) -> row-major [B,H,Sp,D] (Sp = S padded to 64)."""
    Sp = (S + 63) // 64 * 64
    x = np.asarray(buf, np.uint8).reshape(B, H, Sp // 64, 64 * D)
    key = np.arange(64)[:, None]
    d = np.arange(D)[None, :]
    if layout == _native.LAYOUT_KFRAG:
        off = (((key >> 5) * (D // 64) + (d >> 6)) << 11) + (((d >> 5) & 1) << 10) + (((d >> 4) & 1) << 9) + ((key & 31) << 4) + (d & 15)
    else:
        half, w, hh, i = key >> 5, (key >> 3) & 3, (key >> 2) & 1, key & 3
        off = ((d >> 5) << 11) + (hh << 10) + (half << 9) + ((d & 31) << 4) + (w << 2) + i
    return x[..., off].reshape(B, H, Sp, D)


# ---- literal restatement of include/qattn.h's path table (the fused entry), NOT imported from the product -------------------------
def fused_step_uses_block_v(D, scaling, dtype, Skv) -> bool:
    """V format of the fused entry: head-wise scales and at most 256 chunks of 64 keys per head -> one power-of-two scale per chunk
    (every head dim, bf16 and fp16); token-wise scales or a longer head -> one fp32 scale per head."""
    assert D in (64, 128, 256) and dtype in (torch.bfloat16, torch.float16)
    return {"head": True, "head-wise": True, "token": False, "token-wise": False}[scaling] and Skv <= 16384


def early_rows(Sq, Skv, causal, q_offset=0) -> np.ndarray:
    """bool [Sq]: rows of query blocks (256 rows) whose FIRST row sees fewer than 1024 keys -- every mode of the fused entry attends
    the original 16-bit V there (path 2)."""
    first_row = ((np.arange(Sq) + q_offset) // 256) * 256
    return (np.minimum(Skv, first_row + 1) if causal else np.full(Sq, Skv)) < 1024


class PathRef:
    """The two oracles of a FUSED call, side by side: `fp8v` (fp64 SDPA with the build's fp8 V: paths 0 and 1) and `v16` (fp64 SDPA with
    the original 16-bit V: path 2), both [B,H,Sq,D].  `select(path)` picks, per row, THE oracle of the path the kernel reported."""

    def __init__(self, fp8v, v16):
        self.fp8v, self.v16 = np.asarray(fp8v), np.asarray(v16)
        assert self.fp8v.shape == self.v16.shape

    @property
    def shape(self):
        return self.fp8v.shape

    def __getitem__(self, idx):
        return PathRef(self.fp8v[idx], self.v16[idx])

    def select(self, path):
        path = np.asarray(path)
        assert path.shape == self.fp8v.shape[:-1], (path.shape, self.fp8v.shape)
        assert np.isin(path, (PATH_ONE_TERM, PATH_TWO_TERM, PATH_V16)).all(), np.unique(path)
        return np.where((path == PATH_V16)[..., None], self.v16, self.fp8v)


def oracle_for_fp8_path(q8b, k8b, v16b, sq, sk, *, fp8="e4m3", v_dtype=torch.bfloat16, scaling="head", causal=False,
                        sm_scale=0.0, return_lse=False, v_block=False, q_offset=0, fused=False):
    """O3 of SURVEY.md §8c: fp64 SDPA on the same quantised q, k AND the build's quantised v (v_block: the fused step's
    block-scaled V, oracle.quantize_v_block; else one scale per head).
    fused (implied by v_block): the call under test is the fused entry -> a PathRef (fp8-V oracle, 16-bit-V oracle), to be graded with
    the kernel's row_path.  Otherwise (the separate C calls, the op on pre-quantised q / k): a plain array -- every row is fp8 V."""
    fused = fused or v_block
    if v_block:
        _, _, vdq = oracle.quantize_v_block(v16b, fmt16(v_dtype), FMT[fp8])
        res = oracle.attention_forward(q8b, k8b, vdq, FMT[fp8], FMT[fp8], oracle.FMT_BF16, sq, sk, None, scale_mode=scaling,
                                       causal=causal, sm_scale=sm_scale, return_lse=return_lse, q_offset=q_offset)
    else:
        v8, sv = oracle.quantize_fp8(v16b, fmt16(v_dtype), "head", FMT[fp8], "compiled")
        res = oracle.attention_forward(q8b, k8b, v8, FMT[fp8], FMT[fp8], FMT[fp8], sq, sk, sv, scale_mode=scaling,
                                       causal=causal, sm_scale=sm_scale, return_lse=return_lse, q_offset=q_offset)
    if fused:
        alt = oracle.attention_forward(q8b, k8b, v16b, FMT[fp8], FMT[fp8], fmt16(v_dtype), sq, sk, None, scale_mode=scaling, causal=causal,
                                       sm_scale=sm_scale, return_lse=False, q_offset=q_offset)
        res = (PathRef(res[0], alt),) + tuple(res[1:]) if return_lse else PathRef(res, alt)
    return res


def fused_call(q, k, v, *, causal=False, precision="auto", fp8="e4m3", scaling="head-wise", return_lse=False, **kw):
    """The fused entry through the C ABI with its row_path output: (out fp32 numpy, path uint8 numpy[, lse numpy]).
    (qa.fp8_attn_func is this very call with row_path = NULL: tests/test_gpu_attention.py asserts the bits are the same.)"""
    res = _native.fp8_quant_attention_forward(q.cuda(), k.cuda(), v.cuda(), is_causal=causal, scaling=scaling, fp8_dtype=TDT[fp8],
                                              precision=precision, return_lse=return_lse, return_path=True, **kw)
    out, path = res[0], res[-1]
    got = (out_to_f32(out), path.cpu().numpy())
    return got + (res[1].cpu().numpy(),) if return_lse else got


def check_path_structure(path, Sq, Skv, causal, precision, d128_headwise, q_offset=0):
    """What must hold for the path codes whatever the data: early rows on the 16-bit V; FAST never leaves the one-term sweep elsewhere;
    ACCURATE puts every other row on the precise pass (D = 128 head-wise: 16-bit V; templated kernel: two-term fp8 P)."""
    path = np.asarray(path)
    e = early_rows(Sq, Skv, causal, q_offset)
    assert (path[..., e] == PATH_V16).all(), "early rows must attend the 16-bit V"
    rest = path[..., ~e]
    if precision == "fast":
        assert (rest == PATH_ONE_TERM).all(), np.unique(rest)
    elif precision == "accurate":
        assert (rest == (PATH_V16 if d128_headwise else PATH_TWO_TERM)).all(), np.unique(rest)
    if not d128_headwise:
        assert (rest != PATH_V16).all(), "the templated kernel has no 16-bit-V rescue outside the early blocks"


def grade(got: np.ndarray, ref, path=None):
    """(max-abs, rmse, worst): worst = max over elements of |got - ref| / bound, bound per ELEMENT (module docstring); < 1 passes.
    ref: a plain array (every row fp8 V; path must be None), or a PathRef with the kernel's row_path."""
    got = np.asarray(got, np.float64)
    if isinstance(ref, PathRef):
        assert path is not None, "a fused call is graded per row against the oracle of the path the kernel reported: pass its row_path"
        r = ref.select(path).astype(np.float64)
        v16_rows = (np.asarray(path) == PATH_V16)[..., None]
        bound = np.where(v16_rows, TOL_V16 * np.maximum(1.0, np.abs(r)), TOL * np.maximum(1.0, np.abs(r) / 2.0))
    else:
        assert path is None
        r = np.asarray(ref, np.float64)
        bound = TOL * np.maximum(1.0, np.abs(r) / 2.0)
    d = np.abs(got - r)
    return float(d.max()), float(np.sqrt((d ** 2).mean())), float((d / bound).max())


def err_stats(got: np.ndarray, ref, path=None):
    """(max-abs, rmse) of got against THE reference of every row (see grade)."""
    return grade(got, ref, path)[:2]


def assert_within_bound(got, ref, path=None, what=""):
    mx, rmse, worst = grade(got, ref, path)
    assert worst < 1.0, (what, "max-abs", mx, "rmse", rmse, "worst |err| / bound", worst)
    return mx, rmse

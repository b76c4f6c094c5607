"""Helpers shared by the -m gpu parity tests: numpy <-> torch bit views, fragment-layout inverse maps, and the
oracle call that mirrors one HIP attention launch (same quantised q/k/v, fp64 math)."""
import numpy as np
import torch

import oracle
from quantumattention_amd import _native

FMT = {"e4m3": oracle.FMT_E4M3, "e5m2": oracle.FMT_E5M2}
TDT = {"e4m3": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2}


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
    """Invert QATTN_LAYOUT_KFRAG / _VFRAG (include/qattn.h) -> row-major [B,H,Sp,D] (Sp = S padded to 64)."""
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


def fused_step_uses_block_v(D, scaling, dtype, Skv) -> bool:
    """Mirror of qattn_fp8_quant_attention_forward's choice (csrc/qattn_api.hip): block-scaled V with head-wise scales where the
    kernel's PV products take a chunk scale -- the hand-scheduled D = 128 kernel when it quantises bf16 Q itself, the templated
    kernel at D = 64 / 256 -- and a head has at most 256 chunks; everywhere else V has one scale per head."""
    from quantumattention_amd.utils.checks import fused_step_scales_v_per_head
    return scaling in ("head", "head-wise") and not fused_step_scales_v_per_head(D, dtype, scaling, Skv)


class FusedRef(np.ndarray):
    """The reference of the FUSED step: the array itself is the mixed oracle (fp64 SDPA with the block-scaled fp8 V; the original 16-bit V
    on the rows of the early blocks), `.alt` is fp64 SDPA with the original 16-bit V on EVERY row.  Since round 5 the D = 128 kernel
    recomputes the rows its statistics flag (and the blocks it used to run with two-term P) on the 16-bit V -- the reference kernel's own
    numerics, csrc/qattn_pv16.h -- and which rows those are is the kernel's data-dependent decision: a row must meet the bound against
    ONE of the two (err_stats takes, per row, the closer).  Slicing keeps the pair aligned."""

    def __new__(cls, main, alt):
        obj = np.asarray(main).view(cls)
        obj.alt = np.asarray(alt)
        return obj

    def __array_finalize__(self, obj):
        self.alt = getattr(obj, "alt", None)

    def __getitem__(self, idx):
        out = super().__getitem__(idx)
        if isinstance(out, FusedRef):
            out.alt = self.alt[idx] if self.alt is not None and self.alt.shape == np.asarray(self).shape else None
        return out


def oracle_for_fp8_path(q8b, k8b, v16b, sq, sk, *, fp8="e4m3", v_dtype=torch.bfloat16, scaling="head", causal=False,
                        sm_scale=0.0, return_lse=False, v_block=False, q_offset=0, v16_early=None, fused=False):
    """O3 of SURVEY.md §8c: fp64 SDPA on the same quantised q, k AND the build's quantised v (v_block: the fused step's
    block-scaled V, oracle.quantize_v_block; else one scale per head).  fused (implied by v_block): the call under test is the fused
    step, whose early rows attend the original 16-bit V (v16_early overrides the rule either way)."""
    fused = fused or v_block
    if v_block:
        _, _, vdq = oracle.quantize_v_block(v16b, fmt16(v_dtype), FMT[fp8])
        res = oracle.attention_forward(q8b, k8b, vdq, FMT[fp8], FMT[fp8], oracle.FMT_BF16, sq, sk, None, scale_mode=scaling,
                                       causal=causal, sm_scale=sm_scale, return_lse=return_lse, q_offset=q_offset)
    else:
        v8, sv = oracle.quantize_fp8(v16b, fmt16(v_dtype), "head", FMT[fp8], "compiled")
        res = oracle.attention_forward(q8b, k8b, v8, FMT[fp8], FMT[fp8], FMT[fp8], sq, sk, sv, scale_mode=scaling,
                                       causal=causal, sm_scale=sm_scale, return_lse=return_lse, q_offset=q_offset)
    # The fused step (every head dim, any 16-bit dtype, head- or token-wise scales) attends the ORIGINAL 16-bit V in the query blocks (256 rows)
    # whose first row sees fewer than 1024 keys -- the reference kernel's own P.V numerics, csrc/qattn_pv16.h: causal, the leading
    # blocks; every block when Skv < 1024.  (D = 128 bf16 head-wise: a pass inside the fused kernel; elsewhere a launch of its own.)
    Sq, D, Skv = np.asarray(q8b).shape[2], np.asarray(q8b).shape[3], np.asarray(k8b).shape[2]
    if (v16_early is None and fused) or v16_early:
        first_row = ((np.arange(Sq) + q_offset) // 256) * 256
        early = (np.minimum(Skv, first_row + 1) if causal else np.full(Sq, Skv)) < 1024
        n_early = int(early.sum())
        if n_early:
            assert early[:n_early].all()   # a prefix of the rows
            sq_e = np.asarray(sq)[:, :, :n_early] if np.asarray(sq).ndim == 3 else sq      # token-wise: one q scale per row
            r16 = oracle.attention_forward(np.asarray(q8b)[:, :, :n_early], k8b, v16b, FMT[fp8], FMT[fp8], fmt16(v_dtype), sq_e, sk, None,
                                           scale_mode=scaling, causal=causal, sm_scale=sm_scale, return_lse=False, q_offset=q_offset)
            out = res[0] if return_lse else res
            out[:, :, :n_early] = r16
    if fused:   # the 16-bit-V reference beside the mixed one (FusedRef)
        alt = oracle.attention_forward(q8b, k8b, v16b, FMT[fp8], FMT[fp8], fmt16(v_dtype), sq, sk, None, scale_mode=scaling, causal=causal,
                                       sm_scale=sm_scale, return_lse=False, q_offset=q_offset)
        res = (FusedRef(res[0], alt),) + tuple(res[1:]) if return_lse else FusedRef(res, alt)
    return res


def err_stats(got: np.ndarray, ref: np.ndarray):
    """(max-abs, rmse) of got against ref; against a FusedRef, per row (last axis) against the closer of its two references."""
    d = np.abs(got.astype(np.float64) - np.asarray(ref).astype(np.float64))
    alt = getattr(ref, "alt", None)
    if alt is not None and alt.shape == d.shape:
        d2 = np.abs(got.astype(np.float64) - alt.astype(np.float64))
        d = np.where(d2.max(axis=-1, keepdims=True) < d.max(axis=-1, keepdims=True), d2, d)
    return float(d.max()), float(np.sqrt((d ** 2).mean()))

"""-m gpu: accuracy of the fp8-P attention beyond N(0,1) scores.

One e4m3 term of P carries 3 mantissa bits; that is enough only while a row's softmax weight is spread over many keys.
Trained attention heads have score std 2..5: a handful of keys carry the row.  These tests scale q (score std x2, x3, x5),
mix sharpness per row, and plant groups of equally heavy keys; the oracle is fp64 SDPA on the same quantised q, k, v.
Stated tolerance (BASELINE.json north_star): max-abs < 2^-6, NOT scaled by |O| (S >= 1024 here).

  precision="auto"      (default) must meet 2^-6 on every case: peaked blocks are detected (R = l / p_max < 24) and redone
                        with two-term P inside the kernel (D = 128) or by the rescue launch (D = 64 / 256);
  precision="accurate"  two-term everywhere: must meet 2^-6 too, by a wide margin;
  precision="fast"      one-term, unchecked: documented to FAIL the bound on peaked rows -- asserted, so that the test
                        proves the cases really exercise the rescue path."""
import numpy as np
import pytest
import torch

import oracle
import quantumattention_amd as qa
from tests.gpu_utils import bits16, err_stats, oracle_for_fp8_path, out_to_f32

pytestmark = pytest.mark.gpu
TOL = 2.0 ** -6


def _oracle(q, k, v, causal, fp8="e4m3"):
    fmt = oracle.FMT_E4M3 if fp8 == "e4m3" else oracle.FMT_E5M2
    q8, sq = oracle.quantize_fp8(bits16(q), oracle.FMT_BF16, "head", fmt)
    k8, sk = oracle.quantize_fp8(bits16(k), oracle.FMT_BF16, "head", fmt)
    return oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, fp8=fp8, causal=causal, v_block=q.shape[-1] == 128)   # _run is the fused step


def _run(q, k, v, causal, precision, fp8="e4m3"):
    with qa.config.patch({"attention.precision": precision, "attention.fp8_format": fp8}):
        return out_to_f32(qa.fp8_attn_func(q.cuda(), k.cuda(), v.cuda(), is_causal=causal))


def _inputs(S, D, sharp, seed, H=2):
    torch.manual_seed(seed)
    q = torch.randn(1, H, S, D)
    if sharp == "mixed":      # per-row sharpness from flat (x0.5) to very peaked (x6), shuffled over the rows
        q = q * torch.linspace(0.5, 6.0, S)[torch.randperm(S)].view(1, 1, S, 1)
    else:
        q = q * float(sharp)
    k, v = torch.randn(1, H, S, D), torch.randn(1, H, S, D)
    return q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)


SHARP_CASES = [(S, D, sharp, causal) for S in (1024, 4096) for D in (128,) for sharp in (2.0, 3.0, 5.0, "mixed") for causal in (False, True)]
SHARP_CASES += [(2048, 64, 3.0, False), (2048, 64, "mixed", True), (2048, 256, 3.0, True), (2048, 256, "mixed", False)]


@pytest.mark.parametrize("S,D,sharp,causal", SHARP_CASES, ids=lambda x: str(x))
def test_peaked_rows_meet_the_stated_bound(S, D, sharp, causal):
    q, k, v = _inputs(S, D, sharp, seed=S + D)
    ref = _oracle(q, k, v, causal)
    auto = _run(q, k, v, causal, "auto")
    acc = _run(q, k, v, causal, "accurate")
    assert np.isfinite(auto).all() and np.isfinite(acc).all()
    mx_auto, rms_auto = err_stats(auto, ref)
    mx_acc, _ = err_stats(acc, ref)
    assert mx_auto < TOL, (mx_auto, rms_auto)
    assert mx_acc < TOL, mx_acc
    if S == 4096 and sharp in (3.0, 5.0) and not causal:
        mx_fast, _ = err_stats(_run(q, k, v, causal, "fast"), ref)
        assert mx_fast > TOL, ("one-term P was expected to break the bound on these rows", mx_fast)


def test_groups_of_equally_heavy_keys():
    """Adversarial for a max-weight statistic: K keys share the row's weight equally (R ~ K).  K = 8, 16 must be caught by
    the R < 24 test; K = 64 is left on the one-term path, where 64 independent roundings average out."""
    torch.manual_seed(3)
    S, D = 4096, 128
    for K in (8, 16, 64):
        q, k, v = torch.randn(1, 2, S, D), torch.randn(1, 2, S, D), torch.randn(1, 2, S, D)
        u = torch.randn(D)
        u /= u.norm()
        idx = torch.randperm(S)[:K]
        q = q + 4.0 * u                       # every query has a common component ...
        k[:, :, idx] = 0.25 * k[:, :, idx] + 4.0 * u   # ... that K keys share: their scores sit ~16/sqrt(D)*... above the rest
        q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        ref = _oracle(q, k, v, False)
        mx, rms = err_stats(_run(q, k, v, False, "auto"), ref)
        assert mx < TOL, (K, mx, rms)


def test_flat_rows_keep_the_one_term_result_bit_for_bit():
    """The check must not change what flat inputs compute: on N(0,1) data at S = 4096 `auto` and `fast` agree bit for bit
    except in the 32-row groups that were rescued, and those are rare (< 5 % of the rows)."""
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, 8, 4096, 128, dtype=torch.bfloat16) for _ in range(3))
    a, f = _run(q, k, v, False, "auto"), _run(q, k, v, False, "fast")
    changed = (a != f).any(axis=-1)            # rows that differ
    assert changed.mean() < 0.05, changed.mean()
    groups = changed.reshape(2, 8, 128, 32)
    assert ((groups.any(-1)) == (groups.mean(-1) > 0.5)).all()   # a rescued 32-row group changes (almost) all of its rows, others none
    ref = _oracle(q[:1, :2], k[:1, :2], v[:1, :2], False)
    assert err_stats(a[:1, :2], ref)[0] < TOL
    assert err_stats(f[:1, :2], ref)[0] < TOL      # the unchecked one-term path is accurate on flat rows


@pytest.mark.parametrize("D", [128, 64, 256])
def test_heads_with_wide_scores_start_in_two_term_mode(D):
    """The fused step hands the attention kernel every head's sum of squares (abs-max pass, deterministic partial sums); a head
    whose predicted score variance is clearly above 1 starts its blocks in two-term mode instead of sweeping once in vain
    (qattn_attn.h predicted_r; D = 64 / 256: its groups are flagged unswept and the two-term redo launch takes them).  Observable without a clock: for such a head AUTO must return ACCURATE's bits, for a
    unit-variance head (below the dead band) it must not -- there the one-term result stands."""
    torch.manual_seed(5)
    S = 2048
    q, k, v = (torch.randn(1, 2, S, D) for _ in range(3))
    q[:, 1] *= 2.0                                            # head 0: score std 1, head 1: score std 2
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    auto, acc = _run(q, k, v, False, "auto"), _run(q, k, v, False, "accurate")
    np.testing.assert_array_equal(auto[0, 1], acc[0, 1])
    assert not np.array_equal(auto[0, 0], acc[0, 0])
    ref = _oracle(q, k, v, False)
    assert err_stats(auto, ref)[0] < TOL


@pytest.mark.parametrize("causal", [False, True])
def test_anisotropic_heads_take_the_forecast_exit(causal):
    """Two dimensions of q and k carry 3x the amplitude: the pre-pass's isotropic moment estimate says score variance 1.27 (inside
    the dead band), the true one is 2.25.  The one-term sweep measures the spread of its first chunk of scores, stops after three
    chunks and the block repeats in two-term mode (kv_sweep `forecast`): same bound as everywhere, and for a head in which every
    block takes that exit AUTO returns ACCURATE's bits."""
    torch.manual_seed(17)
    S, D = 4096, 128
    q, k, v = (torch.randn(1, 2, S, D) for _ in range(3))
    q[..., :2] *= 3.0
    k[..., :2] *= 3.0
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    ref = _oracle(q, k, v, causal)
    auto, acc = _run(q, k, v, causal, "auto"), _run(q, k, v, causal, "accurate")
    assert err_stats(auto, ref)[0] < TOL and err_stats(acc, ref)[0] < TOL
    if not causal:
        np.testing.assert_array_equal(auto, acc)


def test_gqa_with_one_wide_kv_group():
    """GQA through the fused step on peaked data: 8 query heads share 2 kv heads; the query heads of kv group 1 are scaled x3, so
    the starting-mode prediction (indexed by query head and kv head), the rescue (indexed by kv head) and the redo all meet."""
    torch.manual_seed(11)
    S, D = 2048, 128
    q = torch.randn(1, 8, S, D); k = torch.randn(1, 2, S, D); v = torch.randn(1, 2, S, D)
    q[:, 4:] *= 3.0
    q[:, 1] *= torch.linspace(0.5, 4.0, S)[torch.randperm(S)].view(S, 1)
    q, k, v = (t.to(torch.bfloat16) for t in (q, k, v))
    from tests.gpu_utils import oracle_for_fp8_path
    q8, sq = oracle.quantize_fp8(bits16(q), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    k8, sk = oracle.quantize_fp8(bits16(k), oracle.FMT_BF16, "head", oracle.FMT_E4M3)
    for causal in (False, True):
        ref = oracle_for_fp8_path(q8, k8, bits16(v), sq, sk, fp8="e4m3", causal=causal, v_block=True)
        got = _run(q, k, v, causal, "auto")
        assert np.isfinite(got).all()
        assert err_stats(got, ref)[0] < TOL, (causal, err_stats(got, ref))


def test_non_finite_and_degenerate_heads_do_not_disturb_the_others():
    """A head of zeros, a head with an inf, a head with a NaN and a head of huge values go through the fused step next to normal
    heads: the moments of such heads are 0 / inf / NaN (the starting-mode prediction must not trip on them), the normal heads'
    rows must come out exactly as when they are attended alone, and the call must return."""
    torch.manual_seed(13)
    S, D = 1536, 128
    q, k, v = (torch.randn(1, 8, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q[0, 1] = 0
    q[0, 2, 5, 7] = float("inf")
    k[0, 3, 9, 1] = float("nan")
    q[0, 4] *= 1.0e18
    k[0, 4] *= 1.0e18
    for causal in (False, True):
        out = qa.fp8_attn_func(q, k, v, is_causal=causal)
        torch.cuda.synchronize()
        for h in (0, 5, 6, 7):
            alone = qa.fp8_attn_func(q[:, h:h + 1], k[:, h:h + 1], v[:, h:h + 1], is_causal=causal)
            assert torch.equal(out[:, h:h + 1], alone), (causal, h)
            assert torch.isfinite(out[:, h]).all()
        assert torch.isfinite(out[:, 1]).all()          # zero queries: uniform attention, finite


def test_lse_reference_layout_and_convention():
    """the reference-defined (disabled) vector, tk/attention.py:333-346 / :439-446:
    L = -(ln l + m ln2) sqrt(D), rows of consecutive (b, h) ld = ceil(Sq*4/16)*16/4 floats apart."""
    from quantumattention_amd import _native

    torch.manual_seed(4)
    B, H, S, D = 2, 3, 1001, 128
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    q8, kf, vf, sq, sk, sv = _native.quant_qkv_fp8(q, k, v)
    out_n, lse_n = _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=True,
                                                 return_lse=True)
    out_r, lse_r = _native.fp8_attention_forward(q8, kf, vf, sq, sk, sv, Hkv=H, Skv=S, out_dtype=torch.bfloat16, is_causal=True,
                                                 return_lse=True, lse_layout=_native.LSE_REFERENCE)
    assert torch.equal(out_n, out_r)
    assert lse_n.shape == lse_r.shape == (B, H, S)
    assert lse_n.stride() == (H * S, S, 1) and lse_r.stride() == (H * 1004, 1004, 1)       # 1001 floats padded to 16 bytes
    torch.testing.assert_close(lse_r, -(D ** 0.5) * lse_n, rtol=1e-6, atol=1e-5)


def test_config5_at_its_stated_size_B4_H40_S16384_e5m2_causal():
    """BASELINE config 5 at full size (only B = 1 had run): finite, deterministic, batch-shard equivalent, and
    within the bound on an oracle slice (one head: the first 1280 rows and, via a non-causal Sq != Skv call, the last 256)."""
    torch.manual_seed(5)
    B, H, S, D = 4, 40, 16384, 128
    q, k, v = (torch.randn(B, H, S, D, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    with qa.config.patch({"attention.fp8_format": "e5m2"}):
        out = qa.fp8_attn_func(q, k, v, is_causal=True)
        assert torch.isfinite(out).all()
        assert torch.equal(out, qa.fp8_attn_func(q, k, v, is_causal=True))                              # determinism
        assert torch.equal(out[2:3], qa.fp8_attn_func(q[2:3], k[2:3], v[2:3], is_causal=True))          # batch-shard equivalence
    b, h, top = 3, 17, 1280
    qs, ks, vs = q[b:b + 1, h:h + 1].cpu(), k[b:b + 1, h:h + 1].cpu(), v[b:b + 1, h:h + 1].cpu()
    q8, sq = oracle.quantize_fp8(bits16(qs), oracle.FMT_BF16, "head", oracle.FMT_E5M2)
    k8, sk = oracle.quantize_fp8(bits16(ks), oracle.FMT_BF16, "head", oracle.FMT_E5M2)
    ref_top = oracle_for_fp8_path(q8[:, :, :top], k8, bits16(vs), sq, sk, fp8="e5m2", causal=True, v_block=True)
    mx, rmse = err_stats(out_to_f32(out[b, h, :top]), ref_top[0, 0])
    assert mx < TOL * max(1.0, float(np.abs(ref_top).max()) / 2.0) and rmse < 3e-3, (mx, rmse)   # |O| > 2 only on the first rows
    tail = slice(S - 256, S)
    ref_tail = oracle_for_fp8_path(q8[:, :, tail], k8, bits16(vs), sq, sk, fp8="e5m2", causal=False)
    got_last = out_to_f32(out[b, h, S - 1])      # the last causal row sees every key = the non-causal row
    assert np.abs(got_last - ref_tail[0, 0, -1]).max() < TOL

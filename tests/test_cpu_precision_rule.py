"""The inequality behind the key-count form of the exact-top rule (csrc/qattn_attn.h row_is_peaked), checked
on random and adversarial weight vectors: for non-negative weights w_2 .. w_n with sum A and sum of squares S2,

    S2 < t^2 + (A - t)^2 / (n - 2)   and   t >= A / (n - 1)     imply     max w_j < t,

and with t = 1 / 24, n > 290 the left side also implies S2 < 1 / 192 (the statistical budget).  Also the byte model the
kernel's estimate of S2 rests on: an e4m3 byte read as e5m2 is 0.444 .. 0.5 of the e4m3 value's square."""
import numpy as np
import torch


def test_cauchy_schwarz_bound_on_the_second_largest_weight():
    rng = np.random.default_rng(0)
    t = 1.0 / 24.0
    accepted = 0
    for trial in range(4000):
        n = int(rng.integers(292, 5000))
        kind = trial % 4
        if kind == 0:
            w = np.exp(rng.normal(0.0, rng.uniform(0.2, 1.6), n - 1))            # lognormal rest (flat data)
        elif kind == 1:
            w = np.exp(rng.normal(0.0, 1.0, n - 1)); w[0] = w.sum() * rng.uniform(0.01, 0.08)   # one more outlier
        elif kind == 2:
            w = np.full(n - 1, 1.0); w[: int(rng.integers(1, 60))] = rng.uniform(5, 60)          # a group of heavy keys
        else:
            w = rng.uniform(0, 1, n - 1) ** 8
        top = rng.uniform(0.0, 0.2)                    # the exact top key's weight
        w = w / w.sum() * (1.0 - top)                  # rest weights, A = 1 - top
        A, S2 = w.sum(), (w * w).sum()
        if S2 < t * t + (A - t) ** 2 / (n - 2) and A - t > 0:
            accepted += 1
            assert w.max() < t, (trial, n, w.max())
            assert S2 < 1.0 / 192.0
    assert accepted > 500   # (the rule is not vacuous on these families)
    # the extremal configuration: one key at exactly t, the others equal -> equality, not accepted
    n = 1200
    w = np.full(n - 1, (0.95 - t) / (n - 2)); w[0] = t
    assert not ((w * w).sum() < t * t + (0.95 - t) ** 2 / (n - 2) - 1e-18)


def test_e4m3_byte_read_as_e5m2_is_between_0444_and_05_of_the_square():
    # normal e4m3 numbers up to 2^8 (1 + 3/8) = 352: the fix-up branch keeps P' below 2^(5 + 3) = 256 (bytes from 124 = 384 on
    # read as e5m2 infinities / NaNs: l2 = inf, the row counts as peaked -- the safe side); the subnormals carry no weight
    b = torch.arange(8, 124, dtype=torch.uint8)
    v = b.view(torch.float8_e4m3fn).float().double()
    r = b.view(torch.float8_e5m2).float().double() / (v * v)
    assert 0.444 <= r.min().item() and r.max().item() <= 0.5 + 1e-12, (r.min().item(), r.max().item())

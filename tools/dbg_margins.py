"""Development: the largest max-abs errors among the peaked-row precision cases (how far below 2^-6 the tests sit)."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_gpu_precision as T
from tests.gpu_utils import err_stats
worst = []
for (S, D, sharp, causal) in T.SHARP_CASES:
    q, k, v = T._inputs(S, D, sharp, seed=S + D)
    ref = T._oracle(q, k, v, causal)
    for prec in ("auto", "accurate"):
        mx, rms = err_stats(T._run(q, k, v, causal, prec), ref)
        worst.append((mx, S, D, sharp, causal, prec))
worst.sort(reverse=True)
for w in worst[:10]: print("%.5f" % w[0], w[1:])
print("TOL", T.TOL)

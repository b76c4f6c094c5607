"""-m gpu: a 250-configuration slice of the randomised sweeps in the regular suite (three rule defects were found by
widening these checks; they should not live as logs only).  The sweeps are the scripts under tools/ (one line per configuration, exit
status 1 on any failure); each test runs a fixed-seed slice in a child process and keeps the transcript in the assertion message.

  tools/fuzz_parity.py   random B, Hq / Hkv, Sq != Skv, ragged lengths, D, causal, e4m3 / e5m2, head- / token-wise, bf16 / fp16,
                         precision, score spread x1 .. x3, planted outlier keys: quantiser bit-exact, fused step AND separate calls AND
                         the 16-bit path against the fp64 oracle (2^-6 / 2^-7);
  tools/fuzz_launch.py   shapes that reach the persistent / dynamic / multi-launch paths: every element written, batched == per-element,
                         graph replay == eager, producer hand-off == plain call, all bit for bit."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(script, n, seed, *extra):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(n), str(seed), *extra], capture_output=True, text=True, env=env,
                       timeout=1500)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-40:])
    assert p.returncode == 0, f"{script} {n} {seed} failed:\n{tail}"
    assert f"{n} cases, 0 failures" in p.stdout, tail


@pytest.mark.parametrize("seed", [401, 402, 403, 404])
def test_parity_sweep_slice(seed):
    _run("fuzz_parity.py", 25, seed)


@pytest.mark.parametrize("seed", [421, 422])
def test_adversarial_parity_sweep_slice(seed):
    """Round 5: two configurations of three carry a structure aimed at one rule of the precision machinery (tools/fuzz_parity.py::adversarial:
    planted keys on both sides of the flag and 16-bit-rescue thresholds, a few equal keys, about 96 peaked rows per block, bimodal scores, the
    dominant key in the last chunk, heavy-tailed V, zero rows, inputs near the top of fp16's range)."""
    _run("fuzz_parity.py", 25, seed, "adv")


@pytest.mark.parametrize("seed", [411, 412, 413, 414])
def test_launch_sweep_slice(seed):
    _run("fuzz_launch.py", 25, seed)

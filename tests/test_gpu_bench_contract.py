"""-m gpu: bench.py prints ONE JSON line with the driver's contract fields (metric/value/unit/..., roofline, cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_emits_the_contract_line():
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "1", "--heads", "8",
                        "--seq", "2048"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["unit"] == "TFLOP/s" and d["value"] > 0 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 5000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # roofline.traffic is MEASURED in this very run (round 6): rocprofv3 PMC child passes of the same command, FETCH_SIZE x 2 + WRITE_SIZE
    alg = 1 * 8 * 2048 * 128 * (2 + 1 + 1 + 2)   # Q bf16 + K, V fp8 + O bf16
    assert "measured in this run" in r["traffic_source"], r["traffic_source"]
    assert 0.9 * alg < r["traffic"] < 3.0 * alg, (r["traffic"], alg)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c

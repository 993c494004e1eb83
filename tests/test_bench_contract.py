"""bench.py's output contract: ONE JSON line on stdout with the driver's keys plus `roofline` (and `cpu_baseline` at N = 1);
without a GPU it refuses to run instead of falling back to a CPU path."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = run_bench("--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0
    assert "no CPU fallback" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = run_bench("--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "audio samples/s" and d["value"] > 22050 * 100          # the >= 100 x real-time target, even at B = 4
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    # value and ms_per_step describe the same timed region
    samples = 4 * 768 * 256
    assert abs(d["value"] - samples / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_helpers_without_a_gpu():
    """Host-only pieces of bench.py: the CPUs a process may really use (affinity mask / cgroup quota, not os.cpu_count()), BASELINE
    config 3's length list and config 4's utterance list (config 3's lengths x 8, the same on every rank)."""
    sys.path.insert(0, ROOT)
    import bench
    n, why = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1) and isinstance(why, str) and why
    lens = bench.mixed_lengths(32)
    assert lens.min() == 40 and lens.max() == 200 and int(lens.sum()) * 6 == 23040 and len(lens) == 32
    a, b = bench.c4_id_lists(256), bench.c4_id_lists(256)
    assert a == b and len(a) == 256 and sum(len(x) for x in a) * 6 == 8 * 23040
    assert sorted(len(x) for x in a[:32]) == sorted(lens.tolist())
    assert all(4 <= t <= 130 for x in a[:8] for t in x)

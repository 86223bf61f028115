"""bench.py's output contract on a toy workload (seconds on the GPU): ONE JSON line with the driver's keys, the
roofline and cpu_baseline objects, the PCIe-inclusive host_fed rate, counters that add up; and, with two ranks on the
one GPU over gloo (rehearsal mode), the N > 1 code path: sharded index files, per-rank batches, max over ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = ["--genomes", "24", "--genome-len", "60000", "--reads", "200000", "--steps", "2", "--warmup", "1", "--cpu-sample", "20000"]


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + TOY, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "host_fed"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["unit"] == "Mreads/s" and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["value"] > 0 and j["ms_per_step"] >= j["roofline"]["kernel_ms"]
    assert "workload" in j["config"] and "model" not in j["config"]
    ro = j["roofline"]
    assert ro["bound"] == "hbm" and ro["unit"] == "GB/s" and ro["frac"] > 0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert ro["traffic"] is None            # no PMC pass exists for a toy workload: never a made-up number
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and j["parity_checked_reads"] == 20000
    o = j["outcome"]
    assert o["reads"] == 2 * 200000 and o["nskipped"] == 0
    assert j["host_fed"]["Mreads_s"] > 0 and j["host_fed"]["bytes_per_read_on_the_wire"] == 26


def test_two_ranks_rehearsal_over_gloo():
    env = dict(os.environ, CAMMIQ_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TOY,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and "sharded x2" in j["config"]["parallelism"] and "gloo" in j["config"]["parallelism"]
    assert j["outcome"]["reads"] == 2 * 2 * 200000       # both ranks' reads are in the all-reduced counters
    assert "cpu_baseline" not in j                        # rank 0 at N = 1 only

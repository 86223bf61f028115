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
    assert ro["algorithmic_T"] == 1 and ro["kernel"].startswith("classify_kernel<8,16,false,") and len(ro["kernel_ms_runs"]) == 2
    assert ro["kernel_launch"]["fixed_shape"] == 1      # h = 26, 100-bp batch: the instantiation with the shape folded in
    assert "frac_at_profiled_ms" not in ro              # ... and no profiled duration either
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and j["parity_checked_reads"] == 20000
    assert cb["value_thread_local"] > 0 and cb["value_atomic"] > 0 and cb["host_cores_online"] >= cb["cores"]
    o = j["outcome"]
    assert o["reads"] == 2 * 200000 and o["nskipped"] == 0
    assert j["host_fed"]["Mreads_s"] > 0 and j["host_fed"]["bytes_per_read_on_the_wire"] == 25
    assert j["value_survey_8d_bracket"] == j["host_fed"]["Mreads_s"]
    assert j["reads_door"]["Mreads_s"] > 0 and j["reads_door"]["reads"] > 0


def test_multi_leg_on_one_device():
    """--multi-leg on: the same host-fed query through cq_multi_* (here the one-rank communicator) equals the
    single-device counts and is reported in the line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + TOY + ["--multi-leg", "on", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["multi_in_process"]["equals_single_device"] is True and j["multi_in_process"]["devices"] == [0]


def test_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun: the launcher in bench.py starts two fresh ranks before it touches
    torch or the GPU, forwards rank 0's line and returns the ranks' status (rehearsal mode: both ranks on cuda:0,
    control plane and reduction over gloo -- RCCL needs one GPU per rank)."""
    env = dict(os.environ, CAMMIQ_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TOY, capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and "sharded x2" in j["config"]["parallelism"] and "gloo" in j["config"]["parallelism"]
    assert j["outcome"]["reads"] == 2 * 2 * 200000


def test_a_failing_rank_fails_the_launcher():
    """No fallback collective and no swallowed failures: a rank that cannot run ends `--gpus 2` non-zero."""
    env = dict(os.environ, CAMMIQ_BENCH_REHEARSAL="1", CAMMIQ_LIB="/nonexistent/libcammiq_hip.so")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TOY, capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


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


def test_a_rank_stuck_past_the_deadline_ends_the_run_with_its_stage():
    """One rank sleeps past a 5 s per-stage deadline (after its reads are resident, where a hung RCCL bring-up would
    sit): the launcher stops both ranks by PID, returns 124 and names the stage -- nothing hangs until the harness's
    own limit, no rank is left on the GPU."""
    import re
    env = dict(os.environ, CAMMIQ_BENCH_REHEARSAL="1", CAMMIQ_BENCH_TEST_STALL="1:reads_resident:600",
               CAMMIQ_BENCH_STAGE_TIMEOUT="5", CAMMIQ_BENCH_KILL_GRACE="3")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + TOY, capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 124, r.stderr[-2000:]
    # (rank 0 waits for rank 1 in the next barrier, so whichever entered its stage first trips the deadline)
    assert "deadline: rank" in r.stderr and "spent more than 5 s in stage=" in r.stderr and "rank 1: reads_resident" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    for pid in re.findall(r"\[bench launcher\] rank \d+ pid (\d+)", r.stderr):
        with pytest.raises(ProcessLookupError):
            os.kill(int(pid), 0)

"""bench.py's own launcher for N > 1 (no GPU needed): `python bench.py --gpus 2` without torchrun starts two
fresh ranks, and when the ranks cannot run -- here: no GPU -- it returns non-zero, prints no JSON line and leaves
no process behind."""
import os
import re
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_returns_the_ranks_status_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the GPU suite covers the launcher on one)")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genomes", "4", "--genome-len",
                        "20000", "--reads", "1000", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0
    # a rank said why (the other one may have been stopped by the launcher before it got that far)
    assert 1 <= r.stderr.count("bench.py needs a GPU") <= 2, r.stderr[-1500:]
    assert "[bench launcher] rank" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_configs4_sizing_rule():
    """tools/configs4.py shrinks configs[4]'s genomes to what the host can build (the index build is budgeted at ~90 bytes of
    host memory per marker; a one-GPU lease is capped near 270 GiB): full size on a big box, proportionally less on a
    small one, and the at-size GPU test skips below 400 kbp genomes."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import configs4
    full, _, _ = configs4.size_for_this_box(15_000, 3_450_000, host_budget_bytes=400e9, shm_bytes=400e9)
    assert full == 3_450_000
    half, _, _ = configs4.size_for_this_box(15_000, 3_450_000, host_budget_bytes=100e9, shm_bytes=400e9)
    assert 1_000_000 < half < 3_450_000
    markers = 15_000 * half * configs4.MARKERS_PER_GENOME_BASE
    assert markers * configs4.BYTES_PER_MARKER_HOST_PEAK <= 0.72 * 100e9 * 1.001
    tiny, _, _ = configs4.size_for_this_box(15_000, 3_450_000, host_budget_bytes=10e9, shm_bytes=400e9)
    assert tiny < 400_000


def _clean_env(**kw):
    env = dict(os.environ, **kw)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


TOY = ["--gpus", "2", "--genomes", "4", "--genome-len", "20000", "--reads", "1000", "--steps", "1", "--warmup", "0"]


def _pids(stderr):
    return [int(x) for x in re.findall(r"\[bench launcher\] rank \d+ pid (\d+)", stderr)]


def _gone(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return True
    # still in the process table: a zombie the launcher has already waited for cannot exist, so it is alive
    return False


def test_launcher_deadline_names_the_stage_and_leaves_no_rank_behind():
    """A rank that stays in one stage past the deadline (here: both sleep at `started`, before torch is imported) ends
    the run: non-zero status, the stage named per rank, every rank's PID gone."""
    env = _clean_env(CAMMIQ_BENCH_TEST_STALL="*:started:600", CAMMIQ_BENCH_STAGE_TIMEOUT="2", CAMMIQ_BENCH_KILL_GRACE="2")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + TOY, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 124, r.stderr[-1500:]
    assert time.time() - t0 < 60
    assert "deadline: rank 0 spent more than 2 s in stage=started" in r.stderr and "rank 1: started" in r.stderr, r.stderr[-1500:]
    pids = _pids(r.stderr)
    assert len(pids) == 2 and all(_gone(p) for p in pids)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_sigterm_to_the_launcher_takes_every_rank_down(tmp_path):
    """SIGTERM reaches only the launcher (a harness killing the job): its handler stops every rank by PID."""
    env = _clean_env(CAMMIQ_BENCH_TEST_STALL="*:started:600", CAMMIQ_BENCH_KILL_GRACE="2")
    errf = tmp_path / "stderr.txt"
    with open(errf, "w") as fe:
        p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + TOY, stdout=subprocess.DEVNULL, stderr=fe, env=env)
        t0 = time.time()
        while True:
            err = errf.read_text()
            if len(_pids(err)) == 2 and err.count("stage=started") == 2:
                break
            assert time.time() - t0 < 60 and p.poll() is None, err[-1500:]
            time.sleep(0.05)
        pids = _pids(err)
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=30)
    err = errf.read_text()
    assert p.returncode == 128 + signal.SIGTERM, err[-1500:]
    assert "signal 15 received" in err and all(_gone(x) for x in pids), err[-1500:]

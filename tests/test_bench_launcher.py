"""bench.py's own launcher for N > 1 (no GPU needed): `python bench.py --gpus 2` without torchrun starts two
fresh ranks, and when the ranks cannot run -- here: no GPU -- it returns non-zero, prints no JSON line and leaves
no process behind."""
import os
import subprocess
import sys

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

"""bench.py's own launcher (`python bench.py --gpus N`, N > 1, outside torch.distributed.run) on a box without GPUs: it must refuse
or fail loudly with a non-zero code and never hang -- the rank processes themselves need MI355X GPUs (no CPU fallback)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "VQSEG_DIST_REHEARSAL")}
    env.update(kw)
    return env


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


def test_launcher_refuses_more_ranks_than_gpus():
    if not _no_gpu():
        import pytest
        pytest.skip("box has GPUs")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 2 and "shows 0 GPU(s)" in res.stderr and res.stdout.strip() == ""


def test_launcher_returns_the_ranks_failure_and_ends_the_other_rank():
    if not _no_gpu():
        import pytest
        pytest.skip("box has GPUs")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=_env(VQSEG_DIST_REHEARSAL="1"), capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "needs MI355X GPUs" in res.stderr and "launcher: rank" in res.stderr

"""The data-parallel code path on RCCL itself, as far as a one-GPU box can take it: a ONE-rank `nccl` process group
(VQSEG_DIST_SINGLE=1, vq_seg_amd.dist.collectives_on) under the real CPSTrainer.  Every collective is then the identity,
so the run must reproduce the plain single-process run BIT FOR BIT -- while communicator initialisation with `device_id`,
`ReduceOp.AVG` on the fp32 gradient buckets, the async work handles launched from inside backward, the stream ordering
between the gradient kernels' side streams and RCCL's, the int64 / fp32 k-means all-reduces and the parameter / buffer
broadcasts all run on the library the 8-GPU job uses (tests/test_dp_gpu.py covers world_size 2 through gloo)."""
import os
import subprocess
import sys

import pytest

from tests.test_dp_cpu import _free_port

pytestmark = pytest.mark.gpu

JOB = r"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.environ["VQSEG_ROOT"])
from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
from vq_seg_amd import dist as vdist

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
model = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                             "vq_cfg": {"num_embeddings": [0, 0, 32, 32, 32], "distance": "euclidean", "kmeans_init": True},
                                             "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}

def run():
    tr = CPSTrainer(CPSConfig(model=model, recipe="v1", total_iters=10, amp_dtype=torch.bfloat16, bucket_mb=16.0), dev)
    data = SyntheticCropWeed(64, 2, dev, seed=5)
    in_bwd, losses = [], []
    for _ in range(3):
        (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
        out = tr.step(l_in, l_tg, ul_in)
        losses.append(float(out["loss"]))
        in_bwd.append([list(b.launched_in_backward) for b in tr.buckets])
    tr.sync_buffers()
    torch.cuda.synchronize()
    state = [t.detach().clone() for m in tr.models for t in list(m.parameters()) + list(m.buffers())]
    return state, losses, in_bwd, tr

plain, plain_losses, _, _ = run()                       # no process group: the collectives are off
assert not vdist.collectives_on()

dist.init_process_group("nccl", device_id=dev)          # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1 and vdist.collectives_on()
rccl, rccl_losses, in_bwd, tr = run()
assert all(b._avg == dist.ReduceOp.AVG for b in tr.buckets), "the RCCL path reduces with ReduceOp.AVG"
assert all(len(b.buckets) >= 3 for b in tr.buckets)
for m in range(2):
    assert not all(in_bwd[0][m]) and all(in_bwd[1][m]) and all(in_bwd[2][m]), in_bwd   # as under gloo (test_dp_gpu.py)
assert plain_losses == rccl_losses, (plain_losses, rccl_losses)
assert len(plain) == len(rccl) and all(torch.equal(a, b) for a, b in zip(plain, rccl)), "a one-rank RCCL run must equal the plain run"
t = torch.arange(8, device=dev, dtype=torch.int64)
dist.all_reduce(t)                                      # the k-means count message type
assert torch.equal(t.cpu(), torch.arange(8))
dist.barrier()
dist.destroy_process_group()
print("RCCL_SINGLE_OK", rccl_losses)
"""


def test_cps_trainer_on_a_one_rank_rccl_group_equals_the_plain_run():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQSEG_DIST_SINGLE="1", VQSEG_ROOT=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", JOB], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and "RCCL_SINGLE_OK" in res.stdout, res.stdout[-2000:] + "\n" + res.stderr[-4000:]


def test_bench_two_rank_rehearsal_with_extras_completes():
    """`bench.py` exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), here as the gloo rehearsal
    with both ranks on cuda:0: the timed region AND the extra legs after it (supervised step, all-bf16 step, the event-bracketed
    roofline_conv steps) hold gradient all-reduces, so every rank has to run every one of them -- a leg run by rank 0 alone leaves the
    job hanging in a collective (it did, before round 2).  The rehearsal also asserts bit-identical parameters on both ranks."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VQSEG_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-4000:]
    import json
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and "rehearsal" in line["config"]["collectives"]
    assert line["roofline_conv"]["by_kind"]["3x3 bf16"]["launches_per_step"] > 0 and line["all_bf16_step"]["images_per_sec"] > 0
    assert "identical on all 2 ranks" in res.stderr


def test_plain_bench_gpus_2_launches_its_own_ranks():
    """`python3 bench.py --gpus 2` in exactly the plain form (no torch.distributed.run, no WORLD_SIZE): the launcher starts two fresh
    rank processes before any HIP call, relays rank 0's one JSON line and returns the worst child's code.  On a
    one-GPU box the ranks share cuda:0 over gloo (VQSEG_DIST_REHEARSAL=1)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VQSEG_DIST_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "1", "--no-extras"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["value"] > 0
    assert "identical on all 2 ranks" in res.stderr

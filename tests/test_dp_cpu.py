"""Data-parallel host logic on CPU with the gloo backend, world_size 2 (the GPU path uses the same code over RCCL).

  * GradBuckets: bucketed, hook-driven gradient all-reduce == the average of the per-rank gradients,
    including parameters that receive no gradient (the frozen codebooks / v1 prototypes).
  * distributed k-means init: rank 0's initial means are broadcast and per-iteration sums/counts are
    all-reduced, so two ranks holding half the rows each reproduce single-process Lloyd on all rows.
    (The per-rank accumulate kernel is replaced by the CPU oracle here: no GPU in this test.)
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def spawn(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _grad_job(rank, world):
    from vq_seg_amd.trainer import GradBuckets
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
    frozen = nn.Parameter(torch.ones(5))                     # never receives a gradient
    params = list(net.parameters()) + [frozen]
    buckets = GradBuckets(params, bucket_mb=0.0003)          # tiny buckets -> several all-reduces
    assert len(buckets.buckets) >= 3
    x = torch.arange(8.0).repeat(3, 1) * (rank + 1)
    in_bwd = []
    for _ in range(3):                                       # later iterations check zero() / re-arming
        buckets.zero()
        net(x).pow(2).sum().backward()
        buckets.finish()
        in_bwd.append(list(buckets.launched_in_backward))
    # step 1: the bucket holding the gradient-free parameter cannot complete inside backward; once it has been learnt
    # (finish() of step 1) every bucket is reduced from inside backward
    assert not all(in_bwd[0]) and all(in_bwd[1]) and all(in_bwd[2]), in_bwd
    assert buckets._silent == {id(frozen)}
    # r4: every bucket all-reduce carries its place in the issue order; all ranks must issue the same sequence (a rank that reduced
    # its buckets in another order would pair different buffers in the collectives) -- bench.py asserts the same in every N > 1 run
    seq = torch.tensor(buckets.launch_sequence)
    assert sorted(seq.tolist()) == list(range(len(buckets.buckets)))
    seqs = [torch.zeros_like(seq) for _ in range(world)]
    dist.all_gather(seqs, seq)
    assert all(torch.equal(q, seqs[0]) for q in seqs), seqs
    return [p.grad.clone() for p in params]


def test_grad_buckets_average_over_ranks():
    out = spawn(_grad_job)
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
    ref = [torch.zeros_like(p) for p in net.parameters()]
    for rank in range(2):
        net.zero_grad()
        x = torch.arange(8.0).repeat(3, 1) * (rank + 1)
        net(x).pow(2).sum().backward()
        ref = [r + p.grad / 2 for r, p in zip(ref, net.parameters())]
    for rank in range(2):
        for g, r in zip(out[rank][:-1], ref):
            assert torch.allclose(g, r, rtol=1e-6, atol=1e-7)
        assert torch.equal(out[rank][-1], torch.zeros(5))
    assert all(torch.equal(a, b) for a, b in zip(out[0], out[1]))      # every rank holds the same averaged gradient


def _kmeans_job(rank, world):
    from oracle import torch_ref
    from tests import synth
    from vq_seg_amd import _hip
    from vq_seg_amd.vector_quantizer import vq_img

    def accumulate(samples, means):                          # CPU stand-in for vqseg_kmeans_accumulate_f32
        idx = torch.argmin(torch.cdist(samples, means), dim=-1)
        k, c = means.shape
        sums = torch.zeros(k, c).index_add_(0, idx, samples)
        return sums, torch.bincount(idx, minlength=k)

    def finalize(sums, counts, means):                       # vqseg_kmeans_finalize_f32
        nz = counts > 0
        means[nz] = sums[nz] / counts[nz, None].float()
        return means

    _hip.kmeans_accumulate, _hip.kmeans_finalize = accumulate, finalize
    rows = synth.relu_features(7, (512, 16))
    mine = rows[rank::world].contiguous()
    init = rows[:8].clone() if rank == 0 else torch.zeros(8, 16)      # only rank 0's draw counts
    means, counts = vq_img.kmeans(mine, 8, 10, init_means=init)
    return means, counts


def test_distributed_kmeans_matches_single_process():
    from oracle import torch_ref
    from tests import synth
    out = spawn(_kmeans_job)
    rows = synth.relu_features(7, (512, 16))
    ref_means, ref_bins = torch_ref.kmeans_lloyd(rows, rows[:8].clone(), 10)
    for means, counts in out:
        assert torch.equal(counts, ref_bins)
        assert torch.allclose(means, ref_means, rtol=1e-5, atol=1e-6)
    assert torch.equal(out[0][0], out[1][0])


def test_synthetic_data_is_sharded_by_rank():
    from vq_seg_amd.trainer import SyntheticCropWeed
    a = SyntheticCropWeed(32, 2, "cpu", seed=1)
    img, lab = a.labelled()
    assert img.shape == (2, 3, 32, 32) and lab.shape == (2, 32, 32) and img.min() >= 0 and img.max() <= 1
    assert set(lab.unique().tolist()) <= {0, 1, 2}


def _flat_broadcast_and_resume_job(rank, world):
    """CPSTrainer's rank-0 -> all broadcast of the initial state (one flat buffer per dtype) and the per-rank BatchNorm statistics of
    a data-parallel checkpoint, on the host logic alone (the models here are plain torch modules: no kernel runs)."""
    import tempfile
    from vq_seg_amd import trainer as T
    from vq_seg_amd.utils.ckpoints import load_training_state
    torch.manual_seed(100 + rank)                            # different initial state per rank
    nets = [nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4)) for _ in range(2)]
    tr = T.CPSTrainer.__new__(T.CPSTrainer)                  # the two methods under test need only these attributes
    tr.models, tr.device, tr.iter = nets, torch.device("cpu"), 7
    tr.opts = [torch.optim.Adam(m.parameters(), lr=1e-3) for m in nets]
    for m in tr.models:                                      # what the constructor does under data parallelism
        T.broadcast_module_state(m)
    state = torch.cat([t.detach().double().reshape(-1) for m in nets for t in list(m.parameters()) + list(m.buffers())])
    # -- per-rank statistics, then a checkpoint written by rank 0 that keeps every rank's own
    for m in nets:
        m[1].running_mean.fill_(float(rank + 1))
        m[1].num_batches_tracked.fill_(10 * (rank + 1))
    path = os.path.join(tempfile.gettempdir(), f"vqseg_dp_ckpt_{os.environ['MASTER_PORT']}.pt")
    tr.save_checkpoint(path)
    dist.barrier()
    for m in nets:
        m[1].running_mean.zero_()
        m[1].num_batches_tracked.zero_()
    tr.iter = 0
    tr.load_checkpoint(path)
    ok = all(float(m[1].running_mean[0]) == float(rank + 1) and int(m[1].num_batches_tracked) == 10 * (rank + 1) for m in nets) and tr.iter == 7
    dist.barrier()
    if rank == 0:
        keys = set(load_training_state(path))
        os.remove(path)
        assert {"model_1", "model_2", "epoch", "batch_idx", "optimizer_1", "optimizer_2"} <= keys      # the reference's layout is intact
    return state, ok


def test_initial_broadcast_is_flat_and_dp_resume_keeps_per_rank_bn_statistics():
    out = spawn(_flat_broadcast_and_resume_job)
    assert torch.equal(out[0][0], out[1][0])                 # every rank starts from rank 0's parameters and buffers
    assert out[0][1] and out[1][1]                           # ranks > 0 get THEIR OWN running statistics back on resume

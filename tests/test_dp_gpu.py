"""Data-parallel gradient path on the GPU kernels, world_size 2 (both ranks share cuda:0; gloo stands in for RCCL,
the bucket / sink logic is the same): the HIP weight-gradient kernels accumulate straight into the bucket views
(nnf grad sinks), a module used twice in one graph reports once, every bucket is all-reduced during backward,
and the result equals the average of the per-rank single-process gradients."""
import copy

import pytest
import torch
from torch import nn

from tests import synth
from tests.test_dp_cpu import spawn

pytestmark = pytest.mark.gpu


class TwoUse(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.c1, self.b1 = nn.Conv2d(32, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64)
        self.c2, self.b2 = nn.Conv2d(64, 64, 3, 1, 1, bias=False, padding_mode="reflect"), nn.BatchNorm2d(64)
        self.lin = nn.Conv2d(64, 4, 1)                         # plain torch op: its gradient arrives through autograd

    def forward(self, x):
        from vq_seg_amd import nnf
        h = nnf.conv_bn_act(x, self.c1, self.b1)
        h = nnf.conv_bn_act(h, self.c2, self.b2)
        h = nnf.conv_bn_act(h, self.c2, self.b2)               # second use of the same parameters
        return self.lin(h)


def _inputs(rank):
    return synth.uniform(40 + rank, (2, 32, 16, 16), -1, 1).cuda().contiguous(memory_format=torch.channels_last)


def _job(rank, world):
    from vq_seg_amd.trainer import GradBuckets
    torch.cuda.set_device(0)
    net = TwoUse().cuda()
    buckets = GradBuckets(list(net.parameters()), bucket_mb=0.02)
    assert len(buckets.buckets) >= 3
    launched_in_backward = None
    for _ in range(2):                                         # second iteration checks zero() / re-arming
        buckets.zero()
        net(_inputs(rank)).square().sum().backward()
        launched_in_backward = list(buckets._launched)
        assert all(c == 0 for c in buckets._pending), "every parameter must report exactly once (sink or hook, not both)"
        buckets.finish()
    return [p.grad.detach().cpu().clone() for p in net.parameters()], launched_in_backward


def test_sinks_and_buckets_average_over_ranks():
    out = spawn(_job)
    ref = None
    for rank in range(2):
        net = TwoUse().cuda()
        net(_inputs(rank)).square().sum().backward()
        g = [p.grad.detach().cpu() / 2 for p in net.parameters()]
        ref = g if ref is None else [a + b for a, b in zip(ref, g)]
    for rank in range(2):
        grads, launched = out[rank]
        assert all(launched), "every bucket must have been all-reduced from inside backward"
        for a, b in zip(grads, ref):
            assert ((a - b).abs().max() / (b.abs().max() + 1e-12)).item() < 1e-5
    assert all(torch.equal(a, b) for a, b in zip(out[0][0], out[1][0]))


# ---- EXTENSION: EMA codebook update under data parallelism (per-code sums / counts all-reduced before the update) ----
def _ema_inputs(rank, step):
    return synth.relu_features(70 + 10 * step + rank, (2, 64, 12, 12)).cuda()


def _ema_module():
    from vq_seg_amd.vector_quantizer import VectorQuantizer
    vq = VectorQuantizer(dim=64, num_embeddings=40, decay=0.9, eps=1e-5, ema_update=True).cuda()
    W = synth.relu_features(10, (40, 64)).cuda()
    with torch.no_grad():
        vq.codebook.embedding.weight.copy_(W)
        vq.codebook.embed_avg.copy_(W)
        vq.codebook.cluster_size.fill_(1.0)
    return vq.train()


def _ema_job(rank, world):
    torch.cuda.set_device(0)
    vq = _ema_module()
    for step in range(2):
        vq(_ema_inputs(rank, step))
    torch.cuda.synchronize()
    return vq.codebook.embedding.weight.detach().cpu(), vq.codebook.cluster_size.cpu()


def test_ema_codebook_stays_identical_across_ranks():
    out = spawn(_ema_job)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    vq = _ema_module()                                       # one process on the union of the two ranks' batches
    for step in range(2):
        vq(torch.cat([_ema_inputs(0, step), _ema_inputs(1, step)], dim=0))
    ref = vq.codebook.embedding.weight.detach().cpu()
    assert ((out[0][0] - ref).abs().max() / ref.abs().max()).item() < 1e-5
    assert torch.allclose(out[0][1], vq.codebook.cluster_size.cpu(), rtol=1e-6)


# ---- the real trainer under data parallelism: which buckets are reduced from inside backward, identical replicas ----
def _trainer_job(rank, world):
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    model = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                                 "vq_cfg": {"num_embeddings": [0, 0, 32, 32, 32], "distance": "euclidean", "kmeans_init": True},
                                                 "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    tr = CPSTrainer(CPSConfig(model=model, recipe="v1", total_iters=10, amp_dtype=torch.bfloat16, bucket_mb=16.0), dev)
    data = SyntheticCropWeed(64, 2, dev, seed=5)             # seed * 1000 + rank: every rank its own shard
    in_bwd = []
    for _ in range(3):
        (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
        tr.step(l_in, l_tg, ul_in)
        in_bwd.append([list(b.launched_in_backward) for b in tr.buckets])
    silent = [sorted(n for n, p in m.named_parameters() if id(p) in b._silent) for m, b in zip(tr.models, tr.buckets)]
    torch.cuda.synchronize()
    chk = torch.stack([p.detach().double().sum() for m in tr.models for p in m.parameters()]).cpu()
    tr.sync_buffers()
    buf = torch.stack([b.detach().double().sum() for m in tr.models for b in m.buffers()]).cpu()
    return in_bwd, silent, chk, buf, [len(b.buckets) for b in tr.buckets]


def test_cps_trainer_two_ranks_buckets_launch_inside_backward():
    """the codebooks (and v1's prototypes) never report a gradient; counted as pending they kept their buckets from
    completing inside backward on the real model.  They are learnt on the first step: from step 2 on every bucket of both networks is
    all-reduced from inside backward; replicas stay bit-identical; BatchNorm buffers agree after sync_buffers()."""
    out = spawn(_trainer_job)
    for in_bwd, silent, _chk, _buf, n_buckets in out:
        assert all(n >= 3 for n in n_buckets)
        for m in range(2):
            assert not all(in_bwd[0][m]), "step 1 cannot complete the buckets that hold gradient-free parameters"
            assert all(in_bwd[1][m]) and all(in_bwd[2][m]), in_bwd
            assert silent[m] == sorted([f"codebook.{i}.codebook.embedding.weight" for i in (2, 3, 4)] + ["prototype_loss.embedding.weight"])
    assert torch.equal(out[0][2], out[1][2]), "replicas diverged"
    assert torch.equal(out[0][3], out[1][3]), "buffers differ after sync_buffers()"

"""CPSTrainer.step on the GPU: running the two networks of the pair on their own HIP streams must not change a single
bit of the result (every kernel of the path is deterministic, so any difference would be a missing stream dependency),
in both recipes and both activation precisions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(two_streams, recipe, amp):
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
    dev = torch.device("cuda:0")
    name = "vqreptunet1x1" if recipe == "v1" else "vqreptunet1x1v2"
    model = {"name": name, "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                      "vq_cfg": {"num_embeddings": [0, 0, 64, 64, 64], "distance": "euclidean", "kmeans_init": True},
                                      "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    torch.manual_seed(0)
    cfg = CPSConfig(model=model, recipe=recipe, total_iters=10, amp_dtype=torch.bfloat16 if amp else None, two_streams=two_streams)
    tr = CPSTrainer(cfg, dev)
    assert tr._two_streams == two_streams
    data = SyntheticCropWeed(64, 2, dev, seed=5)
    (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
    losses = []
    for _ in range(3):
        losses.append(tr.step(l_in, l_tg, ul_in)["loss"].item())
    torch.cuda.synchronize()
    chk = torch.stack([p.detach().double().sum() for m in tr.models for p in m.parameters()]).cpu()
    return losses, chk


@pytest.mark.parametrize("recipe,amp", [("v1", True), ("v2", False), ("v2", True), ("v1", False)])
def test_two_streams_change_nothing(recipe, amp):
    la, ca = _run(False, recipe, amp)
    lb, cb = _run(True, recipe, amp)
    assert all(l == l and abs(l) < 1e6 for l in la)           # finite
    # the loss VALUE passes through ATen's cross-entropy forward (v2 recipe), whose atomics make its last bit vary from
    # run to run; every gradient kernel is deterministic, so the updated parameters must agree bit for bit
    assert all(abs(u - v) <= 1e-6 * abs(u) for u, v in zip(la, lb)), (la, lb)
    assert torch.equal(ca, cb)


def test_training_learns_the_synthetic_task():
    """End to end: 80 CPS steps (v1 recipe, bf16 activations, two streams) on the synthetic crop/weed blobs must drive the
    supervised loss down and the mean IoU of the labelled batch up -- every forward kernel, every backward kernel, the
    bucketed gradients and the fused Adam step have to cooperate for that.  (CWFID itself cannot travel to the GPU box;
    tools/learn_synthetic.py prints the whole curve.)"""
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
    dev = torch.device("cuda:0")
    model = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                                 "vq_cfg": {"num_embeddings": [0, 0, 64, 64, 64], "distance": "euclidean", "kmeans_init": True},
                                                 "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    torch.manual_seed(0)
    tr = CPSTrainer(CPSConfig(model=model, recipe="v1", total_iters=200, amp_dtype=torch.bfloat16, learning_rate=1e-3), dev)
    data = SyntheticCropWeed(128, 8, dev, seed=5)
    first, last = [], []
    for i in range(80):
        (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
        out = tr.step(l_in, l_tg, ul_in, epoch_frac=i / 80)
        if i < 5 or i >= 75:
            (first if i < 5 else last).append((out["sup_loss_1"].item(), out["miou"].item()))
    sup0, miou0 = (sum(v) / len(v) for v in zip(*first))
    sup1, miou1 = (sum(v) / len(v) for v in zip(*last))
    assert sup1 < 0.8 * sup0, (sup0, sup1)
    assert miou1 > miou0 + 0.2, (miou0, miou1)


def test_cps_steps_with_the_ema_codebook_extension():
    """vq_cfg {"ema_update": true}: the codebooks move with every training forward (they are frozen otherwise), stay
    finite, and the step still trains in both stream modes with bit-identical parameters."""
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
    dev = torch.device("cuda:0")
    model = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                                 "vq_cfg": {"num_embeddings": [0, 0, 64, 64, 64], "distance": "euclidean", "kmeans_init": True,
                                                            "decay": 0.8, "ema_update": True},
                                                 "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    sums = []
    for two in (False, True):
        torch.manual_seed(0)
        tr = CPSTrainer(CPSConfig(model=model, recipe="v1", total_iters=10, amp_dtype=torch.bfloat16, two_streams=two), dev)
        data = SyntheticCropWeed(64, 2, dev, seed=5)
        (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
        tr.step(l_in, l_tg, ul_in)
        cb = tr.models[0].codebook[2].codebook
        w1 = cb.embedding.weight.detach().clone()
        out = tr.step(l_in, l_tg, ul_in)
        assert torch.isfinite(out["loss"]).item() and torch.isfinite(cb.embedding.weight).all()
        assert not torch.equal(cb.embedding.weight, w1), "the EMA update must move the codebook"
        torch.cuda.synchronize()
        sums.append(torch.stack([p.detach().double().sum() for m in tr.models for p in m.parameters()]).cpu())
    assert torch.equal(sums[0], sums[1])


def test_shared_stem_patches_change_nothing():
    """The step unfolds each batch once for its six stems (nnf.stem_share_*): parameters after two steps are bit-identical
    to unfolding in every forward."""
    from vq_seg_amd import _hip
    res = []
    for share in (1, 0):
        _hip.lib()
        _hip.PY_OPTS["py_stem_share"] = share
        try:
            res.append(_run(True, "v1", True)[1])
        finally:
            _hip.PY_OPTS.pop("py_stem_share", None)
    assert torch.equal(res[0], res[1])


def test_checkpoint_resume_is_bit_exact(tmp_path):
    """Checkpoint / resume: two steps, save_checkpoint (the reference's dictionary layout + `initted` flags +
    the schedule's iteration counter), load into a DIFFERENTLY initialised trainer, two more steps == four uninterrupted steps,
    bit for bit: parameters, BatchNorm buffers, codebooks, Adam moments; and k-means does not re-run on the loaded codebooks."""
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
    from vq_seg_amd.utils.ckpoints import load_ckpoints
    dev = torch.device("cuda:0")
    model = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                                 "vq_cfg": {"num_embeddings": [0, 0, 32, 32, 32], "distance": "euclidean", "kmeans_init": True},
                                                 "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}

    def trainer(seed):
        return CPSTrainer(CPSConfig(model=model, recipe="v1", total_iters=8, amp_dtype=torch.bfloat16, seed=seed), dev)
    data = SyntheticCropWeed(64, 2, dev, seed=21)
    batches = [(data.labelled(), data.unlabelled()) for _ in range(4)]

    def run(tr, sel):
        for (l_in, l_tg), ul_in in sel:
            tr.step(l_in, l_tg, ul_in)

    def state(tr):
        torch.cuda.synchronize()
        out = [t.detach().clone() for m in tr.models for t in m.state_dict().values()]
        for o in tr.opts:
            for st in o.state_dict()["state"].values():
                out += [v.detach().clone() for v in st.values() if torch.is_tensor(v)]
        return out

    a = trainer(42)
    run(a, batches)
    b = trainer(42)
    run(b, batches[:2])
    path = str(tmp_path / "ck.pth")
    b.save_checkpoint(path, epoch=3, batch_idx=7)
    c = trainer(7)                                                   # other initial weights: everything must come from the file
    assert c.load_checkpoint(path) == (3, 7) and c.iter == 2
    assert all(m.codebook[i].codebook.initted for m in c.models for i in (2, 3, 4)) and all(m.prototype_loss.initted for m in c.models)
    codebooks = [m.codebook[2].codebook.embedding.weight.detach().clone() for m in c.models]
    run(c, batches[2:])
    for m, w in zip(c.models, codebooks):
        assert torch.equal(m.codebook[2].codebook.embedding.weight, w), "k-means re-ran over the loaded codebook"
    sa, sc = state(a), state(c)
    assert len(sa) == len(sc) and all(torch.equal(x, y) for x, y in zip(sa, sc)), "resumed run differs from the uninterrupted one"
    # the file is the reference's layout: its loader returns model_2 (training) / model_1 (testing), utils/ckpoints.py:15-21
    m2, epoch, batch_idx, o1, o2 = load_ckpoints(path, True, map_location="cpu")
    assert (epoch, batch_idx) == (3, 7) and set(m2) == set(b.models[1].state_dict()) and "state" in o1 and "param_groups" in o2

"""GPU parity of the model surface (encoder -> VQ -> decoder -> head -> losses) against the golden
vectors captured from the reference, fp32 mode.  Tolerance for logits: 1e-3 relative (north_star)."""
import pytest
import torch
import torch.nn.functional as F

from tests import cases, golden_io, synth

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def close(a, b, rtol, atol=0.0, what=""):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    assert torch.allclose(a, b, rtol=rtol, atol=atol), f"{what}: max abs err {err:.3e} (ref max {b.abs().max().item():.3e})"


def rel_close(a, b, tol, what=""):
    """max |a-b| <= tol * max |b|  (the 'relative to the tensor scale' reading of the 1e-3 logit bar)"""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol:g} * {scale:.3e}"


def grad_close(a, b, l2tol, maxtol, what=""):
    """Gradients pass through ReLU masks: where a pre-activation lies within rounding error of zero, two correct
    implementations may mask differently, which moves a FEW entries by a visible amount.  So gradients are held to
    a tight relative L2 error and a loose max error instead of a tight max error."""
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    l2 = ((a - b).norm() / (b.norm() + 1e-30)).item()
    mx = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
    assert l2 <= l2tol and mx <= maxtol, f"{what}: rel L2 err {l2:.3e} (tol {l2tol:g}), max err {mx:.3e} of scale (tol {maxtol:g})"


def build(name, margin, scale, seed, size=64, inputs=None):
    from vq_seg_amd.models.networks import make_model
    cfg = {"name": name, "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                    "vq_cfg": {"num_embeddings": [0, 0, 512, 512, 512], "distance": "euclidean",
                                               "kmeans_init": True},
                                    "margin": margin, "scale": scale, "use_feature": False, "encoder_weights": None}}
    model = make_model(cfg)
    sd = synth.synth_state_dict(golden_io.layout("vqreptunet1x1"), seed)
    model.load_state_dict(sd)
    model.prototype_loss.initted = True
    model = model.to(dev())
    x, gt = inputs if inputs is not None else cases.model_inputs(s=size)[:2]
    version = 1 if name == "vqreptunet1x1" else 2
    cases.prepare_module_model(model, x.to(dev()), gt.to(dev()), version,
                               to_input=lambda t: t.contiguous(memory_format=torch.channels_last))
    return model


# Gradient bars (relative L2, max of scale) per fixture.  Measured on MI355X: 64^2 1.7e-2 / 6.6e-2, 128^2 2.1e-2 / 3.7e-2.  The
# 128^2 fixture (32 samples per channel at the deepest level instead of 8) was added to test whether the small BatchNorm sample
# count explains the 64^2 error: it does not -- the CPU oracle ITSELF moves these gradients by 3e-3..4.5e-3 of scale under a
# 1e-6 input perturbation at BOTH sizes (tests/diagnostics/conditioning.py 64|128 1e-6: ReLU masks flipping through 53 convolutions),
# i.e. an amplification of ~3000x, and the fp32-precise kernels differ from ATen's CPU fp32 by ~5e-6 per layer.
GRAD_TOL = {"model_v1": (3e-2, 0.10), "model_v2": (3e-2, 0.10), "model_v1_128": (3e-2, 0.06)}


@pytest.mark.parametrize("fixture", ["model_v1", "model_v2", "model_v1_128"])
def test_whole_model_matches_reference_golden(fixture):
    fx = golden_io.load(fixture)
    version, size = fx.meta["version"], fx.meta.get("size", 64)
    model = build(fx.meta["name"], fx.meta["margin"], fx.meta["scale"], fx.meta["model_seed"], size)
    x, gt, scores = cases.model_inputs(s=size)
    x, gt, scores = x.to(dev()), gt.to(dev()), scores.to(dev())
    model.eval()
    with torch.no_grad():
        logits, closs, usage, proto = model(x)
        feats = model.encoder(x.contiguous(memory_format=torch.channels_last))[1:]
        for lvl in (2, 3, 4):
            idx = model.codebook[lvl](feats[lvl])[1]
            assert torch.equal(idx.cpu(), fx[f"eval_idx{lvl}"]), f"level {lvl} code indices differ"
    rel_close(logits, fx["eval_logits"], 1e-3, "eval logits")
    assert usage.device.type == "cpu" and usage.shape == (3,)
    close(usage, fx["eval_usage"], rtol=1e-6, what="usage")
    assert proto is None and closs.shape == (1,) and closs.item() == 0.0

    model.train()
    kw = dict(percent=fx.meta["percent"]) if version == 1 else dict(th=fx.meta["th"])
    logits, closs, usage, proto = model(x, gt, **kw)
    rel_close(logits, fx["train_logits"], 1e-3, "train logits")
    close(closs, fx["train_loss"], rtol=1e-4, what="commitment")
    close(usage, fx["train_usage"], rtol=1e-6, what="usage")
    close(proto, fx["train_proto"], rtol=1e-4, what="prototype loss")
    total = (logits * cases.logits_cotangent(logits.shape).to(dev())).sum() + fx.meta["loss_scale"] * closs.sum()
    if version == 1:
        total = total + fx.meta["proto_scale"] * proto
    total.backward()
    named = dict(model.named_parameters())
    # Tolerance: a 1e-6 perturbation of the INPUT moves these gradients by up to 5e-3 of their scale on the CPU
    # oracle itself (tests/diagnostics/conditioning.py), and a change of the last float bit of the BatchNorm statistics
    # (a different but equally exact merge order) moves them by 1e-2.  Cross-device agreement is therefore asserted at
    # GRAD_TOL here; the per-operator gradient checks in test_nn_gpu.py carry the tight bar (2e-4 of scale in fp32).
    worst = [0.0, 0.0]
    for key in [k[5:] for k in fx if k.startswith("grad/")]:
        a, b = golden_io.probe(named[key].grad).double().cpu(), torch.as_tensor(fx["grad/" + key]).double()
        worst[0] = max(worst[0], ((a - b).norm() / (b.norm() + 1e-30)).item())
        worst[1] = max(worst[1], ((a - b).abs().max() / (b.abs().max() + 1e-30)).item())
    print(f"{fixture}: worst gradient error rel-L2 {worst[0]:.3e}, max-of-scale {worst[1]:.3e}")
    for key in [k[5:] for k in fx if k.startswith("grad/")]:
        grad_close(golden_io.probe(named[key].grad), fx["grad/" + key], *GRAD_TOL[fixture], "grad " + key)
    for i in (2, 3, 4):
        assert named[f"codebook.{i}.codebook.embedding.weight"].grad is None
    post = model.state_dict()
    for key in [k[5:] for k in fx if k.startswith("post/")]:
        rel_close(post[key], fx["post/" + key], 2e-4, key)
    if version == 2:
        model2 = build(fx.meta["name"], fx.meta["margin"], fx.meta["scale"], fx.meta["model_seed"])
        model2.load_state_dict(post)
        model2.train()
        with torch.no_grad():
            out = model2(x, scores, th=fx.meta["th"])
        close(out[3], fx["train_proto_score"], rtol=1e-4, what="prototype loss (pseudo scores)")
        # the reference's in-place margin makes its fp32 backward raise (q10); ours is differentiable
        model2.zero_grad()
        out = model2(x, gt, th=fx.meta["th"])
        out[3].backward()
        assert torch.isfinite(model2.prototype_loss.embedding.weight.grad).all()


def test_plain_unet_matches_reference_golden():
    from vq_seg_amd.loss import make_loss
    from vq_seg_amd.models.networks import make_model
    fx = golden_io.load("model_unet")
    model = make_model({"name": "unet", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                                     "encoder_weights": "imagenet_swsl"}})
    model.load_state_dict(synth.synth_state_dict(golden_io.layout("unet"), fx.meta["model_seed"]))
    model = model.to(dev())
    x, gt, _ = cases.model_inputs(b=2, s=64, seed=6500)
    x, gt = x.to(dev()), gt.to(dev())
    cases.set_bn_momentum(model, 1.0)                             # BN calibration, as in make_golden.gen_unet
    model.train()
    with torch.no_grad():
        model(x)
    cases.set_bn_momentum(model, 0.1)
    model.eval()
    with torch.no_grad():
        y = model(x)
    assert isinstance(y, torch.Tensor)                               # bare tensor (unet/net.py:833-838)
    rel_close(y, fx["eval_logits"], 1e-3, "unet eval logits")
    model.train()
    y = model(x)
    rel_close(y, fx["train_logits"], 1e-3, "unet train logits")
    loss = make_loss("dice_loss", 3, ignore_index=255)(y, gt) + 0.5 * F.cross_entropy(y, gt, ignore_index=255)
    close(loss, fx["loss"], rtol=1e-4, what="loss")
    loss.backward()
    named = dict(model.named_parameters())
    for key in [k[5:] for k in fx if k.startswith("grad/")]:
        grad_close(golden_io.probe(named[key].grad), fx["grad/" + key], 2e-2, 0.15, "grad " + key)


@pytest.mark.parametrize("name", cases.DEC_CASES)
def test_decoder_matches_reference_golden(name):
    from vq_seg_amd.models.networks.unet.decoder import UnetDecoder
    fx = golden_io.load(name)
    feats, sd, g = cases.decoder_inputs(fx.meta)
    dec = UnetDecoder(list(fx.meta["enc"]), list(fx.meta["dec"]))
    dec.load_state_dict(sd)
    dec = dec.to(dev())
    cl = lambda t: t.to(dev()).contiguous(memory_format=torch.channels_last)
    dec.eval()
    with torch.no_grad():
        y = dec(*[cl(f) for f in feats])
    rel_close(y, fx["y_eval"], 1e-3, "decoder eval")
    dec.train()
    fr = [cl(f).requires_grad_(True) for f in feats]
    y = dec(*fr)
    rel_close(y, fx["y_train"], 1e-3, "decoder train")
    (y * g.to(dev())).sum().backward()
    for i, f in enumerate(fr):
        grad_close(f.grad, fx[f"grad_feat{i}"], 1e-2, 0.1, f"grad feat {i}")
    grad_close(dec.blocks[0][0][0].weight.grad, fx["grad_w_first"], 1e-2, 0.1, "grad w first")
    grad_close(dec.blocks[4][1][0].weight.grad, fx["grad_w_last"], 1e-2, 0.1, "grad w last")
    grad_close(dec.blocks[4][1][1].weight.grad, fx["grad_bn_w_last"], 1e-2, 0.1, "grad bn w")
    grad_close(dec.blocks[4][1][1].bias.grad, fx["grad_bn_b_last"], 1e-2, 0.1, "grad bn b")
    post = dec.state_dict()
    close(post["blocks.0.0.1.running_mean"], fx["run_mean_first"], rtol=1e-4, atol=1e-6)
    close(post["blocks.0.0.1.running_var"], fx["run_var_first"], rtol=1e-4, atol=1e-6)
    close(post["blocks.4.1.1.running_var"], fx["run_var_last"], rtol=1e-4, atol=1e-6)


def test_prototype_losses_match_reference_golden():
    from vq_seg_amd.models.modules.prototype import ReliablePrototypeLoss, ReliablePrototypeLossv2
    fx = golden_io.load("prototype")
    feat, gt, scores, protos, entropy = [t.to(dev()) for t in cases.proto_inputs()]
    for tag, margin, scale in (("m0", 0.0, 1.0), ("m05", 0.5, 30.0)):
        m1 = ReliablePrototypeLoss(3, 32, scale=scale, margin=margin, init="normal").to(dev())
        with torch.no_grad():
            m1.embedding.weight.copy_(protos)
        m1.train()
        fr = feat.clone().requires_grad_(True)
        l1 = m1(fr, gt, percent=fx.meta["percent"], entropy=entropy)
        assert l1.dtype == torch.float64
        close(l1, fx[f"v1_{tag}_loss"], rtol=1e-5, what="v1 loss")
        l1.backward()
        rel_close(fr.grad, fx[f"v1_{tag}_grad"], 1e-3, "v1 grad")
        assert m1.embedding.weight.grad is None
        for kind, target in (("gt", gt), ("score", scores)):
            m2 = ReliablePrototypeLossv2(3, 32, scale=scale, margin=margin, init="normal").to(dev())
            with torch.no_grad():
                m2.embedding.weight.copy_(protos)
            m2.train()
            l2 = m2(feat, target, fx.meta["th"])
            close(l2, fx[f"v2_{tag}_{kind}_loss"], rtol=1e-4, what="v2 loss")
            close(m2.embedding.weight, fx[f"v2_{tag}_{kind}_proto_after"], rtol=1e-5, what="v2 protos")


def test_kmeans_prototype_init_and_model_first_train_forward():
    """First TRAINING forward of a kmeans_init model initialises the 3 codebooks and the prototypes (q5)."""
    from vq_seg_amd.models.networks import make_model
    torch.manual_seed(0)
    model = make_model({"name": "vqreptunet1x1v2", "params": {
        "encoder_name": "resnet50", "num_classes": 3, "depth": 5,
        "vq_cfg": {"num_embeddings": [0, 0, 32, 32, 16], "distance": "euclidean", "kmeans_init": True},
        "margin": 0.5, "scale": 30.0, "use_feature": False, "encoder_weights": None}}).to(dev())
    x, gt, _ = cases.model_inputs(b=2, s=128)
    model.eval()
    with torch.no_grad():
        model(x.to(dev()))
    assert not any(model.codebook[i].codebook.initted for i in (2, 3, 4)) and not model.prototype_loss.initted
    model.train()
    out = model(x.to(dev()), gt.to(dev()), th=0.7)
    assert all(model.codebook[i].codebook.initted for i in (2, 3, 4)) and model.prototype_loss.initted
    (out[0].mean() + out[1].sum() + out[3]).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert (out[2] < 100).all()                                   # k-means codes are in use


def test_evaluation_loop_on_hip_model():
    """evaluate.test_loop with the HIP model (eval-mode fused epilogues) == Measurement on that model's own logits."""
    import numpy as np
    from vq_seg_amd.evaluate import test_loop
    from vq_seg_amd.measurement import Measurement
    fx = golden_io.load("model_v1")
    model = build(fx.meta["name"], fx.meta["margin"], fx.meta["scale"], fx.meta["model_seed"])
    x, gt, _ = cases.model_inputs()
    tgt = torch.nn.functional.interpolate(gt[:, None].float(), scale_factor=1.5, mode="nearest")[:, 0].long()
    batches = [(x, tgt), (x.flip(0), tgt.flip(0))]
    got = test_loop(model, batches, 3, device=dev())
    meas = Measurement(3)
    miou = 0.0
    model.eval()
    with torch.no_grad():
        for img, t in batches:
            pred = F.interpolate(model(img.to(dev()))[0].float(), t.shape[-2:], mode="bilinear").cpu().numpy()
            miou += meas(pred, t.numpy())[1]
    assert abs(got["test_miou"] - miou / 2) < 1e-9 and 0.0 <= got["test_acc"] <= 1.0
    assert len(got["test_ious"]) == 3 and all(np.isfinite(got["test_ious"]))


@pytest.mark.gpu
@pytest.mark.parametrize("encoder,mode", [("resnet18", "fp32"), ("resnet34", "bf16"), ("resnet18", "bf16")])
def test_basic_block_encoders_through_the_plain_unet(encoder, mode):
    """The BasicBlock encoders (resnet18 / 34: zero-padded 3x3 convolutions, stride-2 3x3 conv1 + 1x1 projection in the first block of a
    stage, identity shortcuts elsewhere; models/encoders/resnet.py:117-190 on torchvision's BasicBlock) are not what the benchmark
    runs (ResNet-50) and had no test of their own: the plain Unet on them, training-mode forward + backward on the HIP kernels,
    against the SAME modules evaluated with stock torch operators in fp32 on the same device (Unet.forward_plumbing).
    fp32 (precise kernels): logits 2e-4, gradient probes 3e-2 (the whole-model bar of the other model tests: ReLU masks of a randomly
    initialised deep network flip at pre-activations within rounding of zero).
    bf16 under autocast (the stride-2 3x3 data gradient then takes the one-launch zero-padding path of r4): on this 2-image batch the
    gradients of a bf16 run are dominated by those flips and by BatchNorm statistics over 32 samples in layer4 -- stock torch's own
    bf16-autocast run of the same modules is 0.4-0.85 away from fp32 in rel-L2 (tools/micro/basic_block_diag.py).  The bar is therefore
    relative: the HIP run's distance to fp32 is at most 1.25 x the stock bf16 run's, probe by probe; logits 3e-2."""
    import copy
    from vq_seg_amd.models.networks import make_model

    def rel(a, b):
        a, b = a.detach().double(), b.detach().double()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    torch.manual_seed(5)
    model = make_model({"name": "unet", "params": {"encoder_name": encoder, "num_classes": 3, "depth": 5}}).to(dev())
    ref, refb = copy.deepcopy(model), copy.deepcopy(model)
    x = synth.uniform(3, (2, 3, 128, 128)).to(dev()).contiguous(memory_format=torch.channels_last)
    g = synth.uniform(4, (2, 3, 128, 128), -1, 1).to(dev())
    model.train(), ref.train(), refb.train()
    if mode == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
            yb = refb.forward_plumbing(x)
        (yb.float() * g).sum().backward()
    else:
        y = model(x)
    (y.float() * g).sum().backward()
    yr = ref.forward_plumbing(x)
    (yr * g).sum().backward()
    assert rel(y.float(), yr) < (2e-4 if mode == "fp32" else 3e-2)
    names = ["encoder.conv1.weight", "encoder.layer1.0.conv1.weight", "encoder.layer2.0.conv1.weight", "encoder.layer2.0.downsample.0.weight",
             "encoder.layer3.1.conv2.weight", "encoder.layer4.0.conv1.weight", "encoder.layer4.0.bn1.weight", "decoder.blocks.0.0.0.weight",
             "decoder.blocks.4.1.0.weight", "segmentation_head.0.weight", "segmentation_head.0.bias"]
    pm, pr, pb = dict(model.named_parameters()), dict(ref.named_parameters()), dict(refb.named_parameters())
    for nme in names:
        assert pm[nme].grad is not None
        err = rel(pm[nme].grad, pr[nme].grad)
        bar = 3e-2 if mode == "fp32" else 1.25 * rel(pb[nme].grad, pr[nme].grad) + 1e-3
        assert err < bar, (nme, err, bar)
    bb = dict(refb.named_buffers())
    for (k, b1), (_k, b2) in zip(model.named_buffers(), ref.named_buffers()):
        if "running" in k:
            bar = 1e-4 if mode == "fp32" else 1.25 * rel(bb[k], b2) + 5e-3
            assert rel(b1, b2) < bar, (k, rel(b1, b2), bar)

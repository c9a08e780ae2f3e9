"""BASELINE configs[0]: the plain `unet` plumbing model "on CPU".  `Unet.forward_plumbing` runs the network on plain torch operators
(an explicit entry point, not a fallback: `forward` on CPU tensors still raises) -- checked here, on the CPU, against the REFERENCE's
golden vectors (tests/golden/model_unet.npz: the reference's Unet.forward, models/networks/unet/net.py:806-838, on the same state and
inputs), and the configuration's own size (256x256, batch 2) against the oracle."""
import pytest
import torch
import torch.nn.functional as F

from tests import cases, golden_io, synth


def _model():
    from vq_seg_amd.models.networks import make_model
    return make_model({"name": "unet", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5, "encoder_weights": "imagenet_swsl"}})


def test_unet_plumbing_forward_matches_the_reference_golden_vectors_on_the_cpu():
    from vq_seg_amd.loss import make_loss
    fx = golden_io.load("model_unet")
    model = _model()
    model.load_state_dict(synth.synth_state_dict(golden_io.layout("unet"), fx.meta["model_seed"]))
    x, gt, _ = cases.model_inputs(b=2, s=64, seed=6500)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(x)                                                 # the accelerated forward does not take CPU tensors
    cases.set_bn_momentum(model, 1.0)                            # BN calibration, as in make_golden.gen_unet
    model.train()
    with torch.no_grad():
        model.forward_plumbing(x)
    cases.set_bn_momentum(model, 0.1)
    model.eval()
    with torch.no_grad():
        y = model.forward_plumbing(x)
    assert isinstance(y, torch.Tensor) and y.shape == (2, 3, 64, 64)
    ref = fx["eval_logits"]
    assert (y - ref).abs().max() <= 1e-4 * ref.abs().max(), "eval logits"
    model.train()
    y = model.forward_plumbing(x)
    ref = fx["train_logits"]
    assert (y - ref).abs().max() <= 1e-4 * ref.abs().max(), "train logits"
    loss = make_loss("dice_loss", 3, ignore_index=255)(y, gt) + 0.5 * F.cross_entropy(y, gt, ignore_index=255)
    assert abs(float(loss) - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    loss.backward()
    named = dict(model.named_parameters())
    for key in [k[5:] for k in fx if k.startswith("grad/")]:
        a, b = golden_io.probe(named[key].grad).double(), fx["grad/" + key].double()
        assert ((a - b).norm() / (b.norm() + 1e-30)).item() <= 1e-3, key


def test_config_1_at_its_own_size_on_the_cpu_against_the_oracle():
    """config/CWFID_Unet.json: 256x256, batch 2, one step of Dice + 0.5 CE with Adam -- the plumbing the reference validates without a GPU"""
    from oracle import torch_ref
    model = _model()
    sd = synth.synth_state_dict(golden_io.layout("unet"), 4242)
    model.load_state_dict(sd)
    x = synth.uniform(1, (2, 3, 256, 256))
    gt = synth.blob_labels(2, 2, 256, cell=32)
    model.eval()
    with torch.no_grad():
        y = model.forward_plumbing(x)
        ref = torch_ref.unet_forward({k: v.clone() for k, v in sd.items()}, x, training=False)
    assert y.shape == (2, 3, 256, 256)
    assert (y - ref).abs().max() <= 1e-4 * ref.abs().max()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    before = model.segmentation_head[0].weight.detach().clone()
    out = model.forward_plumbing(x)
    loss = 0.5 * F.cross_entropy(out, gt, ignore_index=255)
    loss.backward()
    opt.step()
    assert torch.isfinite(loss) and not torch.equal(before, model.segmentation_head[0].weight.detach())

"""The drop-in boundary: with `<repo>/compat` in front of sys.path the reference trainer's own import block
binds to this repository's package.  Runs in a fresh interpreter whose sys.path holds nothing else of this repository, from
a foreign working directory -- exactly the situation of a maintainer who adds one `sys.path.insert` to the reference trainer."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# train_vqreptunet1x1v2.py:13-26 minus the out-of-scope lines (utils.logger: wandb; utils.visualize: cv2 / matplotlib;
# utils.processing), then :70-80 as the trainer does them
SCRIPT = textwrap.dedent('''
    import json, sys
    sys.path.insert(0, sys.argv[1])                 # the ONE line a maintainer adds: <repo>/compat
    import torch
    import torch.nn as nn

    import models
    from utils.ckpoints import save_ckpoints, load_ckpoints, save_tar
    from utils.load_config import get_config_from_json
    from utils.device import device_setting
    from utils.seg_tools import img_to_label
    from utils.lr_schedulers import WarmUpPolyLR, CosineAnnealingLR
    from utils.seed import seed_everything

    from data.dataset import BaseDataset

    from loss import make_loss
    from measurement import Measurement
    from vector_quantizer import make_vq_module      # models/networks/modified_vqunet/net.py of the reference imports this

    cfg = get_config_from_json(sys.argv[2])
    seed_everything()
    device = device_setting(-1)
    measurement = Measurement(cfg.num_classes)
    model_1 = models.networks.make_model(cfg.model).to(device)
    model_2 = models.networks.make_model(cfg.model).to(device)
    if cfg.train.init_weights:
        models.init_weight([model_1.decoder, model_1.segmentation_head], nn.init.kaiming_normal_,
                           nn.BatchNorm2d, cfg.train.bn_eps, cfg.train.bn_momentum, mode='fan_in', nonlinearity='relu')
        models.init_weight([model_2.decoder, model_2.segmentation_head], nn.init.kaiming_normal_,
                           nn.BatchNorm2d, cfg.train.bn_eps, cfg.train.bn_momentum, mode='fan_in', nonlinearity='relu')
    loss_weight = cfg.train.criterion.get("weight", None)
    ce_loss = nn.CrossEntropyLoss(weight=loss_weight, ignore_index=255)
    dice_loss = make_loss(cfg.train.criterion.name, cfg.num_classes, weight=loss_weight, ignore_index=255)
    sched = CosineAnnealingLR(start_lr=cfg.train.learning_rate, min_lr=cfg.train.lr_scheduler.min_lr, total_iters=100,
                              warmup_steps=cfg.train.lr_scheduler.warmup_steps)
    opt = torch.optim.Adam(model_1.parameters(), lr=cfg.train.learning_rate, betas=(0.9, 0.999))
    import vq_seg_amd
    print(json.dumps({
        "models": models.__name__, "same": models.networks is vq_seg_amd.models.networks,
        "cls": type(model_1).__name__, "keys": len(model_1.state_dict()),
        "n_params": sum(p.numel() for p in model_1.parameters()),
        "bn_mom": model_1.decoder.blocks[0][0][1].momentum, "lr0": sched.get_lr(0),
        "label": img_to_label(torch.tensor([0, 128, 255]), cfg.pixel_to_label).tolist(),
        "dice": type(dice_loss).__name__, "vq": type(model_1.codebook[2]).__module__, "dataset": BaseDataset.__module__,
        "paths": [p for p in sys.path if "repo" in p]}))
''')


def test_reference_trainer_import_block_binds(tmp_path):
    cfg = {"num_classes": 3, "pixel_to_label": {"0": 0, "128": 1, "255": 2}, "resize": 448,        # config/vqreptunet1x1v2.json
           "model": {"name": "vqreptunet1x1v2", "params": {
               "encoder_name": "resnet50", "num_classes": 3, "depth": 5,
               "vq_cfg": {"num_embeddings": [0, 0, 512, 512, 512], "distance": "euclidean", "kmeans_init": True},
               "margin": 0.5, "scale": 30.0, "use_feature": False, "encoder_weights": None}},    # the URL fetch cannot work offline
           "train": {"learning_rate": 1e-4, "lr_scheduler": {"name": "cosineannealing", "min_lr": 1e-7, "warmup_steps": 0},
                     "half": True, "init_weights": True, "bn_eps": 1e-5, "bn_momentum": 0.1, "criterion": {"name": "dice_loss"}}}
    cfg_path = tmp_path / "vqreptunet1x1v2.json"
    cfg_path.write_text(json.dumps(cfg))
    script = tmp_path / "trainer_head.py"
    script.write_text(SCRIPT)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    res = subprocess.run([sys.executable, str(script), os.path.join(ROOT, "compat"), str(cfg_path)], cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["models"] == "vq_seg_amd.models" and out["same"] is True
    assert out["cls"] == "VQRePTUnet1x1v2" and out["keys"] == 383 and 69.2e6 < out["n_params"] < 69.3e6
    assert out["bn_mom"] == 0.1 and abs(out["lr0"] - 1e-4) < 1e-18
    assert out["label"] == [0, 1, 2] and out["dice"] == "DiceLoss" and out["vq"].startswith("vq_seg_amd.vector_quantizer")
    assert out["dataset"] == "vq_seg_amd.data.dataset"


def test_checkpoint_reference_layout_round_trip(tmp_path):
    """utils/ckpoints.py:7-21 of the reference: a {model_1, model_2, epoch, batch_idx, optimizer_1, optimizer_2} dictionary
    written by plain torch.save (what the reference's save_ckpoints does) loads through this repository's load_ckpoints and
    a strict load_state_dict; the reference's return arities are kept (istrain -> 5 values, model_2 first: q17)."""
    import torch
    from vq_seg_amd.models.networks import make_model
    from vq_seg_amd.utils.ckpoints import load_ckpoints, load_training_state, restore_initted, save_ckpoints
    from tests import golden_io, synth
    cfg = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                               "vq_cfg": {"num_embeddings": [0, 0, 32, 32, 32], "distance": "euclidean", "kmeans_init": True},
                                               "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    layout = dict(golden_io.layout("vqreptunet1x1"))
    for i in (2, 3, 4):
        layout[f"codebook.{i}.codebook.embedding.weight"] = (32, layout[f"codebook.{i}.codebook.embedding.weight"][1])
    sd1, sd2 = synth.synth_state_dict(layout, 5), synth.synth_state_dict(layout, 6)
    m = make_model(cfg)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    ref_file = tmp_path / "last.pth"
    torch.save({"model_1": sd1, "model_2": sd2, "epoch": 7, "batch_idx": 3, "optimizer_1": opt.state_dict(),
                "optimizer_2": opt.state_dict()}, ref_file)                                   # the reference's writer, verbatim layout
    got = load_ckpoints(str(ref_file), istrain=True)
    assert len(got) == 5 and got[1] == 7 and got[2] == 3                                     # reference arity
    assert all(torch.equal(got[0][k], sd2[k]) for k in sd2)                                   # ... and its quirk: model_2's weights
    w = load_ckpoints(str(ref_file), istrain=False)
    missing = m.load_state_dict(w, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert all(torch.equal(m.state_dict()[k], sd1[k]) for k in sd1)
    assert list(m.state_dict()) == list(sd1)                                                  # same key order as the reference layout
    # a bare state_dict file (test_detailviz.py:90 `weights.get('model_1', weights)`)
    bare = tmp_path / "bare.pth"
    torch.save(sd2, bare)
    assert all(torch.equal(load_ckpoints(str(bare), istrain=False)[k], sd2[k]) for k in sd2)
    # this repository's writer: same layout + the `initted` flags the reference loses (q7)
    m.codebook[2].codebook.initted = True
    m.prototype_loss.initted = True
    ours = tmp_path / "ours.pth"
    save_ckpoints(m.state_dict(), m.state_dict(), 1, 2, opt.state_dict(), opt.state_dict(), str(ours), models=[m, m])
    state = load_training_state(str(ours))
    assert set(state) == {"model_1", "model_2", "epoch", "batch_idx", "optimizer_1", "optimizer_2", "initted"}
    assert len(load_ckpoints(str(ours), istrain=True)) == 5
    m2 = make_model(cfg)
    assert m2.codebook[2].codebook.initted is False
    m2.load_state_dict(state["model_1"], strict=True)
    restore_initted(m2, state["initted"][0])
    assert m2.codebook[2].codebook.initted is True and m2.codebook[3].codebook.initted is False and m2.prototype_loss.initted is True
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-4)
    opt2.load_state_dict(state["optimizer_1"])


def test_init_weight_contract():
    """models/__init__.py:7-26: init_func on every conv weight of the listed modules (with the keyword arguments passed through),
    every `norm_layer` instance gets eps / momentum / weight 1 / bias 0; nothing outside the list is touched."""
    import torch
    from torch import nn
    from vq_seg_amd.models import init_weight
    from vq_seg_amd.models.networks import make_model
    cfg = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                               "vq_cfg": {"num_embeddings": [0, 0, 16, 16, 16], "distance": "euclidean", "kmeans_init": True},
                                               "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    torch.manual_seed(3)
    m = make_model(cfg)
    enc_before = {k: v.clone() for k, v in m.encoder.state_dict().items()}
    for mod in m.decoder.modules():
        if isinstance(mod, nn.BatchNorm2d):
            nn.init.uniform_(mod.weight, 2, 3), nn.init.uniform_(mod.bias, 2, 3)
    calls = []

    def init_func(w, **kw):
        calls.append((tuple(w.shape), kw))
        nn.init.kaiming_normal_(w, **kw)

    init_weight([m.decoder, m.segmentation_head], init_func, nn.BatchNorm2d, 1e-3, 0.05, mode="fan_in", nonlinearity="relu")
    convs = [mod for part in (m.decoder, m.segmentation_head) for mod in part.modules() if isinstance(mod, nn.Conv2d)]
    assert len(convs) == 11 and len(calls) == 11                                 # 10 decoder convs + the 1x1 head
    assert all(kw == {"mode": "fan_in", "nonlinearity": "relu"} for _, kw in calls)
    for c in convs:                                                              # kaiming_normal_(fan_in, relu): std = sqrt(2 / fan_in)
        fan_in = c.weight[0].numel()
        if c.weight.numel() > 4096:
            assert abs(c.weight.std().item() / (2.0 / fan_in) ** 0.5 - 1) < 0.1
    bns = [mod for mod in m.decoder.modules() if isinstance(mod, nn.BatchNorm2d)]
    assert len(bns) == 10
    for b in bns:
        assert b.eps == 1e-3 and b.momentum == 0.05 and bool((b.weight == 1).all()) and bool((b.bias == 0).all())
    assert all(torch.equal(v, enc_before[k]) for k, v in m.encoder.state_dict().items())     # the encoder is not in the list
    assert m.encoder.bn1.eps == 1e-5 and m.encoder.bn1.momentum == 0.1
    init_weight(m.segmentation_head, init_func, nn.BatchNorm2d, 1e-3, 0.05, mode="fan_in", nonlinearity="relu")   # a bare module too
    assert len(calls) == 12

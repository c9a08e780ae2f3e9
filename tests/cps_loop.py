"""ONE restatement of the reference trainers' loop bodies, written against the reference's own top-level surface
(`models.networks.make_model`, `loss.make_loss`, `measurement.Measurement`, `utils.lr_schedulers.CosineAnnealingLR`) and
nothing else -- TEST INFRASTRUCTURE.

    v2: train_vqreptunet1x1v2.py:137-211            (score-mask CPS, 0.5 CE + Dice, model(x, gt, th=...))
    v1: deprecated/train_with_test_pt_pseudo_entropy_reg.py:141-203   (entropy-percentile pseudo labels, criterion from
        the config, model(x, gt, percent=...))

The trainer FILES cannot be imported (module-level wandb / cv2 / matplotlib, image folders on disk), so their
loop bodies are restated here once and driven from two sides with the same code:
  * oracle/make_golden.py hands in the REFERENCE's modules (imported through oracle/ref_harness.py, CPU) and stores what the
    iterations produce as tests/golden/cps_iter_v{1,2}.npz / cps_curve_v1.npz;
  * tests/test_compat_gpu.py hands in the SAME names resolved through `<repo>/compat` (-> vq_seg_amd, HIP kernels, cuda:0)
    and compares with those fixtures -- the proof that the reference's trainer drives this repository's path.
Only torch / numpy are imported here; every model / loss / metric / schedule object comes in through `ns`.
"""
from __future__ import annotations

import contextlib
from types import SimpleNamespace
from typing import Dict, List

import numpy as np
import torch
from torch import nn

from tests import cases, golden_io, synth

SIZE, BATCH = 64, 2                      # the authors' own debug size (deprecated/train_with_test_pt_pseudo_entropy_reg.py:303-308)
SEEDS = (77, 78)                         # state_dict seeds of model_1 / model_2
TRAIN = dict(learning_rate=1e-4, min_lr=1e-7, cps_loss_weight=1, total_commitment_loss_weight=1,
             total_prototype_loss_weight=0.01, unsup_loss_drop_percent=20, confidence_threshold=0.7,
             criterion="dice_loss")       # config/vqreptunet1x1{,v2}.json "train" section
PROBES = ["segmentation_head.weight", "encoder.conv1.weight", "decoder.blocks.4.1.0.weight", "decoder.blocks.0.0.0.weight",
          "encoder.layer4.2.conv3.weight", "encoder.layer2.0.downsample.0.weight", "encoder.layer1.0.bn1.weight",
          "decoder.blocks.2.0.1.bias"]


def model_cfg(version: int, k=(0, 0, 512, 512, 512)) -> dict:
    """config/vqreptunet1x1.json / vqreptunet1x1v2.json "model" section; encoder_weights None (the URL fetch cannot work)."""
    name, margin, scale = ("vqreptunet1x1", 0.0, 1.0) if version == 1 else ("vqreptunet1x1v2", 0.5, 30.0)
    return {"name": name, "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                     "vq_cfg": {"num_embeddings": list(k), "distance": "euclidean", "kmeans_init": True},
                                     "margin": margin, "scale": scale, "use_feature": False, "encoder_weights": None}}


def batches(n_iters: int, size: int = SIZE, batch: int = BATCH, seed: int = 8100):
    """[(l_input, l_target, ul_input)] * n_iters, deterministic on any machine."""
    out = []
    for i in range(n_iters):
        l_in, l_tg = synth.blob_images(seed + 10 * i, batch, size, cell=8)
        ul_in, _ = synth.blob_images(seed + 10 * i + 5, batch, size, cell=8)
        out.append((l_in, l_tg, ul_in))
    return out


def build_pair(ns, version: int, device, to_cfg=lambda d: d, prepare=None, k=(0, 0, 512, 512, 512), size: int = SIZE,
               batch: int = BATCH):
    """model_1 / model_2 as the trainers build them (train_vqreptunet1x1v2.py:70-71) with synthetic weights instead of the RNG
    dependent ones: state_dict seeds 77 / 78, prototypes installed, BatchNorm statistics and codebooks from the shared
    preparation recipe (tests/cases.py::prepare_module_model == oracle/make_golden.py::prepare_model) on the first batch."""
    prepare = prepare or cases.prepare_module_model
    l_in, l_tg, _ = batches(1, size, batch)[0]
    pair = []
    for seed in SEEDS:
        model = ns.models.networks.make_model(to_cfg(model_cfg(version, k)))
        layout = dict(golden_io.layout("vqreptunet1x1"))                  # insertion order = the reference's key order
        for i, ki in enumerate(k):
            if ki:
                layout[f"codebook.{i}.codebook.embedding.weight"] = (ki, layout[f"codebook.{i}.codebook.embedding.weight"][1])
        model.load_state_dict(synth.synth_state_dict(layout, seed))
        model.prototype_loss.initted = True
        model = model.to(device)
        prepare(model, l_in.to(device), l_tg.to(device), version)
        pair.append(model)
    return pair


def regularized_pseudo_label(raw: torch.Tensor, percent: float) -> torch.Tensor:
    """make_regularized_pseudo_label, deprecated/train_with_test_pt_pseudo_entropy_reg.py:30-39 (numpy percentile on the host)."""
    prob = torch.softmax(raw, dim=1)
    label = torch.argmax(prob, dim=1)
    entropy = -torch.sum(prob * torch.log(prob + 1e-10), dim=1)
    thresh = np.percentile(entropy.detach().cpu().numpy().flatten(), percent)
    label[entropy.ge(thresh).bool()] = 255
    return label


def score_mask(pred: torch.Tensor, pseudo: torch.Tensor, th: float = 0.7) -> torch.Tensor:
    """train_vqreptunet1x1v2.py:43-46."""
    top = torch.softmax(pred, dim=1).max(dim=1)[0]
    return torch.where(top > th, pseudo, 255)


class Loop:
    """State of a training run: two models, two Adam optimisers, schedule, criteria (train_vqreptunet1x1v2.py:70-114)."""

    def __init__(self, ns, version: int, model_1, model_2, total_iters: int, half: bool = False, train: Dict = None,
                 amp_dtype=torch.float16, with_optim: bool = True, scaler: bool = False):
        """`scaler=True`: the trainer's LITERAL mixed-precision region (train_vqreptunet1x1v2.py:114,151,172,199-202):
        `torch.cuda.amp.autocast(enabled=half)` WITHOUT a dtype argument (CUDA/HIP default: float16) and
        `torch.cuda.amp.GradScaler(enabled=half)` with scale(loss).backward() / step(opt1) / step(opt2) / update()."""
        self.ns, self.version = ns, version
        self.literal_amp = bool(scaler)
        self.scaler = torch.cuda.amp.GradScaler(enabled=half) if scaler else None                       # :114
        self.m1, self.m2 = model_1, model_2
        self.t = SimpleNamespace(**dict(TRAIN, **(train or {})))
        self.half, self.amp_dtype = half, amp_dtype
        self.sched = ns.CosineAnnealingLR(start_lr=self.t.learning_rate, min_lr=self.t.min_lr, total_iters=total_iters, warmup_steps=0)
        self.opt1 = torch.optim.Adam(model_1.parameters(), lr=self.t.learning_rate, betas=(0.9, 0.999)) if with_optim else None
        self.opt2 = torch.optim.Adam(model_2.parameters(), lr=self.t.learning_rate, betas=(0.9, 0.999)) if with_optim else None
        self.ce = nn.CrossEntropyLoss(ignore_index=255)
        self.criterion = ns.make_loss(self.t.criterion, 3, weight=None, ignore_index=255)
        self.measurement = ns.Measurement(3)
        self.it = 0

    def _autocast(self):
        if self.literal_amp:
            return torch.cuda.amp.autocast(enabled=self.half)                                             # :151, :172
        if not self.half:
            return contextlib.nullcontext()
        dev = next(self.m1.parameters()).device.type
        return torch.autocast(dev, dtype=self.amp_dtype)

    def iteration(self, l_input, l_target, ul_input, epoch_frac: float = 0.0, backward: bool = True) -> Dict[str, object]:
        m1, m2, t = self.m1, self.m2, self.t
        m1.train(), m2.train()
        if self.opt1 is not None:
            self.opt1.zero_grad(), self.opt2.zero_grad()
        with torch.no_grad():                                           # pseudo labels from eval-mode passes, OUTSIDE autocast
            m1.eval(), m2.eval()
            score_1, score_2 = m1(ul_input)[0], m2(ul_input)[0]
            m1.train(), m2.train()
        if self.version == 1:
            percent = 100 - t.unsup_loss_drop_percent * (1 - epoch_frac)
            kw, gt_ul_1, gt_ul_2 = dict(percent=percent), torch.argmax(score_2, dim=1), torch.argmax(score_1, dim=1)
        else:
            kw, gt_ul_1, gt_ul_2 = dict(th=t.confidence_threshold), score_2, score_1
        with self._autocast():
            ps1, c_l1, u_l1, p_l1 = m1(l_input, l_target, **kw)
            ps2, c_l2, u_l2, p_l2 = m2(l_input, l_target, **kw)
            pu1, c_u1, u_u1, p_u1 = m1(ul_input, gt_ul_1, **kw)
            pu2, c_u2, u_u2, p_u2 = m2(ul_input, gt_ul_2, **kw)
        pred_1, pred_2 = torch.cat([ps1, pu1], dim=0), torch.cat([ps2, pu2], dim=0)
        if self.version == 1:
            mask_1, mask_2 = regularized_pseudo_label(pred_1, percent), regularized_pseudo_label(pred_2, percent)
        else:
            pl1, pl2 = torch.argmax(pred_1, dim=1).long(), torch.argmax(pred_2, dim=1).long()
        with self._autocast():
            if self.version == 1:
                cps = self.criterion(pred_1, mask_2) + self.criterion(pred_2, mask_1)
                sup_1, sup_2 = self.criterion(ps1, l_target), self.criterion(ps2, l_target)
            else:
                mask_1 = score_mask(pred_1, pl1, th=t.confidence_threshold)
                mask_2 = score_mask(pred_2, pl2, th=t.confidence_threshold)
                cps = 0.5 * self.ce(pred_1, mask_2) + 0.5 * self.ce(pred_2, mask_1) + self.criterion(pred_1, mask_2) + \
                    self.criterion(pred_2, mask_1)
                sup_1 = 0.5 * self.ce(ps1, l_target) + self.criterion(ps1, l_target)
                sup_2 = 0.5 * self.ce(ps2, l_target) + self.criterion(ps2, l_target)
            commitment = (c_l1 + c_l2 + c_u1 + c_u2) * t.total_commitment_loss_weight
            prototype = (p_l1 + p_l2 + p_u1 + p_u2) * t.total_prototype_loss_weight
            lr = self.sched.get_lr(self.it)
            if self.opt1 is not None:
                self.opt1.param_groups[0]["lr"] = lr
                self.opt2.param_groups[0]["lr"] = lr
            loss = sup_1 + sup_2 + t.cps_loss_weight * cps + commitment + prototype
        usage = (u_l1 + u_l2 + u_u1 + u_u2) / 4
        grads = {}
        if backward:
            inv = 1.0
            if self.scaler is not None:
                grads["scale_before"] = float(self.scaler.get_scale())
                inv = 1.0 / grads["scale_before"]
                self.scaler.scale(loss).backward()                      # :199
            else:
                loss.backward()                                         # GradScaler(enabled=False) is the identity
            for tag, m in (("m1", m1), ("m2", m2)):
                named = dict(m.named_parameters())
                for key in PROBES:
                    grads[f"grad/{tag}/{key}"] = (golden_io.probe(named[key].grad).detach().float() * inv).cpu().clone()
            grads["grad_none/m1"] = sorted(k for k, p in m1.named_parameters() if p.grad is None)
            if self.scaler is not None:
                self.scaler.step(self.opt1)                             # :200-202
                self.scaler.step(self.opt2)
                self.scaler.update()
                grads["scale_after"] = float(self.scaler.get_scale())
            else:
                self.opt1.step(), self.opt2.step()
        conf = self.measurement._make_confusion_matrix(ps1.detach().float().cpu().numpy(), l_target.detach().cpu().numpy())
        miou, ious = self.measurement.miou(conf)
        self.it += 1
        out = dict(loss=loss, sup_loss_1=sup_1, sup_loss_2=sup_2, cps_loss=cps, commitment_loss=commitment, prototype_loss=prototype)
        out = {k: float(torch.as_tensor(v).detach().double().sum().cpu()) for k, v in out.items()}
        for k in ("scale_before", "scale_after"):
            if k in grads:
                out[k] = grads.pop(k)
        out.update(lr=float(lr), step_miou=float(miou), mask_1=mask_1.detach().cpu().to(torch.uint8), mask_2=mask_2.detach().cpu().to(torch.uint8),
                   score_1=score_1.detach().float().cpu(), pred_sup_1=ps1.detach().float().cpu(), pred_ul_2=pu2.detach().float().cpu(),
                   code_usage=torch.as_tensor(usage).detach().float().cpu(), **grads)
        return out

    def probes(self) -> Dict[str, torch.Tensor]:
        out = {}
        for tag, m in (("m1", self.m1), ("m2", self.m2)):
            sd = m.state_dict()
            for key in PROBES:
                out[f"param/{tag}/{key}"] = golden_io.probe(sd[key]).detach().float().cpu().clone()
            out[f"param/{tag}/encoder.bn1.running_var"] = sd["encoder.bn1.running_var"].detach().float().cpu().clone()
        return out

    def evaluate(self, images, labels) -> float:
        """test() of the trainers (train_vqreptunet1x1v2.py:28-41): mean over the images of the per-image mIoU of model_1."""
        total = 0.0
        self.m1.eval()
        with torch.no_grad():
            for i in range(images.shape[0]):
                pred = self.m1(images[i:i + 1])[0]
                conf = self.measurement._make_confusion_matrix(pred.detach().float().cpu().numpy(), labels[i:i + 1].cpu().numpy())
                total += float(self.measurement.miou(conf)[0])
        self.m1.train()
        return total / images.shape[0]


def run_iterations(ns, version: int, device, n_iters: int = 2, backward: bool = True, to_cfg=lambda d: d, prepare=None,
                   half: bool = False, amp_dtype=torch.float16, scaler: bool = False) -> List[Dict[str, object]]:
    """fixture (9) of SURVEY 8c: `n_iters` CPS iterations from the prepared pair; per-iteration dictionaries, the last one
    also carries the parameter probes after the final optimiser step."""
    m1, m2 = build_pair(ns, version, device, to_cfg, prepare)
    loop = Loop(ns, version, m1, m2, total_iters=1000, half=half, amp_dtype=amp_dtype, with_optim=backward, scaler=scaler)
    outs = []
    for l_in, l_tg, ul_in in batches(n_iters):
        outs.append(loop.iteration(l_in.to(device), l_tg.to(device), ul_in.to(device), backward=backward))
    outs[-1].update(loop.probes())
    return outs


CURVE = dict(size=64, batch=4, steps=40, eval_every=10, eval_images=16, learning_rate=1e-3)
# r4: the thicker mIoU-parity run -- 128x128, the shipped codebook size K = 512 at all three levels, 200 steps
CURVE128 = dict(size=128, batch=4, steps=200, eval_every=25, eval_images=16, learning_rate=1e-3)
K128 = (0, 0, 512, 512, 512)


def run_curve(ns, device, to_cfg=lambda d: d, prepare=None, half: bool = False, amp_dtype=torch.float16, k=(0, 0, 64, 64, 64),
              spec: Dict = None, scaler: bool = False) -> Dict[str, object]:
    """The mIoU-parity run (north_star: "mIoU within +-0.2"): `steps` v1 iterations on the synthetic crop/weed blobs, test-set
    mIoU of model_1 (the trainers' test()) every `eval_every` steps."""
    s = SimpleNamespace(**dict(CURVE, **(spec or {})))
    m1, m2 = build_pair(ns, 1, device, to_cfg, prepare, k=k, size=s.size, batch=s.batch)
    loop = Loop(ns, 1, m1, m2, total_iters=s.steps, half=half, amp_dtype=amp_dtype, train=dict(learning_rate=s.learning_rate), scaler=scaler)
    test_img, test_lab = synth.blob_images(9900, s.eval_images, s.size, cell=8)
    test_img, test_lab = test_img.to(device), test_lab.to(device)
    mious, losses, step_mious, scales = [loop.evaluate(test_img, test_lab)], [], [], []
    data = batches(s.steps, s.size, s.batch, seed=8500)
    for i, (l_in, l_tg, ul_in) in enumerate(data):
        out = loop.iteration(l_in.to(device), l_tg.to(device), ul_in.to(device), epoch_frac=0.0)
        losses.append(out["sup_loss_1"])
        step_mious.append(out["step_miou"])
        scales.append(out.get("scale_after", 1.0))
        if (i + 1) % s.eval_every == 0:
            mious.append(loop.evaluate(test_img, test_lab))
    return dict(test_miou=np.array(mious), sup_loss_1=np.array(losses), step_miou=np.array(step_mious), scale=np.array(scales))

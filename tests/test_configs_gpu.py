"""BASELINE.json configurations 2 and 4 at model level (3 is the bench -- here once more at batch 8 for the near-tie report; 1 is
tests/test_model_gpu.py::test_plain_unet...):
  #2  vqreptunet1x1   512 x 512, K = 256,  bf16 activations (autocast)
  #4  vqreptunet1x1v2 1024 x 1024, K = 1024, bf16 activations
Two CPSTrainer steps each (finite terms, codebooks initialised by k-means in the first training forward), then the
size-independent VQ properties of tests/test_vq_gpu.py::test_full_size_properties on the model's OWN bf16 encoder features
and k-means codebooks.  The near-tie census runs on LIVE codebooks against the CPU oracle
(test_indices_on_own_features_with_live_codebooks_match_the_oracle): calibrated network state, k-means codebooks with no dead code.
Also: one decoder block at the benchmark's own channel / pixel counts against the reference's double_conv_block."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests import cases, golden_io, synth

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("recipe,size,k,batch", [("v1", 512, 256, 2), ("v2", 1024, 1024, 1), ("v1", 512, 512, 8)])
def test_baseline_config_steps_and_vq_properties_on_own_features(recipe, size, k, batch):
    from vq_seg_amd import _hip
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
    name = "vqreptunet1x1" if recipe == "v1" else "vqreptunet1x1v2"
    margin, scale = (0.0, 1.0) if recipe == "v1" else (0.5, 30.0)
    model = {"name": name, "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                      "vq_cfg": {"num_embeddings": [0, 0, k, k, k], "distance": "euclidean", "kmeans_init": True},
                                      "margin": margin, "scale": scale, "use_feature": False, "encoder_weights": None}}
    torch.manual_seed(0)
    tr = CPSTrainer(CPSConfig(model=model, recipe=recipe, total_iters=10, amp_dtype=torch.bfloat16), dev())
    data = SyntheticCropWeed(size, batch, dev(), seed=3)
    (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
    for _ in range(2):
        out = tr.step(l_in, l_tg, ul_in)
        for key, v in out.items():
            assert torch.isfinite(torch.as_tensor(v)).all(), key
    m = tr.models[0]
    assert all(m.codebook[i].codebook.initted for i in (2, 3, 4))
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        feats = m.encode(ul_in.contiguous(memory_format=torch.channels_last))
    report = []
    for lvl in (2, 3, 4):
        f = feats[lvl]
        assert f.dtype == torch.bfloat16
        b, c, h, w = f.shape
        rows = f.permute(0, 2, 3, 1).reshape(-1, c).contiguous()                   # bf16 rows as the layer sees them
        W = m.codebook[lvl].codebook.embedding.weight.detach()
        n = rows.shape[0]
        quant, idx, _loss, dead, dmin = _hip.vq_forward(rows, W, False, 1.0, want_dmin=True)
        # (1) epsilon-argmin in fp64 on a row sample.  Two steps from random init leave k-means codebooks with > 99 % dead codes, so a
        # near-tie census here would be vacuous: the census -- against the CPU ORACLE, on live codebooks -- is
        # test_indices_on_own_features_with_live_codebooks_match_the_oracle below.
        sub = torch.arange(0, n, max(n // 4096, 1), device=dev())
        d2 = torch.cdist(rows[sub].double(), W.double()).pow(2)
        best = d2.min(dim=1).values
        chosen = d2.gather(1, idx[sub, None])[:, 0]
        assert ((chosen - best) <= 1e-5 * best.clamp_min(1e-6)).all(), f"level {lvl}: argmin is not a minimiser"
        report.append(f"{recipe} {size}^2 K={k} level {lvl}: N={n} C={c}, dead codes {float(dead):.1f} %")
        # (2) exact gather (bf16 rows out: the fp32 code rounded once)
        assert torch.equal(quant, W[idx].to(torch.bfloat16))
        # (3) histogram
        cnt = torch.bincount(idx, minlength=k)
        assert float(dead) == float(100 * ((cnt == 0).sum() / k))
        # (4) the fp32-rows entry point sees the same bf16 values -> the same indices; permutation equivariance
        assert torch.equal(_hip.vq_assign(rows.float(), W), idx)
        perm = torch.randperm(n, device=dev())
        assert torch.equal(_hip.vq_assign(rows[perm].contiguous(), W), idx[perm])
    print("\n".join(report))


@pytest.mark.parametrize("size,k", [(512, 256), (512, 512), (1024, 1024)])
def test_indices_on_own_features_with_live_codebooks_match_the_oracle(size, k):
    """The near-tie census on REAL structure, against the CPU oracle (oracle/torch_ref.py::vq_forward, pinned to the reference's
    EuclideanCodebook.forward by tests/test_oracle_golden.py incl. vq_big.npz) instead of GPU-vs-GPU: the calibrated network state of
    the model fixtures at the BASELINE configurations' sizes, codebooks = the package's 10-iteration k-means on the network's own
    fp32 eval features (every code alive).  Per level: the HIP kernel's indices for the fp32 rows and for the bf16-rounded rows (what
    the autocast path hands over) must EQUAL the oracle's on the same rows, except near-ties below fp32 resolution (fp64 gap < 1e-5
    relative: the CPU's blocked sgemm and the MFMA chain round such a pair differently), which are counted and reported."""
    from oracle import torch_ref as R
    from tests.test_model_gpu import build
    from vq_seg_amd import _hip
    from vq_seg_amd.vector_quantizer.vq_img import kmeans
    m = build("vqreptunet1x1", 0.0, 1.0, 77, size=size)
    x = cases.model_inputs(s=size)[0].to(dev()).contiguous(memory_format=torch.channels_last)
    m.eval()
    torch.manual_seed(5)
    lines = []
    with torch.no_grad():
        feats = m.encoder(x)[1:]
    for lvl in (2, 3, 4):
        f = feats[lvl].float()
        rows32 = f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]).contiguous()
        W, bins = kmeans(rows32, min(k, rows32.shape[0] // 2), 10)
        W = W.contiguous()
        kk = W.shape[0]
        for tag, rows in (("fp32", rows32), ("bf16", rows32.bfloat16())):
            quant, idx, _l, dead = _hip.vq_forward(rows, W, False, 1.0)[:4]
            rc = rows.float().cpu()
            n, c = rc.shape
            with torch.no_grad():
                _q, ref_idx, _loss, ref_dead = R.vq_forward(rc.reshape(1, n, 1, c).permute(0, 3, 1, 2), W.cpu(), training=False)
            ref_idx = ref_idx.reshape(-1)
            d2 = torch.cdist(rows.double(), W.double()).pow(2)
            top2 = torch.topk(d2, 2, dim=1, largest=False).values
            gap = (top2[:, 1] - top2[:, 0]) / top2[:, 1].clamp_min(1e-30)
            wrong, wgap, wexc = cases.near_tie_audit(rows.float(), W, idx, ref_idx.to(idx.device))
            lines.append(f"{size}^2 K={kk} level {lvl} {tag}: N={n} C={c}, dead codes {float(dead):.1f} % (oracle {float(ref_dead):.1f} %), rows with "
                         f"top-2 gap < 1e-5: {int((gap < 1e-5).sum())}, < 1e-6: {int((gap < 1e-6).sum())}, min gap {gap.min().item():.2e}; "
                         f"index != oracle in {wrong}" + (f" (their fp64 gap <= {wgap:.1e})" if wrong else ""))
            # only near-ties below fp32 resolution may differ (cases.near_tie_audit)
            assert wrong <= int((gap < 1e-5).sum()) and wgap < 1e-5 and wexc < 1e-5, lines[-1]
            assert float(dead) == float(ref_dead)
    print("\n".join(lines))


def _block(sd):
    from vq_seg_amd.models.networks.unet.decoder import double_conv_block
    blk = double_conv_block(2048, 1024)
    blk.load_state_dict(sd)
    return blk.to(dev())


def _probe_close(t, want, want_stats, tol, what, bad):
    got = golden_io.probe(t).double().cpu()
    want = want.double()
    scale = float(want_stats[2])                                     # max |.| of the WHOLE reference tensor
    err = (got - want).abs().max().item() / scale
    l2 = ((got - want).norm() / (want.norm() + 1e-30)).item()
    s2 = t.detach().double().pow(2).sum().item()
    ds2 = abs(s2 - float(want_stats[1])) / float(want_stats[1])
    line = f"{what}: probe max err {err:.2e} of scale, rel L2 {l2:.2e}, sum of squares off by {ds2:.2e} (tol {tol:g})"
    print(line)
    if err > tol or l2 > tol or ds2 > 4 * tol:
        bad.append(line)


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_decoder_block_at_bench_scale_matches_reference(mode):
    """double_conv_block(2048, 1024) on 32 x 16 x 16 pixels (models/networks/unet/decoder.py:7-12; fixture decoder_block0_b32 from
    oracle/make_golden.py::gen_decoder_block): the natural dispatch at this size is what bench.py runs there -- bf16: multi-chunk
    patch-reuse forward + zero-pad data gradient (conv3x3_patch_kernel<128,...,64>, 32 / 16 channel chunks), nine-tap weight gradient
    (conv_wgrad3x3_kernel), BN statistics in the epilogue; fp32: the bf16x3 precise kernels.  Strided probes + fp64 checksums."""
    from tests.cases import block_inputs
    fx = golden_io.load("decoder_block0_b32")
    x, sd, g = block_inputs(fx.meta)
    assert synth.checksum(x) == fx.meta["x_sum"]
    amp = mode == "bf16"
    # Gradients: the cotangent here is white noise, so every gradient entry is a random-sign sum over 8192 pixels, and ONE ReLU whose
    # pre-activation lies within rounding error of zero (a few dozen of the 8.4 M do) moves the affected entries by ~1 % of scale
    # whichever side is "right" (tools/block_diag.py: the same size of error with eval-mode BatchNorm and in grad_beta, a plain
    # masked sum; the CPU reference itself is within 1e-6 of float64).  Forward values carry the tight bar.
    tol_f, tol_g = (2e-2, 0.1) if amp else (1e-4, 2e-2)      # bf16 gradients measured: 6-7 % rel L2 on white noise
    xg = x.to(dev()).contiguous(memory_format=torch.channels_last)
    if amp:
        from vq_seg_amd import nnf
        xg = nnf.cast_act(xg, torch.bfloat16)                          # what the decoder does to its inputs under autocast
    blk = _block(sd)
    blk.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = blk(xg)
    bad = []
    _probe_close(y.float(), fx["y_eval"], fx["y_eval_stats"], tol_f, "eval output", bad)
    blk.train()
    xr = xg.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        y = blk(xr)
    _probe_close(y.float(), fx["y_train"], fx["y_train_stats"], tol_f, "train output", bad)
    (y.float() * g.to(dev())).sum().backward()
    assert xr.grad.dtype == (torch.bfloat16 if amp else torch.float32) and y.dtype == xr.grad.dtype
    _probe_close(xr.grad.float(), fx["grad_x"], fx["grad_x_stats"], tol_g, "grad x", bad)
    _probe_close(blk[0][0].weight.grad, fx["grad_w0"], fx["grad_w0_stats"], tol_g, "grad w0", bad)
    _probe_close(blk[1][0].weight.grad, fx["grad_w1"], fx["grad_w1_stats"], tol_g, "grad w1", bad)
    assert not bad, bad
    for name, got in (("grad_bn_w1", blk[1][1].weight.grad), ("grad_bn_b1", blk[1][1].bias.grad), ("grad_bn_w0", blk[0][1].weight.grad)):
        want = fx[name].double()
        assert ((got.double().cpu() - want).norm() / want.norm()).item() <= tol_g, name
    post = blk.state_dict()
    for name, key in (("run_mean0", "0.1.running_mean"), ("run_var0", "0.1.running_var"), ("run_var1", "1.1.running_var")):
        assert torch.allclose(post[key].cpu(), fx[name], rtol=2e-2 if amp else 1e-4, atol=1e-3 if amp else 1e-6), name


def test_shipped_config_size_448_matches_oracle_and_trains():
    """The reference's SHIPPED configs resize to 448 x 448 at batch 4 (config/vqreptunet1x1.json:10,34; 512 / 64 are BASELINE's
    choices): feature maps of 224 / 112 / 56 / 28 / 14 pixels, so the deep levels are NOT multiples of the 3x3 kernels' 16- and
    32-pixel tiles and take the generic implicit-GEMM kernels.  (1) eval forward in fp32 against the CPU oracle on the same state
    (logits 1e-3 of scale -- north_star --, code indices of the three levels equal, dead-code percentages equal);
    (2) two bf16 CPSTrainer steps at the shipped batch size, every term finite."""
    from oracle import torch_ref as R
    from tests.test_model_gpu import build, rel_close
    from vq_seg_amd.trainer import CPSConfig, CPSTrainer, SyntheticCropWeed
    size = 448
    model = build("vqreptunet1x1", 0.0, 1.0, 77, size=size)           # synthetic state + the calibration recipe of the fixtures
    x, _gt, _ = cases.model_inputs(s=size)
    model.eval()
    with torch.no_grad():
        logits, closs, usage, proto = model(x.to(dev()))
        feats = model.encoder(x.to(dev()).contiguous(memory_format=torch.channels_last))[1:]
        idx = [model.codebook[lvl](feats[lvl])[1].cpu() for lvl in (2, 3, 4)]
    assert [tuple(f.shape[-2:]) for f in feats] == [(224, 224), (112, 112), (56, 56), (28, 28), (14, 14)]
    sd = {k: v.detach().float().cpu() if v.is_floating_point() else v.cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref_logits, _closs, ref_usage, _proto, aux = R.vq_unet_forward(sd, x, False, (0, 0, 512, 512, 512), version=1)
    rel_close(logits, ref_logits, 1e-3, "eval logits at 448^2")
    for got, want, lvl in zip(idx, aux["indices"], (2, 3, 4)):
        assert torch.equal(got, want), f"level {lvl}: code indices differ from the oracle's"
    assert torch.allclose(usage.double(), torch.stack([u.double() for u in ref_usage]).cpu(), rtol=1e-6)

    cfg = {"name": "vqreptunet1x1", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5,
                                               "vq_cfg": {"num_embeddings": [0, 0, 512, 512, 512], "distance": "euclidean", "kmeans_init": True},
                                               "margin": 0.0, "scale": 1.0, "use_feature": False, "encoder_weights": None}}
    torch.manual_seed(0)
    tr = CPSTrainer(CPSConfig(model=cfg, recipe="v1", total_iters=10, amp_dtype=torch.bfloat16), dev())
    data = SyntheticCropWeed(size, 4, dev(), seed=11)
    (l_in, l_tg), ul_in = data.labelled(), data.unlabelled()
    for _ in range(2):
        out = tr.step(l_in, l_tg, ul_in)
        for key, v in out.items():
            assert torch.isfinite(torch.as_tensor(v)).all(), key


@pytest.mark.parametrize("h,w", [(200, 328), (96, 160)])
def test_ragged_non_square_sizes_match_oracle(h, w):
    """Sizes the tile geometry of no kernel divides (feature maps 100x164 / 50x82 / 25x41 / 13x21 / 7x11 for 200 x 328: odd extents
    from the stem's and the stride-2 bottlenecks' rounding, non-square, the decoder's bilinear resize to a skip of a different aspect):
    eval AND train-mode forward against the CPU oracle on the same state -- logits 1e-3 of scale, commitment loss, code indices."""
    from oracle import torch_ref as R
    from tests.test_model_gpu import build, close, rel_close
    x = synth.uniform(7100 + h, (2, 3, h, w))
    gt = synth.blob_labels(7101 + h, 2, max(h, w), cell=8)[:, :h, :w].contiguous()
    model = build("vqreptunet1x1", 0.0, 1.0, 77, inputs=(x, gt))
    sd = {k: v.detach().float().cpu() if v.is_floating_point() else v.cpu() for k, v in model.state_dict().items()}
    ks = (0, 0, 512, 512, 512)
    model.eval()
    with torch.no_grad():
        logits = model(x.to(dev()))[0]
        feats = model.encoder(x.to(dev()).contiguous(memory_format=torch.channels_last))[1:]
        idx = [model.codebook[lvl](feats[lvl])[1].cpu() for lvl in (2, 3, 4)]
        ref_logits, _c, _u, _p, aux = R.vq_unet_forward(sd, x, False, ks, version=1)
    assert logits.shape == (2, 3, h, w)
    rel_close(logits, ref_logits, 1e-3, f"eval logits at {h}x{w}")
    for got, want, lvl in zip(idx, aux["indices"], (2, 3, 4)):
        assert torch.equal(got, want), f"level {lvl}: code indices differ from the oracle's"
    model.train()
    with torch.no_grad():
        logits, closs, usage, proto = model(x.to(dev()), gt.to(dev()), percent=80.0)
        p = {k: v.clone() for k, v in sd.items()}
        ref = R.vq_unet_forward(p, x, True, ks, gt=gt, version=1, percent=80.0)
    rel_close(logits, ref[0], 1e-3, f"train logits at {h}x{w}")
    close(closs, ref[1], rtol=1e-4, what="commitment")
    close(proto, ref[3], rtol=1e-4, what="prototype loss")


def test_bf16_argmin_mismatch_rate_is_reported():
    """SURVEY 5 (mixed precision): bf16 is a build-side speed mode whose argmin mismatch rate has to be REPORTED, not assumed zero.
    The benchmark's network shape (K = 512, 512^2) on the calibrated synthetic state of the fixtures, codebooks = the package's own
    10-iteration k-means on the network's fp32 eval features (every code alive: real near-tie structure).  Code indices of an eval
    forward under bf16 autocast against the same forward in fp32 (the parity mode), per level, and the logits' distance.  The
    mismatches are rows whose two best codes are closer than the bf16 rounding of the features moves them -- nothing the VQ kernel
    does (its arithmetic is exact fp32 on whatever rows it is given: tests/test_vq_gpu.py)."""
    from tests.test_model_gpu import build
    from vq_seg_amd.vector_quantizer.vq_img import kmeans
    size = 512
    m = build("vqreptunet1x1", 0.0, 1.0, 77, size=size)
    x = cases.model_inputs(s=size)[0].to(dev()).contiguous(memory_format=torch.channels_last)
    m.eval()
    torch.manual_seed(5)
    with torch.no_grad():
        feats = m.encoder(x)[1:]
        for lvl in (2, 3, 4):
            f = feats[lvl].float()
            rows = f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]).contiguous()
            means, bins = kmeans(rows, 512, 10)
            cb = m.codebook[lvl].codebook
            cb.embedding.weight.data.copy_(means)
            cb.initted = True
        from vq_seg_amd import nnf
        nnf.invalidate_weight_caches(m)                               # `.data` writes: drop the prepared codebooks (_wcache)

    def indices_and_logits(autocast):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            feats = m.encoder(x)[1:]
            out = [m.codebook[lvl](feats[lvl]) for lvl in (2, 3, 4)]
            logits = m(x)[0].float()
        return [o[1] for o in out], [float(o[3]) for o in out], logits
    idx32, dead32, log32 = indices_and_logits(False)
    idx16, dead16, log16 = indices_and_logits(True)
    lines = []
    for lvl, a, b, d in zip((2, 3, 4), idx32, idx16, dead32):
        rate = (a != b).float().mean().item()
        lines.append(f"level {lvl}: {a.numel()} rows, {d:.1f} % dead codes, bf16 index != fp32 index in {100 * rate:.2f} % of the rows")
        assert rate < 0.5
    rel = ((log16 - log32).abs().max() / log32.abs().max()).item()
    agree = (log16.argmax(1) == log32.argmax(1)).float().mean().item()
    lines.append(f"logits: max |bf16 - fp32| = {rel:.2e} of scale; arg-max class agrees on {100 * agree:.2f} % of the pixels")
    print("\n".join(lines))
    assert rel < 0.2 and agree > 0.9


def test_whole_model_with_live_kmeans_codebooks_matches_the_oracle():
    """the model-level fixtures build their codebooks from copies of the model's own feature rows (well separated
    winners by construction).  Here the calibrated network at 512^2 gets LIVE codebooks -- the package's 10-iteration k-means on its
    own fp32 eval features, K = 512 at every level, no dead code -- and its no-grad eval forward (the split-3 path of the pseudo-label
    passes) is compared with the CPU oracle on the same state: code indices of the three levels equal except rows whose two candidates
    are within 1e-4 relative in float64 (the features themselves differ by ~1e-5 between the devices, so a near-tie below that may
    legitimately flip; each such row is audited and counted), dead-code percentages equal, and -- when no index differs -- logits
    within north_star's 1e-3 of scale."""
    from oracle import torch_ref as R
    from tests.test_model_gpu import build, rel_close
    from vq_seg_amd import nnf
    from vq_seg_amd.vector_quantizer.vq_img import kmeans
    size = 512
    model = build("vqreptunet1x1", 0.0, 1.0, 77, size=size)
    x, _gt, _ = cases.model_inputs(s=size)
    xd = x.to(dev()).contiguous(memory_format=torch.channels_last)
    model.eval()
    torch.manual_seed(11)
    with torch.no_grad():
        feats = model.encoder(xd)[1:]
        for lvl in (2, 3, 4):
            f = feats[lvl].float()
            rows = f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]).contiguous()
            k = min(512, rows.shape[0] // 2)
            means, _bins = kmeans(rows, k, 10)
            cb = model.codebook[lvl].codebook
            w = cb.embedding.weight.data
            w.copy_(means.repeat((512 + k - 1) // k, 1)[:512])          # K = 512 slots; levels with fewer rows repeat their means
            cb.initted = True                                             # (duplicates: the LOWEST index must win, on both sides)
        nnf.invalidate_weight_caches(model)
        logits, _closs, usage, _proto = model(xd)
        idx = [model.codebook[lvl](feats[lvl])[1].cpu() for lvl in (2, 3, 4)]
    sd = {k_: v.detach().float().cpu() if v.is_floating_point() else v.cpu() for k_, v in model.state_dict().items()}
    with torch.no_grad():
        ref_logits, _c, ref_usage, _p, aux = R.vq_unet_forward(sd, x, False, (0, 0, 512, 512, 512), version=1)
    report, total_bad = [], 0
    for got, want, lvl in zip(idx, aux["indices"], (2, 3, 4)):
        f = feats[lvl].float()
        rows = f.permute(0, 2, 3, 1).reshape(-1, f.shape[1]).contiguous()
        W = model.codebook[lvl].codebook.embedding.weight.detach()
        n_bad, gap, excess = cases.near_tie_audit(rows, W, got.reshape(-1).to(dev()), want.reshape(-1).to(dev()))
        total_bad += n_bad
        report.append(f"level {lvl}: {rows.shape[0]} rows, index != oracle in {n_bad}" + (f" (fp64 gap of the two candidates <= {gap:.1e})" if n_bad else ""))
        assert n_bad <= 3 and gap < 1e-4 and excess < 1e-4, report[-1]
    assert torch.allclose(usage.double().cpu(), torch.stack([u.double() for u in ref_usage]).cpu(), atol=0.25)
    if total_bad == 0:
        rel_close(logits, ref_logits, 1e-3, "eval logits with live codebooks")
    report.append(f"dead codes GPU {usage.tolist()} oracle {[float(u) for u in ref_usage]}; indices differing in total: {total_bad}")
    print("\n".join(report))


def test_config_1_plain_unet_at_its_own_size_matches_the_oracle():
    """BASELINE config #1 (`config/CWFID_Unet.json`, plain UNet, 256 x 256, batch 2 -- "on CPU" in the reference; this repository has
    no CPU path by design, so the configuration runs on the MI355X and the CPU side is the ORACLE): eval and train-mode forward at
    256^2 against oracle/torch_ref.py::unet_forward on the same state (logits 1e-3 of scale), and one step of the intended
    supervised recipe (deprecated/train_baseline.py:128-140: loss = Dice + 0.5 CE, Adam) -- loss within 1e-4 of the oracle's, the
    sampled weight gradients within the model-level bar, every parameter finite after the step."""
    from oracle import torch_ref as R
    from tests.test_model_gpu import close, grad_close, rel_close
    from vq_seg_amd.loss import make_loss
    from vq_seg_amd.models.networks import make_model
    size, batch = 256, 2
    model = make_model({"name": "unet", "params": {"encoder_name": "resnet50", "num_classes": 3, "depth": 5, "encoder_weights": "imagenet_swsl"}})
    sd = synth.synth_state_dict(golden_io.layout("unet"), 91)
    model.load_state_dict(sd)
    model = model.to(dev())
    x, gt, _ = cases.model_inputs(b=batch, s=size, seed=6700)
    cases.set_bn_momentum(model, 1.0)                                 # calibrate the running statistics on the batch (as the fixtures do)
    model.train()
    with torch.no_grad():
        model(x.to(dev()))
    cases.set_bn_momentum(model, 0.1)
    state = {k: (v.detach().float().cpu() if v.is_floating_point() else v.cpu()) for k, v in model.state_dict().items()}
    model.eval()
    with torch.no_grad():
        y_eval = model(x.to(dev()))
        ref_eval = R.unet_forward({k: v.clone() for k, v in state.items()}, x, False)
    assert y_eval.shape == (batch, 3, size, size)
    rel_close(y_eval, ref_eval, 1e-3, "config 1 eval logits at 256^2")
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    y = model(x.to(dev()))
    loss = make_loss("dice_loss", 3, ignore_index=255)(y, gt.to(dev())) + 0.5 * F.cross_entropy(y, gt.to(dev()), ignore_index=255)
    p = {k: v.clone() for k, v in state.items()}
    probes = ["segmentation_head.0.weight", "encoder.conv1.weight", "decoder.blocks.4.1.0.weight"]
    for k in probes:
        p[k].requires_grad_(True)
    ref_y = R.unet_forward(p, x, True)
    ref_loss = R.dice_loss(ref_y, gt) + 0.5 * F.cross_entropy(ref_y, gt, ignore_index=255)
    rel_close(y, ref_y, 1e-3, "config 1 train logits at 256^2")
    close(loss, ref_loss, rtol=1e-4, what="Dice + 0.5 CE")
    loss.backward()
    ref_loss.backward()
    named = dict(model.named_parameters())
    for k in probes:
        grad_close(golden_io.probe(named[k].grad), golden_io.probe(p[k].grad), 3e-2, 0.15, "grad " + k)
    opt.step()
    assert all(torch.isfinite(v).all() for v in model.parameters())

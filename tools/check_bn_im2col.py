"""Debug aid: vqseg_im2col_f against F.unfold (must be bit-exact) and vqseg_bn_finalize_f against a numpy
double-precision merge of the same partials."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream

for reflect in (0, 1):
    for dt in (torch.float32, torch.bfloat16):
        n, h, w = 2, 40, 36
        x = torch.randn(n, h, w, 3, device=dev)
        ho, wo, kp = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1, 160
        out = torch.full((n, ho, wo, kp), 7.0, dtype=dt, device=dev)
        assert L.vqseg_im2col_f(int(dt == torch.bfloat16), x.data_ptr(), n, h, w, 3, 7, 7, 2, 3, reflect, ho, wo, kp, out.data_ptr(), st()) == 0
        xp = F.pad(x.permute(0, 3, 1, 2), (3, 3, 3, 3), mode="reflect" if reflect else "constant")
        u = F.unfold(xp, 7, stride=2)                                   # (n, 3*49 [ci, kh, kw], L)
        u = u.view(n, 3, 7, 7, ho, wo).permute(0, 4, 5, 2, 3, 1).reshape(n, ho, wo, 147).to(dt)
        print("im2col reflect", reflect, dt, "equal:", torch.equal(out[..., :147], u), "pad zero:", bool((out[..., 147:] == 0).all()))

for (M, C) in ((2048, 64), (2048, 32), (100000, 64), (777, 256), (2 * 1024 * 1024, 32)):
    rps = 64 if C >= 64 else 32
    n_slots = (M + rps - 1) // rps
    slots_alloc = L.vqseg_conv_stat_slots(M, C)
    rng = np.random.default_rng(M + C)
    data = rng.normal(0.3, 1.7, size=(M, C)).astype(np.float32)
    part = np.zeros((slots_alloc, 2, C), np.float32)
    for s in range(n_slots):
        blk = data[s * rps:(s + 1) * rps].astype(np.float64)
        part[s, 0] = blk.mean(0)
        part[s, 1] = ((blk - blk.mean(0)) ** 2).sum(0)
    # double merge of the float partials
    ns = np.minimum(rps, M - np.arange(n_slots) * rps).astype(np.float64)[:, None]
    p64 = part[:n_slots].astype(np.float64)
    mean = (ns * p64[:, 0]).sum(0) / M
    m2 = (p64[:, 1] + ns * (p64[:, 0] - mean) ** 2).sum(0)
    var = m2 / M
    pt = torch.from_numpy(part).to(dev)
    gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
    rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
    coef = torch.empty(4, C, device=dev)
    assert L.vqseg_bn_finalize_f(pt.data_ptr(), M, C, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, 1,
                                 coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(), None, None, st()) == 0
    torch.cuda.synchronize()
    em = np.abs(coef[2].cpu().numpy() - mean).max() / np.abs(mean).max()
    ei = np.abs(coef[3].cpu().numpy() - 1 / np.sqrt(var + 1e-5)).max()
    erv = np.abs(rv.cpu().numpy() - (0.9 + 0.1 * m2 / (M - 1))).max()
    print(f"bn_finalize M={M} C={C}: mean rel err {em:.2e} invstd abs err {ei:.2e} running_var err {erv:.2e}")

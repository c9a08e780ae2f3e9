"""bf16 candidate filter vs the exact kernel on the bench's three VQ levels (B = 32 at 512^2, K = 512; bf16 rows), each level alone and
the grouped launch of a forward; k-means codebooks on the rows themselves (the trained state: no dead code), so the share of rows
left to the exact kernel is the realistic one.   python tools/vq_filter_bench.py [K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vq_seg_amd import _hip
from vq_seg_amd.vector_quantizer.vq_img import kmeans
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
shapes = ((131072, 512), (32768, 1024), (8192, 2048))
rows, books, preps = [], [], []
for n, c in shapes:
    # post-ReLU features with channel-wise structure (smooth mixtures), bf16
    base = torch.relu(torch.randn(64, c, device=dev))
    mix = torch.softmax(2.0 * torch.randn(n, 64, device=dev), dim=1)
    x = (mix @ base + 0.1 * torch.relu(torch.randn(n, c, device=dev))).to(torch.bfloat16)
    W, _ = kmeans(x.float(), K, 10)
    rows.append(x); books.append(W.contiguous()); preps.append(_hip.vq_prepare(W))


def timed(fn, reps=20):
    for _ in range(3): fn()
    _hip.profile_begin(4 * reps)
    for _ in range(reps): fn()
    recs = _hip.profile_collect(4 * reps, with_kind=True)
    per = {}
    for n, c, k, ms, kd in recs: per.setdefault((n, c, k, kd), []).append(ms)
    return {key: sorted(v)[len(v) // 2] for key, v in per.items()}

for mode in ("filter", "exact"):
    _hip.set_option("vq_bf16_filter", 1 if mode == "filter" else 0)
    line = []
    for x, W, p in zip(rows, books, preps):
        t = timed(lambda: _hip.vq_assign(x, W, prepared=p))
        (key, ms), = t.items()
        _, amb = _hip.vq_assign(x, W, prepared=p, want_filter_count=True)
        torch.cuda.synchronize()
        line.append(f"N{key[0]}xC{key[1]}: {ms * 1e3:6.1f} us ({2.0 * key[0] * key[1] * key[2] / ms / 1e9:6.1f} TF alg.)" + (f" pairs/row {int(amb) / key[0]:.3f}" if (amb is not None and mode == "filter") else ""))
    t = timed(lambda: _hip.vq_forward_group(rows, books, preps, True, [1.0, 1.0, 1.0]))
    tot = sum(t.values())
    flops = sum(2.0 * n * c * K for n, c in shapes)
    print(f"[{mode:6s}] " + " | ".join(line) + f" | grouped launch {tot * 1e3:6.1f} us = {flops / tot / 1e9:6.1f} TF algorithmic", flush=True)
_hip.set_option("vq_bf16_filter", 1)

import sys, torch
sys.path.insert(0, '/root/repo')
from vq_seg_amd import _hip
from tests import synth
dev = torch.device('cuda:0')
for (n, c, k) in [(64, 64, 32), (64, 64, 64), (256, 64, 32), (64, 128, 32), (64, 256, 32), (64, 64, 256), (64, 512, 512), (3000, 512, 512), (64, 72, 32), (64, 200, 40)]:
    rows = synth.relu_features(n + c, (n, c)).to(dev).bfloat16()
    cb = synth.relu_features(k + 5, (k, c)).to(dev)
    i32 = _hip.vq_forward(rows.float(), cb, False, 0.0)[1]
    i16 = _hip.vq_forward(rows, cb, False, 0.0)[1]
    bad = (i32 != i16).nonzero().flatten()
    print((n, c, k), 'mismatch', bad.numel(), bad[:8].tolist())

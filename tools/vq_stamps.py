import os, sys, torch, ctypes
sys.path.insert(0, '/root/repo')
from vq_seg_amd import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
n, c, k = 131072, 512, 512
x = torch.relu(torch.randn(n, c, device=dev)); W = torch.relu(torch.randn(k, c, device=dev))
prep = _hip.vq_prepare(W)
st = torch.zeros(2048 * 6, dtype=torch.int64, device=dev)
for _ in range(3): _hip.vq_assign(x, W, prepared=prep)
L.vqseg_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
_hip.vq_assign(x, W, prepared=prep)
torch.cuda.synchronize()
L.vqseg_debug_set_stamps(None)
s = st.view(2048, 6).cpu().double()
d = s[:, 1:] - s[:, :-1]
print("mean cycles per phase [prologue, mainloop, dist, butterfly, atomics]:", d.mean(0).tolist())
print("total per WG:", (s[:, 5] - s[:, 0]).mean().item(), "span all:", (s[:, 5].max() - s[:, 0].min()).item())
t0 = s[:, 0] - s[:, 0].min()
import numpy as np
print("start time quantiles (cycles):", np.quantile(t0.numpy(), [0, .2, .25, .3, .5, .75, 1.0]))

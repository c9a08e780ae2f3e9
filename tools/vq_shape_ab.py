"""v_mfma_f32_16x16x4_f32 INSIDE the real distance + argmin kernel, with clock stamps in both arms.
    python tools/vq_shape_ab.py <libvqseg_hip_tl.so | libvqseg_hip_s16.so>
Both libraries are timeline builds (-DVQ_TIMELINE=1); the second also has -DVQ_SHAPE16=1: every 32x32x2 MFMA of the main loop issued as
two 16x16x4 MFMAs on quarters of its accumulator tile (same flops, same operand traffic; results garbage -- timing only).  One grouped
launch of the bench's three levels on fp32 rows (what the kernel runs on in the step); prints launch time (HIP events), the shader clock
inside the kernel (s_memtime / s_memrealtime per workgroup) and the main-loop time per level."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VQSEG_LIB"] = os.path.abspath(sys.argv[1])
import torch  # noqa: E402
from vq_seg_amd import _hip  # noqa: E402
dev = torch.device("cuda:0")
L = _hip.lib()
L.vqseg_debug_timeline.argtypes = [ctypes.c_void_p]
shapes = ((131072, 512, 512), (32768, 1024, 512), (8192, 2048, 512))
torch.manual_seed(0)
rows = [torch.relu(torch.randn(n, c, device=dev)) for n, c, k in shapes]
books = [torch.relu(torch.randn(k, c, device=dev)) for n, c, k in shapes]
preps = [_hip.vq_prepare(w) for w in books]
n_wg = sum((n // 128 + 7) // 8 * 8 * (k // 32) for n, c, k in shapes)
for _ in range(5):
    _hip.vq_forward_group(rows, books, preps, False, [1.0] * 3)
torch.cuda.synchronize()
res = []
for rep in range(5):
    tl = torch.zeros(n_wg, 16, dtype=torch.int64, device=dev)
    L.vqseg_debug_timeline(tl.data_ptr())
    L.vqseg_profile_begin(16)
    _hip.vq_forward_group(rows, books, preps, False, [1.0] * 3)
    recs = _hip.profile_collect(16)
    torch.cuda.synchronize()
    L.vqseg_debug_timeline(None)
    t = tl.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] != 0]
    rt = t[:, 0:12:2].astype(np.float64) * 0.01
    ck = t[:, 1:12:2].astype(np.float64)
    mhz = (ck[:, 5] - ck[:, 0]) / np.maximum(rt[:, 5] - rt[:, 0], 1e-9)
    loops = {int(lv): float(np.median((rt[:, 2] - rt[:, 1])[t[:, 14] == lv])) for lv in np.unique(t[:, 14])}
    res.append((sum(r[3] for r in recs) * 1e3, float(np.median(mhz)), loops))
us = sorted(r[0] for r in res)[len(res) // 2]
flops = sum(2.0 * n * c * k for n, c, k in shapes)
print(f"{os.path.basename(sys.argv[1]):24s} launch {us:7.1f} us = {flops / us / 1e6:6.1f} TF/s ({flops / us / 1e6 / 157.3:.3f} of peak)  in-kernel clock "
      f"{np.median([r[1] for r in res]):6.0f} MHz  main loop per workgroup (us): " +
      ", ".join(f"C={ {0: 2048, 1: 1024, 2: 512}[lv] }: {np.median([r[2][lv] for r in res]):.1f}" for lv in sorted(res[0][2])), flush=True)

"""vq_seg_amd -- MI355X (gfx950) native hot path of the VQ-UNet segmentation trainer.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed/RCCL only).
Compute:   hand-written HIP kernels in libvqseg_hip.so behind the C ABI of include/vqseg.h.

The sub-packages mirror the reference's nn.Module surface so the reference's
trainer can drive them:  vq_seg_amd.vector_quantizer  <->  vector_quantizer/,
vq_seg_amd.models  <->  models/.  There is NO CPU or eager fallback: calling a forward
without the HIP library or on a CPU tensor raises.
"""
__version__ = "0.1.0"

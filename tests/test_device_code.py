"""Properties of the generated gfx950 code that performance depends on and that no numerical test sees: the hot MFMA
kernels must not spill registers (a spill in the tap loop of the 3x3 kernel put full `s_waitcnt vmcnt(0)` drains of its
weight-DMA ring there; the host object's disassembly shows nothing of it -- this reads the DEVICE assembly)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vq_seg_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# kernel-name fragments (mangled) that must be spill free
HOT = {
    "vq_kernels.hip": ["vq_assign_f32_kernelILi8E", "vq_assign_f32_kernelILi4E", "vq_assign_f32_kernelILi2E", "vq_gather_kernel"],
    "conv_kernels.hip": ["conv3x3_patch_kernelILi128ELi3ELb1ELi64E", "conv3x3_patch_kernelILi64ELi3ELb1ELi64E",
                         "conv3x3_patch_kernelILi32ELi3ELb1ELi64E", "conv3x3_patch_kernelILi128ELi3ELb1ELi32E",
                         "conv3x3_patch_kernelILi32ELi0ELb1ELi32E", "conv3x3_patch_kernelILi64ELi0ELb1ELi32ELb0ELi512E",
                         "conv3x3_patch_kernelILi32ELi0ELb1ELi32ELb1ELi512E", "conv_igemm_glds_kernelILi256ELi128ELi8ELi3ELi1E",
                         "conv_igemm_glds_kernelILi128ELi128ELi4ELi2ELi1E", "conv_wgrad3x3_kernelILi4ELi2E",
                         "conv_wgrad1x1_kernelILi8ELi4E"],
}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("source", sorted(HOT))
def test_hot_kernels_do_not_spill(source):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dev.s")
        subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", os.path.join(CSRC, source), "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        asm = open(out).read()
    meta = re.findall(r"\.name:\s+(\S+)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size: (\d+)(?:.*\n){0,12}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)", asm)
    assert len(meta) > 10, "kernel metadata not found in the device assembly"
    for frag in HOT[source]:
        hits = [(n, int(p), int(s)) for n, p, _v, s in meta if frag in n]
        assert hits, f"{frag}: no such kernel in {source}"
        for name, private, spilled in hits:
            assert spilled == 0 and private == 0, f"{name}: {spilled} spilled VGPRs, {private} B of scratch"

#!/bin/bash
# Lists the gfx950 kernels that spill registers (device assembly; the host object's disassembly shows nothing of this):
#   bash tools/check_spills.sh [file.hip ...]          (default: every kernel file)
cd "$(dirname "$0")/../vq_seg_amd/csrc"
for f in ${@:-vq_kernels.hip conv_kernels.hip nn_kernels.hip loss_kernels.hip}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S "$f" -o /tmp/spills_$$.s 2>/dev/null || { echo "$f: compile failed"; continue; }
  python3 - "$f" /tmp/spills_$$.s <<'PY'
import re, sys
s = open(sys.argv[2]).read()
n = 0
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size: (\d+)(?:.*\n){0,12}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count: (\d+)", s):
    name, priv, vg, sp = m.groups()
    n += 1
    if int(sp) or int(priv):
        print(f"{sys.argv[1]}: {name[:100]}  scratch {priv} B  vgprs {vg}  spilled {sp}")
print(f"{sys.argv[1]}: {n} kernels checked")
PY
  rm -f /tmp/spills_$$.s
done

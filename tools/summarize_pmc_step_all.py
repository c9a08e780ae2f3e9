"""Summarise tools/pmc_step_all.sh: fabric-side bytes (FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024; Infinity-Cache hits counted) of EVERY
kernel of the bench step, by kernel family, per step.   python tools/summarize_pmc_step_all.py gpurun_out/pmc_step_all profiles/r04 [steps = 3]"""
import csv, os, re, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
acc = defaultdict(lambda: defaultdict(float))
def short(n):
    n = re.sub(r"^void ", "", n).replace("vqseg::", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"^_ZN5vqseg\d+", "", n)
    n = re.sub(r"^_GLOBAL__N_1\d+", "", n)
    n = n.split("(")[0]
    n = re.sub(r"<.*", "", n) if not n.startswith(("conv", "vq_assign")) else n
    n = re.sub(r"I(DF16b|f)[A-Za-z0-9_]*$", "", n)
    return n[:70]
for name, ctr, scale in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
    for r in csv.DictReader(open(os.path.join(src, name + ".csv"))):
        if r["Counter_Name"] != ctr:
            continue
        k = short(r["Kernel_Name"])
        acc[k][name] += float(r["Counter_Value"]) * scale
        if name == "fetch":
            acc[k]["n"] += 1
            acc[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
def family(k):
    if k.startswith("bn_"): return "BatchNorm"
    if k.startswith(("conv_wgrad", "wgrad_reduce", "reduce_partials")): return "convolution weight gradients (+ slab sums)"
    if k.startswith(("conv", "stem7", "im2col", "reflect")): return "convolution forward / data gradient (+ stem patches, reflect folds)"
    if k.startswith(("vq_", "km_")): return "vector quantisation"
    if "bilinear" in k or "maxpool" in k or k.startswith("s3_"): return "resize / pool / split-3 elementwise"
    if k.startswith(("dice", "softmax_stats", "kth", "proto", "confusion", "head", "loss_combine")): return "losses, head, metrics"
    if k.startswith("adam"): return "optimiser"
    return "ATen / runtime"
fam = defaultdict(float)
tot = 0.0
rows = []
for k, c in acc.items():
    gb = (c["fetch"] + c["write"]) / steps / 1e9
    fam[family(k)] += gb; tot += gb
    rows.append((gb, k, c))
rows.sort(reverse=True)
with open(dst + "_step_traffic.md", "w") as f:
    f.write("# Fabric-side bytes of the whole bench step, by kernel (PMC)\n\n`bash tools/pmc_step_all.sh` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench.py --no-cpu-baseline --no-extras "
            "--steps 2 --warmup 1`, every kernel); FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count) + WRITE_SIZE KiB x 1024; Infinity-Cache hits are counted, so this is "
            "traffic past the XCDs' L2s, an upper bound of HBM traffic.  Durations are those of the serialised counter run.\n\n")
    f.write(f"**{tot:.0f} GB per step** (B = 32 + 32 images): against the ~150 ms step = {tot / 0.150 / 1000:.1f} TB/s averaged over the whole step.\n\n| family | GB / step | share |\n|---|---|---|\n")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
        f.write(f"| {k} | {v:.1f} | {100 * v / tot:.1f} % |\n")
    f.write("\n| kernel | launches / step | us (serialised) | fetched MB / launch | written MB / launch | GB / step |\n|---|---|---|---|---|---|\n")
    for gb, k, c in rows[:60]:
        n = max(c["n"], 1)
        f.write(f"| {k} | {c['n'] / steps:.0f} | {c['ns'] / n / 1e3:.1f} | {c['fetch'] / n / 1e6:.1f} | {c['write'] / n / 1e6:.1f} | {gb:.2f} |\n")
print(f"{tot:.1f} GB per step")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f"  {k:70s} {v:7.1f}")

"""gpurun_out/pmc_weak/<layer>.{fetch,write,mfma}.csv (tools/pmc_weak.sh) -> profiles/<round>_conv_weak_layers_pmc.md
usage: python tools/summarize_pmc_weak.py gpurun_out/pmc_weak profiles/r04
HBM-side bytes as MI355X_MICROARCH.md prescribes: FETCH_SIZE (KiB units) x 1024 x 2 (gfx950 counts a 128-B streaming read as one
64-B request), WRITE_SIZE x 1024; separate passes; per launch = median over the repetitions of the same dispatch position."""
import csv, os, re, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
LAYERS = ["l1.conv2", "l2.0.conv2", "l3.0.conv2", "l4.0.conv2", "dec4.0", "dec4.1"]
out = ["# Convolution layers below both roofs: counters per kernel (round 4)\n",
       "`bash tools/pmc_weak.sh` (rocprofv3 --kernel-trace --pmc, one counter group per pass) over `tools/weak_layers.py <layer>`: B = 32, bf16, "
       "training-mode forward + backward of ONE layer at the bench's shape, 5 repetitions after a warm-up; medians per launch.  Bytes: "
       "FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024 (MI355X_MICROARCH.md); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).\n"]
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void vqseg::", "").replace("vqseg::", "")
    return n[:60]
for layer in LAYERS:
    per = defaultdict(lambda: defaultdict(list))
    desc = ""
    for pname in ("fetch", "write", "mfma"):
        f = os.path.join(src, f"{layer}.{pname}.csv")
        if not os.path.exists(f):
            continue
        log = open(os.path.join(src, f"{layer}.{pname}.log")).read().splitlines()
        desc = next((l for l in log if l.startswith(layer + ":")), desc)
        rows = list(csv.DictReader(open(f)))
        for r in rows:
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            per[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if pname == "mfma" and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                per[key]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out.append(f"\n## {desc or layer}\n\n| kernel | workgroups | launches per pass | us | fetched MB | written MB | MFMA busy | effective clock |\n|---|---|---|---|---|---|---|---|\n")
    med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
    for (kn, grid), c in sorted(per.items(), key=lambda kv: -med(kv[1].get("ns", [0])) ):
        ns = med(c.get("ns", []))
        gui = med(c.get("GRBM_GUI_ACTIVE", [])) / 8
        busy = med(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / (gui * 1024) * 100 if gui == gui and gui > 0 else float("nan")
        n = len(c.get("ns", [])) / 6.0
        out.append(f"| {kn} | {grid // 256 if 'patch' not in kn else grid // 512} | {n:.1f} | {ns / 1e3:.1f} | {med(c.get('FETCH_SIZE', [])) * 2048 / 1e6:.1f} | "
                   f"{med(c.get('WRITE_SIZE', [])) * 1024 / 1e6:.1f} | {busy:.1f} % | {gui / ns:.2f} GHz |\n")
open(dst + "_conv_weak_layers_pmc.md", "w").write("".join(out))
print("".join(out))

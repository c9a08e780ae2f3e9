"""Summarise tools/pmc_conv.sh (gpurun_out/pmc_conv/pass*.csv) for conv3x3_patch_kernel into profiles/<round>_conv_patch_pmc.md.
usage: python tools/summarize_pmc_conv.py gpurun_out/pmc_conv profiles/r01"""
import csv, glob, os, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
# tools/ab_conv.py launches its layers in this order, 7 repetitions x (1 + 10) launches each with one variant: rows of a pass in
# dispatch order map to layers by position (name, C1, C2, Cout, H = W)
LAYERS = [("dec0.0", 2048, 0, 1024, 16), ("dec0.1", 1024, 0, 1024, 16), ("dec1.0", 1024, 1024, 512, 32), ("dec1.1", 512, 0, 512, 32),
          ("dec2.0", 512, 512, 256, 64), ("dec2.1", 256, 0, 256, 64), ("dec3.0", 256, 256, 128, 128), ("dec3.1", 128, 0, 128, 128),
          ("l3.conv2", 256, 0, 256, 32), ("l2.conv2", 128, 0, 128, 64)]
PER_LAYER, B = 77, 32
traffic = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pass*.csv"))):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    counters = sorted({r["Counter_Name"] for r in rows})
    for name in counters:
        sel = [r for r in rows if r["Counter_Name"] == name]
        for i, r in enumerate(sel):
            if name in ("FETCH_SIZE", "WRITE_SIZE") and len(sel) == PER_LAYER * len(LAYERS):
                traffic[i // PER_LAYER][name].append(float(r["Counter_Value"]))
    for r in rows:
        g = int(r["Grid_Size"]) // 512
        acc[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[g]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(dst + "_conv_patch_pmc.md", "w") as f:
    f.write("# conv3x3_patch_kernel<128,3,true,64>: PMC summary (tools/ab_conv.py decoder layers, B = 32, one stream)\n\n"
            "GRBM_GUI_ACTIVE is summed over 8 XCDs; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); the wait / active\n"
            "columns are fractions of SQ_WAVE_CYCLES (WAIT_ANY: parked on s_waitcnt / barrier, WAIT_INST_ANY: issue stalls).  The effective clock\n"
            "reads high on dispatches well under 0.3 ms (MI355X_MICROARCH.md, DVFS give-back); profiled passes clock a few % lower than plain runs.\n\n"
            "| workgroups | us | effective clock (GRBM_GUI_ACTIVE / 8 / time) | MFMA busy | LDS bank-conflict cycles / LDS active | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | WAIT_INST_LDS |\n|---|---|---|---|---|---|---|---|---|\n")
    for g, c in sorted(acc.items()):
        a = {k: sum(v) / len(v) for k, v in c.items()}
        cyc = a["GRBM_GUI_ACTIVE"] / 8
        wc = a["SQ_WAVE_CYCLES"]
        f.write(f"| {g} | {a['ns'] / 1e3:.0f} | {cyc / a['ns']:.2f} GHz | {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100:.1f} % | "
                f"{a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1) * 100:.1f} % | {a['SQ_WAIT_ANY'] / wc * 100:.1f} % | "
                f"{a['SQ_WAIT_INST_ANY'] / wc * 100:.1f} % | {a['SQ_ACTIVE_INST_ANY'] / wc * 100:.1f} % | {a['SQ_WAIT_INST_LDS'] / wc * 100:.1f} % |\n")
    if traffic:
        f.write("\n## HBM traffic per launch (FETCH_SIZE KiB x 1024 x 2 -- the gfx950 half-count of wide streaming reads --, WRITE_SIZE KiB x 1024; "
                "separate passes)\n\nAlgorithmic bytes = input rows + output rows + packed weights, bf16.  Operands are the same buffers every "
                "launch (tools/ab_conv.py), so layers whose operands fit the 256 MiB Infinity Cache can read BELOW their algorithmic bytes; conversely FETCH_SIZE counts the L2's "
                "memory-side requests, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM): the 2-4x of the large layers is the input patch "
                "re-read once per 128-channel Cout chunk (4-8 chunks), mostly served by that cache, not by HBM -- placing the chunks of a pixel "
                "tile on one XCD at the same time (conv3x3_patch_xcd_pair=1) changes the time of no layer by more than 1 %.\n\n"
                "| layer | Cin -> Cout @ HxW | algorithmic MB | fetched MB | written MB | (fetched + written) / algorithmic |\n|---|---|---|---|---|---|\n")
        for i, (name, c1, c2, cout, hw) in enumerate(LAYERS):
            t = traffic.get(i)
            if not t or "FETCH_SIZE" not in t or "WRITE_SIZE" not in t:
                continue
            med = lambda v: sorted(v)[len(v) // 2]
            fetched, written = med(t["FETCH_SIZE"]) * 1024 * 2, med(t["WRITE_SIZE"]) * 1024
            alg = (B * hw * hw * (c1 + c2 + cout) + 9 * (c1 + c2) * cout) * 2
            f.write(f"| {name} | {c1 + c2} -> {cout} @ {hw}x{hw} | {alg / 1e6:.1f} | {fetched / 1e6:.1f} | {written / 1e6:.1f} | {(fetched + written) / alg:.2f} |\n")
print(open(dst + "_conv_patch_pmc.md").read())

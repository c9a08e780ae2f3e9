"""Summarise tools/pmc_conv.sh (gpurun_out/pmc_conv/pass*.csv) for conv3x3_patch_kernel into profiles/<round>_conv_patch_pmc.md.
usage: python tools/summarize_pmc_conv.py gpurun_out/pmc_conv profiles/r01"""
import csv, glob, os, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pass*.csv"))):
    for r in csv.DictReader(open(f)):
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
print(open(dst + "_conv_patch_pmc.md").read())

"""Summarise the rocprofv3 --pmc passes of tools/pmc_vq.sh (gpurun_out/pmc_r1/*.csv) for vq_assign_f32_kernel into
profiles/<round>_vq_assign_pmc.{json,md}.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE is in KiB and reports HALF of the bytes of wide coalesced reads on gfx950 -> x 1024 x 2; WRITE_SIZE is
in KiB and exact -> x 1024.  GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy fraction =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
usage: python tools/summarize_pmc.py gpurun_out/pmc_r1 profiles/r01"""
import csv, json, os, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
K = 512                                                      # bench.py's codebook size
acc = defaultdict(lambda: defaultdict(list))
filt = defaultdict(lambda: defaultdict(list))                # r4: the bf16 candidate filter's three kernels, keyed by kernel and grid
for name in ("fetch", "write", "mfma"):
    for r in csv.DictReader(open(os.path.join(src, name + ".csv"))):
        mf = re.search(r"(vq_filter_bf16_kernel|vq_resolve_kernel|vq_rescore_kernel)", r["Kernel_Name"])
        if mf:
            key = f"{mf.group(1)}_WG{int(r['Grid_Size']) // 256}"
            filt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            filt[key]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            continue
        mt = re.search(r"vq_assign_f32_kernel(?:<(\d+)|ILi(\d+)E)", r["Kernel_Name"])   # demangled or mangled (bf16 rows) name
        t = int(mt.group(1) or mt.group(2))
        wgs = int(r["Grid_Size"]) // 256
        if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 100000:
            continue                                         # the gated overflow launch of the bf16 filter: every workgroup exits at once
        rows = "bf16" if "DF16b" in r["Kernel_Name"] or "__bf16" in r["Kernel_Name"] or "bfloat" in r["Kernel_Name"] else "f32"
        key = f"WG{wgs}_T{t}_{rows}"                            # one launch serves the three levels of a forward -> keyed by its grid and row type
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[key]["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for key, c in sorted(acc.items()):
    if not all(k in c for k in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")):
        continue
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    e = {"launches_sampled": len(c["FETCH_SIZE"]), "fetch_bytes": avg["FETCH_SIZE"] * 1024 * 2, "write_bytes": avg["WRITE_SIZE"] * 1024,
         "mfma_busy_frac": avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024),
         "cycles_per_xcd": avg["GRBM_GUI_ACTIVE"] / 8}
    e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    out[key] = e
fout = {}
for key, c in sorted(filt.items()):
    avg = {k: sum(v) / len(v) for k, v in c.items()}
    e = {"launches_sampled": len(c.get("FETCH_SIZE", [])), "fetch_bytes": avg.get("FETCH_SIZE", 0.0) * 1024 * 2, "write_bytes": avg.get("WRITE_SIZE", 0.0) * 1024,
         "us": avg["ns"] / 1e3}
    if "GRBM_GUI_ACTIVE" in avg:
        e["mfma_busy_frac"] = avg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (avg["GRBM_GUI_ACTIVE"] / 8 * 1024)
        e["clock_ghz"] = avg["GRBM_GUI_ACTIVE"] / 8 / avg["ns"]
    e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
    fout[key] = e
json.dump({"kernel": "vq_assign_f32_kernel", "source": "rocprofv3 --pmc passes over `python bench.py --steps 2 --warmup 1` (tools/pmc_vq.sh)",
           "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 half-count), WRITE_SIZE KiB x1024, GRBM_GUI_ACTIVE / 8 XCDs", "shapes": out,
           "bf16_filter": fout},
          open(dst + "_vq_assign_pmc.json", "w"), indent=1)
with open(dst + "_vq_assign_pmc.md", "w") as f:
    f.write("# vq_assign_f32_kernel: PMC summary (per launch, averages)\n\n")
    f.write("| launch (workgroups, tiles/wave, row type) | launches | HBM fetch MB (x2 corrected) | HBM write MB | MFMA busy | cycles / XCD |\n|---|---|---|---|---|---|\n")
    for k, e in out.items():
        f.write(f"| {k} | {e['launches_sampled']} | {e['fetch_bytes'] / 1e6:.1f} | {e['write_bytes'] / 1e6:.2f} | {e['mfma_busy_frac'] * 100:.1f} % | {e['cycles_per_xcd']:.0f} |\n")
    if fout:
        f.write("\n# bf16 candidate filter (bf16 rows): its three kernels per grouped launch\n\n"
                "| kernel, workgroups | launches | us | HBM fetch MB (x2 corrected) | HBM write MB | MFMA busy | effective clock |\n|---|---|---|---|---|---|---|\n")
        for k, e in fout.items():
            f.write(f"| {k} | {e['launches_sampled']} | {e['us']:.1f} | {e['fetch_bytes'] / 1e6:.1f} | {e['write_bytes'] / 1e6:.2f} | "
                    f"{e.get('mfma_busy_frac', float('nan')) * 100:.1f} % | {e.get('clock_ghz', float('nan')):.2f} GHz |\n")
print(open(dst + "_vq_assign_pmc.md").read())

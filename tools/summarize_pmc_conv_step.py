"""Summarise tools/pmc_conv_step.sh (gpurun_out/pmc_conv_step/{fetch,write}.csv): HBM-side bytes of every convolution kernel of the bench step,
per kernel name and as the aggregate of the 3x3 bf16 forward / data-gradient launches (what bench.py's roofline_conv covers).
Corrections per MI355X_MICROARCH.md: FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count), WRITE_SIZE KiB x 1024; Infinity-Cache hits are counted.
    python tools/summarize_pmc_conv_step.py gpurun_out/pmc_conv_step profiles/r04 [steps_in_the_run = 3]"""
import csv, json, os, re, sys
from collections import defaultdict
src, dst = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
acc = defaultdict(lambda: defaultdict(float))
def short(n):
    n = re.sub(r"^void ", "", n).replace("vqseg::", "")
    return n.split("(")[0]
for name, ctr, scale in (("fetch", "FETCH_SIZE", 2048.0), ("write", "WRITE_SIZE", 1024.0)):
    for r in csv.DictReader(open(os.path.join(src, name + ".csv"))):
        if r["Counter_Name"] != ctr:
            continue
        k = short(r["Kernel_Name"])
        acc[k][name] += float(r["Counter_Value"]) * scale
        acc[k]["n_" + name] += 1
        if name == "fetch":
            acc[k]["ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
def is_3x3_bf16(k):
    m = re.match(r"conv3x3_patch_kernel<(.*)>", k)
    if m:
        return m.group(1).split(", ")[4] == "false"            # not the split-3 instantiation
    m = re.match(r"conv_igemm_glds_kernel<(.*)>", k)
    if m:
        a = m.group(1).split(", ")
        return a[5] == "false" and a[6] == "false"              # not split-3, not the linear-pixel (1x1) prologue: 3x3 stride-2 / ring / parity classes (+ the stride-2 1x1 projections)
    return False
rows, agg = [], defaultdict(float)
for k, c in sorted(acc.items(), key=lambda kv: -(kv[1]["fetch"] + kv[1]["write"])):
    n = max(c["n_fetch"], 1)
    e = {"kernel": k, "launches_per_step": round(c["n_fetch"] / steps, 1), "us": c["ns"] / n / 1e3, "fetch_bytes_per_launch": c["fetch"] / n,
         "write_bytes_per_launch": c["write"] / max(c["n_write"], 1), "hbm_gb_per_step": (c["fetch"] + c["write"]) / steps / 1e9}
    rows.append(e)
    if is_3x3_bf16(k):
        agg["launches"] += c["n_fetch"]; agg["fetch"] += c["fetch"]; agg["write"] += c["write"]; agg["ns"] += c["ns"]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1` (tools/pmc_conv_step.sh); "
                 "two streams: a launch's counters are its own, its duration is not (kernels are serialised by the counter collection)",
       "corrections": "FETCH_SIZE KiB x1024 x2 (gfx950 half-count), WRITE_SIZE KiB x1024; Infinity-Cache hits counted",
       "conv3x3_bf16_fwd_dgrad": {"launches_per_step": agg["launches"] / steps, "hbm_bytes_per_launch": (agg["fetch"] + agg["write"]) / max(agg["launches"], 1),
                                  "fetch_bytes_per_launch": agg["fetch"] / max(agg["launches"], 1), "write_bytes_per_launch": agg["write"] / max(agg["launches"], 1),
                                  "hbm_gb_per_step": (agg["fetch"] + agg["write"]) / steps / 1e9},
       "kernels": rows}
json.dump(out, open(dst + "_conv_step_pmc.json", "w"), indent=1)
with open(dst + "_conv_step_pmc.md", "w") as f:
    f.write("# Convolution kernels of the bench step: HBM-side bytes per launch (PMC)\n\n" + out["source"] + ".  " + out["corrections"] + ".\n\n")
    a = out["conv3x3_bf16_fwd_dgrad"]
    f.write(f"3x3 bf16 forward + data-gradient launches (bench.py `roofline_conv`): {a['launches_per_step']:.0f} per step, "
            f"{a['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch on average ({a['fetch_bytes_per_launch'] / 1e6:.1f} fetched + "
            f"{a['write_bytes_per_launch'] / 1e6:.1f} written), {a['hbm_gb_per_step']:.1f} GB per step.\n\n")
    f.write("| kernel | launches / step | us (serialised) | fetched MB / launch | written MB / launch | GB / step |\n|---|---|---|---|---|---|\n")
    for e in rows:
        f.write(f"| {e['kernel']} | {e['launches_per_step']} | {e['us']:.1f} | {e['fetch_bytes_per_launch'] / 1e6:.1f} | {e['write_bytes_per_launch'] / 1e6:.1f} | {e['hbm_gb_per_step']:.2f} |\n")
print(json.dumps(out["conv3x3_bf16_fwd_dgrad"]))

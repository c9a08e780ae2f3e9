"""Which kernels of the step are bound by VALU instruction issue?  From tools/pmc_step_insts.sh's per-kernel totals:
VALU issue time = instructions x 4 cycles (a wave64 VALU instruction occupies its SIMD's 16 lanes for 4 cycles) / 1024 SIMDs / clock,
as a fraction of the kernel's (serialised: counter passes run kernels one at a time) duration.   python tools/pmc_step_insts.py file.csv"""
import csv, sys
CLK = 2.1e9
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["total_ns"]) for r in rows)
print(f"{'kernel':70s} {'disp':>5s} {'ms':>8s} {'%':>5s} {'VALU/wave':>9s} {'VALU issue / time':>17s}")
for r in rows[:40]:
    ns, waves, valu = int(r["total_ns"]), max(int(r["waves"]), 1), int(r["valu"])
    frac = valu * 4 / 1024 / CLK / (ns * 1e-9) if ns else 0
    name = r["kernel"]
    name = name[name.index("vqseg::") + 7:] if "vqseg::" in name else name
    print(f"{name[:70]:70s} {int(r['dispatches']):5d} {ns / 1e6:8.2f} {100 * ns / tot:5.1f} {valu / waves:9.0f} {frac:17.2f}")

"""VALU / LDS / SALU instructions per wave of the convolution kernels from a rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
SQ_WAVES pass (tools/pmc_1x1.sh): python tools/pmc_insts.py gpurun_out/r3/pmc_1x1_<tag>.csv"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[(r["Kernel_Name"], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), c in acc.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    w = a.get("SQ_WAVES", 1)
    short = name[name.index("vqseg::") + 7:][:66] if "vqseg::" in name else name[:66]
    print(f"{short:66s} grid {grid:>8s} waves {int(w):6d}  VALU/wave {a.get('SQ_INSTS_VALU', 0) / w:6.0f}  LDS/wave {a.get('SQ_INSTS_LDS', 0) / w:5.0f}  SALU/wave {a.get('SQ_INSTS_SALU', 0) / w:5.0f}")

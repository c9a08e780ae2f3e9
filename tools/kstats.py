"""Print the rows of a rocprofv3 kernel_stats.csv whose kernel name contains one of the given substrings:
   python tools/kstats.py <kernel_stats.csv> <steps> name1 [name2 ...]      (times per step in ms)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
for r in rows:
    if any(k in r["Name"] for k in sys.argv[3:]) or len(sys.argv) == 3:
        print(f"{r['Name'][:96]:96s} calls/step {int(r['Calls']) / steps:7.1f}  ms/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f}  avg us {float(r['AverageNs']) / 1e3:8.1f}")

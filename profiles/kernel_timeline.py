"""Last occurrence of each kernel chain in a rocprofv3 kernel_trace.csv: name, duration, gap to the previous kernel (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
prev = None
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:8.2f} us  gap {gap:7.2f}  {r['Kernel_Name'][:90]}")
    prev = e

"""Developer probe: the last N kernel dispatches of a rocprofv3 kernel trace (csv) as a timeline in µs.   python ktimeline.py DIR [N]"""
import csv
import sys
from pathlib import Path

f = next(Path(sys.argv[1]).rglob("*kernel_trace.csv"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = list(csv.DictReader(open(f)))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(f"{r['Kernel_Name'][:70]:70s} queue {r.get('Queue_Id', '?'):>3s}  {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us")

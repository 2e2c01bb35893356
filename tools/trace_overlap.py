"""Reads a rocprofv3 kernel trace (csv) and prints, for the last N kernel dispatches, start / end relative to the first of them and the
stream / queue - to see whether kernels enqueued on two streams actually ran side by side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} us  q={r.get("Queue_Id", "?")} st={r.get("Stream_Id", "?")} grid={r.get("Grid_Size_X", r.get("Grid_Size", "?"))}  {r["Kernel_Name"][:70]}')

"""Timeline of the last kernels of a rocprofv3 --kernel-trace CSV: start / end in microseconds, hardware queue, stream, kernel, grid.
usage: trace_timeline.py <kernel_trace.csv> [rows from the end] [rows to drop at the end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
drop = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[-n:len(rows) - drop] if drop else rows[-n:]
t0 = int(sel[0]["Start_Timestamp"])
print("start_us    end_us   queue stream kernel (threads)")
for r in sel:
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:58]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}   q{r['Queue_Id']}    s{r['Stream_Id']:<3} {name} ({r['Grid_Size_X']})")

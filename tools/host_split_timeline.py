#!/usr/bin/env python3
"""Print the last call's timeline (kernels and copies, microseconds from its first event) out of a rocprofv3 --kernel-trace --memory-copy-trace -f csv directory."""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gtcrn_chunk" in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel q%s" % r.get("Queue_Id", "?")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
last = ev[-n:]
t0 = last[0][0]
for a, b, what in last:
    print(f"{(a - t0) / 1e3:9.1f} -> {(b - t0) / 1e3:9.1f} us  ({(b - a) / 1e3:7.1f})  {what}")

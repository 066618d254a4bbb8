#!/usr/bin/env python3
"""Timeline of a tools/pipeline_probe.py run traced with rocprofv3 --kernel-trace --memory-copy-trace: per batch the copy-in, kernel and copy-out intervals (us, relative).
    python tools/pipeline_timeline.py <trace dir>"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gtcrn_chunk" in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K"))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = int(r.get("Bytes", r.get("Size", 0)) or 0) if ("Bytes" in r or "Size" in r) else 0
        kind = r.get("Direction", r.get("Kind", ""))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ("H2D" if "HOST_TO_DEVICE" in kind.upper() or "H2D" in kind.upper() else "D2H") + f"({n >> 20}MB)"))
ev.sort()
t0 = ev[-120][0] if len(ev) > 120 else ev[0][0]
for a, b, k in ev[-120:-60] if len(ev) > 120 else ev:
    print(f"{(a - t0) / 1e3:10.1f} {(b - t0) / 1e3:10.1f}  {(b - a) / 1e3:8.1f} us  {k}")

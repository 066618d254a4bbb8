#!/usr/bin/env python3
"""Per-kernel MFMA-busy fraction from a tools/pmc_mfma.sh pass, the gfx94x MfmaUtil formula (ROCm 7.2 has no gfx950 derived-counter
section) with the MI355X's aggregation made explicit: rocprofv3 SUMS a counter over its instances, so SQ_VALU_MFMA_BUSY_CYCLES is the sum
over all CUs x 4 SIMDs and GRBM_GUI_ACTIVE the sum over the 8 XCDs (checked below: GUI_ACTIVE / 8 / kernel duration must come out as
the shader clock, ~2.1 GHz) -> busy fraction = MFMA_BUSY / ((GUI_ACTIVE / 8) * CUs * 4).  Time share from the same pass's trace."""
import collections, csv, glob, sys

d = sys.argv[1]
json_out = None
if "--json" in sys.argv:                       # also write {whole_step_busy_pct, kernels: [{name, time_pct, mfma_busy_pct}]} for bench.py's roofline.mfma_busy
    i = sys.argv.index("--json")
    json_out = sys.argv[i + 1]
    del sys.argv[i:i + 2]
cus = int(sys.argv[2]) if len(sys.argv) > 2 else 256
XCDS = 8
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
for r in csv.DictReader(open(glob.glob(d + "/*/*counter_collection.csv")[0])):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(float)
for r in csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])):
    dur[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
tot = sum(dur.values())
tm = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"] for c in agg.values()); tg = sum(c["GRBM_GUI_ACTIVE"] for c in agg.values())
print(f"{'kernel':70s} {'disp':>5s} {'time%':>6s} {'mfma_busy%':>10s} {'mops_f32/disp':>14s}")
rows = []
for k, c in sorted(agg.items(), key=lambda kv: -dur[kv[0]])[:14]:
    ga = c["GRBM_GUI_ACTIVE"]
    util = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (ga / XCDS * cus * 4) if ga else 0.0
    print(f"{k[:70]:70s} {len(nd[k]):5d} {100 * dur[k] / tot:6.1f} {util:10.1f} {c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) / len(nd[k]):14.3e}")
    rows.append({"kernel": k[:160], "dispatches": len(nd[k]), "time_pct": round(100 * dur[k] / tot, 1), "mfma_busy_pct": round(util, 1)})
if json_out:
    import json
    json.dump({"whole_step_busy_pct": round(100.0 * tm / (tg / XCDS * cus * 4), 1), "clock_check_cycles_per_ns": round(tg / XCDS / tot, 2),
               "method": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ...; busy = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)",
               "kernels": rows}, open(json_out, "w"), indent=1)
print(f"whole run: mfma busy {100.0 * tm / (tg / XCDS * cus * 4):.1f} % of (GUI-active cycles x {cus} CUs x 4 SIMDs); "
      f"clock check: GUI_ACTIVE / {XCDS} / traced kernel time = {tg / XCDS / tot:.2f} cycles per ns")

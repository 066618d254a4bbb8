#!/usr/bin/env python3
"""Pretty-print the `dynamic estimate` CSV block of tools/isa_census.py (stdin) as a fixed-width table."""
import sys
rows = [l.rstrip("\n").split(",") for l in sys.stdin if "," in l and not l.startswith(" ")]
short = {"VALU_non_mfma": "VALU", "fma_packed": "pkfma", "fma_scalar": "fma", "transcendental": "trans", "ieee_div_fixup": "div", "minmax_med3_clamp": "med3",
         "cross_lane_dpp_permlane": "dpp", "readlane_writelane": "rdlane", "select_compare": "select", "address_int": "addr", "vmem_load": "vload", "vmem_store": "vstore",
         "nop_hazard": "s_nop", "salu_other": "salu", "valu_other": "v_oth", "waitcnt": "wait", "barrier": "barr", "convert": "cvt", "branch": "branch"}
keep = [i for i in range(len(rows[0])) if i == 0 or any(r[i] not in ("0", "") for r in rows[1:])]
for r in rows:
    print(" ".join((f"{short.get(r[i], r[i])[:6]:>6s}" if i else f"{r[i]:22s}") for i in keep))

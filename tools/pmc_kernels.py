#!/usr/bin/env python3
"""Per-kernel summary of the passes of tools/pmc_workload.sh: per dispatch (the largest dispatches of each kernel only) stall split, instructions per wave,
matrix-core busy, LDS conflict share, fabric bytes.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half the bytes of wide streaming reads)."""
import collections, csv, glob, re, sys

def short(name):
    name = name.replace('ade::(anonymous namespace)::', '').replace('ade::gemm16::', 'g16::').replace('ade::gemm::', 'g::').replace('void ', '')
    return name.split('(')[0][:64]

def load(d, p):
    files = glob.glob(f'{d}/{p}/*/*counter_collection.csv')
    per = collections.defaultdict(lambda: collections.defaultdict(dict))      # kernel -> dispatch -> counter -> value
    if not files:
        return per
    for r in csv.DictReader(open(files[0])):
        k = short(r['Kernel_Name'])
        dd = per[k][r['Dispatch_Id']]
        dd[r['Counter_Name']] = dd.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        dd['_grid'] = float(r.get('Grid_Size', 0) or 0)
    return per

def big(per_k):
    """mean counters over the dispatches whose grid is the kernel's largest"""
    g = max(v['_grid'] for v in per_k.values())
    sel = [v for v in per_k.values() if v['_grid'] == g]
    out = collections.defaultdict(float)
    for v in sel:
        for c, x in v.items():
            out[c] += x / len(sel)
    out['_n'] = len(sel)
    return out

d = sys.argv[1]
P = {p: load(d, p) for p in ('p1', 'p2', 'p3', 'p4', 'p5')}
print(f"{'kernel':64s} {'n':>3s} {'waves':>8s} {'kcyc/w':>7s} {'wait%':>6s} {'winst%':>6s} {'valu%':>6s} | {'VALU/w':>7s} {'SALU/w':>6s} {'VMRD/w':>6s} {'VMWR/w':>6s} {'LDS/w':>6s} {'MFMA/w':>6s} | {'mfma_busy%':>10s} {'lds_act%':>8s} {'lds_conf%':>9s} | {'fetch_GB':>8s} {'write_GB':>8s}")
for k in sorted(P['p1'], key=lambda k: -big(P['p1'][k]).get('SQ_WAVE_CYCLES', 0)):
    a = big(P['p1'][k])
    if not a.get('SQ_WAVES'):
        continue
    b = big(P['p2'][k]) if k in P['p2'] else {}
    c = big(P['p3'][k]) if k in P['p3'] else {}
    f = big(P['p4'][k]) if k in P['p4'] else {}
    w = big(P['p5'][k]) if k in P['p5'] else {}
    wv, wc = a['SQ_WAVES'], a['SQ_WAVE_CYCLES']
    gui = c.get('GRBM_GUI_ACTIVE', 0)
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SQs' SIMDs as exposed by the 8 XCD aggregates: busy% = busy / (GUI_ACTIVE x 256 CUs x 4 SIMDs / 8 XCD-sums) as tools/pmc_mfma_summary.py derives it
    busy = 100.0 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 4 * 256 / 8) if gui else float('nan')      # gui sums the 8 XCDs' counters
    ldsa = 100.0 * c.get('SQ_LDS_IDX_ACTIVE', 0) / (gui * 256 / 8) if gui else float('nan')
    conf = 100.0 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1)
    print(f"{k:64s} {int(a['_n']):3d} {wv:8.0f} {4*wc/wv/1e3:7.1f} {100*a['SQ_WAIT_ANY']/wc:6.1f} {100*a['SQ_WAIT_INST_ANY']/wc:6.1f} {100*a['SQ_ACTIVE_INST_VALU']/wc:6.1f} | "
          f"{b.get('SQ_INSTS_VALU',0)/wv:7.0f} {b.get('SQ_INSTS_SALU',0)/wv:6.0f} {b.get('SQ_INSTS_VMEM_RD',0)/wv:6.0f} {b.get('SQ_INSTS_VMEM_WR',0)/wv:6.0f} {b.get('SQ_INSTS_LDS',0)/wv:6.0f} {b.get('SQ_INSTS_MFMA',0)/wv:6.0f} | "
          f"{busy:10.1f} {ldsa:8.1f} {conf:9.1f} | {2*f.get('FETCH_SIZE',0)*1024/1e9:8.2f} {w.get('WRITE_SIZE',0)*1024/1e9:8.2f}")

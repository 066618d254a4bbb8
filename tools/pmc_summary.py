#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (tools/pmc_pass.sh) per kernel: stall split and instruction mix per wave."""
import collections, csv, glob, re, sys

def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
        name = m.group(1) if m else r['Kernel_Name'][:20]
        agg[name][r['Counter_Name']] += float(r['Counter_Value']); seen[name].add(r['Dispatch_Id'])
    return agg, {k: len(v) for k, v in seen.items()}

d = sys.argv[1]
a1, n1 = load(glob.glob(d + '/p1/*/*counter_collection.csv')[0])
a2, n2 = load(glob.glob(d + '/p2/*/*counter_collection.csv')[0])
print(f"{'kernel':14s} {'disp':>4s} {'waves':>7s} {'cyc/wave':>8s} {'busy_cyc':>8s} {'wait_any%':>9s} {'wait_inst%':>10s} {'act%':>5s} {'valu%':>5s} | {'VALU/w':>6s} {'SALU/w':>6s} {'SMEM/w':>6s} {'VMRD/w':>6s} {'VMWR/w':>6s} {'LDS/w':>5s}")
for k in a1:
    c = a1[k]; n = n1[k]
    if not c['SQ_WAVES']: continue
    d2 = a2.get(k, {}); wv = c['SQ_WAVES'] / n * n2.get(k, n)
    wc = c['SQ_WAVE_CYCLES']
    print(f"{k[:14]:14s} {n:4d} {c['SQ_WAVES']/n:7.0f} {4*wc/c['SQ_WAVES']:8.0f} {c['SQ_BUSY_CYCLES']/n:8.0f} {100*c['SQ_WAIT_ANY']/wc:9.1f} "
          f"{100*c['SQ_WAIT_INST_ANY']/wc:10.1f} {100*c['SQ_ACTIVE_INST_ANY']/wc:5.1f} {100*c['SQ_ACTIVE_INST_VALU']/wc:5.1f} | "
          f"{d2.get('SQ_INSTS_VALU',0)/wv:6.0f} {d2.get('SQ_INSTS_SALU',0)/wv:6.0f} {d2.get('SQ_INSTS_SMEM',0)/wv:6.0f} "
          f"{d2.get('SQ_INSTS_VMEM_RD',0)/wv:6.0f} {d2.get('SQ_INSTS_VMEM_WR',0)/wv:6.0f} {d2.get('SQ_INSTS_LDS',0)/wv:5.0f}")

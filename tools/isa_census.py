#!/usr/bin/env python3
"""Instruction-class census of a gfx950 kernel from `hipcc -S` output, split at its workgroup barriers.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -I include -S --cuda-device-only audio_denoiser_onnx_amd/csrc/ade_fused.hip -o /tmp/fused.s
    python tools/isa_census.py /tmp/fused.s 'k_gtcrn_chunkINS_5stage3GeoILi256ELi16ELi4EEELb0' [--regions] [--weights tools/isa_census_gtcrn.json]

Static counts per region (a region = the code between two s_barrier instructions, in program order; loops are reported with their label span).  With --weights, a JSON
file {"regions": {"<index>": {"name": ..., "trips": n}}, ...} turns them into dynamic per-wavefront estimates (straight-line code x 1, loop bodies x their trip count)."""
import collections
import json
import re
import sys

CLASSES = [
    ("mfma", r"^v_mfma"),
    ("fma_packed", r"^v_pk_(fma|mul|add)_f32"),
    ("fma_scalar", r"^v_(fma|fmac|mul|add|sub|mac|subrev)_f32"),
    ("transcendental", r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_"),
    ("ieee_div_fixup", r"^v_div_(scale|fmas|fixup)"),
    ("minmax_med3_clamp", r"^v_(med3|max|min|max3|min3)_(f32|i32|u32)"),
    ("cross_lane_dpp_permlane", r"(_dpp$|^v_permlane|^ds_bpermute|^ds_swizzle|^v_mov_b32_dpp)"),
    ("readlane_writelane", r"^v_(readlane|writelane|readfirstlane)"),
    ("select_compare", r"^v_(cndmask|cmp|cmpx)"),
    ("address_int", r"^v_(add|sub|subrev|lshl|lshr|ashr|and|or|xor|mad|mul_lo|mul_hi|bfe|add3|lshl_add|lshl_or|and_or|or3|mul_u32|mul_i32|addc|add_co|subb|not|bfi|alignbit|lshlrev|lshrrev|ashrrev|mad_u64|mad_i64)_?"),
    ("mov", r"^v_(mov|accvgpr|swap)"),
    ("convert", r"^v_cvt"),
    ("lds", r"^ds_"),
    ("vmem_load", r"^(global|buffer|flat|scratch)_load"),
    ("vmem_store", r"^(global|buffer|flat|scratch)_(store|atomic)"),
    ("smem", r"^s_(load|buffer_load)"),
    ("waitcnt", r"^s_waitcnt"),
    ("barrier", r"^s_barrier"),
    ("nop_hazard", r"^s_nop"),
    ("branch", r"^s_(cbranch|branch|setpc|call)"),
    ("salu_other", r"^s_"),
    ("valu_other", r"^v_"),
]
VALU = {"fma_packed", "fma_scalar", "transcendental", "ieee_div_fixup", "minmax_med3_clamp", "cross_lane_dpp_permlane", "readlane_writelane", "select_compare", "address_int", "mov",
        "convert", "valu_other"}


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "other"


def kernel_body(path, needle):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(needle) + r"\S*:", l)]
    if not starts:
        raise SystemExit(f"no function matching {needle!r}")
    i0 = starts[0]
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[i0 + 1:i1]


def parse(body):
    """-> list of (kind, text, chain): kind in {'label', 'op'}; chain = tuple of the loop headers (outermost first) around the instruction, read from the
    asm printer's own loop annotations on the basic-block labels."""
    out = []
    chains = {}            # block label -> chain
    cur_chain = ()
    i = 0
    while i < len(body):
        l = body[i]
        code = l.split(";")[0].strip()
        if code.endswith(":") and code.startswith(".LBB"):
            name = code[:-1].lstrip(".L")
            notes = [l.split(";", 1)[1] if ";" in l else ""]
            j = i + 1
            while j < len(body) and body[j].strip().startswith(";") and not body[j].split(";")[0].strip():
                notes.append(body[j].split(";", 1)[1])
                j += 1
            parents, header_of, is_header = [], None, False
            for n in notes:
                m = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", n)
                if m:
                    parents.append(m.group(1))
                m = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", n)
                if m:
                    header_of = m.group(1)
                if "Loop Header" in n:
                    is_header = True
            if is_header:
                cur_chain = tuple(parents) + (name,)
            elif header_of:
                cur_chain = chains.get(header_of, (header_of,))
            else:
                cur_chain = ()
            chains[name] = cur_chain
            out.append(("label", code[:-1], cur_chain))
            i = j
            continue
        if code and not code.startswith((".", "//")):
            out.append(("op", code, cur_chain))
        i += 1
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path, needle = args[0], args[1]
    weights = None
    if "--weights" in sys.argv:
        weights = json.load(open(sys.argv[sys.argv.index("--weights") + 1]))
        args = [a for a in args if a != sys.argv[sys.argv.index("--weights") + 1]]
    items = parse(kernel_body(path, needle))
    names = [n for n, _ in CLASSES] + ["other"]
    # regions by barrier
    regions, cur = [], []
    for k, t, chain in items:
        if k != "op":
            continue
        cur.append((t.split()[0], chain))
        if t.startswith("s_barrier"):
            regions.append(cur)
            cur = []
    regions.append(cur)
    total = collections.Counter()
    for ops in regions:
        total.update(classify(o) for o, _ in ops)
    if "--regions" in sys.argv:
        for r, ops in enumerate(regions):
            c = collections.Counter(classify(o) for o, _ in ops)
            loops = collections.OrderedDict()
            for o, chain in ops:
                if chain:
                    loops.setdefault(chain, collections.Counter())[classify(o)] += 1
            print(f"region {r:3d}  {len(ops):5d} ops  valu {sum(c[k] for k in VALU):5d}  mfma {c['mfma']:3d}  pkfma {c['fma_packed']:4d}  exp/rcp {c['transcendental']:3d}  lds {c['lds']:3d}  "
                  f"vmem {c['vmem_load']:3d}/{c['vmem_store']:3d}  rdlane {c['readlane_writelane']:3d}")
            for chain, lc in loops.items():
                print(f"            loop {'>'.join(chain):40s} {sum(lc.values()):5d} ops  valu {sum(lc[k] for k in VALU):5d}  pkfma {lc['fma_packed']:4d}  exp/rcp {lc['transcendental']:3d}")
    print("static total:", sum(total.values()), " VALU (non-MFMA):", sum(total[k] for k in VALU))
    for k in names:
        if total[k]:
            print(f"  {k:26s} {total[k]:6d}")
    if weights:
        # A loop header that spans several regions is an OUTER loop (the tile loop of the front / back stage, the three-block loops of the network): its trip count is the
        # region's "mult".  The loops inside ONE region are numbered in order of appearance; "trips"[i] is the trip count of the i-th, "share" the fraction of the workgroup's
        # wavefronts that execute the region at all (1.0 unless a phase runs on some wavefronts only).
        span = collections.defaultdict(set)
        for r, ops in enumerate(regions):
            for _, chain in ops:
                for h in chain:
                    span[h].add(r)
        phases = collections.OrderedDict()
        for r, ops in enumerate(regions):
            spec = weights["regions"].get(str(r))
            if spec is None:
                raise SystemExit(f"region {r} has no entry in the weights file")
            ph = phases.setdefault(spec["name"], collections.Counter())
            order = []
            for _, chain in ops:
                for h in chain:
                    if len(span[h]) == 1 and h not in order:
                        order.append(h)
            trips = spec.get("trips", [])
            for o, chain in ops:
                w = float(spec.get("share", 1.0)) * float(spec.get("mult", 1))
                for h in chain:
                    if h in order:
                        i = order.index(h)
                        w *= trips[i] if i < len(trips) else 1
                ph[classify(o)] += w
        dyn_total = collections.Counter()
        print("\ndynamic estimate per wavefront (phase x class):")
        print(",".join(["phase", "VALU_non_mfma"] + names))
        for ph, c in phases.items():
            print(",".join([ph, str(int(round(sum(c[k] for k in VALU))))] + [str(int(round(c[k]))) for k in names]))
            dyn_total.update(c)
        print(",".join(["TOTAL", str(int(round(sum(dyn_total[k] for k in VALU))))] + [str(int(round(dyn_total[k]))) for k in names]))


if __name__ == "__main__":
    main()

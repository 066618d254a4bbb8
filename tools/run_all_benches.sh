#!/bin/bash
# Every throughput figure quoted in README.md / DESIGN.md, re-measured in one go on one MI355X.  Usage: tools/run_all_benches.sh <out.txt>
OUT=${1:-gpurun_out/all_benches.txt}
mkdir -p "$(dirname "$OUT")"
{
  echo "== bench.py (north star: GTCRN 256 x 1 s)"; python bench.py 2>/dev/null | tail -1
  echo "== tools/bench_host_path.py"; python tools/bench_host_path.py 2>/dev/null | tail -2
  echo "== tools/bench_streaming.py"; python tools/bench_streaming.py 2>/dev/null | tail -6
  echo "== tools/bench_dfsmn.py"; python tools/bench_dfsmn.py 2>/dev/null | tail -4
  echo "== tools/bench_ulunas.py"; python tools/bench_ulunas.py 2>/dev/null | tail -3
  echo "== tools/bench_hgtcrn.py"; python tools/bench_hgtcrn.py --batches 1,64,256 2>/dev/null | tail -3
  echo "== tools/bench_melband.py"; python tools/bench_melband.py 2>/dev/null | tail -6
  echo "== tools/bench_mossformer.py"; python tools/bench_mossformer.py 2>/dev/null | tail -6
} > "$OUT" 2>&1

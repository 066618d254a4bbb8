#!/usr/bin/env python3
"""bench.py — RTF / audio-seconds-per-second of the chunk path; default = GTCRN at batch = 256 x 1 s chunks (BASELINE.json configs[1]).

`--workload zipenhancer | melband | mossformer` selects the other BASELINE configs (configs[2] / [3] / [4]) at their own batch shapes as the headline;
the default invocation's line also carries them, briefly timed, in `other_workloads` (so that the driver's clock covers every BASELINE config; `--other`, `--other-steps`).

A "step" is one pass of the hot path (int16 PCM in HBM -> STFT -> GTCRN -> mask -> ISTFT/OLA -> int16 PCM in HBM)
over one batch of 256 synthetic 1 s chunks per GPU (`configs[1]`), through libade's C ABI on device buffers.
Weights: seeded reference-architecture weights (tests/golden/gtcrn_seed0.adew; the reference ships no checkpoint).

    python bench.py --gpus 1 --steps 100 --warmup 10
    python bench.py --gpus N ...                      # N > 1 outside a launcher: re-executes itself under torch.distributed.run with N ranks
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 8 --workload melband --dtype bf16 --stitch      # BASELINE configs[3] with its RCCL all-gather of the outputs, one command

`--gpus N` MEANS N ranks: the line is refused (non-zero exit, nothing printed) when the launcher's WORLD_SIZE differs from it.

Multi-GPU: chunks are independent reference calls, so each rank runs its own 256 chunks (weak scaling) with NO
collective in the data path; `--stitch` additionally all-gathers the int16 outputs over RCCL inside the timed region
(what a file-level job needs to write one wav).  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

CHUNK = 16000
SR = 16000

# Algorithmic work per 1 s chunk (T = 63 frames) of each kernel family, per LAUNCH-SET in one forward:
# MACs counted from the layer shapes (DESIGN.md "Kernels"); bytes = activation tensors the op must read + write once.
_T = 63
KERNEL_MODEL = {
    # name: (MACs per chunk per forward, algorithmic bytes per chunk per forward)
    "pcm_mean": (0, 32000),
    "stft_feat": (int(_T * (2.5 * 512 * 9 / 2 + 3 * 382)), 32000 + _T * (2 * 257 + 3 * 129) * 4),
    "conv0": (_T * 65 * 16 * 45, _T * (3 * 129 + 65 * 16) * 4),
    "conv1": (_T * 33 * 16 * 40, _T * (65 * 16 + 33 * 16) * 4),
    "gt_pw1": (6 * _T * 33 * 384, 6 * _T * 33 * (8 + 16) * 4),
    "gt_dw_pw2": (6 * _T * 33 * 272, 6 * _T * 33 * (16 + 8 + 16) * 4),
    "tra_gru": (6 * _T * 1280, 6 * _T * 16 * 4),
    "intra_gru": (2 * _T * 33 * 4 * 144, 2 * _T * 33 * 32 * 4),
    "inter_gru": (2 * _T * 33 * 2 * 384, 2 * _T * 33 * 32 * 4),
    "fc_ln_res": (4 * _T * 33 * 256, 4 * _T * 33 * 48 * 4),
    "deconv3": (_T * 33 * 16 * 8 * 5, _T * (2 * 33 * 16 + 65 * 16) * 4),
    "deconv4": (_T * 65 * 16 * 2 * 5, _T * (2 * 65 * 16 + 2 * 129) * 4),
    "istft_mask": (int(_T * (2.5 * 512 * 9 / 2 + 2 * 382 + 4 * 257)), _T * (2 * 257 + 2 * 129 + 512) * 4),
    "ola_pcm": (0, _T * 512 * 4 + 31744),
    # fused per-chunk stage kernels (ade_fused.hip): a GTConvBlock = pw1 + dw/pw2 + TRA ; a DPGRNN = 2 GRNN + 2 (FC+LN)
    "gtblock": (6 * _T * (33 * (384 + 272) + 1280), (3 * 2 + 3 * 3) * _T * 33 * 16 * 4),
    "dpgrnn": (2 * _T * 33 * (4 * 144 + 2 * 384 + 2 * 256), 2 * 2 * _T * 33 * 16 * 4),
    # front = mean + STFT/feat + conv0 + conv1 ; back = deconv3 + deconv4 + mask/irFFT/OLA/PCM   (ade_stage_frontback.h)
    "front": (int(_T * (2.5 * 512 * 9 / 2 + 3 * 382)) + _T * 65 * 16 * 45 + _T * 33 * 16 * 40,
              32000 + _T * (2 * 257 + 65 * 16 + 33 * 16) * 4),
    "back": (_T * 33 * 16 * 8 * 5 + _T * 65 * 16 * 2 * 5 + int(_T * (2.5 * 512 * 9 / 2 + 2 * 382 + 4 * 257)),
             _T * (2 * 33 * 16 + 65 * 16 + 2 * 257) * 4 + 31744),
    # the whole chunk path as ONE kernel: SURVEY.md 8(d3) per-chunk figures — 54.5 MFLOP (= 27.25 MMAC-equivalents) and
    # 63 744 B (int16 in + int16 out; every intermediate is LDS / L2-resident scratch, not algorithmic traffic)
    "gtcrn_chunk": (27_250_000, 32000 + 31744),
}
PIPELINE_BYTES_PER_CHUNK = 32000 + 31744          # int16 in + int16 out
PIPELINE_FLOP_PER_CHUNK = 54.5e6                   # 2 x 26.52 MMAC network + FFT-form STFT/ISTFT
HBM_PEAK_GBS = 8000.0                              # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3                           # fp32 dense: f32-MFMA rate == fp32 VALU rate on gfx950
BF16_PEAK_TFLOPS = 2500.0                          # dense bf16 MFMA (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="gtcrn", choices=["gtcrn", "zipenhancer", "melband", "mossformer", "dfsmn"],
                    help="BASELINE.json config: gtcrn = configs[1] (default), zipenhancer = [2], melband = [3], mossformer = [4]; dfsmn = north_star's fourth network "
                         "(256 x 2 s @ 48 kHz, no BASELINE config of its own)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = exact fp32 matrix-core products (parity path, default); bf16 = bf16 activations and weights stored in HBM (--workload melband only: "
                         "csrc/ade_gemm16.h).  The deviation from the f32 path is measured and reported")
    ap.add_argument("--batch", type=int, default=0, help="chunks per GPU per step (0 = the workload's BASELINE batch: 256 / 128 / 32 / 64)")
    ap.add_argument("--host-steps", type=int, default=20, help="steps of the host-inclusive leg (pinned host buffers through ade_process; 0 = skip)")
    ap.add_argument("--stitch", action="store_true", help="all-gather the int16 outputs (RCCL) inside the timed region")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): every rank runs its own --batch chunks; strong: ONE batch of --batch chunks (default: the workload's BASELINE batch) is dealt over the "
                         "ranks in contiguous blocks of ceil(B / N), the way the file drivers shard a file's slices (distributed.shard_bounds; trailing ranks may be idle)")
    ap.add_argument("--other-cpu-seconds", type=float, default=60.0, help="budget of the CPU-oracle legs of the `other_workloads` block (0 = skip them)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle baseline leg (0 = skip)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-deviation", action="store_true", help="skip the bf16-vs-f32 deviation run of a --dtype bf16 line (profiling runs)")
    ap.add_argument("--geometry", default="auto", choices=["auto", "0", "1", "2"],
                    help="GTCRN fused-path workgroup geometry: 2 = four 256-thread workgroups per CU, each a 16-frame segment of a chunk (default where it fits); "
                         "1 = two 512-thread workgroups per CU (32-frame segments); 0 = one 1024-thread workgroup per chunk (the round-1/2 kernel)")
    ap.add_argument("--other", default="zipenhancer,melband,mossformer,dfsmn", help="which of the other BASELINE configs the default line times in its `other_workloads` block")
    ap.add_argument("--other-steps", type=int, default=3, help="timed steps of the `other_workloads` leg of the default line (ZipEnhancer 128 x 1 s, f32; 0 = skip)")
    ap.add_argument("--ramp-ms", type=float, default=100.0, help="untimed power-state ramp before the W warm-up steps (0 = none)")
    ap.add_argument("--no-stft-operator", action="store_true", help="skip the `stft_operator` block (the generic STFT_Process operator's HBM roofline) of the default line")
    return ap.parse_args()


# The translation units and headers k_gtcrn_chunk and the per-stage kernels are compiled from; the other families' sources do not change those kernels.
GTCRN_PATH_SOURCES = ("ade_device.h", "ade_engine.hip", "ade_fft.h", "ade_fused.hip", "ade_gtcrn_pack.h", "ade_internal.h", "ade_kernels.hip", "ade_stage_frontback.h",
                      "ade_stage_net.h")


def source_sha1() -> str:
    """Identity of the kernel sources the GTCRN traffic measurement belongs to (GTCRN_PATH_SOURCES under csrc/)."""
    import hashlib
    h = hashlib.sha1()
    for name in GTCRN_PATH_SOURCES:
        with open(os.path.join(REPO, "audio_denoiser_onnx_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def workload_traffic(name: str, dtype: str, B: int, default_B: int):
    """Fabric-side bytes of one step of a GEMM-family workload from the committed FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload <name>` (tools/pmc_traffic_workload.sh);
    the newest round's file wins.  -> (bytes per step scaled to B rows, note) or (None, None)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        try:
            with open(os.path.join(REPO, "profiles", f"{rnd}_{name}_{dtype}_traffic.json")) as f:
                tp = json.load(f)
            return int(tp["bytes_per_step"] * B / default_B), (f"profiles/{rnd}_{name}_{dtype}_traffic.json: bytes per step = (2*FETCH_SIZE + WRITE_SIZE)*1024 summed over every kernel "
                                                                "of a step, separate rocprofv3 --pmc passes; fabric-side incl. Infinity-Cache hits")
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def kernel_family_roofline(name: str, dtype: str, B: int, frames: int):
    """Per-kernel roofline of a workload's DOMINANT product family (the dominant kernel priced on its own, beside the whole-step figure): the family's
    algorithmic flops per step (from the model's dimensions) / the summed device time of its launches in the committed rocprofv3 --kernel-trace --stats summary of the same
    command (profiles/rNN_z_<workload>_<dtype>_kernel_stats.csv, the newest round's file) / the dtype's dense matrix peak.  -> dict or None (no committed summary)."""
    import csv
    peak = FP32_PEAK_TFLOPS if dtype == "f32" else BF16_PEAK_TFLOPS
    if name == "zipenhancer":
        from audio_denoiser_onnx_amd import zipenhancer as zp
        cfg = zp.ZipConfig()
        C, F0, F = cfg.channels, 201, zp.freq_len()
        dense_macs = sum(6 * C * C * (i + 1) for i in range(cfg.dense_depth)) * frames * (F0 + 2 * F)
        pat, what, flops, default_B = "k_zip_dense", "the causal dense blocks' layers (12 launches per step: 4 encoder + 2 x 4 decoder layers, cin 64 .. 256, implicit GEMM)", 2.0 * dense_macs * B, 128
    elif name == "melband":
        di, rows = 8 * 64, 60 * frames
        fam = 2.0 * rows * 384 * ((3 * di + 8) + di + 2 * 4 * 384) * 2 * 6                        # q|k|v|gates, out, FFN in + out; two transformers per depth, depth 6
        pat = "gemm16::k_gemm16<" if dtype == "bf16" else "gemm::k_gemm128<"
        what, flops, default_B = "the transformers' Linear products (in-projection, out-projection, FFN in / out: 4 launches per transformer, 12 transformers)", fam * B, 32
    elif name == "mossformer":
        pat, what, default_B = "gemm::k_gemm128<", "the FLASH layers' Linear products (non-batched 128 x 128 tiles)", 64
        flops = 24 * 2.0 * frames * (512 * 2176 + 1024 * 512 + 512 * 256 + 256 * 512 + 2 * 256 * 256 + 256 * 512) * B
    else:
        return None
    for rnd in ("r06", "r05", "r04"):
        path = os.path.join(REPO, "profiles", f"{rnd}_z_{name}_{dtype}_kernel_stats.csv")
        if not os.path.exists(path):
            continue
        rows_ = list(csv.DictReader(open(path)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows_)
        fam_rows = [r for r in rows_ if pat in r["Name"]]
        if not fam_rows or tot <= 0:
            continue
        steps_in_trace = 4 if name == "zipenhancer" else 4                                   # bench.py --steps 3 --warmup 1: four steps in the trace
        fam_ns = sum(float(r["TotalDurationNs"]) for r in fam_rows) / steps_in_trace
        tf = flops * (default_B / B) / (fam_ns * 1e-9) / 1e12 if B else 0.0                  # the summary was taken at the workload's BASELINE batch
        return {"kernel_family": pat, "what": what, "launches_per_step": int(sum(int(r["Calls"]) for r in fam_rows) / steps_in_trace), "ms_per_step": round(fam_ns * 1e-6, 3),
                "share_of_step_pct": round(100.0 * fam_ns * steps_in_trace / tot, 1), "algorithmic_tflop_per_step": round(flops * (default_B / B) / 1e12, 3),
                "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 4), "source": f"profiles/{rnd}_z_{name}_{dtype}_kernel_stats.csv",
                "measured": "FROM THE COMMITTED PROFILE named in `source` (rocprofv3 cannot run inside bench.py), not from this run: compare `ms_per_step` here with the line's own"}
    return None


def other_workload_line(name: str, steps: int, local_rank: int, stream, cpu_budget_s: float = 0.0, dtype: str = "f32"):
    """A short timed run of another BASELINE config inside the default invocation, so that the driver's own clock covers it too: `steps` (>= 3) individually timed steps after
    one warm-up step (mean in ms_per_step, spread in ms_min / ms_max), the whole-step roofline with the committed traffic figure, and the workload's CPU-oracle baseline on a
    bounded sample."""
    import torch
    global SKIP_DEVIATION
    skip, SKIP_DEVIATION = SKIP_DEVIATION, True            # (the deviation leg belongs to the --workload line of the dtype)
    try:
        wl = build_workload(name, 0, 0, local_rank, dtype)
    finally:
        SKIP_DEVIATION = skip
    sess, B, x = wl["sess"], wl["B"], wl["x"]
    sess.reserve(B)
    d_in = torch.from_numpy(x).cuda()
    d_out = torch.empty((B, sess.row_out), dtype=torch.int16, device="cuda")
    sess.run_device(d_in, d_out, stream=stream)            # warm-up (graph capture, clocks)
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        sess.run_device(d_in, d_out, stream=stream)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sum(times) / steps
    audio = B * sess.out_len / wl["sr"]
    tf = wl["flop"] * B / dt / 1e12
    traffic, note = workload_traffic(name, dtype, B, B)
    peak = FP32_PEAK_TFLOPS if dtype == "f32" else BF16_PEAK_TFLOPS
    line = {"workload": wl["workload"], "steps": steps, "warmup": 1, "ms_per_step": round(dt * 1e3, 3), "ms_min": round(min(times) * 1e3, 3), "ms_max": round(max(times) * 1e3, 3),
            "value": round(audio / dt, 1), "unit": "audio-s/s",
            "rtf": float(f"{dt / audio:.3e}"), "dtype": dtype, "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s",
                                                                             "frac": round(tf / peak, 4), "traffic": traffic, "traffic_note": note,
                                                                             "algorithmic_bytes_per_step": int(B * (sess.row_in + sess.row_out) * 2)}}
    fam = kernel_family_roofline(name, dtype, B, sess.frames)
    if fam:
        line["roofline"]["dominant_kernel"] = fam
    if wl.get("target_rtf"):
        line["target_rtf"] = wl["target_rtf"]
    del sess, d_in, d_out
    torch.cuda.empty_cache()
    if cpu_budget_s > 0 and wl.get("cpu") is not None:
        try:
            line["cpu_baseline"] = wl["cpu"]()
        except Exception as ex:   # a baseline leg must not take the line down
            line["cpu_baseline"] = {"error": repr(ex)}
    return line


def cpu_model() -> str:
    """The host CPU's model string (/proc/cpuinfo) x sockets, for the cpu_baseline block."""
    try:
        names, phys = [], set()
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    names.append(ln.split(":", 1)[1].strip())
                elif ln.startswith("physical id"):
                    phys.add(ln.split(":", 1)[1].strip())
        return f"{max(1, len(phys))} x {names[0]} ({len(names)} hardware threads)" if names else "unknown"
    except OSError:
        return "unknown"


def cpu_baseline(blob: bytes, x: np.ndarray, budget_s: float):
    """Oracle (C restatement of the reference, OpenMP over chunks) timed on this host: the reported CPU baseline.  Two legs: every core
    (`value`), and ONE thread (`threads_1`: the reference's published numbers are single-stream CPU RTFs)."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle_lib import GtcrnOracle   # test infrastructure used as the *baseline*, never as the product
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    o = GtcrnOracle(blob, CHUNK)
    t0 = time.perf_counter()
    o.process(x[:1], threads=1)                              # single-thread leg: one chunk to size it, then ~3 s worth
    one = time.perf_counter() - t0
    n1 = int(min(32, max(2, 3.0 / max(one, 1e-3))))
    t0 = time.perf_counter()
    o.process(x[np.arange(n1) % x.shape[0]], threads=1)
    dt1 = time.perf_counter() - t0
    threads_1 = {"value": round(n1 * (15872 / SR) / dt1, 3), "unit": "audio-s/s", "cores": 1, "rtf": round(dt1 / (n1 * 15872 / SR), 5),
                 "sample": f"{n1} synthetic 1 s chunks one after the other on ONE thread, {dt1:.1f} s wall"}
    t0 = time.perf_counter()
    o.process(x[:cores], threads=cores)                     # probe: one chunk per core
    probe = time.perf_counter() - t0
    n = int(max(cores, (budget_s / max(probe, 1e-3)) * cores))
    n = min(max(cores, (n // cores) * cores), 4096)
    xs = x[np.arange(n) % x.shape[0]]                        # bounded sample: the same chunks, repeated to fill the budget
    t0 = time.perf_counter()
    o.process(xs, threads=cores)
    dt = time.perf_counter() - t0
    out = {"value": round(n * (15872 / SR) / dt, 2), "unit": "audio-s/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
           "sample": f"{n} of the same synthetic 1 s chunks, oracle/ade_oracle.c (dense-DFT reference arithmetic), "
                     f"OpenMP over chunks, {dt:.1f} s wall", "rtf": round(dt / (n * 15872 / SR), 5), "threads_1": threads_1}
    # the reference's OWN forward in PyTorch eager mode, timed once in the build container (where /root/reference exists) by
    # tools/ref_eager_cpu_timing.py and committed: a recorded figure from another CPU, shown beside the live legs, never mixed with them.
    try:
        with open(os.path.join(REPO, "profiles", "r06_ref_eager_cpu.json")) as f:
            ref = json.load(f)
        out["reference_eager_build_container"] = {"from": "profiles/r06_ref_eager_cpu.json (tools/ref_eager_cpu_timing.py; NOT measured on this host)", "cpu": ref.get("cpu"),
                                                  "hardware_threads": ref.get("hardware_threads"), "torch": ref.get("torch"), "runs": ref.get("runs")}
    except (OSError, ValueError):
        pass
    return out


def cpu_baseline_numpy(make_oracle, x_row: np.ndarray, seconds_per_row: float, what: str, takes_batch: bool = False):
    """The numpy oracle of a GEMM-shaped family timed on ONE row of the workload at the workload's OWN window length (these oracles take 10 - 90 s per row on a host CPU)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    o = make_oracle()
    t0 = time.perf_counter()
    o.process(x_row)
    dt = time.perf_counter() - t0
    return {"value": round(seconds_per_row / dt, 3), "unit": "audio-s/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"{'rows' if takes_batch else '1 row'} of the workload ({what}) through the numpy fp32 oracle (BLAS threads = host default), {dt:.1f} s wall",
            "rtf": round(dt / seconds_per_row, 4)}


def stft_operator_lines(stream):
    """The generic STFT_Process operator (csrc/ade_stft.hip) on its own: the one HBM-BOUND kernel family of the path (section 8 d3).  Analysis
    and synthesis at GTCRN's shape (256 x 1 s, n_fft 512 / hop 256) and Mel-Band's (64 x 1.5 s, n_fft 2048 / hop 441): algorithmic bytes (fp32 samples in or out + the
    packed (B, 2F, T) spectrum out or in, each once) / the average launch duration (HIP events on the launch stream) / 8 TB/s."""
    import torch
    from audio_denoiser_onnx_amd.stft_process import STFT_Process
    out = {}
    for name, n_fft, hop, L, B in (("gtcrn_512_256", 512, 256, 16000, 256), ("melband_2048_441", 2048, 441, 66150, 64)):
        fwd = STFT_Process("stft_B", n_fft, n_fft, hop, 0, "hann", True, "reflect")
        T = fwd.frames(L)
        inv = STFT_Process("istft_B", n_fft, n_fft, hop, T, "hann", True, "reflect")
        x = torch.randn(B, 1, L, device="cuda") * 0.1
        for _ in range(3):
            sp = fwd(x, stream=stream)
            y = inv(sp, stream=stream)
        # the timed launches go straight through the C ABI into preallocated tensors: the Python mirror's per-call work (output allocation, shape checks: ~40 us) is longer
        # than the 512-point kernels and would be what the events measure
        import ctypes as C
        lib = fwd._lib.c
        sp2, y2 = torch.empty_like(sp), torch.empty_like(y)
        n = 50
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(n):
            st = lib.ade_stft_analyze(fwd._h, C.c_void_p(x.data_ptr()), B, L, C.c_void_p(sp2.data_ptr()), C.c_void_p(stream))
        ev[1].record()
        for _ in range(n):
            st |= lib.ade_stft_synthesize(inv._h, C.c_void_p(sp.data_ptr()), B, T, C.c_void_p(y2.data_ptr()), C.c_void_p(stream))
        ev[2].record()
        torch.cuda.synchronize()
        assert st == 0 and torch.equal(sp2, sp) and torch.equal(y2, y)
        ta, ts = ev[0].elapsed_time(ev[1]) / n * 1e-3, ev[1].elapsed_time(ev[2]) / n * 1e-3
        bytes_spec, bytes_a, bytes_s = B * (n_fft + 2) * T * 4, B * L * 4, int(y.numel()) * 4
        err = float((y.reshape(B, -1) - x.reshape(B, -1)[:, :y.numel() // B]).abs().max())
        out[name] = {"batch": B, "samples": L, "frames": T,
                     "analysis": {"us": round(ta * 1e6, 1), "algorithmic_bytes": bytes_a + bytes_spec, "achieved": round((bytes_a + bytes_spec) / ta / 1e9, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round((bytes_a + bytes_spec) / ta / 1e9 / HBM_PEAK_GBS, 4)},
                     "synthesis": {"us": round(ts * 1e6, 1), "algorithmic_bytes": bytes_s + bytes_spec, "achieved": round((bytes_s + bytes_spec) / ts / 1e9, 1), "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": round((bytes_s + bytes_spec) / ts / 1e9 / HBM_PEAK_GBS, 4)},
                     "round_trip_max_abs_err": float(f"{err:.2e}")}
        fwd.close()
        inv.close()
    out["bound"] = "hbm"
    out["note"] = ("GTCRN / ZipEnhancer / Mel-Band / MossFormer2 / DFSMN engines do not call this operator (their front / back stages fuse the transforms with the network); "
                   "H-GTCRN and UL-UNAS do, and it is the drop-in for the reference's STFT_Process module: the only HBM-bound kernel family of the path.  Timed through the C ABI "
                   "into preallocated tensors; kernel durations of the same shapes: profiles/r06_l_stft_kernel_us.txt")
    return out


SKIP_DEVIATION = False
DEVIATION_DTYPE = "bf16"


def deviation_from_f32(make_session, x, rows: int = 2):
    """What the reduced-precision GEMM inputs cost: the same rows through the exact (f32) path and the selected one."""
    if SKIP_DEVIATION:
        return None
    with make_session("f32") as ref:
        want, wf = ref.process(x[:rows], want_f32=True)
    with make_session(DEVIATION_DTYPE) as low:
        got, gf = low.process(x[:rows], want_f32=True)
    err = gf.astype(np.float64) - wf.astype(np.float64)
    sig = float((wf.astype(np.float64) ** 2).mean())
    return {"vs": f"the f32 path on {rows} rows of the workload", "max_abs": round(float(np.abs(err).max()), 6), "rms": round(float(np.sqrt((err ** 2).mean())), 6),
            "signal_rms": round(sig ** 0.5, 6), "units": "the engine's fp32 pre-PCM waveform", "snr_db": round(float(10 * np.log10(sig / max((err ** 2).mean(), 1e-30))), 1),
            "max_pcm_lsb": int(np.abs(got.astype(np.int32) - want.astype(np.int32)).max())}


def build_workload(name: str, batch: int, rank: int, local_rank: int, dtype: str = "f32"):
    """-> dict(sess, B, x_host (B, row_in) int16, sr, out_seconds_per_row, flop_per_row, metric, workload, weights, cpu)"""
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.synth import synth_batch, synth_chunk
    from audio_denoiser_onnx_amd.weights import pack_blob
    if name == "zipenhancer":                                      # BASELINE configs[2]: 128 x 1 s chunks
        from audio_denoiser_onnx_amd import zipenhancer as zp
        cfg, B = zp.ZipConfig(), batch or 128
        tensors = zp.synthetic_tensors(cfg)
        x = synth_batch(B, CHUNK, first_index=rank * B)
        blob = pack_blob(tensors)
        deviation = deviation_from_f32(lambda dt: InferenceSession(weights=blob, metadata=zp.metadata(CHUNK, gemm_dtype=dt), device_id=local_rank), x, 4) if dtype != "f32" else None

        sess = InferenceSession(weights=blob, metadata=zp.metadata(CHUNK, gemm_dtype=dtype), device_id=local_rank)

        def cpu():
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            from zipenhancer_oracle import ZipEnhancerOracle
            return cpu_baseline_numpy(lambda: ZipEnhancerOracle(tensors, CHUNK), x[:1], 1.0, "one 1 s chunk, 161 frames x 101 sub-bands")
        return dict(sess=sess, B=B, x=x, sr=16000, flop=2.0 * zp.macs_per_window(sess.frames, cfg)["total"], cpu=cpu,
                    metric="audio_seconds_per_second (ZipEnhancer 16 kHz, batch=128 x 1 s chunks; RTF = 1/value)",
                    workload="ZipEnhancer 16 kHz, batch=128 x 1 s chunks (161 frames x 101 sub-bands), int16 PCM in/out resident in HBM, " +
                             ("bf16 weights and activations stored in HBM (BASELINE.json configs[2]'s dtype; csrc/ade_zip16.h: v_mfma_f32_32x32x16_bf16), fp32 residual stream / "
                              "InstanceNorm statistics / softmax / STFT front / PCM tail" if dtype == "bf16" else "fp32 matrix cores (the parity dtype; --dtype bf16 is configs[2]'s dtype)"),
                    weights="random-init weights of the architecture (zipenhancer.synthetic_tensors, 2.1 M parameters; no checkpoint is available offline)",
                    target_rtf=0.01, deviation=deviation)
    if name == "melband":                                          # BASELINE configs[3]: 32 x 8 s stereo segments @ 44.1 kHz
        from audio_denoiser_onnx_amd import melband, weightgen
        B, L, depth = batch or 32, 352800, 6
        w = weightgen.materialise(melband.synthetic_spec(depth))
        blob = pack_blob(melband.model_tensors(w))
        del w
        from audio_denoiser_onnx_amd.synth import synth_stereo
        x = np.stack([synth_stereo(rank * B + i, L, 44100).reshape(-1) for i in range(B)])
        deviation = deviation_from_f32(lambda dt: InferenceSession(weights=blob, metadata=melband.metadata(L, gemm_dtype=dt), device_id=local_rank), x, 1) if dtype != "f32" else None
        sess = InferenceSession(weights=blob, metadata=melband.metadata(L, gemm_dtype=dtype), device_id=local_rank)
        del blob

        def cpu():          # ONE row of the workload at its own length (8 s stereo, 801 frames, depth 6): 60 - 120 s of numpy on the host
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            from melband_oracle import MelBandOracle
            from audio_denoiser_onnx_amd import mel_bands
            wts = weightgen.materialise(melband.synthetic_spec(depth))
            freq_indices, dim_inputs = mel_bands.band_tables()[:2]
            return cpu_baseline_numpy(lambda: MelBandOracle(wts, freq_indices, dim_inputs, L // 441 + 1, depth), synth_stereo(0, L, 44100), L / 44100.0,
                                      "one 8 s stereo segment, 801 frames x 60 bands, depth 6 -- the workload's own row")
        return dict(sess=sess, B=B, x=x, sr=44100, flop=melband.flops_per_clip(sess.frames, depth), cpu=cpu, deviation=deviation,
                    metric="audio_seconds_per_second (Mel-Band-Roformer stereo 44.1 kHz, batch=32 x 8 s segments; RTF = 1/value)",
                    workload="Mel-Band-Roformer stereo 44.1 kHz, depth 6, batch=32 x 8 s segments (801 frames), " +
                             ("bf16 activations and weights stored in HBM, v_mfma_f32_32x32x16_bf16 / 16x16x32 (csrc/ade_gemm16.h), fp32 residual stream / norms / softmax "
                              "statistics / STFT / band split / mask / ISTFT" if dtype == "bf16" else "fp32 matrix cores (the parity dtype; --dtype bf16 is the configs[3] dtype)") +
                             ", int16 PCM resident in HBM (BASELINE.json configs[3])",
                    weights="random-init weights of the architecture (melband.synthetic_spec, depth 6)", target_rtf=None)
    if name == "mossformer":                                       # BASELINE configs[4]: 64 x 4 s
        from audio_denoiser_onnx_amd import mossformer
        B, L, layers = batch or 64, 64000, 24
        frames = mossformer.frames_of(L)
        fused = {n: mossformer.synthetic_tensor(n, sh, sc, frames) for n, sh, sc in mossformer.synthetic_spec(layers)}
        scalars = dict(mossformer.DEFAULT_SCALARS, fs_front_alpha=[0.25] * layers)
        blob = pack_blob(mossformer.model_tensors(fused, scalars, L))
        del fused
        x = np.stack([(synth_chunk(rank * B + i, L).astype(np.int32) + synth_chunk(10000 + rank * B + i, L)).clip(-32768, 32767).astype(np.int16) for i in range(B)])
        deviation = deviation_from_f32(lambda dt: InferenceSession(weights=blob, metadata=mossformer.metadata(L, gemm_dtype=dt), device_id=local_rank), x, 1) if dtype != "f32" else None
        sess = InferenceSession(weights=blob, metadata=mossformer.metadata(L, gemm_dtype=dtype), device_id=local_rank)
        del blob

        def cpu():          # ONE row of the workload at its own length (4 s, 7999 frames, 24 layers): 30 - 60 s of numpy on the host
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            from mossformer_oracle import MossFormerOracle
            tensors = {n: mossformer.synthetic_tensor(n, sh, sc, frames) for n, sh, sc in mossformer.synthetic_spec(layers)}
            tensors.update(mossformer.position_tables(frames, int(scalars["rot_dim"])))
            xs = (synth_chunk(0, L).astype(np.int32) + synth_chunk(10000, L)).clip(-32768, 32767).astype(np.int16)[None]
            return cpu_baseline_numpy(lambda: MossFormerOracle(tensors, scalars, layers, L), xs, L / 16000.0, "one 4 s window, 7999 frames, 24 layers -- the workload's own row")
        return dict(sess=sess, B=B, x=x, sr=16000, flop=mossformer.flops_per_window(sess.frames, layers), cpu=cpu, deviation=deviation,
                    metric="audio_seconds_per_second (MossFormer2-SS-16K, batch=64 x 4 s; RTF = 1/value)",
                    workload="MossFormer2-SS-16K two-speaker separation, 24 layers, batch=64 x 4 s (7999 frames), fp32 matrix cores, int16 PCM resident in HBM "
                             "(BASELINE.json configs[4])",
                    weights="random-init weights of the published geometry (mossformer.synthetic_spec, 24 layers)", target_rtf=None)
    if name == "dfsmn":                                            # north_star's DFSMN: no BASELINE config of its own; 256 x 2 s @ 48 kHz
        B, L = batch or 256, 96000
        with open(os.path.join(REPO, "tests", "golden", "dfsmn_seed0.adew"), "rb") as f:
            blob = f.read()
        meta = build_audio_metadata_dfsmn(L)
        rng = np.random.default_rng(1000 + rank)
        x = (rng.standard_normal((B, L)) * 1500).astype(np.int16)
        sess = InferenceSession(weights=blob, metadata=meta, device_id=local_rank)
        frames = sess.frames
        macs_frame = 120 * 1025 + 256 * 120 + 9 * 2 * 256 * 256 + 961 * 256          # mel bank + Linear(120,256) + 9 x (Linear, Linear) + Linear(256,961) (Export_DFSMN.py:216-230)
        fft_flop_frame = 5.0 * (2048 * 11 + 2 * 1920 * 10.9) / 2                        # two real forward transforms riding one complex FFT per frame pair + one inverse
        # (the nine layers' depthwise memories, lorder 20: 9 x 256 x 20 MACs per frame)
        flop_row = frames * (2.0 * (macs_frame + 9 * 256 * 20) + fft_flop_frame)

        def cpu():
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            from dfsmn_oracle import DfsmnOracle
            from audio_denoiser_onnx_amd.weights import load_blob
            tensors = load_blob(os.path.join(REPO, "tests", "golden", "dfsmn_seed0.adew"))
            return cpu_baseline_numpy(lambda: DfsmnOracle(tensors, L), x[:min(B, 256)], min(B, 256) * L / 48000.0, f"{min(B, 256)} rows of 2 s @ 48 kHz, 99 frames each", takes_batch=True)
        return dict(sess=sess, B=B, x=x, sr=48000, flop=flop_row, cpu=cpu, deviation=None,
                    metric="audio_seconds_per_second (DFSMN 48 kHz, batch=256 x 2 s chunks; RTF = 1/value)",
                    workload="DFSMN 48 kHz acoustic noise suppression (north_star's fourth network; no BASELINE config of its own), batch=256 x 2 s chunks (99 frames of 1920 / hop 960), "
                             "fp32: FFT analysis / synthesis + the mask network on fp32 matrix cores, int16 PCM resident in HBM",
                    weights="seeded reference-architecture DFSMN (tests/golden/dfsmn_seed0.adew; the reference takes its parameters from modelscope, absent offline)", target_rtf=None)
    raise SystemExit(f"unknown workload {name}")


def build_audio_metadata_dfsmn(length: int):
    from audio_denoiser_onnx_amd.metadata import build_audio_metadata
    return build_audio_metadata(producer="bench.py", model_name="DFSMN", task="denoise", model_family="dfsmn", input_audio_length=length, in_sample_rate=48000,
                                out_sample_rate=48000, model_sample_rate=48000, nfft=1920, window_length=1920, hop_length=960, window_type="hamming", center_pad=False,
                                pad_mode="constant", feature_kind="kaldi_fbank_stft")


def relaunch_command(n: int, argv, port: int):
    """The launcher line `--gpus N` (N > 1) stands for when bench.py is started bare: one rank per GPU on this node, rendezvous on the loopback address."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + list(argv)


def check_world(gpus: int, env=os.environ):
    """--gpus N means N ranks.  -> ("run", world) inside a launcher whose WORLD_SIZE equals N (or bare with N = 1); ("relaunch", N) when N > 1 and no launcher
    is present; SystemExit(2) when a launcher's WORLD_SIZE contradicts --gpus (a line with another n_gpus than the one asked for is never printed)."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in env:
        return ("run", 1) if gpus == 1 else ("relaunch", gpus)
    world = int(env["WORLD_SIZE"])
    if world != gpus:
        sys.stderr.write(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE = {world} rank(s); refusing to report a line for another GPU count\n")
        raise SystemExit(2)
    return ("run", world)


def main():
    global SKIP_DEVIATION, DEVIATION_DTYPE
    args = parse_args()
    action, _n = check_world(args.gpus)
    if action == "relaunch":
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(relaunch_command(args.gpus, sys.argv[1:], port), env=env))
    SKIP_DEVIATION = args.no_deviation
    DEVIATION_DTYPE = args.dtype
    if args.dtype != "f32" and args.workload not in ("melband", "zipenhancer"):
        raise SystemExit("--dtype bf16 (bf16 stored in HBM) exists for the melband and zipenhancer workloads; the other workloads run f32")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    assert world == args.gpus, (world, args.gpus)            # (check_world: a line is only ever printed for the GPU count that was asked for)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU execution path")
    if torch.cuda.device_count() < (int(os.environ.get("LOCAL_WORLD_SIZE", world)) if distributed else 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    from audio_denoiser_onnx_amd.distributed import shard_bounds, stitch_device
    from audio_denoiser_onnx_amd.metadata import build_audio_metadata
    from audio_denoiser_onnx_amd.session import InferenceSession
    from audio_denoiser_onnx_amd.synth import synth_batch

    gtcrn = args.workload == "gtcrn"
    strong = args.scaling == "strong"
    if gtcrn:
        with open(os.path.join(REPO, "tests", "golden", "gtcrn_seed0.adew"), "rb") as f:
            blob = f.read()
        meta = build_audio_metadata(producer="bench.py", model_name="GTCRN", task="denoise", model_family="gtcrn",
                                    input_audio_length=CHUNK)
        sess = InferenceSession(weights=blob, metadata=meta, device_id=local_rank)
        B_total = args.batch or 256
        lo, hi = shard_bounds(B_total, world, rank) if strong else (rank * B_total, (rank + 1) * B_total)
        B = hi - lo
        x_host = synth_batch(B, CHUNK, first_index=lo)
        sr, wl = SR, None
    else:
        default_B = {"zipenhancer": 128, "melband": 32, "mossformer": 64, "dfsmn": 256}[args.workload]
        B_total = args.batch or default_B
        lo, hi = shard_bounds(B_total, world, rank) if strong else (rank * B_total, (rank + 1) * B_total)
        B = hi - lo
        wl = build_workload(args.workload, max(B, 1), rank, local_rank, args.dtype)      # (an idle rank of a strong-scaling run still opens its session)
        sess, x_host, sr = wl["sess"], wl["x"][:B], wl["sr"]
        if args.steps == 100 and args.warmup == 10:      # the defaults are sized for GTCRN's 0.4 ms steps; these steps take 0.2 - 1.2 s
            args.steps, args.warmup = 10, 2
        args.ramp_ms = 0.0
    if args.no_graph:
        sess.set_option("graph", "0")
    if gtcrn and args.geometry != "auto":
        sess.set_option("geometry", args.geometry)
    sess.reserve(max(B, 1))
    d_in = torch.from_numpy(x_host).cuda()
    per = (B_total + world - 1) // world if strong else B             # rows of a rank's (zero-padded) gather block
    d_block = torch.zeros((per, sess.row_out), dtype=torch.int16, device="cuda")
    d_out = d_block[:B]
    gathered = torch.empty((world * per, sess.row_out), dtype=torch.int16, device="cuda") if (args.stitch and distributed) else None
    # A real (non-null) stream: ade_process_device treats a NULL stream handle as "run synchronously", which would put a
    # host round trip between consecutive steps.  Steps are enqueued back to back; the timed region is still bracketed
    # by barrier + torch.cuda.synchronize() on both sides.
    launch_stream = torch.cuda.Stream()
    torch.cuda.set_stream(launch_stream)
    stream = launch_stream.cuda_stream

    def step():
        if B > 0:
            sess.run_device(d_in, d_out, stream=stream)
        if gathered is not None:
            stitch_device(d_block, gathered)

    # Power-state ramp (untimed, disclosed in config.clock_ramp_ms): after process start the GPU needs tens of milliseconds of work before its
    # clocks settle, far more than W = 10 steps of 0.4 ms; without it the first timed steps run at a lower clock and the result depends on
    # how long the process has been alive.  The W warm-up steps and the K timed steps below are unchanged.
    if args.ramp_ms > 0:
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_ms = [1e3 * elapsed / max(1, args.steps)]
    if distributed:
        mine = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        every = torch.empty(world, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(every, mine)
        rank_ms = [1e3 * float(v) / max(1, args.steps) for v in every.tolist()]
        elapsed = float(every.max().item())                 # the job's time is its slowest rank's

    if gtcrn and sess.tap("xchg_error", 1)[0] != 0.0:      # a segment hand-off of the fused path timed out inside the timed loop: the steps after it are not a measurement
        raise SystemExit("bench.py: the fused path reported an inter-workgroup time-out (xchg_error) during the timed loop; no number is reported")
    out_seconds_per_row = sess.out_len / sr
    audio_s_per_step = (B_total if strong else world * B) * out_seconds_per_row
    ms_per_step = 1e3 * elapsed / max(1, args.steps)
    value = audio_s_per_step * args.steps / elapsed

    # the reference's own timing convention (wall clock around the slice loop INCLUDING the host <-> device copies, Inference_GTCRN_ONNX.py:323-343;
    # SURVEY.md section 8 d1): the same batch through the host-buffer entry ade_process on page-locked caller buffers.  Reported beside the
    # device-resident headline, never as `value`.
    host_inclusive = None
    if rank == 0 and args.host_steps > 0:
        pin_in = torch.from_numpy(x_host.copy()).pin_memory()
        pin_out = torch.empty((B, sess.row_out), dtype=torch.int16).pin_memory()
        p_in, p_out = pin_in.numpy(), pin_out.numpy()
        hs = args.host_steps if gtcrn else min(args.host_steps, 3)
        for _ in range(5 if gtcrn else 1):       # (the first calls create the sub-batch streams and page the buffers in)
            sess.process_into(p_in, p_out)
        t_h = time.perf_counter()
        for _ in range(hs):
            sess.process_into(p_in, p_out)
        h_ms = (time.perf_counter() - t_h) / hs * 1e3
        host_inclusive = {"ms_per_step": round(h_ms, 4), "value": round(B * out_seconds_per_row / (h_ms * 1e-3), 1), "unit": "audio-s/s",
                          "rtf": float(f"{h_ms * 1e-3 / (B * out_seconds_per_row):.3e}"), "steps": hs,
                          "note": "synchronous ade_process on page-locked host buffers: H2D + kernels + D2H per step, one GPU"}
        # The same loop PIPELINED (ade_submit / ade_wait, include/ade.h): what a file of many batches costs per batch once the copy-in of call k + 1 and the copy-out of
        # call k - 1 run under call k's kernels.  A ring of `depth` page-locked buffer sets whose inputs differ per slot; wall clock over 100 back-to-back submissions.
        if hasattr(sess, "submit"):
            depth = 3
            ring_in = [torch.from_numpy(np.roll(x_host, k, axis=0).copy()).pin_memory() for k in range(depth)]
            ring_out = [torch.empty((B, sess.row_out), dtype=torch.int16).pin_memory() for _ in range(depth)]

            def pipelined(n):
                tickets = []
                for k in range(n):
                    if len(tickets) >= depth:
                        sess.wait(tickets.pop(0))
                    tickets.append(sess.submit(ring_in[k % depth].numpy(), ring_out[k % depth].numpy()))
                for t in tickets:
                    sess.wait(t)
            ps = max(100, hs) if gtcrn else max(4, hs)               # (fill + drain of the pipeline -- one copy-in and one copy-out -- are inside the clock: 0.3 ms over the run)
            pipelined(8 if gtcrn else 2)
            t_p = time.perf_counter()
            pipelined(ps)
            p_ms = (time.perf_counter() - t_p) / ps * 1e3
            same = bool(np.array_equal(ring_out[0].numpy(), sess.process(ring_in[0].numpy())[0]))
            host_inclusive.update({"steady_state_ms_per_step": round(p_ms, 4), "steady_state_value": round(B * out_seconds_per_row / (p_ms * 1e-3), 1), "steady_state_steps": ps,
                                   "steady_state_equals_ade_process": same,
                                   "steady_state_note": f"ade_submit / ade_wait, {depth} submissions in flight on page-locked buffers: wall clock per batch over {ps} back-to-back batches"})

    roofline = cpu = kernels = others = stft_op = None
    if rank == 0 and gtcrn:
        # per-kernel device time, HIP events on the launch stream (ade_profile_last), averaged over a few forwards
        def timed(mode, reps=10):
            sess.profile(mode)
            out = {}
            for _ in range(reps):
                sess.run_device(d_in, d_out, stream=stream)
                for k, v in sess.kernel_times().items():
                    a = out.setdefault(k, {"ms": 0.0, "launches": v["launches"]})
                    a["ms"] += v["ms"] / reps
            sess.profile(0)
            return {k: v for k, v in out.items() if v["launches"] > 0}

        acc = timed(2)            # the launch sequence exactly as timed above (ONE kernel on the fused path)
        stages = timed(1, 5)      # informational: the same work as one kernel per network stage
        kernels = {k: {"ms_per_forward": round(v["ms"], 4), "launches": v["launches"]} for k, v in acc.items()}
        kernels.update({"stage:" + k: {"ms_per_forward": round(v["ms"], 4), "launches": v["launches"]} for k, v in stages.items()})
        dom = max(acc, key=lambda k: acc[k]["ms"])
        macs, nbytes = KERNEL_MODEL[dom]
        t_events = acc[dom]["ms"] * 1e-3 / acc[dom]["launches"]           # HIP events around the launch (profile mode, a synchronise per step)
        # When the step IS this one kernel, its average launch duration is the timed loop's own step time (K back-to-back launches between the
        # two synchronise pairs); the event figure is reported beside it.
        one_kernel_step = len(acc) == 1 and acc[dom]["launches"] == 1
        t_launch = (elapsed / max(1, args.steps)) if one_kernel_step else t_events
        flops_launch = 2.0 * macs * B / acc[dom]["launches"]
        bytes_launch = float(nbytes) * B / acc[dom]["launches"]
        tf = flops_launch / t_launch / 1e12
        gbs = bytes_launch / t_launch / 1e9
        if tf / FP32_PEAK_TFLOPS >= gbs / HBM_PEAK_GBS:
            roofline = {"kernel": dom, "bound": "fp32_valu", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / FP32_PEAK_TFLOPS, 4), "traffic": None,
                        "peak_note": "fp32 peak, 157.3 TFLOP/s: the packed-fp32 VALU rate and the dense f32-MFMA rate (v_mfma_f32_16x16x4_f32) are the same 64 FLOP/clk/SIMD on gfx950; the kernel uses both (pointwise convolutions of the GTConvBlocks and conv0 on the matrix cores, the rest packed VALU)",
                        "avg_launch_us": round(t_launch * 1e6, 2), "avg_launch_us_events": round(t_events * 1e6, 2),
                        "avg_launch_from": "the timed loop (K launches back to back)" if one_kernel_step else "HIP events",
                        "alt_hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
        else:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_us": round(t_launch * 1e6, 2),
                        "alt_fp32_frac": round(tf / FP32_PEAK_TFLOPS, 4)}
        # HBM-side traffic of the dominant kernel: bench.py cannot run PMC passes itself, so it reports the committed
        # rocprofv3 FETCH_SIZE / WRITE_SIZE measurement of this kernel (profiles/traffic_pmc.json, tools/pmc_traffic.sh)
        try:
            with open(os.path.join(REPO, "profiles", "traffic_pmc.json")) as f:
                tp = json.load(f)
            ent = tp["kernels"].get(dom)
            if ent and tp.get("source_sha1") != source_sha1():
                roofline["traffic_note"] = ("profiles/traffic_pmc.json was measured on another build of csrc/ (" + str(tp.get("build")) + "): not reported; "
                                            "re-run tools/pmc_traffic.sh")
            elif ent:
                roofline["traffic"] = int((2.0 * ent["fetch_kib"] + ent["write_kib"]) * 1024 * B / tp["batch"])
                roofline["traffic_note"] = ("bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes "
                                            f"({tp['build']}); fabric-side incl. Infinity-Cache hits; algorithmic bytes per launch = "
                                            f"{int(bytes_launch)}")
        except (OSError, KeyError, ValueError):
            pass
        per_gpu_step_s = elapsed / max(1, args.steps)
        roofline["pipeline"] = {
            "hbm_frac": round(PIPELINE_BYTES_PER_CHUNK * B / per_gpu_step_s / 1e9 / HBM_PEAK_GBS, 6),
            "fp32_frac": round(PIPELINE_FLOP_PER_CHUNK * B / per_gpu_step_s / 1e12 / FP32_PEAK_TFLOPS, 4),
            "sum_kernel_ms": round(sum(v["ms"] for v in acc.values()), 4),
        }
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(blob, x_host, args.cpu_seconds)
        if world == 1 and not args.no_stft_operator:
            try:
                stft_op = stft_operator_lines(stream)
            except Exception as ex:   # the headline must not depend on it
                stft_op = {"error": repr(ex)}
        if world == 1 and args.other_steps > 0:
            # ZipEnhancer (the second north-star target) with the full step count; the two one-second-per-step transformer configs with ONE timed step
            # each after their warm-up step, so that the driver's clock covers every BASELINE config and the default invocation still ends within minutes.
            others = {}
            for name, steps, dt_ in (("zipenhancer", max(3, args.other_steps), "f32"), ("zipenhancer", max(3, args.other_steps), "bf16"), ("melband", 3, "f32"), ("melband", 3, "bf16"), ("mossformer", 3, "f32"), ("dfsmn", max(3, args.other_steps), "f32")):
                if name not in args.other.split(","):
                    continue
                key = name if dt_ == "f32" else f"{name}_{dt_}"
                try:
                    others[key] = other_workload_line(name, steps, local_rank, stream, args.other_cpu_seconds if dt_ == "f32" else 0.0, dt_)
                except Exception as ex:   # the headline must not depend on it
                    others[key] = {"error": repr(ex)}
    elif rank == 0:
        # A GEMM-shaped family is hundreds of launches per step (fp32 matrix-core GEMMs + row kernels), so the roofline object prices the WHOLE
        # step against the dense fp32 matrix rate: achieved = algorithmic flops of the step / the step's device time.  Per-kernel device times and the
        # MFMA-busy counters of the same command are the committed rocprofv3 summaries under profiles/ (named in `evidence`).
        per_gpu_step_s = elapsed / max(1, args.steps)
        tf = wl["flop"] * B / per_gpu_step_s / 1e12
        roofline = {"kernel": "whole step (all launches)", "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / FP32_PEAK_TFLOPS, 4), "traffic": None,
                    "peak_note": "dense f32-MFMA rate (v_mfma_f32_16x16x4_f32), 157.3 TFLOP/s; flops = 2 x MACs of the model's matrix products per row x rows",
                    "algorithmic_bytes_per_step": int(B * (sess.row_in + sess.row_out) * 2)}
        for cand in (f"r06_{args.workload}{'' if args.dtype == 'f32' else '_' + args.dtype}_mfma_busy.json", f"r05_{args.workload}{'' if args.dtype == 'f32' else '_' + args.dtype}_mfma_busy.json", f"r04_{args.workload}{'' if args.dtype == 'f32' else '_' + args.dtype}_mfma_busy.json",
                     f"r02_{args.workload}{'' if args.dtype == 'f32' else '_' + args.dtype}_mfma_busy.json"):
            if "mfma_busy" in roofline:
                break
            try:
                with open(os.path.join(REPO, "profiles", cand)) as f:
                    roofline["mfma_busy"] = json.load(f)
                    roofline["evidence"] = "profiles/" + cand
            except (OSError, ValueError):
                pass
        roofline["traffic"], roofline["traffic_note"] = workload_traffic(args.workload, args.dtype, B, default_B)
        fam = kernel_family_roofline(args.workload, args.dtype, B, sess.frames)
        if fam:
            roofline["dominant_kernel"] = fam
        if world == 1 and args.cpu_seconds > 0 and wl["cpu"] is not None:
            cpu = wl["cpu"]()

    if rank == 0:
        geo = sess.tap("fused_geometry", 2) if gtcrn else None
        line = {
            "metric": "audio_seconds_per_second (GTCRN 16 kHz, batch=256 x 1 s chunks; RTF = 1/value)" if gtcrn else wl["metric"],
            "value": round(value, 1),
            "unit": "audio-s/s",
            "rtf": float(f"{1.0 / value:.3e}"),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype if not gtcrn else "f32",
            "data": "synthetic",
            "config": {"workload": ("GTCRN 16 kHz, batch=256 x 1 s chunks, fp32, int16 PCM in/out resident in HBM "
                                    "(BASELINE.json configs[1])") if gtcrn else wl["workload"],
                       "chunks_per_gpu": B, "chunks_total": B_total if strong else world * B, "ranks": world, "chunk_samples": sess.in_len, "out_samples": sess.out_len, "clock_ramp_ms": args.ramp_ms,
                       "weights": "seeded reference-architecture GTCRN (tests/golden/gtcrn_seed0.adew)" if gtcrn else wl["weights"],
                       "launch": (f"one kernel per step (k_gtcrn_chunk, geometry {int(geo[0])}: {int(geo[1])} workgroup(s) of {(1024, 512, 256)[int(geo[0])]} threads "
                                  "per chunk), plain launch") if gtcrn
                                 else "the sub-engine's launch sequence (replayed from a captured hipGraph unless --no-graph)",
                       "rank_ms_per_step": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4)},
                       "stitch_all_gather": bool(gathered is not None),
                       "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if distributed else None)},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "host_inclusive": host_inclusive,
            "kernels": kernels,
        }
        if others:
            line["other_workloads"] = others
        if stft_op:
            line["stft_operator"] = stft_op
        if not gtcrn and wl.get("target_rtf"):
            line["target_rtf"] = wl["target_rtf"]
        if not gtcrn and args.dtype != "f32":
            line["deviation_from_f32"] = wl.get("deviation")
            roofline["frac_of_f32_peak"] = roofline["frac"]                    # comparability with the f32 line (can exceed 1: bf16 inputs run on a faster pipe)
            roofline["peak"] = BF16_PEAK_TFLOPS
            roofline["frac"] = round(roofline["achieved"] / BF16_PEAK_TFLOPS, 4)
            roofline["peak_note"] = ("dense bf16-MFMA rate, ~2.5 PFLOP/s (MI355X_MICROARCH.md; not the 2:1-sparsity figure); bf16 operands stored in HBM, 32x32x16 / 16x16x32 "
                                     "bf16 matrix instructions; at K = 384 .. 1536 and 1.5 M rows the products are bound by their stores and the residual stream's traffic, "
                                     "not by the matrix cores (DESIGN.md section 6d)")
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

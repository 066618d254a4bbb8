"""Multi-GPU sharding of independent slices (one process per GPU, ``torch.distributed``; "nccl" == RCCL on ROCm).

Slices / chunks are independent reference calls that the reference simply concatenates
(GTCRN/Inference_GTCRN_ONNX.py:326-332): there is no halo, no state and no reduction between them, so the N>1
path is a contiguous block partition of the slice axis with NO collective in the compute path.  The only exchange
step is the final stitch — every rank (or the writer) needs the whole output waveform — which is one all-gather of
int16 rows (8.1 MB total for 256 x 1 s chunks; latency-bound on xGMI, not bandwidth-bound).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of ``n_rows`` owned by ``rank`` (blocks of ceil(n/world); trailing ranks may be empty)."""
    if world <= 1:
        return 0, n_rows
    per = (n_rows + world - 1) // world
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def stitch_rows(local: np.ndarray, n_rows: int, world: int, rank: int, group=None) -> np.ndarray:
    """All-gather the per-rank int16 output rows back into the full ``(n_rows, out_len)`` array on every rank.

    Uses ``all_gather_into_tensor`` on equal-sized (padded) blocks: on GPUs the tensors stay on the device and the
    collective runs over RCCL/xGMI; with the gloo backend (CPU tests) the same code path runs on host tensors."""
    import torch
    import torch.distributed as dist

    per = (n_rows + world - 1) // world
    out_len = local.shape[1]
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    block = torch.zeros((per, out_len), dtype=torch.int16, device=device)
    if local.shape[0]:
        block[: local.shape[0]] = torch.from_numpy(np.ascontiguousarray(local)).to(device)
    gathered = torch.empty((world * per, out_len), dtype=torch.int16, device=device)
    # neither RCCL/NCCL nor gloo has an int16 datatype: move the rows as raw bytes
    dist.all_gather_into_tensor(gathered.view(torch.uint8), block.view(torch.uint8), group=group)
    return gathered[:n_rows].cpu().numpy()


def sharded_run(session, rows: np.ndarray, world: int, rank: int, group=None) -> np.ndarray:
    """``rows`` (n, row_in) int16 -> (n, row_out) int16 on every rank: this rank's contiguous block goes through ``run_device`` on DEVICE
    tensors straight into its padded gather block, the blocks are all-gathered where they are (RCCL over xGMI with the "nccl" backend), and the
    stitched result crosses to the host once.  No host round trip between the engine and the collective.  With the gloo backend (CPU tests on
    the host-simulated engine, whose device memory is host memory) the same code runs on CPU tensors."""
    import torch
    import torch.distributed as dist

    n = rows.shape[0]
    per = (n + world - 1) // world
    lo, hi = shard_bounds(n, world, rank)
    on_gpu = torch.cuda.is_available() and getattr(session, "device_id", -1) >= 0 and not getattr(session._lib, "is_simulator", False)
    compute = torch.device("cuda", session.device_id) if on_gpu else torch.device("cpu")
    comm = compute if dist.get_backend(group) == "nccl" else torch.device("cpu")
    block = torch.zeros((per, session.row_out), dtype=torch.int16, device=compute)
    if hi > lo:
        d_in = torch.from_numpy(np.ascontiguousarray(rows[lo:hi], dtype=np.int16)).to(compute)
        session.run_device(d_in, block[: hi - lo])                         # synchronous on the engine's own stream
    block = block.to(comm)
    gathered = torch.empty((world * per, session.row_out), dtype=torch.int16, device=comm)
    # neither RCCL/NCCL nor gloo has an int16 datatype: move the rows as raw bytes
    dist.all_gather_into_tensor(gathered.view(torch.uint8), block.view(torch.uint8), group=group)
    return gathered[:n].cpu().numpy()


def stitch_device(d_local, d_gathered, group=None) -> None:
    """Device-tensor form used by bench.py --stitch: ``d_gathered[(world*B), out_len] <- all ranks' d_local[B, out_len]``."""
    import torch.distributed as dist

    import torch

    dist.all_gather_into_tensor(d_gathered.view(torch.uint8), d_local.view(torch.uint8), group=group)

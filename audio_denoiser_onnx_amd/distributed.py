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
    """All-gather the per-rank output rows (int16 PCM, or the float32 rows of a float-output export) back into the full ``(n_rows, out_len)``
    array on every rank, in ``local``'s dtype.

    Uses ``all_gather_into_tensor`` on equal-sized (padded) blocks: on GPUs the tensors stay on the device and the
    collective runs over RCCL/xGMI; with the gloo backend (CPU tests) the same code path runs on host tensors."""
    import torch
    import torch.distributed as dist

    per = (n_rows + world - 1) // world
    out_len = local.shape[1]
    backend = dist.get_backend(group)
    device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    local = np.ascontiguousarray(local)
    tdtype = torch.from_numpy(np.zeros(0, local.dtype)).dtype                # the rows travel in their own dtype (a float32 block cast to int16 would be all zeros)
    block = torch.zeros((per, out_len), dtype=tdtype, device=device)
    if local.shape[0]:
        block[: local.shape[0]] = torch.from_numpy(local).to(device)
    gathered = torch.empty((world * per, out_len), dtype=tdtype, device=device)
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


def init_from_env(prefer_backend: str = "") -> tuple:
    """torchrun's environment -> ``(rank, world, local_rank)``.  One process (WORLD_SIZE unset or 1) touches nothing.  With more, the process group is
    created here: "nccl" (= RCCL over xGMI) when this rank has a GPU, "gloo" otherwise (CPU tests on the host-simulated engine) or when ``ADE_DIST_BACKEND`` /
    ``prefer_backend`` says so.  Rendezvous defaults to 127.0.0.1 (the container hostname may not resolve)."""
    import os

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world <= 1:
        return 0, 1, local if "LOCAL_RANK" in os.environ else 0
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    backend = prefer_backend or os.environ.get("ADE_DIST_BACKEND", "") or ("nccl" if torch.cuda.is_available() else "gloo")
    if not dist.is_initialized():
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shutdown() -> None:
    """Counterpart of init_from_env (no-op for a single process)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def run_rows(session, rows: np.ndarray, rank: int = 0, world: int = 1, group=None):
    """One batched graph call on ``rows`` (n, channels, in_len) -> a list with one (n, out_channels, out_len) array per graph output, the SAME on every rank.

    ``world == 1``: ``session.run``.  Otherwise the rows are dealt in contiguous blocks (shard_bounds; rows are independent calls of the reference's graph) and the
    outputs stitched with one all-gather: int16 static exports through ``sharded_run`` (device block -> all-gather over RCCL -> one D2H), float / dynamic-length
    exports through ``session.run`` on the rank's block + ``stitch_rows`` in the output's dtype.  Used by the Mel-Band-Roformer (2-channel rows), MossFormer2-SS
    (two outputs) and H-GTCRN (2 channels in, 1 out) drivers; the GTCRN-style drivers go through ``inference_gtcrn.denoise``."""
    name = session.get_inputs()[0].name
    rows = np.ascontiguousarray(rows)
    n = rows.shape[0]
    if world <= 1:
        return session.run(None, {name: rows})
    n_out = getattr(session, "n_outputs", 1)
    oc, ol = getattr(session, "out_channels", 1), session.out_len
    int16_io = getattr(session, "in_dtype", np.int16) == np.int16 and getattr(session, "out_dtype", np.int16) == np.int16
    meta = getattr(session, "metadata", None)
    dynamic = bool(meta and meta.optional_bool("dynamic_axes", False))
    if int16_io and not dynamic and hasattr(session, "run_device"):
        flat = sharded_run(session, rows.reshape(n, -1), world, rank, group).reshape(n, n_out, oc, ol)
        return [np.ascontiguousarray(flat[:, i]) for i in range(n_out)]
    lo, hi = shard_bounds(n, world, rank)
    outs = session.run(None, {name: rows[lo:hi]}) if hi > lo else [np.zeros((0, oc, ol), getattr(session, "out_dtype", np.int16)) for _ in range(n_out)]
    return [stitch_rows(np.ascontiguousarray(o).reshape(hi - lo, -1), n, world, rank, group).reshape(n, oc, -1) for o in outs]

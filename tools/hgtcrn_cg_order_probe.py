#!/usr/bin/env python3
"""How much of H-GTCRN's WPE instability is summation order?  

The reference solves a 36 x 36 complex system per bin with six conjugate-gradient steps in fp32 (H-GTCRN/Export_H_GTCRN.py:499-555).  This probe restates that solve
three ways on the fixture's spectra and measures each against the reference's own WPE output (tests/golden/hgtcrn_seed0.npz `wpe_r / wpe_i`):
  (a) the oracle (numpy, i.e. OpenBLAS sgemm + numpy reductions),
  (b) the same expressions on torch tensors (torch's own sgemm / reductions: the reference's arithmetic, op for op),
  (c) the same expressions with every product sum taken sequentially in fp32 (the order a GPU lane takes).
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import hgtcrn_oracle as ho  # noqa: E402

F32 = np.float32


def wpe_generic(Xr, Xi, mm, rsum):
    """hgtcrn_oracle.wpe with the matrix product and the column reduction injected."""
    B, Fq, M, T = Xr.shape
    LG, DELAY, CG = ho.LG, ho.DELAY, ho.CG_ITER
    Dr = np.zeros((B, Fq, LG, M, T), F32); Di = np.zeros_like(Dr)
    for l in range(LG):
        sh = DELAY + l
        if sh < T:
            Dr[:, :, l, :, sh:] = Xr[..., :T - sh]; Di[:, :, l, :, sh:] = Xi[..., :T - sh]
    Dr, Di = Dr.reshape(B, Fq, LG * M, T), Di.reshape(B, Fq, LG * M, T)
    mag = (Xr * Xr + Xi * Xi).astype(F32)
    eps = (F32(1e-3) * mag.max(axis=(-2, -1)).mean(axis=-1, dtype=F32)).astype(F32).reshape(B, 1, 1, 1)
    inv = (F32(1.0) / np.maximum(mag.mean(axis=2, keepdims=True, dtype=F32), eps)).astype(F32)
    tr, ti = (Dr * inv).astype(F32), (Di * inv).astype(F32)
    DrT, DiT, XrT, XiT = Dr.transpose(0, 1, 3, 2), Di.transpose(0, 1, 3, 2), Xr.transpose(0, 1, 3, 2), Xi.transpose(0, 1, 3, 2)
    Rr = (mm(tr, DrT) + mm(ti, DiT)).astype(F32); Ri = (mm(ti, DrT) - mm(tr, DiT)).astype(F32)
    Pr = (mm(tr, XrT) + mm(ti, XiT)).astype(F32); Pi = (mm(ti, XrT) - mm(tr, XiT)).astype(F32)
    Rr = (Rr + eps * np.eye(LG * M, dtype=F32)).astype(F32)
    xr = np.zeros_like(Pr); xi = np.zeros_like(Pi)
    r_r, r_i, pr, pi = Pr, Pi, Pr, Pi
    rr = (rsum(r_r * r_r + r_i * r_i) + F32(1e-12)).astype(F32)
    for _ in range(CG):
        Apr = (mm(Rr, pr) - mm(Ri, pi)).astype(F32); Api = (mm(Rr, pi) + mm(Ri, pr)).astype(F32)
        pAp = (rsum(pr * Apr + pi * Api) + F32(1e-12)).astype(F32)
        alpha = (rr / pAp).astype(F32)[..., None, :]
        xr, xi = (xr + alpha * pr).astype(F32), (xi + alpha * pi).astype(F32)
        r_r, r_i = (r_r - alpha * Apr).astype(F32), (r_i - alpha * Api).astype(F32)
        rr_new = (rsum(r_r * r_r + r_i * r_i) + F32(1e-12)).astype(F32)
        beta = (rr_new / rr).astype(F32)[..., None, :]
        pr, pi = (r_r + beta * pr).astype(F32), (r_i + beta * pi).astype(F32)
        rr = rr_new
    Gr, Gi = xr.transpose(0, 1, 3, 2), (-xi).transpose(0, 1, 3, 2)
    return (Xr - (mm(Gr, Dr) - mm(Gi, Di))).astype(F32), (Xi - (mm(Gi, Dr) + mm(Gr, Di))).astype(F32)


def mm_numpy(a, b): return np.matmul(a, b).astype(F32)
def mm_torch(a, b): return torch.matmul(torch.from_numpy(np.ascontiguousarray(a)), torch.from_numpy(np.ascontiguousarray(b))).numpy()
def mm_seq(a, b):
    out = np.zeros(a.shape[:-1] + (b.shape[-1],), F32)
    for k in range(a.shape[-1]):
        out = (out + a[..., :, k:k + 1] * b[..., k:k + 1, :]).astype(F32)
    return out
def rs_numpy(x): return x.sum(axis=-2, dtype=F32)
def rs_torch(x): return torch.from_numpy(np.ascontiguousarray(x)).sum(dim=-2).numpy()
def rs_seq(x):
    out = np.zeros(x.shape[:-2] + x.shape[-1:], F32)
    for k in range(x.shape[-2]):
        out = (out + x[..., k, :]).astype(F32)
    return out


def main():
    z = np.load(os.path.join(REPO, "tests", "golden", "hgtcrn_seed0.npz"))
    pcm = z["pcm_in"]
    from audio_denoiser_onnx_amd import hgtcrn
    W = pcm.shape[-1]
    o = ho.HgtcrnOracle(hgtcrn.fold_state_dict({str(k): z["w:" + str(k)] for k in z["keys"]}), W)
    with np.errstate(all="ignore"):
        o.process(pcm)
    Xr, Xi = np.ascontiguousarray(o.taps["stft_r"].transpose(0, 2, 1, 3)), np.ascontiguousarray(o.taps["stft_i"].transpose(0, 2, 1, 3))     # (B, F, 2, T)
    ref_r, ref_i = z["wpe_r"].transpose(0, 2, 1, 3), z["wpe_i"].transpose(0, 2, 1, 3)
    spread = z["wpe_ref_spread"]                         # the reference's own fp32-vs-fp64 distance per (row, bin)
    for name, mm, rs in (("numpy (OpenBLAS)", mm_numpy, rs_numpy), ("torch ops", mm_torch, rs_torch), ("sequential fp32", mm_seq, rs_seq)):
        with np.errstate(all="ignore"):
            a_r, a_i = wpe_generic(Xr, Xi, mm, rs)
        err = np.maximum(np.abs(a_r - ref_r), np.abs(a_i - ref_i)).max(axis=(2, 3))        # (rows, bins)
        unstable = spread >= 1e-4
        print(f"{name:18s}: bit-equal to the reference {bool(np.array_equal(a_r, ref_r) and np.array_equal(a_i, ref_i))};  max err on the reference's stable bins "
              f"{np.nanmax(err[~unstable]):.2e}, on its unstable bins {np.nanmax(err[unstable]):.2e} (median {np.nanmedian(err[unstable]):.2e});  bins off by > 1e-3: {(err > 1e-3).sum()} "
              f"of {err.size} (the reference's own fp32-vs-fp64 set: {unstable.sum()})")


if __name__ == "__main__":
    main()

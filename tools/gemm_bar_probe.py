#!/usr/bin/env python3
"""What the vendor library reaches on Mel-Band-Roformer's four bf16 products (a bar for csrc/ade_gemm16.h, not a dependency: the engine does not call it).

    python tools/gemm_bar_probe.py          # torch.nn.functional.linear on bf16 (hipBLASLt / rocBLAS underneath), plain product + bf16 store, no epilogue
"""
import time

import torch

M = 32 * 801 * 60            # tokens of 32 x 8 s at depth-6 Mel-Band-Roformer (60 bands x 801 frames per clip)
SHAPES = [("in-projection", M, 384, 1544, torch.bfloat16), ("FFN-in", M, 384, 1536, torch.bfloat16), ("FFN-out", M, 1536, 384, torch.bfloat16),
          ("out-projection", M, 512, 384, torch.bfloat16),
          # the same four on the f32 path, and MossFormer2's two large products (64 x 4 s: 511 936 rows; the engine's shifted-operand in-projection and its gated out-projection)
          ("f32 in-projection", M, 384, 1544, torch.float32), ("f32 FFN-in", M, 384, 1536, torch.float32), ("f32 FFN-out", M, 1536, 384, torch.float32),
          ("f32 out-projection", M, 512, 384, torch.float32), ("MossFormer2 in", 64 * 7999, 512, 2176, torch.float32), ("MossFormer2 out", 64 * 7999, 1024, 512, torch.float32)]


def main():
    dev = torch.device("cuda", 0)
    torch.backends.cuda.matmul.allow_tf32 = False
    for name, M, K, N, dt in SHAPES:
        x = torch.randn((M, K), device=dev, dtype=dt)
        w = torch.randn((N, K), device=dev, dtype=dt)
        y = torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            y = torch.nn.functional.linear(x, w)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 10 * 1e3
        print(f"{name:18s} M={M} K={K:4d} N={N:4d}: {ms:7.3f} ms  {2.0 * M * K * N / (ms * 1e-3) / 1e12:7.1f} TFLOP/s  ({(M * K + M * N + N * K) * x.element_size() / (ms * 1e-3) / 1e9:6.0f} GB/s of operands)", flush=True)
        del x, w, y
    # the time-attention core of the same step: 32 clips x 60 bands = 1920 sequences of 801 frames, 8 heads of 64 (k_attention16<2>: q | k | v read in place from the projection)
    for name, nseq, n in (("time attention", 32 * 60, 801), ("frequency attention", 32 * 801, 60)):
        try:
            q, k, v = (torch.randn((nseq, 8, n, 64), device=dev, dtype=torch.bfloat16) for _ in range(3))
            y = torch.nn.functional.scaled_dot_product_attention(q, k, v)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(10):
                y = torch.nn.functional.scaled_dot_product_attention(q, k, v)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / 10 * 1e3
            print(f"{name:18s} {nseq} x 8 heads x {n} x 64: {ms:7.3f} ms  {4.0 * nseq * 8 * n * n * 64 / (ms * 1e-3) / 1e12:7.1f} TFLOP/s", flush=True)
            del q, k, v, y
        except Exception as e:  # noqa: BLE001
            print(f"{name}: scaled_dot_product_attention unavailable here: {e}")


if __name__ == "__main__":
    main()

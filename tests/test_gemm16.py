"""csrc/ade_gemm16.h (bf16 operands stored in HBM, v_mfma_f32_32x32x16_bf16) and the bf16 attention core of csrc/ade_melband.hip, unit-checked against
double-precision host products on the same bf16 inputs (tests/unit/*.hip).  CPU: the same sources under the host simulator (tests/hipsim, which emulates the two
gfx950 bf16 MFMA instructions lane for lane); GPU: built by hipcc for gfx950 on the box and run at sizes that exercise many tiles, tails and both attention axes."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
BUILD = os.path.join(HERE, "unit", "_build")


def _build_sim(name):
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, name + "_sim")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wno-psabi", "-I", os.path.join(HERE, "hipsim"), "-I", os.path.join(REPO, "include"), "-x", "c++",
                    os.path.join(HERE, "unit", name + ".hip"), "-x", "c++", os.path.join(HERE, "hipsim", "hipsim.cpp"), "-o", out], check=True, cwd=REPO)
    return out


def _build_gpu(name):
    os.makedirs(BUILD, exist_ok=True)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(BUILD, name + "_gpu")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-Wno-unused-value", "-I", os.path.join(REPO, "include"),
                    os.path.join(HERE, "unit", name + ".hip"), "-o", out], check=True, cwd=REPO)
    return out


def _run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.hipsim
def test_hipsim_gemm16_tiles_tails_and_stores():
    _run(_build_sim("gemm16_unit"), 130, 70, 72, 200, 136, 192, 64, 260, 64)


@pytest.mark.hipsim
def test_hipsim_attention16_both_axes():
    _run(_build_sim("melband16_unit"), 70, 1, 2, 0, 40, 2, 3, 1, 129, 1, 1, 0)


@pytest.mark.hipsim
def test_hipsim_gemm32_fetch_paths_and_k_tails():
    """csrc/ade_gemm.h's fp32 tile: the four operand fetch paths, a K that ends inside a slab (the fetch is loads only; the tail is zeroed on the way into LDS), partial tiles"""
    _run(_build_sim("gemm32_unit"), 130, 70, 37, 129, 200, 64, 40, 300, 20)


@pytest.mark.gpu
def test_gpu_gemm32_fetch_paths_and_k_tails():
    _run(_build_gpu("gemm32_unit"), 130, 70, 37, 129, 200, 64, 1000, 333, 250, 257, 640, 1029)


@pytest.mark.gpu
def test_gpu_gemm16_many_tiles():
    _run(_build_gpu("gemm16_unit"), 1000, 1544, 384, 777, 384, 1536, 130, 70, 72, 64, 25633, 64, 4096, 512, 512)


@pytest.mark.gpu
def test_gpu_attention16_both_axes():
    _run(_build_gpu("melband16_unit"), 801, 8, 3, 0, 60, 8, 50, 1, 151, 8, 2, 0, 64, 8, 4, 1)

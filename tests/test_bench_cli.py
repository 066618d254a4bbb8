"""bench.py's launcher contract: `--gpus N` means N ranks.  A line is only ever printed for the GPU count that was asked for:
started bare with N > 1 the script re-executes itself under torch.distributed.run with N ranks; inside a launcher whose WORLD_SIZE differs it exits non-zero."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_gpus_flag_decides_the_world(bench):
    assert bench.check_world(1, {}) == ("run", 1)
    assert bench.check_world(2, {}) == ("relaunch", 2)
    assert bench.check_world(8, {"WORLD_SIZE": "8"}) == ("run", 8)
    assert bench.check_world(1, {"WORLD_SIZE": "1"}) == ("run", 1)
    for gpus, world in ((2, "1"), (8, "4"), (1, "2")):
        with pytest.raises(SystemExit) as e:
            bench.check_world(gpus, {"WORLD_SIZE": world})
        assert e.value.code == 2
    with pytest.raises(SystemExit):
        bench.check_world(0, {})


def test_relaunch_is_one_rank_per_gpu_on_loopback(bench):
    cmd = bench.relaunch_command(4, ["--gpus", "4", "--steps", "5"], 29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")


def _lines_with_n_gpus(text):
    out = []
    for ln in text.splitlines():
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        if isinstance(d, dict) and "n_gpus" in d:
            out.append(d)
    return out


def test_mismatched_launcher_prints_no_line():
    """What the round-5 script did: `--gpus 2` inside a one-rank environment ran one rank and printed n_gpus: 1.  Now: exit code 2, no JSON line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and not _lines_with_n_gpus(r.stdout), (r.returncode, r.stdout[-400:])
    assert "WORLD_SIZE = 1" in r.stderr


def test_bare_gpus_2_relaunches_two_ranks_and_never_reports_one():
    """Bare `python bench.py --gpus 2`: two ranks are started (torch.distributed.run on 127.0.0.1).  In this container there is no GPU, so both ranks stop at the
    engine's no-CPU-path check: the exit code is non-zero and, whatever happened, no line with n_gpus == 1 was printed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--host-steps", "0", "--cpu-seconds", "0",
                        "--other-steps", "0", "--no-stft-operator"], env=env, capture_output=True, text=True, timeout=600)
    lines = _lines_with_n_gpus(r.stdout)
    assert all(d["n_gpus"] == 2 for d in lines), lines
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and not lines
        assert (r.stdout + r.stderr).count("bench.py needs a GPU") >= 1

"""N>1 path on CPU: world_size-2 gloo processes shard the slices of one file, run a stand-in compute on their block
and stitch with the same all-gather the GPU path uses (audio_denoiser_onnx_amd/distributed.py).  The compute stand-in
is a pure function of each row, like the real engine (rows are independent reference calls)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from audio_denoiser_onnx_amd.distributed import shard_bounds

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def test_shard_bounds_cover_and_order():
    for n in (0, 1, 2, 7, 10, 256, 257):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            flat = [i for lo, hi in blocks for i in range(lo, hi)]
            assert flat == list(range(n)), (n, world, blocks)
    assert shard_bounds(10, 8, 7) == (10, 10)          # trailing ranks may be empty (B < G: replicas idle)


def test_world_size_2_gloo_shard_and_stitch(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {REPO!r})
        import numpy as np, torch.distributed as dist
        from audio_denoiser_onnx_amd.inference_gtcrn import cut_slices, denoise

        class FakeSession:                      # stand-in for the engine: out row = f(in row), rows independent
            in_len, out_len = 16000, 15872
            def process(self, pcm, want_f32=False):
                return (pcm[:, :15872].astype(np.int32) // 2 + 7).astype(np.int16), None

        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        audio = (np.arange(156302) % 20011 - 10000).astype(np.int16)
        out = denoise(FakeSession(), audio, rank=rank, world=world)
        ref = denoise(FakeSession(), audio)                       # single-process answer
        assert out.shape == ref.shape == (156302,) and np.array_equal(out, ref), rank
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    assert all("ok" in o for o in outs)

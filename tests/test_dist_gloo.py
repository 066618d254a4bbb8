"""N>1 path on CPU: world_size-2 gloo processes shard the slices of one file, run THE ENGINE on their block -- the same csrc/*.hip sources
under the host simulator (tests/hipsim, test-only), called through libade's C ABI exactly like on a GPU -- and stitch with the same
all-gather the GPU path uses (audio_denoiser_onnx_amd/distributed.py::sharded_run).  The stitched file must equal the single-process answer
bit for bit and the oracle within the parity tolerance."""
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
        sys.path.insert(0, {REPO!r}); sys.path.insert(0, {HERE!r})
        import numpy as np, torch.distributed as dist
        from ade_testlib import golden_blob, golden_inputs, hipsim_library, make_session
        from audio_denoiser_onnx_amd.inference_gtcrn import cut_slices, denoise
        from oracle_lib import GtcrnOracle

        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        ins = golden_inputs()
        audio = np.concatenate((ins["wav0"], ins["randn"], ins["wav1"]))[:40000]       # 3 slices of 16000 at stride 15872: ranks get 2 + 1
        sess = make_session(hipsim_library(), seed=0)                                   # the engine (host-simulated), one per rank
        out = denoise(sess, audio, rank=rank, world=world)                              # sharded_run: device block -> all-gather
        assert out.shape == (40000,), out.shape
        if rank == 0:
            ref = denoise(sess, audio)                                                  # single-process answer, same engine
            assert np.array_equal(out, ref)
            slices, _ = cut_slices(audio, 16000, 15872)
            want = GtcrnOracle(golden_blob(0), 16000).process(slices)[0].reshape(-1)[:40000]
            assert np.abs(out.astype(np.int32) - want.astype(np.int32)).max() <= 1
        dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """))
    from ade_testlib import hipsim_library
    hipsim_library()                                   # build once here, not concurrently in both workers
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("ok" in o for o in outs)

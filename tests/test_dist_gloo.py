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


def test_world_size_4_gloo_idle_ranks_and_uneven_last_block(tmp_path):
    """World 4 on the real engine (host simulator): a file of 2 slices (fewer rows than ranks: ranks 2 and 3 are idle replicas that still take part in the all-gather) and a
    file of 5 slices (blocks of ceil(5 / 4) = 2: 2 + 2 + 1 + 0 -- an uneven last block AND an idle rank), both through sharded_run's padded gather blocks, bit-equal to one
    process.  Short slices (4096 samples, 17 frames) keep the simulator's cost down."""
    script = tmp_path / "worker4.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {REPO!r}); sys.path.insert(0, {HERE!r})
        import numpy as np, torch.distributed as dist
        from ade_testlib import hipsim_library, make_session
        from audio_denoiser_onnx_amd.distributed import shard_bounds
        from audio_denoiser_onnx_amd.inference_gtcrn import denoise, plan_slices
        from audio_denoiser_onnx_amd.synth import synth_chunk

        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        assert world == 4
        sess = make_session(hipsim_library(), seed=1, length=4096)                       # 4096 in -> 4096 out (16 hops)
        for n_slices in (2, 5):
            n = (n_slices - 1) * sess.out_len + sess.in_len - 100                        # the last slice is partly padding
            assert plan_slices(n, sess.in_len, sess.out_len)[1] == n_slices
            audio = np.concatenate([synth_chunk(7 + i, 4096) for i in range(n_slices + 1)])[:n]
            lo, hi = shard_bounds(n_slices, world, rank)
            assert (hi - lo) == ([1, 1, 0, 0] if n_slices == 2 else [2, 2, 1, 0])[rank]
            out = denoise(sess, audio, rank=rank, world=world)
            assert out.shape == (n,), out.shape
            if rank == 0:
                assert np.array_equal(out, denoise(sess, audio)), n_slices              # the single-process answer of the same engine
            dist.barrier()
        dist.destroy_process_group()
        print("rank", rank, "ok")
    """))
    from ade_testlib import hipsim_library
    hipsim_library()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="4")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(4)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("ok" in o for o in outs)


def test_world_size_2_gloo_stitch_keeps_float_rows(tmp_path):
    """stitch_rows moves the rows in THEIR dtype: a float-output export's rows (normalised samples) must not be squeezed through int16 (they would all become 0)."""
    script = tmp_path / "worker_f32.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {REPO!r})
        import numpy as np, torch.distributed as dist
        from audio_denoiser_onnx_amd.distributed import shard_bounds, stitch_rows
        dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        full = (np.arange(5 * 7, dtype=np.float32).reshape(5, 7) - 17.0) / 64.0          # values inside (-1, 1): int16 would truncate them to 0
        lo, hi = shard_bounds(5, world, rank)
        for dt in (np.float32, np.int16):
            rows = (full * (1 if dt == np.float32 else 4096)).astype(dt)
            got = stitch_rows(rows[lo:hi], 5, world, rank)
            assert got.dtype == dt and np.array_equal(got, rows), (dt, got)
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
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def _run_two_ranks(script, timeout=900, extra_env=None):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", ADE_DIST_BACKEND="gloo", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    return outs


def test_world_size_2_gloo_run_rows_two_outputs_two_channels(tmp_path):
    """distributed.run_rows -- the batched call of the Mel-Band (2-channel rows), MossFormer2-SS (two graph outputs) and H-GTCRN drivers -- on a stand-in session
    whose outputs are a known function of the rows: 5 rows on 2 ranks (3 + 2), both dtypes, every rank gets every row of every output."""
    script = tmp_path / "worker_rows.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {REPO!r})
        import numpy as np
        from audio_denoiser_onnx_amd.distributed import init_from_env, run_rows, shutdown

        class In:
            name = "mix_audio"

        class Fake:                                             # two outputs of 2 channels x 6 samples from rows of 2 channels x 8 samples
            n_outputs, out_channels, out_len, in_len, channels = 2, 2, 6, 8, 2
            def __init__(self, dt): self.in_dtype = np.int16; self.out_dtype = dt
            def get_inputs(self): return [In()]
            def run(self, _, feed):
                x = feed["mix_audio"].astype(np.float32)
                a = x[:, :, :6] * 2 + 1
                b = x[:, ::-1, 2:] - 3
                return [a.astype(self.out_dtype), b.astype(self.out_dtype)]

        rank, world, _ = init_from_env()
        assert world == 2
        rows = (np.arange(5 * 2 * 8).reshape(5, 2, 8) * 7 % 101).astype(np.int16)
        for dt in (np.int16, np.float32):
            s = Fake(dt)
            want = s.run(None, {{"mix_audio": rows}})
            got = run_rows(s, rows, rank, world)
            assert len(got) == 2 and all(g.dtype == dt and np.array_equal(g, w) for g, w in zip(got, want)), dt
        shutdown()
        print("rank", rank, "ok")
    """))
    assert all("ok" in o for o in _run_two_ranks(script, 300))


def test_world_size_2_gloo_hgtcrn_driver_two_channel_rows(tmp_path):
    """`torchrun --nproc-per-node 2 -m audio_denoiser_onnx_amd.inference_hgtcrn` in miniature: two gloo ranks run the driver's main() on the host-simulated
    engine (2-channel rows in, 1 channel out; sharded_run), rank 0 writes the wav; it equals the one-process file bit for bit."""
    from ade_testlib import hipsim_library
    hipsim_library()
    script = tmp_path / "worker_hg.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {REPO!r}); sys.path.insert(0, {HERE!r})
        import numpy as np
        from ade_testlib import hipsim_library
        from audio_denoiser_onnx_amd import _lib, hgtcrn, inference_hgtcrn
        from audio_denoiser_onnx_amd.metadata import write_metadata
        from audio_denoiser_onnx_amd.synth import synth_chunk
        from audio_denoiser_onnx_amd.wavio import write_pcm16
        from audio_denoiser_onnx_amd.weights import save_blob
        _lib._default = hipsim_library()                       # the drivers open the default library: point it at the simulator build (test-only)
        rank = int(os.environ["RANK"]); tmp = {str(tmp_path)!r}
        z = np.load(os.path.join({HERE!r}, "golden", "hgtcrn_seed0.npz"))
        state = {{str(k): z["w:" + str(k)] for k in z["keys"]}}
        W = 2048
        model = os.path.join(tmp, f"hg_{{rank}}.adew")
        save_blob(model, hgtcrn.fold_state_dict(state)); write_metadata(model, hgtcrn.metadata(W))
        audio = np.stack([synth_chunk(3, 3000), synth_chunk(4, 3000)])          # 2 slices of 2048 (stride = output length 2048): one per rank
        write_pcm16(os.path.join(tmp, f"in_{{rank}}.wav"), audio, 16000)
        out = os.path.join(tmp, f"out_{{rank}}.wav")
        assert inference_hgtcrn.main([model, os.path.join(tmp, f"in_{{rank}}.wav"), out]) == 0
        print("rank", rank, "ok")
    """))
    _run_two_ranks(script, 900)
    # one process, same engine
    import numpy as np
    from audio_denoiser_onnx_amd import inference_hgtcrn
    from audio_denoiser_onnx_amd.inference_gtcrn import read_wav_int16
    assert os.path.exists(tmp_path / "out_0.wav") and not os.path.exists(tmp_path / "out_1.wav")      # only rank 0 writes
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK") if k in os.environ}
    try:
        single = tmp_path / "worker_single.py"
        single.write_text(textwrap.dedent(f"""
            import os, sys
            sys.path.insert(0, {REPO!r}); sys.path.insert(0, {HERE!r})
            from ade_testlib import hipsim_library
            from audio_denoiser_onnx_amd import _lib, inference_hgtcrn
            _lib._default = hipsim_library()
            tmp = {str(tmp_path)!r}
            assert inference_hgtcrn.main([os.path.join(tmp, "hg_0.adew"), os.path.join(tmp, "in_0.wav"), os.path.join(tmp, "single.wav")]) == 0
        """))
        subprocess.run([sys.executable, str(single)], check=True, timeout=900)
    finally:
        os.environ.update(env)
    assert np.array_equal(read_wav_int16(tmp_path / "out_0.wav", 16000), read_wav_int16(tmp_path / "single.wav", 16000))

"""``InferenceSession`` — host-side mirror of the object the reference drives per slice.

The reference's inference scripts talk to ``onnxruntime.InferenceSession`` (GTCRN/Inference_GTCRN_ONNX.py:237,
262-267, 307-317): ``get_inputs()/get_outputs()`` metadata, pre-bound buffers, one ``run_with_iobinding`` per
slice.  This class exposes the same surface over libade's C ABI (include/ade.h); the compute is the hand-written
gfx950 path and nothing else — construction fails loudly without the built library or a GPU.

Differences from ORT that are the point of this engine: ``run`` accepts ``(B, 1, L)`` and treats the B rows as
B independent reference calls executed as one batch, and :meth:`run_device` takes device tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .metadata import MetadataReader, load_runtime_metadata, validate_audio_metadata

INPUT_NAME = "noisy_audio"       # GTCRN/Export_GTCRN.py:768
OUTPUT_NAME = "denoised_audio"   # GTCRN/Export_GTCRN.py:769


class NodeArg:
    """What ``session.get_inputs()[i]`` looks like to the reference driver (name / shape / type)."""

    def __init__(self, name: str, shape: Sequence[int], type_: str = "tensor(int16)"):
        self.name, self.shape, self.type = name, list(shape), type_

    def __repr__(self):
        return f"NodeArg(name={self.name!r}, type={self.type!r}, shape={self.shape})"


class ModelMeta:
    def __init__(self, custom_metadata_map: Dict[str, str]):
        self.custom_metadata_map = dict(custom_metadata_map)


def resolve_model_path(path) -> Path:
    """A model directory resolves to its single ``*.adew`` blob (the reference resolves a dir to ``GTCRN.onnx``,
    Inference_GTCRN_ONNX.py:26-32)."""
    p = Path(path).expanduser()
    if p.is_dir():
        blobs = sorted(p.glob("*.adew"))
        if len(blobs) != 1:
            raise FileNotFoundError(f"expected exactly one .adew weight blob in {p}, found {len(blobs)}")
        return blobs[0]
    if not p.exists():
        raise FileNotFoundError(f"model file not found: {p}")
    return p


class InferenceSession:
    def __init__(self, model_path=None, *, weights: Optional[bytes] = None, metadata: Optional[Dict[str, str]] = None,
                 device_id: int = 0, library: Optional[_lib.AdeLibrary] = None):
        """``model_path``: ``<name>.adew`` (or its directory) with ``<name>_Metadata.json`` beside it; or pass
        ``weights`` + ``metadata`` directly.  ``library`` defaults to the in-tree gfx950 build."""
        self._lib = library or _lib.get_library()
        self._h = C.c_void_p()
        self._in_flight = {}            # ticket -> (pcm, out, f32): the buffers of submissions that have not been waited for
        if model_path is not None:
            model_path = resolve_model_path(model_path)
            reader = load_runtime_metadata(model_path)          # FileNotFoundError / KeyError like the reference
            with open(model_path, "rb") as f:
                weights = f.read()
        else:
            if weights is None or metadata is None:
                raise ValueError("pass either model_path or both weights and metadata")
            reader = MetadataReader(metadata)
        self.metadata = reader
        blob = bytes(weights)                            # libade parses and uploads it inside ade_create; nothing keeps it afterwards
        st = self._lib.c.ade_create(reader.to_json().encode(), blob, len(blob), int(device_id), C.byref(self._h))
        del blob, weights
        self._lib.check(st, None)
        io = _lib.IoDesc()
        self._lib.check(self._lib.c.ade_get_io(self._h, C.byref(io)), self._h)
        self.in_len, self.out_len, self.frames = io.in_len, io.out_len, io.frames
        self.out_channels = io.out_channels
        self.channels = io.in_channels                    # 1 (GTCRN, DFSMN) or 2 (Mel-Band-Roformer stereo)
        self.n_outputs = io.n_outputs                     # 1, or 2 for MossFormer2-SS ("separated_0", "separated_1")
        self.row_in, self.row_out = io.in_channels * io.in_len, io.n_outputs * io.out_channels * io.out_len   # one batch item, planar
        self.sample_rate = io.model_sample_rate
        self.in_sample_rate, self.out_sample_rate = io.in_sample_rate, io.out_sample_rate
        self.device_id = io.device
        in_name = "mix_audio" if reader.string("model_family", "") == "mossformer2_ss" else INPUT_NAME        # :688
        # audio tensor dtypes of the export (Export_GTCRN.py:47-48, 757-760): INT16 PCM, or normalised F32 / F16 (an F16 tensor crosses libade's ABI as IEEE half: ade_process_f16)
        self.in_dtype = {"INT16": np.int16, "F32": np.float32, "F16": np.float16}[reader.string("input_audio_dtype", "INT16")]
        self.out_dtype = {"INT16": np.int16, "F32": np.float32, "F16": np.float16}[reader.string("output_audio_dtype", "INT16")]
        tname = {np.int16: "tensor(int16)", np.float32: "tensor(float)", np.float16: "tensor(float16)"}
        self._inputs = [NodeArg(in_name, [1, io.in_channels, io.in_len], tname[self.in_dtype])]
        out_names = [OUTPUT_NAME] if io.n_outputs == 1 else [f"separated_{i}" for i in range(io.n_outputs)]   # Export_MossFormer2_SS_16K.py:689-690
        self._outputs = [NodeArg(name, [1, io.out_channels, io.out_len], tname[self.out_dtype]) for name in out_names]
        self._inputs_meta, self._outputs_meta = self._inputs, self._outputs   # names the reference script touches
        validate_audio_metadata(reader, self)

    # -- ORT-shaped surface --------------------------------------------------------------------------------
    def get_inputs(self) -> List[NodeArg]:
        return self._inputs

    def get_outputs(self) -> List[NodeArg]:
        return self._outputs

    def get_modelmeta(self) -> ModelMeta:
        return ModelMeta(self.metadata.metadata)

    def get_providers(self) -> List[str]:
        return ["AdeMI355XExecutionProvider"]

    def run(self, output_names, input_feed: Dict[str, np.ndarray], return_f32: bool = False):
        """``session.run(None, {"noisy_audio": int16 (B,1,L)})`` -> ``[int16 (B,1,L_out)]`` (+ fp32 pre-PCM tap)."""
        if self._inputs[0].name not in input_feed:
            raise KeyError(f"missing input {self._inputs[0].name!r}")
        x = np.asarray(input_feed[self._inputs[0].name])
        if x.dtype != self.in_dtype:
            raise ValueError(f"{INPUT_NAME} must be {np.dtype(self.in_dtype).name}, got {x.dtype}")
        if x.ndim != 3 or x.shape[1] != self.channels or x.shape[2] != self.in_len:
            raise ValueError(f"{INPUT_NAME} must have shape (B, {self.channels}, {self.in_len}), got {x.shape}")
        float_out = self.out_dtype != np.int16
        if (self.in_dtype == np.float16 or self.out_dtype == np.float16) and self.in_dtype != np.float32 and not return_f32:
            # a float16 graph tensor on either side: the half entry (the engine widens / narrows on the device)
            pcm, f16 = self.process_f16(x.reshape(x.shape[0], self.row_in), want_pcm=not float_out, want_f16=float_out)
            if float_out:
                f16 = f16.reshape(-1, self.n_outputs, self.out_channels, self.out_len)
                return [np.ascontiguousarray(f16[:, i]) for i in range(self.n_outputs)]
            pcm = pcm.reshape(-1, self.n_outputs, self.out_channels, self.out_len)
            return [np.ascontiguousarray(pcm[:, i]) for i in range(self.n_outputs)]
        if self.in_dtype == np.int16:
            pcm, f32 = self.process(x.reshape(x.shape[0], self.row_in), want_f32=return_f32 or float_out)
        else:
            pcm, f32 = self.process_f32(x.reshape(x.shape[0], self.row_in).astype(np.float32, copy=False), want_pcm=not float_out, want_f32=return_f32 or float_out)
        if float_out:        # the export's F32 / F16 output: the waveform without the PCM scale and clamp (Export_GTCRN.py:689-693)
            f32 = f32.reshape(-1, self.n_outputs, self.out_channels, self.out_len)
            return [np.ascontiguousarray(f32[:, i]).astype(self.out_dtype, copy=False) for i in range(self.n_outputs)]
        pcm = pcm.reshape(-1, self.n_outputs, self.out_channels, self.out_len)
        out = [np.ascontiguousarray(pcm[:, i]) for i in range(self.n_outputs)]               # one array per graph output
        if return_f32:
            f32 = f32.reshape(-1, self.n_outputs, self.out_channels, self.out_len)
            out += [np.ascontiguousarray(f32[:, i]) for i in range(self.n_outputs)]
        return out

    # -- batch call on host buffers ---------------------------------------------------------------------------
    def process(self, pcm: np.ndarray, want_f32: bool = False):
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        if pcm.ndim != 2 or pcm.shape[1] != self.row_in:
            raise ValueError(f"expected int16 (B, {self.row_in}), got {pcm.shape}")
        B = pcm.shape[0]
        out = np.empty((B, self.row_out), np.int16)
        f32 = np.empty((B, self.row_out), np.float32) if want_f32 else None
        st = self._lib.c.ade_process(self._h, pcm.ctypes.data, B, out.ctypes.data, f32.ctypes.data if want_f32 else None)
        self._lib.check(st, self._h)
        return out, f32

    def process_f16(self, x: np.ndarray, want_pcm: bool = True, want_f16: bool = True):
        """IEEE-half audio tensors (``ade_process_f16``): x (B, row_in) in the handle's input dtype -- float16 for a float-input manifest, int16 PCM otherwise --
        -> (int16 PCM or None, float16 waveform or None)."""
        want = np.int16 if self.in_dtype == np.int16 else np.float16
        x = np.ascontiguousarray(x, dtype=want)
        if x.ndim != 2 or x.shape[1] != self.row_in:
            raise ValueError(f"input must have shape (B, {self.row_in}), got {x.shape}")
        if not (want_pcm or want_f16):
            raise ValueError("nothing to compute")
        B = x.shape[0]
        out = np.empty((B, self.row_out), np.int16) if want_pcm else None
        f16 = np.empty((B, self.row_out), np.float16) if want_f16 else None
        st = self._lib.c.ade_process_f16(self._h, x.ctypes.data, B, out.ctypes.data if want_pcm else None, f16.ctypes.data if want_f16 else None)
        self._lib.check(st, self._h)
        return out, f16

    def process_f32(self, x: np.ndarray, want_pcm: bool = True, want_f32: bool = True):
        """fp32 audio in (input_audio_dtype F32 / F16; ``ade_process_f32``): normalised samples (B, row_in) -> (int16 PCM or None, fp32 waveform or None)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.row_in:
            raise ValueError(f"expected float32 (B, {self.row_in}), got {x.shape}")
        if not (want_pcm or want_f32):
            raise ValueError("ask for at least one output")
        B = x.shape[0]
        out = np.empty((B, self.row_out), np.int16) if want_pcm else None
        f32 = np.empty((B, self.row_out), np.float32) if want_f32 else None
        st = self._lib.c.ade_process_f32(self._h, x.ctypes.data, B, out.ctypes.data if want_pcm else None, f32.ctypes.data if want_f32 else None)
        self._lib.check(st, self._h)
        return out, f32

    # -- batch call on device buffers (torch tensors on this session's GPU) ------------------------------------
    def process_into(self, pcm: np.ndarray, out: np.ndarray, f32: Optional[np.ndarray] = None) -> None:
        """``process`` on caller-owned, re-used buffers -- the reference's io_binding pattern (Inference_GTCRN_ONNX.py:307-317): nothing is
        allocated per call.  Page-locked buffers (hipHostMalloc / a pinned torch tensor's ``.numpy()``) make both copies plain DMA."""
        B = pcm.shape[0]
        if pcm.dtype != np.int16 or out.dtype != np.int16 or pcm.shape != (B, self.row_in) or out.shape != (B, self.row_out):
            raise ValueError(f"expected int16 (B, {self.row_in}) -> int16 (B, {self.row_out})")
        if not pcm.flags.c_contiguous or not out.flags.c_contiguous or (f32 is not None and not f32.flags.c_contiguous):
            raise ValueError("buffers must be C-contiguous")
        if f32 is not None and (f32.dtype != np.float32 or f32.shape != out.shape):
            raise ValueError("f32 must be float32 with the shape of out")
        st = self._lib.c.ade_process(self._h, pcm.ctypes.data, B, out.ctypes.data, f32.ctypes.data if f32 is not None else None)
        self._lib.check(st, self._h)

    # -- the pipelined form (ade_submit / ade_wait): a file of many batches overlaps copy-in (k + 1), kernels (k) and copy-out (k - 1) -------------------------------
    def submit(self, pcm: np.ndarray, out: np.ndarray, f32: Optional[np.ndarray] = None) -> int:
        """Enqueue one ``process_into`` call without waiting; returns the ticket ``wait`` takes.  The three buffers must stay alive and untouched until that
        ``wait`` -- this object keeps references to them meanwhile.  At most ``pipe_depth`` (option, default 3) tickets between waits."""
        B = pcm.shape[0]
        if pcm.dtype != np.int16 or out.dtype != np.int16 or pcm.shape != (B, self.row_in) or out.shape != (B, self.row_out):
            raise ValueError(f"expected int16 (B, {self.row_in}) -> int16 (B, {self.row_out})")
        if not pcm.flags.c_contiguous or not out.flags.c_contiguous or (f32 is not None and (not f32.flags.c_contiguous or f32.dtype != np.float32 or f32.shape != out.shape)):
            raise ValueError("buffers must be C-contiguous (f32: float32 with the shape of out)")
        t = C.c_uint64(0)
        st = self._lib.c.ade_submit(self._h, pcm.ctypes.data, B, out.ctypes.data, f32.ctypes.data if f32 is not None else None, C.byref(t))
        self._lib.check(st, self._h)
        self._in_flight[int(t.value)] = (pcm, out, f32)
        return int(t.value)

    def wait(self, ticket: int):
        """Block until the submission is complete; returns its (out, f32) buffers.  Raises what ``process`` would have raised for that call."""
        st = self._lib.c.ade_wait(self._h, C.c_uint64(int(ticket)))
        bufs = self._in_flight.pop(int(ticket), None)
        self._lib.check(st, self._h)
        return (bufs[1], bufs[2]) if bufs else (None, None)

    def run_device(self, d_in, d_out, d_f32=None, stream: Optional[int] = None) -> None:
        """``d_in`` int16 (B, L) / ``d_out`` int16 (B, L_out) CUDA(HIP) tensors; enqueues on ``stream`` (a raw
        hipStream_t handle, e.g. ``torch.cuda.current_stream().cuda_stream``) or runs synchronously when None.
        A model whose input tensor is float (input_audio_dtype F32 / F16) takes a float32 ``d_in`` (``ade_process_device_f32``); ``d_out`` may be None when only the
        float waveform ``d_f32`` is wanted (a float OUTPUT model)."""
        B = int(d_in.shape[0])
        if tuple(d_in.shape) != (B, self.row_in) or (d_out is not None and tuple(d_out.shape) != (B, self.row_out)):
            raise ValueError("device tensors must be (B, channels * in_len) -> (B, channels * out_len)")
        if not d_in.is_contiguous() or (d_out is not None and not d_out.is_contiguous()):
            raise ValueError("device tensors must be contiguous")
        float_in = self.in_dtype != np.int16
        if str(d_in.dtype) != ("torch.float32" if float_in else "torch.int16") or (d_out is not None and str(d_out.dtype) != "torch.int16"):
            raise ValueError(f"d_in must be {'float32' if float_in else 'int16'} for this model, d_out int16")
        if d_out is None and d_f32 is None:
            raise ValueError("ask for at least one output")
        f32_ptr = None
        if d_f32 is not None:
            if tuple(d_f32.shape) != (B, self.row_out) or not d_f32.is_contiguous() or str(d_f32.dtype) != "torch.float32":
                raise ValueError("d_f32 must be a contiguous (B, channels * out_len) float32 tensor")
            f32_ptr = C.c_void_p(d_f32.data_ptr())
        fn = self._lib.c.ade_process_device_f32 if float_in else self._lib.c.ade_process_device
        st = fn(self._h, C.c_void_p(d_in.data_ptr()), B, C.c_void_p(d_out.data_ptr()) if d_out is not None else None, f32_ptr,
                C.c_void_p(stream) if stream else None)
        self._lib.check(st, self._h)

    def reserve(self, batch: int) -> None:
        self._lib.check(self._lib.c.ade_reserve(self._h, int(batch)), self._h)

    def set_option(self, key: str, value: str) -> None:
        self._lib.check(self._lib.c.ade_set_option(self._h, key.encode(), str(value).encode()), self._h)

    # -- parity / timing taps ---------------------------------------------------------------------------------
    def tap(self, name: str, count: int) -> np.ndarray:
        buf = np.empty(int(count), np.float32)
        n = C.c_size_t()
        self._lib.check(self._lib.c.ade_debug_tap(self._h, name.encode(), buf.ctypes.data, buf.size, C.byref(n)), self._h)
        return buf[: n.value]

    def profile(self, enable) -> None:
        """0/False: off; 1/True: one kernel per stage (+ phase clocks); 2: the shipped launch sequence timed as launched;
        3: the single-launch kernel's phase-clock build (tap ``phase_clock`` with 640 slots)."""
        self._lib.check(self._lib.c.ade_profile_last(self._h, int(enable)), self._h)

    def kernel_times(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for i in range(self._lib.c.ade_kernel_count(self._h)):
            ms, n = C.c_float(), C.c_int()
            self._lib.check(self._lib.c.ade_kernel_ms(self._h, i, C.byref(ms), C.byref(n)), self._h)
            out[self._lib.c.ade_kernel_name(self._h, i).decode()] = {"ms": float(ms.value), "launches": int(n.value)}
        return out

    # -- STFT_Process operator on device tensors ---------------------------------------------------------------
    def stft_device(self, d_x, d_spec, stream: Optional[int] = None) -> None:
        B, L = int(d_x.shape[0]), int(d_x.shape[1])
        st = self._lib.c.ade_stft_forward(self._h, C.c_void_p(d_x.data_ptr()), B, L, C.c_void_p(d_spec.data_ptr()),
                                          C.c_void_p(stream) if stream else None)
        self._lib.check(st, self._h)

    def istft_device(self, d_spec, d_y, stream: Optional[int] = None) -> None:
        B, T = int(d_spec.shape[0]), int(d_spec.shape[2])
        st = self._lib.c.ade_istft_forward(self._h, C.c_void_p(d_spec.data_ptr()), B, T, C.c_void_p(d_y.data_ptr()),
                                           C.c_void_p(stream) if stream else None)
        self._lib.check(st, self._h)

    def close(self) -> None:
        for ref in list(getattr(self, "_streams", ())):          # child streams first: they hold device state of this engine
            child = ref()
            if child is not None:
                child.close()
        self._streams = []
        if getattr(self, "_h", None) and self._h:
            self._lib.c.ade_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class StreamingSession:
    """Stateful streaming over a GTCRN session (``ade_stream_*``, include/ade.h).

    ``n_streams`` independent audio streams advance together, ``frames_per_push`` hops (x 256 samples) at a time.  The STFT overlap,
    the causal-convolution histories and every GRU hidden state are carried on the device between pushes, so the concatenated outputs
    equal what the reference's graph computes on the whole signal in ONE call -- one hop (256 samples) later, the stream's first hop
    zero, and without the reference's whole-call DC removal (not causal).  The reference itself has no such mode: its driver calls a
    stateless graph per slice (Inference_GTCRN_ONNX.py:307-317)."""

    def __init__(self, session: InferenceSession, n_streams: int, frames_per_push: int):
        self._lib, self._session = session._lib, session
        self.n_streams, self.frames_per_push, self.samples_per_push = int(n_streams), int(frames_per_push), int(frames_per_push) * 256
        self._h = C.c_void_p()
        self._lib.check(self._lib.c.ade_stream_create(session._h, self.n_streams, self.frames_per_push, C.byref(self._h)), session._h)
        import weakref
        if not hasattr(session, "_streams"):
            session._streams = []
        session._streams.append(weakref.ref(self))

    def push(self, pcm: np.ndarray, want_f32: bool = False):
        """int16 (n_streams, samples_per_push) -> int16 of the same shape (+ fp32 pre-PCM waveform)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        if pcm.shape != (self.n_streams, self.samples_per_push):
            raise ValueError(f"expected int16 ({self.n_streams}, {self.samples_per_push}), got {pcm.shape}")
        out = np.empty_like(pcm)
        f32 = np.empty(pcm.shape, np.float32) if want_f32 else None
        st = self._lib.c.ade_stream_push(self._h, pcm.ctypes.data, out.ctypes.data, f32.ctypes.data if want_f32 else None)
        self._lib.check(st, self._session._h)
        return (out, f32) if want_f32 else out

    def push_device(self, d_in, d_out, d_f32=None, stream: Optional[int] = None) -> None:
        if tuple(d_in.shape) != (self.n_streams, self.samples_per_push) or tuple(d_out.shape) != tuple(d_in.shape):
            raise ValueError("device tensors must be (n_streams, samples_per_push) int16")
        st = self._lib.c.ade_stream_push_device(self._h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()),
                                                C.c_void_p(d_f32.data_ptr()) if d_f32 is not None else None, C.c_void_p(stream) if stream else None)
        self._lib.check(st, self._session._h)

    def flush(self, want_f32: bool = False):
        """End of the signal: the last hop, int16 (n_streams, 256).  ``concatenate(pushes + [flush])[:, 256:]`` is then exactly the one-shot
        output of the whole signal.  ``reset()`` before pushing again."""
        out = np.empty((self.n_streams, 256), np.int16)
        f32 = np.empty((self.n_streams, 256), np.float32) if want_f32 else None
        self._lib.check(self._lib.c.ade_stream_flush(self._h, out.ctypes.data, f32.ctypes.data if want_f32 else None), self._session._h)
        return (out, f32) if want_f32 else out

    def reset(self) -> None:
        self._lib.check(self._lib.c.ade_stream_reset(self._h), self._session._h)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.c.ade_stream_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

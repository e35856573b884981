"""ctypes binding of libpgan_hip.so (the C-ABI declared in include/pgan_hip.h) and HipOps, the op
backend the engine drives.  There is no fallback: if the library is missing, import fails; if an
entry point rejects its arguments or HIP reports a launch error, PganError is raised.

Tensors are torch CUDA tensors used as plain device buffers (data_ptr + shape); every call only
enqueues kernels on torch's current stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_longlong, c_void_p
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGAN_HIP_LIBRARY names another build of the same library (debugging: tools/dbg/racefind5.py loads a packed-fp32 build)
LIB_PATH = os.environ.get("PGAN_HIP_LIBRARY") or os.path.join(_HERE, "libpgan_hip.so")


class PganError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(LIB_PATH)
    # Device code built with packed-fp32 vector instructions returned wrong values beside other streams' kernels on this
    # part: a library that does not state the build without them is refused (PGAN_PACKED_FP32=1: comparison builds).
    try:
        lib.pgan_build_info.restype = ctypes.c_char_p
        info = (lib.pgan_build_info() or b"").decode()
    except AttributeError:
        info = "unknown (no pgan_build_info)"
    if "packed-fp32-ops=off" not in info and os.environ.get("PGAN_PACKED_FP32", "0") != "1":
        raise ImportError(f"{LIB_PATH} was built as '{info}': rebuild it through __graft_entry__.build(), which compiles "
                          f"the device code without packed-fp32 instructions")
    return lib


_lib = _load()

P, I, L, F = c_void_p, c_int, c_longlong, c_float
_SIGS = {
    "pgan_conv_down": [P, P, P, I, I, I, I, P],
    "pgan_conv_down_gn_sums": [P, P, P, I, I, I, I, P, L, P],
    "pgan_conv_up": [P, P, P, I, I, I, I, I, P],
    "pgan_conv_wgrad": [P, P, P, P, L, I, I, I, I, I, I, I, P],
    "pgan_conv_x3_prep": [P, P, I, I, P],
    "pgan_conv_down_x3": [P, P, P, I, I, I, I, P, I, I, P],
    "pgan_conv_up_x3": [P, P, P, I, I, I, I, P, I, I, P],
    "pgan_conv_wgrad_x3": [P, P, P, P, L, I, I, I, I, I, I, I, P, I, I, P, I, I, P],
    "pgan_conv_h2_prep": [P, P, I, I, P],
    "pgan_conv_h2_forget": [P, P],
    "pgan_conv_h2_prep_multi": [P, P, P, P, I, P],
    "pgan_conv_x3_prep_multi": [P, P, P, P, I, P],
    "pgan_conv_down_h2": [P, P, P, I, I, I, I, P, I, I, P],
    "pgan_conv_down_h2_zeroed": [P, P, P, I, I, I, I, P, I, I, P],
    "pgan_conv_down_x3_zeroed": [P, P, P, I, I, I, I, P, I, I, P],
    "pgan_conv_down_h2_adds": [I, I, I, I],
    "pgan_conv_down_x3_adds": [I, I, I, I],
    "pgan_zero_multi": [P, P, I, P],
    "pgan_conv_up_h2": [P, P, P, I, I, I, I, P, I, I, P],
    "pgan_conv_wgrad_h2": [P, P, P, P, L, I, I, I, I, I, I, I, P, I, I, P, I, I, P],
    "pgan_rows_reduce": [P, I, P, I, I, I, I, I, I, P],
    "pgan_gn_fwd": [P, P, P, P, P, P, L, I, I, I, P, P],
    "pgan_gn_fwd_presummed": [P, P, P, P, P, P, L, I, I, I, P, P],
    "pgan_gn_bwd": [P, P, P, P, P, P, P, P, P, P, I, P, L, I, I, I, P, P],
    "pgan_gn_bwd_onepass": [P, P, P, P, P, P, P, P, P, P, I, P, L, I, I, I, P, P],
    "pgan_gn_onepass_workspace_bytes": [I],
    "pgan_gn_onepass_workspace_init": [P, L, I, P],
    "pgan_gn_onepass_workspace_error": [P, I, P],
    "pgan_rows_reduce_multi": [P, P, P, P, I, I, I, I, I, P],
    "pgan_gn_jvp": [P, P, P, P, P, P, P, P, P, L, I, I, I, P, P],
    "pgan_gn_bwd2": [P, P, P, P, P, P, P, P, P, P, P, P, P, I, P, L, I, I, I, P, P],
    "pgan_bn_fwd": [P, P, P, P, P, P, P, P, L, I, I, I, F, I, P, P],
    "pgan_bn_bwd": [P, P, P, P, P, P, P, P, P, L, I, I, I, P, P],
    "pgan_bn_fwd_sums": [P, P, I, I, I, I, P, P],
    "pgan_bn_fwd_apply": [P, P, P, P, P, P, P, P, I, I, I, I, F, I, P, P],
    "pgan_bn_bwd_sums": [P, P, P, P, P, I, I, I, P, P],
    "pgan_bn_bwd_apply": [P, P, P, P, P, P, P, P, P, I, I, I, I, P, P],
    "pgan_head_fwd": [P, P, P, P, P, I, I, I, P],
    "pgan_head_seed": [P, P, P, I, I, P],
    "pgan_head_wgrad": [P, P, P, I, I, P],
    "pgan_interpolate": [P, P, P, P, I, I, P],
    "pgan_gp_penalty": [P, P, P, P, L, F, P],
    "pgan_div_penalty": [P, P, P, P, I, I, P],
    "pgan_critic_loss": [P, P, P, I, P],
    "pgan_sum_scale": [P, P, I, F, P],
    "pgan_act_bwd": [P, P, P, L, I, P],
    "pgan_affine": [P, P, L, F, F, P],
    "pgan_adam": [P, P, P, P, L, F, F, F, F, F, P, I, P],
    "pgan_philox": [P, L, P, I, P],
    "pgan_philox_slice": [P, L, L, L, P, I, P],
    "pgan_take_batch": [P, P, P, P, P, I, I, I, I, P],
    "pgan_take_batch_slice": [P, P, P, P, P, I, I, I, I, I, I, P],
    "pgan_label_smooth": [P, L, F, F, I, P, P, L, L, P],
    "pgan_mono_switch": [P, P, I, I, I, I, I, P, P, P, P, I, I, P],
    "pgan_event_record": [P, P],
    "pgan_stream_wait_event": [P, P],
    "pgan_stream_edge": [P, P, P],
    "pgan_d1d_fwd": [P, P, P, P, P, P, I, P],
    "pgan_d1d_bwd": [P, P, P, P, P, P, P, P, I, P, P, I, P],
    "pgan_d1d_fwd_len": [P, P, P, P, P, P, I, I, P],
    "pgan_d1d_bwd_len": [P, P, P, P, P, P, P, P, I, P, P, I, I, P],
    "pgan_ct1d_up": [P, P, P, I, I, I, I, I, P],
    "pgan_ct1d_down": [P, P, P, I, I, I, I, P],
    "pgan_ct1d_wgrad": [P, P, P, I, I, I, I, P],
    "pgan_gemm": [P, P, P, I, I, I, I, P],
    "pgan_parse_alignments": [P, P, I, P, P, P, I],
    "pgan_unpack_sites": [P, P, L, F, F, P],
    "pgan_pack_sites": [P, P, L, P],
    "pgan_gn_fwd_static": [P, P, P, P, P, I, I, I, P],
    "pgan_fakebuf_step": [P, P, I, P, P, P, P, P, P, I, I, I, P, P, P],
    "pgan_sample_genotypes": [P, P, L, I, I, P, I, P],
    "pgan_parse_ms": [ctypes.c_char_p, L, I, I, I, P, P],
    "pgan_step_graph_begin": [P],
    "pgan_step_graph_end": [P, ctypes.POINTER(ctypes.c_void_p)],
    "pgan_step_graph_launch": [P, P],
    "pgan_step_graph_destroy": [P],
    "pgan_event_create": [ctypes.POINTER(ctypes.c_void_p)],
    "pgan_event_destroy": [P],
    "pgan_tape_destroy": [P],
    "pgan_tape_arg_count": [ctypes.c_char_p],
    "pgan_tape_append": [P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_ulonglong), I],
    "pgan_tape_length": [P],
    "pgan_tape_replay": [P, ctypes.POINTER(c_int)],
}
EXPORTS = sorted(list(_SIGS) + ["pgan_conv_wgrad_workspace_floats", "pgan_set_option", "pgan_x3_set_option",
                                "pgan_conv_x3_prep_bytes", "pgan_conv_wgrad_x3_workspace_floats", "pgan_h2_set_option",
                                "pgan_conv_h2_prep_bytes", "pgan_conv_wgrad_h2_workspace_floats",
                                "pgan_gn_workspace_bytes", "pgan_bn_workspace_bytes", "pgan_bn_partials",
                                "pgan_bn_fwd_partials", "pgan_gn_fwd_static_max_samples", "pgan_build_info",
                                "pgan_events_created", "pgan_tape_create", "pgan_tape_entry_name", "pgan_gn_fwd_static_init"])
for _name, _args in _SIGS.items():
    _fn = getattr(_lib, _name)
    _fn.argtypes = _args
    _fn.restype = c_int
_lib.pgan_conv_wgrad_workspace_floats.argtypes = [I, I, I, I]
_lib.pgan_conv_wgrad_workspace_floats.restype = c_longlong
_lib.pgan_bn_workspace_bytes.argtypes = [I]
_lib.pgan_bn_workspace_bytes.restype = c_longlong
_lib.pgan_bn_partials.argtypes = [I, I, I]
_lib.pgan_bn_partials.restype = c_longlong
_lib.pgan_bn_fwd_partials.argtypes = [I, I, I, I]
_lib.pgan_bn_fwd_partials.restype = c_longlong
_lib.pgan_gn_fwd_static_max_samples.argtypes = []
_lib.pgan_gn_fwd_static_max_samples.restype = c_int
_lib.pgan_gn_fwd_static_init.argtypes = []
_lib.pgan_gn_fwd_static_init.restype = c_int
_lib.pgan_gn_workspace_bytes.argtypes = [I]
_lib.pgan_gn_workspace_bytes.restype = c_longlong
_lib.pgan_gn_onepass_workspace_bytes.restype = c_longlong
_lib.pgan_events_created.argtypes = []
_lib.pgan_events_created.restype = c_longlong
_lib.pgan_tape_create.argtypes = []
_lib.pgan_tape_create.restype = c_void_p
_lib.pgan_tape_entry_name.argtypes = [P, I]
_lib.pgan_tape_entry_name.restype = ctypes.c_char_p
for _fn in (_lib.pgan_conv_wgrad_x3_workspace_floats, _lib.pgan_conv_wgrad_h2_workspace_floats):
    _fn.argtypes = [I, I, I, I]
    _fn.restype = c_longlong
for _fn in (_lib.pgan_conv_x3_prep_bytes, _lib.pgan_conv_h2_prep_bytes):
    _fn.argtypes = [I, I]
    _fn.restype = c_longlong
for _fn in (_lib.pgan_set_option, _lib.pgan_x3_set_option, _lib.pgan_h2_set_option):
    _fn.argtypes = [ctypes.c_char_p, c_int]
    _fn.restype = c_int

import pgan_tape  # noqa: E402

pgan_tape.error_type = PganError


class _Enqueue:
    """An entry point that enqueues work on a stream: callable like the ctypes function; while a launch tape is being
    recorded (pgan_tape.tape) the call is also noted there with its resolved arguments."""
    __slots__ = ("fn",)

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *args):
        t = pgan_tape.tape
        if t is not None:
            t.append((self.fn, args, True))
        return self.fn(*args)


class _Lib:
    """The loaded library with its enqueue entry points wrapped for tape recording (queries and host-side functions
    pass through untouched)."""
    _HOST = ("pgan_parse_alignments", "pgan_parse_ms", "pgan_step_graph_begin", "pgan_step_graph_end", "pgan_step_graph_launch",
             "pgan_step_graph_destroy", "pgan_event_create", "pgan_event_destroy", "pgan_tape_destroy", "pgan_tape_arg_count",
             "pgan_tape_append", "pgan_tape_length", "pgan_tape_replay", "pgan_gn_onepass_workspace_bytes",
             "pgan_gn_onepass_workspace_init", "pgan_gn_onepass_workspace_error", "pgan_conv_down_h2_adds",
             "pgan_conv_down_x3_adds")

    def __init__(self, lib):
        self._real = lib
        for name in _SIGS:
            if name not in self._HOST:
                setattr(self, name, _Enqueue(getattr(lib, name)))

    def __getattr__(self, name):          # everything that was not wrapped above
        return getattr(self._real, name)


_lib = _Lib(_lib)


class Event:
    """An event of the library (created once, reused): the marks and fork / join edges between the engine's streams.  Edges
    made through these calls create and destroy nothing, so they are safe while a graph capture is open, and a launch
    tape notes them like any other enqueue call."""
    __slots__ = ("handle",)

    def __init__(self):
        h = ctypes.c_void_p()
        _check(_lib.pgan_event_create(ctypes.byref(h)), "pgan_event_create")
        self.handle = h.value

    def record(self, stream: int) -> None:
        _check(_lib.pgan_event_record(self.handle, stream), "pgan_event_record")

    def wait(self, stream: int) -> None:
        """Work enqueued on `stream` (raw handle) from now on waits for the event's last record."""
        _check(_lib.pgan_stream_wait_event(stream, self.handle), "pgan_stream_wait_event")

    def edge(self, src: int, dst: int) -> None:
        """Work enqueued on dst from now on waits for everything src holds now."""
        _check(_lib.pgan_stream_edge(self.handle, src, dst), "pgan_stream_edge")

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            _lib.pgan_event_destroy(h)


def events_created() -> int:
    """Events the library has created so far (a capture asserts that the count does not move while it is open)."""
    return int(_lib.pgan_events_created())


def _arg_word(kind, v) -> int:
    """One tape word for argument v of ctypes kind `kind` (see pgan_tape_append)."""
    if kind is c_float:
        import struct
        return struct.unpack("<I", struct.pack("<f", float(v)))[0]
    if kind is c_void_p:
        if v is None:
            return 0
        if isinstance(v, int):
            return v & 0xFFFFFFFFFFFFFFFF
        if isinstance(v, ctypes.c_void_p):
            return (v.value or 0) & 0xFFFFFFFFFFFFFFFF
        return ctypes.addressof(v)              # a ctypes array / structure kept alive by the tape's entry list
    return int(v) & 0xFFFFFFFFFFFFFFFF          # c_int / c_longlong: two's complement, sign-extended


class NativeTape:
    """A stretch of library enqueue calls replayed by one pgan_tape_replay call."""

    def __init__(self, entries):
        self.handle = _lib.pgan_tape_create()
        if not self.handle:
            raise PganError("pgan_tape_create failed")
        self.keep = entries                     # host arrays the noted pointers refer to
        for fn, args, _ in entries:
            name = fn.__name__
            kinds = _SIGS[name]
            words = (ctypes.c_ulonglong * len(kinds))(*[_arg_word(k, a) for k, a in zip(kinds, args)])
            _check(_lib.pgan_tape_append(self.handle, name.encode(), words, len(kinds)), f"pgan_tape_append({name})")
        self._failed = c_int(-1)

    @staticmethod
    def accepts(fn) -> bool:
        name = getattr(fn, "__name__", None)
        return name in _SIGS and _lib.pgan_tape_arg_count(name.encode()) == len(_SIGS[name])

    def __len__(self):
        return int(_lib.pgan_tape_length(self.handle))

    def replay(self) -> None:
        rc = _lib.pgan_tape_replay(self.handle, ctypes.byref(self._failed))
        if rc:
            name = _lib.pgan_tape_entry_name(self.handle, self._failed.value)
            raise PganError(f"{(name or b'?').decode()} (call {self._failed.value} of the tape) failed with code {rc} "
                            f"during tape replay")

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            _lib.pgan_tape_destroy(h)


pgan_tape.native_tape = NativeTape


def set_option(name: str, value: int) -> None:
    """Kernel-variant override for tuning runs (0 = size-based choice)."""
    _check(_lib.pgan_set_option(name.encode(), int(value)), f"pgan_set_option({name}, {value})")


def set_x3_option(name: str, value: int) -> None:
    """Tile override of the split-operand convolution kernels (both splits) for tuning runs (0 = size-based choice)."""
    _check(_lib.pgan_x3_set_option(name.encode(), int(value)), f"pgan_x3_set_option({name}, {value})")
    _check(_lib.pgan_h2_set_option(name.encode(), int(value)), f"pgan_h2_set_option({name}, {value})")


GN_AMAX_SLOTS = 8      # entries per sample that the GroupNorm ops write into an `amax` tensor ([N, 8])


def _bound(t: Optional[torch.Tensor], rows: Optional[int] = None):
    """(pointer, entries per sample, entries) of a magnitude bound for an fp16-split convolution operand: None, a
    [N, k] tensor (k bounds per sample, rows aligned with the operand's samples) or a 1-D tensor (every entry bounds
    the whole operand)."""
    if t is None:
        return None, 0, 0
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise PganError("expected a contiguous float32 CUDA tensor as magnitude bound")
    if t.dim() == 2:
        if rows is not None and t.shape[0] < rows:
            raise PganError(f"magnitude bound has {t.shape[0]} rows for {rows} samples")
        return t.data_ptr(), t.shape[1], t.numel()
    return t.data_ptr(), 0, t.numel()


def _amax_out(t: Optional[torch.Tensor], need: int):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() < need:
        raise PganError(f"expected a contiguous float32 CUDA tensor of at least {need} elements for amax")
    return t.data_ptr()


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise PganError(f"expected a contiguous fp32 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def _rows_ptr(t: torch.Tensor):
    """Pointer of a 2-D fp32 view whose rows are contiguous (columns of a wider row-major buffer)."""
    if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
        raise PganError("expected a 2-D fp32 CUDA view with unit column stride")
    return t.data_ptr()


_DEVICE_INDEX = [0]                 # set by HipOps: one device per process
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Handle of torch's current stream on the op backend's device (every launch asks: the direct query is ~20x
    cheaper than building a torch.cuda.Stream object)."""
    if _raw_stream is not None:
        return _raw_stream(_DEVICE_INDEX[0])
    return torch.cuda.current_stream(_DEVICE_INDEX[0]).cuda_stream


class StepGraph:
    """A step recorded through the library's own capture entry points (no torch graph machinery): everything the
    `with` block enqueues on the current stream becomes one executable hipGraph; launch() replays it there."""

    def __init__(self):
        self.handle = ctypes.c_void_p()

    def __enter__(self):
        _check(_lib.pgan_step_graph_begin(_stream()), "pgan_step_graph_begin")
        return self

    def __exit__(self, *exc):
        rc = _lib.pgan_step_graph_end(_stream(), ctypes.byref(self.handle))
        if exc[0] is None:
            _check(rc, "pgan_step_graph_end")
        return False

    def launch(self):
        _check(_lib.pgan_step_graph_launch(self.handle, _stream()), "pgan_step_graph_launch")

    def __del__(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            _lib.pgan_step_graph_destroy(self.handle)
            self.handle = ctypes.c_void_p()


def _check(rc: int, what: str):
    if rc != 0:
        why = {-1: "bad shape/argument", -2: "HIP launch error", -3: "size beyond the entry point's fixed maximum"}
        raise PganError(f"{what} failed with code {rc} ({why.get(rc, 'unknown')})")


def _hw(t: torch.Tensor) -> int:
    n = 1
    for d in t.shape[2:]:
        n *= d
    return n


D1_PACKED = (64, 16, 16, 16, 2048, 32, 32, 32, 8192, 64, 64, 64)   # cw cb gw gb per 1-D stage


class HipOps:
    """Same method signatures as tests/torch_ops.TorchOps; every method writes into caller tensors."""
    name = "hip"
    feeds_tape = True       # every enqueue goes through _Enqueue (pgan_tape)

    def __init__(self, device=None, ws_floats: int = 16 * 1024 * 1024, max_samples: int = 1024,
                 precision: Optional[str] = None):
        if not torch.cuda.is_available():
            raise PganError("HipOps needs a GPU")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _DEVICE_INDEX[0] = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.ws = torch.empty(ws_floats, dtype=torch.float32, device=self.device)
        self.ws_small = torch.empty(4096, dtype=torch.float32, device=self.device)
        self.d1_part = torch.zeros(max_samples * 16384, dtype=torch.float32, device=self.device)  # gaps stay 0
        self.max_samples = max_samples
        # Norm-kernel scratch (slice partial sums) is owned per stream: the engine runs chains of GroupNorm and
        # BatchNorm launches concurrently on several streams, and a sums launch of one chain must not overwrite the
        # partial sums another chain's apply launch is about to read.  Streams are registered up front
        # (reserve_stream); a stream first seen while a capture is open is an error, not an allocation.
        self._gn_bytes = int(_lib.pgan_gn_workspace_bytes(4 * max_samples))
        self._gn1_samples = 4 * max_samples
        self._gn1_bytes = int(_lib.pgan_gn_onepass_workspace_bytes(self._gn1_samples))
        self._bn_bytes = int(_lib.pgan_bn_workspace_bytes(512))
        self._scratch = {}
        self.capturing = False
        self.reserve_stream(_stream())
        self._hold = False
        self._keep = []   # host-side arrays that must outlive an enqueue call
        # "x3": 4x4 convolutions as six exact bf16 partial products per fp32 product on the bf16 matrix pipe
        # (fp32 accuracy); "h2": three products of a two-plane fp16 split on the same pipe; "fp32": the fp32 matrix
        # pipe.  PGAN_CONV_PRECISION overrides the default.
        self.precision = precision or os.environ.get("PGAN_CONV_PRECISION", "h2")
        if self.precision not in ("x3", "h2", "fp32"):
            raise PganError(f"unknown convolution precision {self.precision!r}")
        # prepared bf16 planes per weight tensor; static_weights = the caller promises to call
        # invalidate_weight_planes() after every weight update (otherwise planes are rebuilt per call)
        self._planes = {}
        self.static_weights = False
        self.wgrad_x3 = os.environ.get("PGAN_WGRAD_X3", "1") != "0"     # weight gradients on the bf16 pipe too

    def invalidate_weight_planes(self, within: Optional[torch.Tensor] = None, reset: bool = False):
        """Mark prepared weight copies stale: all of them, or those whose weights live inside `within`.
        reset: the weights may be unrelated to the ones prepared before (a load, another tensor at the same address), so
        the fp16 split's running record of their largest magnitude starts again; without it (an optimizer step) the
        record is kept, which holds the scale steady."""
        lo = hi = None
        if within is not None:
            lo = within.data_ptr()
            hi = lo + within.numel() * within.element_size()
        for key, ent in self._planes.items():
            if lo is None or lo <= key[0] < hi:
                ent[1] = False
                ent[2] = ent[2] or reset

    @property
    def split(self) -> bool:
        """Convolutions run on split operands (prepared weight planes) rather than on the fp32 matrix pipe."""
        return self.precision in ("x3", "h2")

    def _planes_entry(self, w):
        Cs, Cl = w.shape[0], w.shape[1]
        h2 = self.precision == "h2"
        key = (w.data_ptr(), Cs, Cl, h2)
        ent = self._planes.get(key)
        if ent is None:
            if self.capturing:
                raise PganError("weight planes were first requested inside a graph capture")
            nbytes = (_lib.pgan_conv_h2_prep_bytes if h2 else _lib.pgan_conv_x3_prep_bytes)(Cs, Cl)
            ent = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device), False, False]   # [planes, fresh, reset]
            self._planes[key] = ent
        return ent

    def _forget(self, ent) -> None:
        """Start the planes buffer's record of the largest |w| again (a library call on the current stream, so a launch
        tape or graph that is being recorded holds it like the preparation that follows)."""
        _check(_lib.pgan_conv_h2_forget(ent[0].data_ptr(), _stream()), "pgan_conv_h2_forget")

    def settle_weight_planes(self) -> None:
        """Apply pending resets of the running weight maxima now, on the current stream.  Called before an iteration is
        recorded (tape or graph) so that the recorded launches hold preparations only: a replay then keeps the maxima
        growing exactly as eager iterations do, whatever happened to the weights before the recording."""
        if self.precision == "h2":
            for ent in self._planes.values():
                if ent[2]:
                    self._forget(ent)
                    ent[2] = False
                    ent[1] = False

    def _planes_stale(self, w) -> bool:
        return not (self._planes_entry(w)[1] and (self.static_weights or self._hold))

    def _weight_planes(self, w):
        Cs, Cl = w.shape[0], w.shape[1]
        h2 = self.precision == "h2"
        ent = self._planes_entry(w)
        keep = self.static_weights or self._hold
        if not (ent[1] and keep):
            if h2 and (ent[2] or not keep):      # header word 0: largest |w| prepared so far (see pgan_conv_h2_prep)
                self._forget(ent)
            ent[2] = False
            prep = _lib.pgan_conv_h2_prep if h2 else _lib.pgan_conv_x3_prep
            _check(prep(_ptr(w), ent[0].data_ptr(), Cs, Cl, _stream()), "pgan_conv_prep")
            ent[1] = True
        return ent[0].data_ptr()

    # ---------------- conv family ----------------
    # Debugging mode (PGAN_CHECK_BOUNDS=1, eager launches only): before every split convolution that is given magnitude
    # bounds, wait for the stream and compare them with the operand's true magnitudes -- a bound below the truth could
    # overflow fp16 inside the kernel without any error.  bounds_checked counts the comparisons made.
    check_bounds = os.environ.get("PGAN_CHECK_BOUNDS", "0") == "1"
    bounds_checked = 0

    def _verify_bound(self, what: str, operand: torch.Tensor, n: int, bound: Optional[torch.Tensor]) -> None:
        if bound is None or not self.check_bounds:
            return
        if self.capturing or pgan_tape.tape is not None:
            raise PganError("PGAN_CHECK_BOUNDS needs eager launches (it synchronises and reads the operands)")
        torch.cuda.current_stream().synchronize()
        x = operand[:n]
        if bound.dim() == 2:      # k entries per sample, each bounding one slice: their maximum bounds the sample
            true = x.abs().reshape(n, -1).amax(dim=1)
            have = bound[:n].amax(dim=1)
        else:                     # every entry bounds the whole operand; the kernels use the largest
            true = x.abs().amax().reshape(1)
            have = bound.amax().reshape(1)
        ok = have >= true         # False for NaN on either side
        HipOps.bounds_checked += 1
        if not bool(ok.all()):
            i = int((~ok).nonzero()[0])
            raise PganError(f"{what}: magnitude bound {float(have[i]):.6g} is below the operand's {float(true[i]):.6g} "
                            f"(row {i} of {tuple(bound.shape)}; operand {tuple(x.shape)})")

    def conv_down_adds(self, N: int, Cs: int, Cl: int, hs: int) -> bool:
        """True if conv_down at this shape adds into its output (a launch that splits its reduction over two blocks per
        tile): such a launch clears the output first with a launch of its own unless conv_down(..., zeroed=True)."""
        if not (self.split and Cl > 1):
            return False
        fn = _lib.pgan_conv_down_h2_adds if self.precision == "h2" else _lib.pgan_conv_down_x3_adds
        return fn(N, Cs, Cl, hs) == 1

    def zero_multi(self, tensors) -> None:
        """Clear up to 16 contiguous tensors in one launch on the current stream."""
        ts = [t for t in tensors if t.numel()]
        for i in range(0, len(ts), 16):
            part = ts[i:i + 16]
            for t in part:
                if not t.is_cuda or not t.is_contiguous() or (t.numel() * t.element_size()) % 16 or t.data_ptr() % 16:
                    raise PganError("zero_multi: contiguous 16-byte aligned CUDA tensors of a multiple of 16 bytes")
            n = len(part)
            pa = (c_void_p * n)(*[t.data_ptr() for t in part])
            ba = (c_longlong * n)(*[t.numel() * t.element_size() for t in part])
            _check(_lib.pgan_zero_multi(pa, ba, n, _stream()), "pgan_zero_multi")

    def conv_down(self, x, w, y, bound=None, zeroed=False):
        """bound: magnitude bounds of x from its producer (see _bound); without them the fp16-split kernel scans x.
        zeroed: y already holds zeros (cleared on a stream this launch is ordered after): a launch that adds into its
        output (conv_down_adds) then does not clear it itself."""
        N, Cl, HL = x.shape[0], x.shape[1], x.shape[2]
        Cs, hs = w.shape[0], HL // 2
        assert tuple(y.shape) == (N, Cs, hs, hs) and w.shape[1] == Cl
        if self.split and Cl > 1:
            self._verify_bound("conv_down", x, N, bound)
            if zeroed:
                fn = _lib.pgan_conv_down_h2_zeroed if self.precision == "h2" else _lib.pgan_conv_down_x3_zeroed
                _check(fn(_ptr(x), self._weight_planes(w), _ptr(y), N, Cs, Cl, hs, *_bound(bound, N), _stream()),
                       "pgan_conv_down (split, zeroed output)")
                return
            fn = _lib.pgan_conv_down_h2 if self.precision == "h2" else _lib.pgan_conv_down_x3
            _check(fn(_ptr(x), self._weight_planes(w), _ptr(y), N, Cs, Cl, hs, *_bound(bound, N), _stream()),
                   "pgan_conv_down (split)")
            return
        _check(_lib.pgan_conv_down(_ptr(x), _ptr(w), _ptr(y), N, Cs, Cl, hs, _stream()), "pgan_conv_down")

    # the discriminator's first stage: convolution that leaves the following GroupNorm's slice sums in this stream's norm
    # scratch (PGAN_FUSE_STAGE1=0: separate sums launch, for comparison)
    fuse_stage1 = os.environ.get("PGAN_FUSE_STAGE1", "1") != "0"

    def conv_down_gn_sums(self, x, w, y) -> bool:
        """conv_down that also leaves the GroupNorm slice partial sums of y in the current stream's norm scratch, where the
        shape allows (then returns True: follow it with gn_fwd(..., presummed=True) on this stream); plain conv_down and
        False otherwise."""
        N, Cl, HL = x.shape[0], x.shape[1], x.shape[2]
        Cs, hs = w.shape[0], HL // 2
        if not (self.fuse_stage1 and Cl == 1 and hs == 32 and Cs == 64):
            self.conv_down(x, w, y)
            return False
        assert tuple(y.shape) == (N, Cs, hs, hs) and w.shape[1] == Cl
        _check(_lib.pgan_conv_down_gn_sums(_ptr(x), _ptr(w), _ptr(y), N, Cs, Cl, hs, *self._gnws(), _stream()),
               "pgan_conv_down_gn_sums")
        return True

    def conv_up(self, y, w, x, act=0, bound=None):
        N, Cs, hs = y.shape[0], y.shape[1], y.shape[2]
        Cl = w.shape[1]
        assert tuple(x.shape) == (N, Cl, 2 * hs, 2 * hs) and w.shape[0] == Cs
        if self.split and Cl > 1 and act == 0:
            self._verify_bound("conv_up", y, N, bound)
            fn = _lib.pgan_conv_up_h2 if self.precision == "h2" else _lib.pgan_conv_up_x3
            _check(fn(_ptr(y), self._weight_planes(w), _ptr(x), N, Cs, Cl, hs, *_bound(bound, N), _stream()),
                   "pgan_conv_up (split)")
            return
        _check(_lib.pgan_conv_up(_ptr(y), _ptr(w), _ptr(x), N, Cs, Cl, hs, act, _stream()), "pgan_conv_up")

    def conv_wgrad(self, ya, x, dw, n_keep=None, n_swap=0, accumulate=False, bound_s=None, bound_l=None):
        """bound_s / bound_l: magnitude bounds of ya / x (rows of a per-sample bound of x follow x's own sample order:
        the kernel applies the same slot swap to them as to x)."""
        N, Cs, hs = ya.shape[0], ya.shape[1], ya.shape[2]
        Cl = x.shape[1]
        if n_keep is None:
            n_keep = N
        assert tuple(dw.shape[:2]) == (Cs, Cl) and x.shape[0] >= N and x.shape[2] == 2 * hs
        if self.split and self.wgrad_x3 and Cl > 1:
            self._verify_bound("conv_wgrad (S operand)", ya, N, bound_s)
            self._verify_bound("conv_wgrad (L operand)", x, N, bound_l)
            h2 = self.precision == "h2"
            need = (_lib.pgan_conv_wgrad_h2_workspace_floats if h2 else _lib.pgan_conv_wgrad_x3_workspace_floats)(N, Cs, Cl, hs)
            if need < 0 or need > self.ws.numel():
                raise PganError(f"conv_wgrad workspace: need {need} floats, have {self.ws.numel()}")
            fn = _lib.pgan_conv_wgrad_h2 if h2 else _lib.pgan_conv_wgrad_x3
            _check(fn(_ptr(ya), _ptr(x), _ptr(dw), _ptr(self.ws), self.ws.numel(), N, Cs, Cl, hs, n_keep, n_swap,
                      int(accumulate), *_bound(bound_s, N), *_bound(bound_l, N), _stream()), "pgan_conv_wgrad (split)")
            return
        need = _lib.pgan_conv_wgrad_workspace_floats(N, Cs, Cl, hs)
        if need < 0 or need > self.ws.numel():
            raise PganError(f"conv_wgrad workspace: need {need} floats, have {self.ws.numel()}")
        _check(_lib.pgan_conv_wgrad(_ptr(ya), _ptr(x), _ptr(dw), _ptr(self.ws), self.ws.numel(), N, Cs, Cl, hs,
                                    n_keep, n_swap, int(accumulate), _stream()), "pgan_conv_wgrad")

    def rows_reduce(self, part, out, r0, r1, r2, r3, accumulate=False):
        C = part.shape[1]
        assert out.numel() == C
        _check(_lib.pgan_rows_reduce(_rows_ptr(part), part.stride(0), _ptr(out), C, r0, r1, r2, r3, int(accumulate),
                                     _stream()), "pgan_rows_reduce")

    def rows_reduce_multi(self, parts, outs, r0, r1, r2, r3):
        """outs[i][c] = sum of parts[i][r, c] over rows [r0,r1) and [r2,r3); up to 4 pairs, one launch."""
        n = len(parts)
        pa = (c_void_p * n)(*[_rows_ptr(t) for t in parts])
        oa = (c_void_p * n)(*[_ptr(t) for t in outs])
        ld = (c_int * n)(*[t.stride(0) for t in parts])
        cc = (c_int * n)(*[t.shape[1] for t in parts])
        for t, o in zip(parts, outs):
            assert o.numel() == t.shape[1]
        _check(_lib.pgan_rows_reduce_multi(pa, ld, oa, cc, n, r0, r1, r2, r3, _stream()), "pgan_rows_reduce_multi")

    # ---------------- GroupNorm family ----------------
    def reserve_stream(self, handle: int) -> None:
        """Allocate the norm-kernel scratch of a stream (raw handle) before work is enqueued on it."""
        if handle not in self._scratch:
            if self.capturing or pgan_tape.tape is not None:
                raise PganError("a stream without reserved scratch was used inside a graph capture or a tape recording")
            one = torch.empty(self._gn1_bytes, dtype=torch.uint8, device=self.device)
            # the one-launch GroupNorm calls exchange partial sums through this buffer: initialised once, restored by every call
            _check(_lib.pgan_gn_onepass_workspace_init(one.data_ptr(), self._gn1_bytes, self._gn1_samples, _stream()),
                   "pgan_gn_onepass_workspace_init")
            torch.cuda.current_stream().synchronize()
            self._scratch[handle] = (torch.empty(self._gn_bytes, dtype=torch.uint8, device=self.device),
                                     torch.empty(self._bn_bytes, dtype=torch.uint8, device=self.device), one)

    def _gnws(self):
        """Slice-sum scratch of the GroupNorm kernels for the current stream."""
        key = _stream()
        ent = self._scratch.get(key)
        if ent is None:
            self.reserve_stream(key)
            ent = self._scratch[key]
        return ent[0].data_ptr(), self._gn_bytes

    # PGAN_GN_ONEPASS=1: the GroupNorm backward of the two large stages in one launch (pgan_gn_bwd_onepass) instead of the
    # sums / apply pair; =2: of every stage the one-launch kernel supports.  Bit-identical results; measured a few percent
    # faster per call on the large stages, slower on the small ones, nothing on the whole iteration: off by default.
    gn_onepass = int(os.environ.get("PGAN_GN_ONEPASS", "0"))

    def _gnws1(self):
        """Exchange workspace of the one-launch GroupNorm kernels for the current stream."""
        key = _stream()
        ent = self._scratch.get(key)
        if ent is None:
            self.reserve_stream(key)
            ent = self._scratch[key]
        return ent[2].data_ptr(), self._gn1_bytes

    def gn_onepass_error(self) -> bool:
        """True if a one-launch GroupNorm call on any stream of this backend ever gave up waiting (synchronises)."""
        bad = False
        for ent in self._scratch.values():
            rc = _lib.pgan_gn_onepass_workspace_error(ent[2].data_ptr(), self._gn1_samples, _stream())
            _check(min(rc, 0), "pgan_gn_onepass_workspace_error")
            bad = bad or rc == 1
        return bad

    def _onepass_shape(self, a) -> bool:
        m = a.shape[1] * _hw(a)
        return m % 8192 == 0 and m // 8192 in ((4, 8) if self.gn_onepass == 1 else (1, 2, 4, 8))

    def _bnws(self):
        """Slice-sum scratch of the BatchNorm kernels for the current stream."""
        key = _stream()
        ent = self._scratch.get(key)
        if ent is None:
            self.reserve_stream(key)
            ent = self._scratch[key]
        return ent[1].data_ptr(), self._bn_bytes

    def prepare_weights(self, weights, hold: bool = True) -> None:
        """Build the prepared copies of these convolution weights now, on the current stream, so that later uses on
        several streams (all ordered after this point) find them ready."""
        if self.split:
            self._hold = False
            todo = [w for w in weights if w.shape[1] > 1 and self._planes_stale(w)]
            if len(todo) > 1:         # one launch pair for all of them
                h2 = self.precision == "h2"
                ents = [self._planes_entry(w) for w in todo]
                for ent in ents:
                    if h2 and (ent[2] or not self.static_weights):
                        self._forget(ent)
                    ent[2] = False
                n = len(todo)
                wa = (c_void_p * n)(*[_ptr(w) for w in todo])
                pa = (c_void_p * n)(*[ent[0].data_ptr() for ent in ents])
                cs = (c_int * n)(*[w.shape[0] for w in todo])
                cl = (c_int * n)(*[w.shape[1] for w in todo])
                fn = _lib.pgan_conv_h2_prep_multi if h2 else _lib.pgan_conv_x3_prep_multi
                _check(fn(wa, pa, cs, cl, n, _stream()), "pgan_conv_prep_multi")
                for ent in ents:
                    ent[1] = True
            else:
                for w in todo:
                    self._weight_planes(w)
            self._hold = bool(hold)   # until release_weights(): no re-preparation from a second stream

    def release_weights(self) -> None:
        self._hold = False

    def gn_fwd(self, a, gamma, beta, h, stats, amax=None, presummed=False):
        """amax (every GroupNorm op): optional [N, GN_AMAX_SLOTS] output, the largest magnitude of each of the eight
        slices of every sample of the op's main output -- what conv_* take as `bound`.
        presummed: the slice sums are already in this stream's norm scratch (conv_down_gn_sums returned True)."""
        N, C = a.shape[0], a.shape[1]
        if presummed:
            _check(_lib.pgan_gn_fwd_presummed(_ptr(a), _ptr(gamma), _ptr(beta), _ptr(h), _ptr(stats), *self._gnws(), N, C,
                                              _hw(a), _amax_out(amax, N * GN_AMAX_SLOTS), _stream()),
                   "pgan_gn_fwd_presummed")
            return
        _check(_lib.pgan_gn_fwd(_ptr(a), _ptr(gamma), _ptr(beta), _ptr(h), _ptr(stats), *self._gnws(), N, C, _hw(a),
                                _amax_out(amax, N * GN_AMAX_SLOTS), _stream()), "pgan_gn_fwd")

    def gn_fwd_static(self, a, gamma, beta, h, stats):
        """gn_fwd with the partial sums in the library's own fixed array (nothing allocated, no workspace)."""
        N, C = a.shape[0], a.shape[1]
        _check(_lib.pgan_gn_fwd_static(_ptr(a), _ptr(gamma), _ptr(beta), _ptr(h), _ptr(stats), N, C, _hw(a), _stream()),
               "pgan_gn_fwd_static")

    def gn_bwd(self, u, h, a, stats, gamma, d, stats_b, pg, pb, amax=None, beta=None):
        """beta (gn_bwd, gn_jvp, gn_bwd2): given, the LeakyReLU mask is recomputed from a and h is not read."""
        N, C = a.shape[0], a.shape[1]
        if self.gn_onepass and self._onepass_shape(a) and N <= self._gn1_samples:
            _check(_lib.pgan_gn_bwd_onepass(_ptr(u), _ptr(h), _ptr(a), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(d),
                                            _ptr(stats_b), _rows_ptr(pg), _rows_ptr(pb), pg.stride(0), *self._gnws1(), N, C,
                                            _hw(a), _amax_out(amax, N * GN_AMAX_SLOTS), _stream()), "pgan_gn_bwd_onepass")
            return
        _check(_lib.pgan_gn_bwd(_ptr(u), _ptr(h), _ptr(a), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(d), _ptr(stats_b),
                                _rows_ptr(pg), _rows_ptr(pb), pg.stride(0), *self._gnws(), N, C, _hw(a),
                                _amax_out(amax, N * GN_AMAX_SLOTS), _stream()), "pgan_gn_bwd")

    def gn_jvp(self, t, h, a, stats, gamma, hd, stats_t, amax=None, beta=None):
        N, C = a.shape[0], a.shape[1]
        _check(_lib.pgan_gn_jvp(_ptr(t), _ptr(h), _ptr(a), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(hd), _ptr(stats_t),
                                *self._gnws(), N, C, _hw(a), _amax_out(amax, N * GN_AMAX_SLOTS), _stream()), "pgan_gn_jvp")

    def gn_bwd2(self, hbar, u, t, h, a, stats, stats_b, stats_t, gamma, p, pg, pb, amax=None, beta=None):
        N, C = a.shape[0], a.shape[1]
        _check(_lib.pgan_gn_bwd2(_ptr(hbar), _ptr(u), _ptr(t), _ptr(h), _ptr(a), _ptr(stats), _ptr(stats_b),
                                 _ptr(stats_t), _ptr(gamma), _ptr(beta), _ptr(p), _rows_ptr(pg), _rows_ptr(pb), pg.stride(0),
                                 *self._gnws(), N, C, _hw(a), _amax_out(amax, N * GN_AMAX_SLOTS), _stream()), "pgan_gn_bwd2")

    # ---------------- BatchNorm ----------------
    def bn_fwd(self, a, gamma, beta, rmean, rvar, h, stats, momentum=0.1, groups=1, amax=None):
        """groups > 1: a holds that many consecutive forward passes (a.shape[0] / groups samples each), stats is
        [groups, 2, C].  amax (every BatchNorm op): optional [C] output, the largest magnitude per channel of the
        op's main output over all samples (and groups) of the call -- a 1-D `bound` for conv_*."""
        B, C = a.shape[0] // groups, a.shape[1]
        assert B * groups == a.shape[0] and stats.numel() >= groups * 2 * C
        _check(_lib.pgan_bn_fwd(_ptr(a), _ptr(gamma), _ptr(beta), _ptr(rmean), _ptr(rvar), _ptr(h), _ptr(stats),
                                *self._bnws(), B, C, _hw(a), momentum, int(groups), _amax_out(amax, C), _stream()),
               "pgan_bn_fwd")

    def bn_fwd_partials(self, B, C, HW, groups=1) -> int:
        return int(_lib.pgan_bn_fwd_partials(B, C, HW, int(groups)))

    # BatchNorm over the global batch of a data-parallel run: sums -> (caller adds `part` over the ranks) -> apply
    def bn_partials(self, B, C, HW) -> int:
        return int(_lib.pgan_bn_partials(B, C, HW))

    @staticmethod
    def _part_ptr(part, need):
        if not part.is_cuda or part.dtype != torch.float64 or not part.is_contiguous() or part.numel() < need:
            raise PganError("expected a contiguous float64 CUDA tensor of bn_partials() elements")
        return part.data_ptr()

    def bn_fwd_sums(self, a, part, groups=1, amax=None):
        """amax: the tensor the following bn_fwd_apply will fill (this call clears it)."""
        B, C = a.shape[0] // groups, a.shape[1]
        _check(_lib.pgan_bn_fwd_sums(_ptr(a), self._part_ptr(part, self.bn_fwd_partials(B, C, _hw(a), groups)), B, C,
                                     _hw(a), int(groups), _amax_out(amax, C), _stream()), "pgan_bn_fwd_sums")

    def bn_fwd_apply(self, a, gamma, beta, rmean, rvar, h, stats, part, world, momentum=0.1, groups=1, amax=None):
        B, C = a.shape[0] // groups, a.shape[1]
        _check(_lib.pgan_bn_fwd_apply(_ptr(a), _ptr(gamma), _ptr(beta), _ptr(rmean), _ptr(rvar), _ptr(h), _ptr(stats),
                                      self._part_ptr(part, self.bn_fwd_partials(B, C, _hw(a), groups)), B, C, _hw(a),
                                      int(world), momentum, int(groups), _amax_out(amax, C), _stream()), "pgan_bn_fwd_apply")

    def bn_bwd_sums(self, dh, h, a, stats, part, amax=None):
        B, C = a.shape[:2]
        _check(_lib.pgan_bn_bwd_sums(_ptr(dh), _ptr(h), _ptr(a), _ptr(stats),
                                     self._part_ptr(part, self.bn_partials(B, C, _hw(a))), B, C, _hw(a),
                                     _amax_out(amax, C), _stream()), "pgan_bn_bwd_sums")

    def bn_bwd_apply(self, dh, h, a, stats, gamma, da, dgamma, dbeta, part, world, amax=None):
        B, C = a.shape[:2]
        _check(_lib.pgan_bn_bwd_apply(_ptr(dh), _ptr(h), _ptr(a), _ptr(stats), _ptr(gamma), _ptr(da), _ptr(dgamma),
                                      _ptr(dbeta), self._part_ptr(part, self.bn_partials(B, C, _hw(a))), B, C, _hw(a),
                                      int(world), _amax_out(amax, C), _stream()), "pgan_bn_bwd_apply")

    def bn_bwd(self, dh, h, a, stats, gamma, da, dgamma, dbeta, amax=None):
        B, C = a.shape[0], a.shape[1]
        _check(_lib.pgan_bn_bwd(_ptr(dh), _ptr(h), _ptr(a), _ptr(stats), _ptr(gamma), _ptr(da), _ptr(dgamma),
                                _ptr(dbeta), *self._bnws(), B, C, _hw(a), _amax_out(amax, C), _stream()),
               "pgan_bn_bwd")

    # ---------------- head ----------------
    def head_fwd(self, h4, f1, w, b, score):
        N = h4.shape[0]
        K2, K1 = h4[0].numel(), f1[0].numel()
        assert w.numel() == K2 + K1 and score.numel() == N
        _check(_lib.pgan_head_fwd(_ptr(h4), _ptr(f1), _ptr(w), _ptr(b), _ptr(score), N, K2, K1, _stream()),
               "pgan_head_fwd")

    def head_seed(self, w, coef, u4):
        N, K = u4.shape[0], u4[0].numel()
        assert coef.numel() == N
        _check(_lib.pgan_head_seed(_ptr(w), _ptr(coef), _ptr(u4), N, K, _stream()), "pgan_head_seed")

    def head_wgrad(self, x, coef, dw):
        N, J = x.shape[0], x[0].numel()
        assert coef.numel() == N and dw.numel() == J
        _check(_lib.pgan_head_wgrad(_ptr(x), _ptr(coef), _ptr(dw), N, J, _stream()), "pgan_head_wgrad")

    # ---------------- penalties / losses ----------------
    def interpolate(self, t, real, fake, out):
        B, M = real.shape[0], real[0].numel()
        _check(_lib.pgan_interpolate(_ptr(t), _ptr(real), _ptr(fake), _ptr(out), B, M, _stream()), "pgan_interpolate")

    def gp_penalty(self, g, lam, v, pen):
        _check(_lib.pgan_gp_penalty(_ptr(g), _ptr(v), _ptr(pen), _ptr(self.ws_small), g.numel(), float(lam), _stream()),
               "pgan_gp_penalty")

    def div_penalty(self, g, nb, v, pen):
        assert g.shape[0] == 2 * nb and 2 * nb <= self.ws_small.numel()
        _check(_lib.pgan_div_penalty(_ptr(g), _ptr(v), _ptr(pen), _ptr(self.ws_small), nb, g[0].numel(), _stream()),
               "pgan_div_penalty")

    def critic_loss(self, score, nb, pen, out):
        _check(_lib.pgan_critic_loss(_ptr(score), _ptr(pen), _ptr(out), nb, _stream()), "pgan_critic_loss")

    def neg_mean(self, score, out):
        n = score.numel()
        _check(_lib.pgan_sum_scale(_ptr(score), _ptr(out), n, -1.0 / n, _stream()), "pgan_sum_scale")

    def act_bwd(self, dy, y, dx, act):
        _check(_lib.pgan_act_bwd(_ptr(dy), _ptr(y), _ptr(dx), y.numel(), act, _stream()), "pgan_act_bwd")

    def affine(self, x, scale, shift, out):
        _check(_lib.pgan_affine(_ptr(x), _ptr(out), x.numel(), float(scale), float(shift), _stream()), "pgan_affine")

    def adam(self, p, g, m, v, step, lr, b1, b2, eps, step_dev=None, grad_scale=1.0):
        """grad_scale: factor applied to g as it is read (1 / ranks when g holds a sum over data-parallel ranks)."""
        sd = None if step_dev is None else step_dev.data_ptr()
        _check(_lib.pgan_adam(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, b1, b2, eps, float(grad_scale), sd,
                              int(step), _stream()), "pgan_adam")

    def philox(self, out, state, normal: bool, skip: int = 0, total: Optional[int] = None):
        """out = values [skip, skip + out.numel()) of a draw of `total` values (default: the whole draw); the state
        advances by the whole draw."""
        n = out.numel()
        _check(_lib.pgan_philox_slice(_ptr(out), n, int(skip), n if total is None else int(total), state.data_ptr(),
                                      int(normal), _stream()), "pgan_philox_slice")

    def take_batch(self, pool_s, pool_p, out_s, out_p, cursor, offset: int = 0, advance: Optional[int] = None):
        """offset / advance: the batch starts `offset` alignments past the cursor, which then moves on by `advance`
        (default: the batch size) -- a rank's rows of a global batch."""
        B = out_s.shape[0]
        _check(_lib.pgan_take_batch_slice(_ptr(pool_s), _ptr(pool_p), _ptr(out_s), _ptr(out_p), cursor.data_ptr(),
                                          pool_s.shape[0], B, out_s[0].numel(), out_p[0].numel(), int(offset),
                                          B if advance is None else int(advance), _stream()), "pgan_take_batch_slice")

    # ---------------- per-batch augmentations of the real alignments ----------------
    def label_smooth(self, sites, noise, centre, normalized, rng, ey=None, skip: int = 0, total: Optional[int] = None):
        """sites <- sites (1 - ey) + centre ey in place, ey ~ U(0, noise) per element from the device generator `rng`
        (or the given draws `ey`, same shape); normalized: sites hold (raw - 0.5) / 0.5."""
        n = sites.numel()
        if ey is not None and (ey.numel() != n or not ey.is_cuda):
            raise PganError("label_smooth: the given draws must match the sites")
        _check(_lib.pgan_label_smooth(_ptr(sites), n, float(noise), float(centre), int(bool(normalized)),
                                      None if rng is None else rng.data_ptr(), _ptr(ey), int(skip),
                                      n if total is None else int(total), _stream()), "pgan_label_smooth")

    def mono_switch(self, sites, pos, lower, upper, rng, given=None, first: int = 0, total: Optional[int] = None):
        """switch_monomorphic in place on sites [B,1,rows,S] / pos [B,1,S]; given = (n [B] int32, cols [B, upper] int32,
        u [B, upper] float32) replaces the device draws."""
        B, rows, S = sites.shape[0], sites.shape[2], sites.shape[3]
        assert tuple(pos.shape) == (B, 1, S) and sites.shape[1] == 1
        gn = gc = gu = None
        if given is not None:
            n_g, c_g, u_g = given
            assert n_g.dtype == torch.int32 and c_g.dtype == torch.int32 and n_g.numel() >= B
            assert c_g.numel() >= B * upper and u_g.numel() >= B * upper and n_g.is_cuda and c_g.is_cuda
            gn, gc, gu = n_g.data_ptr(), c_g.data_ptr(), _ptr(u_g)
        _check(_lib.pgan_mono_switch(_ptr(sites), _ptr(pos), B, rows, S, int(lower), int(upper),
                                     None if rng is None else rng.data_ptr(), gn, gc, gu, int(first),
                                     B if total is None else int(total), _stream()), "pgan_mono_switch")

    # ---------------- replay buffer / genotype sampling ----------------
    def fakebuf_step(self, buf_s, buf_p, state, slots, new_s, new_p, out_s, out_p, rng, picks=None):
        """FakeBuffer.add(new) then FakeBuffer.sample(B) on the device-resident ring (state: int32[4] head, count,
        error flag, spare; slots: int32[B] scratch; picks: optional int32[B] logical indices instead of draws)."""
        B, size = new_s.shape[0], buf_s.shape[0]
        assert state.dtype == torch.int32 and state.numel() >= 4 and slots.dtype == torch.int32 and slots.numel() >= B
        assert out_s.shape[0] == B and buf_p.shape[0] == size
        if picks is not None:
            assert picks.dtype == torch.int32 and picks.numel() >= B and picks.is_cuda
        _check(_lib.pgan_fakebuf_step(_ptr(buf_s), _ptr(buf_p), size, state.data_ptr(), slots.data_ptr(), _ptr(new_s),
                                      _ptr(new_p), _ptr(out_s), _ptr(out_p), B, new_s[0].numel(), new_p[0].numel(),
                                      rng.data_ptr(), None if picks is None else picks.data_ptr(), _stream()),
               "pgan_fakebuf_step")

    def sample_genotypes(self, p, out, rng, require_variation=True):
        assert p.dim() == 4 and p.shape[1] == 1 and tuple(out.shape) == tuple(p.shape)
        _check(_lib.pgan_sample_genotypes(_ptr(p), _ptr(out), p.shape[0], p.shape[2], p.shape[3], rng.data_ptr(),
                                          int(require_variation), _stream()), "pgan_sample_genotypes")

    # ---------------- discriminator 1-D tower ----------------
    @staticmethod
    def _param_array(cw, cb, gw, gb):
        arr = (c_void_p * 12)()
        for i in range(3):
            arr[4 * i + 0] = _ptr(cw[i])
            arr[4 * i + 1] = _ptr(cb[i])
            arr[4 * i + 2] = _ptr(gw[i])
            arr[4 * i + 3] = _ptr(gb[i])
        return arr

    def d1d_fwd(self, xp, cw, cb, gw, gb, a_save, feat):
        N = xp.shape[0]
        arr = self._param_array(cw, cb, gw, gb)
        _check(_lib.pgan_d1d_fwd_len(_ptr(xp), arr, _ptr(a_save[0]), _ptr(a_save[1]), _ptr(a_save[2]), _ptr(feat), N,
                                     xp.shape[-1], _stream()), "pgan_d1d_fwd_len")

    def d1d_bwd(self, xp, cw, cb, gw, gb, a_save, coef, w_tail, d_cw, d_cb, d_gw, d_gb, dxp):
        N = xp.shape[0]
        arr = self._param_array(cw, cb, gw, gb)
        part, ld, offs, dsts = None, 0, None, None
        if d_cw is not None:
            if N > self.max_samples:
                raise PganError(f"d1d_bwd: {N} samples exceed the partial-gradient scratch ({self.max_samples})")
            dsts = []
            for i in range(3):
                dsts += [d_cw[i], d_cb[i], d_gw[i], d_gb[i]]
            base = min(t.data_ptr() for t in dsts)
            rel = [(t.data_ptr() - base) // 4 for t in dsts]
            span = max(r + t.numel() for r, t in zip(rel, dsts))
            flat = span <= sum((t.numel() + 63) // 64 * 64 for t in dsts) and all(_ptr(t) is not None for t in dsts)
            if not flat:   # destinations are unrelated tensors: packed rows, one reduction per tensor
                rel, acc = [], 0
                for s in D1_PACKED:
                    rel.append(acc)
                    acc += s
                span = acc
            ld = span
            offs = (c_int * 12)(*rel)
            part = self.d1_part
        _check(_lib.pgan_d1d_bwd_len(_ptr(xp), arr, _ptr(a_save[0]), _ptr(a_save[1]), _ptr(a_save[2]), _ptr(coef),
                                     _ptr(w_tail), _ptr(part), ld, offs, _ptr(dxp), N, xp.shape[-1], _stream()),
               "pgan_d1d_bwd_len")
        if d_cw is not None:
            if flat:
                first = min(dsts, key=lambda t: t.data_ptr())
                _check(_lib.pgan_rows_reduce(_ptr(part), ld, first.data_ptr(), ld, 0, N, 0, 0, 0, _stream()),
                       "pgan_rows_reduce")
            else:
                for r, t in zip(rel, dsts):
                    _check(_lib.pgan_rows_reduce(part.data_ptr() + 4 * r, ld, _ptr(t), t.numel(), 0, N, 0, 0, 0,
                                                 _stream()), "pgan_rows_reduce")

    # ---------------- generator pieces ----------------
    def gemm_nn(self, a, b, c):
        M, K = a.shape
        N = b.shape[1]
        assert b.shape[0] == K and tuple(c.shape) == (M, N)
        _check(_lib.pgan_gemm(_ptr(a), _ptr(b), _ptr(c), M, N, K, 0, _stream()), "pgan_gemm")

    def gemm_tn(self, a, b, c):
        K, M = a.shape
        N = b.shape[1]
        assert b.shape[0] == K and tuple(c.shape) == (M, N)
        _check(_lib.pgan_gemm(_ptr(a), _ptr(b), _ptr(c), M, N, K, 1, _stream()), "pgan_gemm")

    def convt1d_up(self, y, w, x, act=0):
        N, Cs, Ls = y.shape
        Cl = w.shape[1]
        assert tuple(x.shape) == (N, Cl, 2 * Ls)
        _check(_lib.pgan_ct1d_up(_ptr(y), _ptr(w), _ptr(x), N, Cs, Cl, Ls, act, _stream()), "pgan_ct1d_up")

    def convt1d_down(self, x, w, y):
        N, Cs, Ls = y.shape
        Cl = w.shape[1]
        assert tuple(x.shape) == (N, Cl, 2 * Ls)
        _check(_lib.pgan_ct1d_down(_ptr(x), _ptr(w), _ptr(y), N, Cs, Cl, Ls, _stream()), "pgan_ct1d_down")

    def convt1d_wgrad(self, ya, x, dw):
        N, Cs, Ls = ya.shape
        Cl = x.shape[1]
        assert tuple(dw.shape) == (Cs, Cl, 4)
        _check(_lib.pgan_ct1d_wgrad(_ptr(ya), _ptr(x), _ptr(dw), N, Cs, Cl, Ls, _stream()), "pgan_ct1d_wgrad")

"""ctypes binding of liblq_hip.so (the C ABI of include/lq_hip.h).

There is NO fallback: if the HIP library is missing, or a tensor is not a
contiguous float32 tensor on a HIP device, the call raises.  PyTorch is used
only for device memory and streams (``tensor.data_ptr()``,
``torch.cuda.current_stream().cuda_stream``).
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Dict, Optional, Tuple

import torch

from .descriptor import memory_order

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads THE in-tree library and reads no environment variable.  Development builds of the same ABI (tuning knobs,
# experiments) are selected explicitly with ``use_library(path)`` before the first call -- only tools/ does that (tools/_devlib.py).
LIB_PATH = os.path.join(_HERE, "csrc", "liblq_hip.so")
_lib: Optional[ctypes.CDLL] = None
_pending: Optional[ctypes.CDLL] = None
_selftest_error: Optional[str] = None


def use_library(path: str) -> None:
    """Development only: load another build of the same C ABI instead of the shipped one.  Must be called before the first op."""
    global LIB_PATH
    if _lib is not None or _pending is not None:
        raise RuntimeError("use_library() must be called before the library is loaded")
    LIB_PATH = os.path.abspath(path)

LQ_Q_NONE, LQ_Q_F32, LQ_Q_I32, LQ_Q_I8 = 0, 1, 2, 3
LQ_ADAM_KERAS, LQ_ADAM_TORCH = 0, 1
LQ_PENALTY_ACCUMULATE_DS = 0x100

_c_i64 = ctypes.c_int64
_c_f = ctypes.c_float
_c_d = ctypes.c_double
_c_p = ctypes.c_void_p
_c_sz = ctypes.c_size_t
_c_int = ctypes.c_int

#: name -> (restype, argtypes); one entry per symbol declared in include/lq_hip.h
SIGNATURES = {
    "lq_version": (_c_int, []),
    "lq_last_error": (ctypes.c_char_p, []),
    "lq_status_string": (ctypes.c_char_p, [_c_int]),
    "lq_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "lq_fq_forward": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_fq_scale_grad": (_c_int, [_c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p, _c_sz, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_fq_fwd_bwd_fused": (_c_int, [_c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p, _c_sz, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_penalty_maxbin_fwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_sz, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_penalty_maxbin_bwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_penalty_difference_fwd": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_sz, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_penalty_difference_bwd": (_c_int, [_c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p, _c_sz, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_penalty_inverse_fwd": (_c_int, [_c_p, _c_p, _c_i64, _c_p]),
    "lq_penalty_inverse_bwd": (_c_int, [_c_p, _c_p, _c_f, _c_p, _c_i64, _c_p]),
    "lq_scale_adam_step": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_d, _c_d, _c_d, _c_d, _c_i64, _c_f, _c_int, _c_p]),
    "lq_scale_adam_step_dev": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_d, _c_d, _c_d, _c_d, _c_p, _c_f, _c_int, _c_p]),
    "lq_min_value_project": (_c_int, [_c_p, _c_i64, _c_f, _c_p]),
    "lq_q_absmax_over_axis": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
}



class TensorDesc(ctypes.Structure):
    """lq_tensor_desc of include/lq_hip.h."""
    _fields_ = [("P", _c_p), ("s", _c_p), ("dy", _c_p), ("out", _c_p), ("ds", _c_p), ("m", _c_p), ("v", _c_p),
                ("outer", _c_i64), ("G", _c_i64), ("inner", _c_i64), ("lambda_", _c_f), ("min_value", _c_f),
                ("out_oihw", _c_p), ("dp", _c_p), ("conv_hw", _c_i64), ("conv_ci", _c_i64), ("conv_co", _c_i64)]


SIGNATURES.update({
    "lq_batch_create": (_c_int, [ctypes.POINTER(TensorDesc), _c_int, ctypes.POINTER(_c_p)]),
    "lq_batch_destroy": (_c_int, [_c_p]),
    "lq_batch_workspace_bytes": (_c_sz, [_c_p]),
    "lq_batch_forward": (_c_int, [_c_p, _c_p]),
    "lq_batch_scale_grad": (_c_int, [_c_p, ctypes.POINTER(_c_p), _c_p, _c_sz, _c_p]),
    "lq_batch_scale_grad_step": (_c_int, [_c_p, ctypes.POINTER(_c_p), _c_int, _c_p, _c_sz, _c_d, _c_d, _c_d, _c_d, _c_i64, _c_p, _c_int, _c_p]),
    "lq_batch_scale_grad_oihw": (_c_int, [_c_p, ctypes.POINTER(_c_p), _c_p, _c_sz, _c_p]),
    "lq_conv_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "lq_conv_tile_supported": (_c_int, [_c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64]),
    "lq_fq_forward_oihw": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_fq_scale_grad_oihw": (_c_int, [_c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p, _c_sz, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_batch_scale_adam": (_c_int, [_c_p, _c_d, _c_d, _c_d, _c_d, _c_i64, _c_p, _c_int, _c_p]),
    "lq_batch_penalty_grads": (_c_int, [_c_p, _c_int, ctypes.POINTER(_c_f), ctypes.POINTER(_c_p), _c_p, _c_sz, _c_p]),
    "lq_adam_set_create": (_c_int, [ctypes.POINTER(_c_p), ctypes.POINTER(_c_p), ctypes.POINTER(_c_p), ctypes.POINTER(_c_i64),
                                    ctypes.POINTER(_c_f), _c_int, ctypes.POINTER(_c_p)]),
    "lq_adam_set_destroy": (_c_int, [_c_p]),
    "lq_adam_set_step": (_c_int, [_c_p, ctypes.POINTER(_c_p), _c_d, _c_d, _c_d, _c_d, _c_i64, _c_p, _c_int, _c_p]),
    "lq_selftest_ratio_division": (_c_int, [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _c_p, _c_p]),
    "lq_selftest_uniform_division": (_c_int, [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _c_p, _c_p]),
    "lq_q_minmax": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_q_histogram": (_c_int, [_c_p, _c_p, ctypes.c_int32, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p]),
    "lq_profile_events": (_c_int, [_c_p, _c_p]),
})

_lock = threading.RLock()


class LQError(RuntimeError):
    """A C-ABI call returned a negative lq_status."""


def load() -> ctypes.CDLL:
    """Loads liblq_hip.so, binds every prototype and runs the device self-test.  Raises loudly if the library is absent;
    a failed self-test is remembered and re-raised by EVERY later call.  The self-test needs a visible GPU and must not run
    inside a stream capture: while either holds the library is handed out with the test still pending (CPU-side argument
    validation, calls recorded into a hipGraph) and every later call retries it -- the first call that can run it does, so no
    kernel result leaves a process whose device/compiler combination fails the test."""
    global _lib, _pending
    if _selftest_error is not None:
        raise RuntimeError(_selftest_error)
    if _lib is not None:
        return _lib
    with _lock:
        if _selftest_error is not None:
            raise RuntimeError(_selftest_error)
        if _lib is not None:
            return _lib
        lib = _pending
        if lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"learned_quantization_amd: HIP extension not built ({LIB_PATH} missing). "
                    "Run `python -c 'import __graft_entry__ as g; g.build()'` or "
                    "`make -C learned_quantization_amd/csrc`.  There is no CPU fallback."
                )
            lib = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in SIGNATURES.items():
                fn = getattr(lib, name)   # AttributeError here = ABI mismatch, also loud
                fn.restype = restype
                fn.argtypes = argtypes
            _pending = lib
        if _device_selftest(lib):
            _lib = lib                    # published only once the self-test has actually run and passed
        return lib                        # test deferred (no GPU visible yet, or a stream capture is running): next call retries


def _device_selftest(lib) -> bool:
    """On the first load with a GPU present: 2^24 random operand pairs through each of the two fast division forms
    against the IEEE '/' ON THIS DEVICE.  The bit-exact-integer guarantee rests on them; a compiler or hardware
    combination that ever disagrees must fail loudly, not quantise differently.  Returns True when the test ran and passed,
    False when it has to be deferred; raises -- now and on every later load() -- when it failed."""
    global _selftest_error
    if not torch.cuda.is_available():
        return False
    if torch.cuda.is_current_stream_capturing():
        return False                      # cannot synchronise inside a capture: deferred, not cancelled
    bad = torch.zeros(2, dtype=torch.int64, device="cuda")
    rc = lib.lq_selftest_ratio_division(0x5EED, 256, 256, bad.data_ptr(), None)
    rc |= lib.lq_selftest_uniform_division(0x5EED, 256, 256, bad.data_ptr() + 8, None)
    torch.cuda.synchronize()
    n_ratio, n_uniform = (int(v) for v in bad.tolist())
    if rc or n_ratio or n_uniform:
        _selftest_error = (f"learned_quantization_amd: device self-test failed (rc={rc}, ratio division mismatches={n_ratio}, "
                           f"uniform division mismatches={n_uniform}): the fast division forms do not reproduce IEEE fp32 "
                           "division on this device/compiler; refusing to run")
        raise RuntimeError(_selftest_error)
    return True


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.lq_last_error().decode("utf-8", "replace")
        raise LQError(f"{what} failed: {lib.lq_status_string(rc).decode()} ({rc}): {msg}")


def same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Same shape and the same element order in memory (strides of unit axes address nothing and are ignored)."""
    return a.shape == b.shape and all(d == 1 or x == y for d, x, y in zip(a.shape, a.stride(), b.stride()))


def require_device_f32(t: torch.Tensor, name: str, dense_ok: bool = False, like: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Checks device / dtype and returns a tensor the kernels can read: contiguous by default; ``dense_ok``: a dense permutation
    of a contiguous array is kept as it is (conv kernels stored in OIHW order behind an HWIO shape: the caller describes the
    groups in memory order, descriptor.memory_descriptor); ``like``: the element order of that tensor (copied into it if needed)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the learned-quantization ops run only on a HIP device "
            "(MI355X); there is no CPU fallback"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        # the launches go to the current device's stream: a tensor of another GPU would be dereferenced from the wrong device
        raise RuntimeError(
            f"{name} lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}: this library runs one "
            "process per GPU -- call torch.cuda.set_device(...) (or use `with torch.cuda.device(...)`) before using its ops"
        )
    if like is not None:
        if t.shape != like.shape:
            raise ValueError(f"{name} shape {tuple(t.shape)} != parameter shape {tuple(like.shape)}")
        return t if same_layout(t, like) else torch.empty_like(like).copy_(t)
    if t.is_contiguous() or (dense_ok and memory_order(t.shape, t.stride()) is not None):
        return t
    return t.contiguous()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device) -> Optional[int]:
    s = torch.cuda.current_stream(device).cuda_stream
    return s if s else None


# ------------------------------------------------------------------ workspace cache
_ws: Dict[Tuple[int, int], torch.Tensor] = {}


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream); stream-ordered reuse is safe because
    every consumer of the partials is enqueued on the same stream."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        nbytes = max(int(nbytes), 1 << 16)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def workspace_for(device: torch.device, outer: int, G: int, inner: int) -> torch.Tensor:
    return workspace(device, load().lq_workspace_bytes(outer, G, inner))

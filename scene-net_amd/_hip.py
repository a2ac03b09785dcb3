"""ctypes binding of the C-ABI HIP library (include/scenenet_hip.h).

There is NO CPU fallback: if the library is missing or a tensor is not on a HIP
device, the call raises.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes
import functools
import os
from ctypes import c_char_p, c_int, c_void_p
from typing import Optional, Sequence, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SN_HIP_LIB: another build of the same library (A/B timing of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get("SN_HIP_LIB") or os.path.join(_HERE, "lib", "libscenenet_hip.so")

SN_F32, SN_F64, SN_U8, SN_OCC8, SN_BF16 = 0, 1, 2, 3, 4
SN_GENEO_CY, SN_GENEO_CONE, SN_GENEO_NEG = 0, 1, 2
SN_GENEO_CY_V1, SN_GENEO_CONE_V1, SN_GENEO_NEG_V1 = 3, 4, 5
SN_P_RADIUS, SN_P_SIGMA, SN_P_APEX, SN_P_CONE_RADIUS, SN_P_CONE_INC, SN_P_NEG_FACTOR = 0, 1, 2, 3, 4, 5
SN_NPARAM = 8

# every symbol include/scenenet_hip.h declares: (restype, argtypes)
_P, _I = c_void_p, c_int
SYMBOLS = {
    "sn_version": (c_int, []),
    "sn_prepare_device": (c_int, []),
    "sn_device_status": (c_int, [_P, _P]),
    "sn_device_status_clear": (c_int, []),
    "sn_last_error": (c_char_p, []),
    "sn_device_count": (c_int, []),
    "sn_conv_i8_path_counts": (c_int, [_P]),
    "sn_set_option": (c_int, [c_char_p, _I]),
    "sn_get_option": (c_int, [c_char_p]),
    "sn_geneo_bank": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sn_effective_lambdas": (c_int, [_P, _P, _I, _I, _P, _P]),
    "sn_conv_bank": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "sn_geneo_bank_prep": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "sn_conv_bank_prep": (c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "sn_conv_bank_prepared": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "sn_conv_bank_prepared_served": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "sn_conv_prep_verdict_offset": (c_int, []),
    "sn_conv_i8_spin_timeouts": (c_int, [_P]),
    "sn_launch_timing_events": (c_int, [_P, _P]),
    "sn_conv_fused": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "sn_conv_fused_supported": (c_int, [_I, _I, _I, _I, _I, _I, _I]),
    "sn_forward_auto": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "sn_voxel_bbox": (c_int, [_P, _P, _I, _P, _P]),
    "sn_voxel_desc": (c_int, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "sn_voxel_desc_from_bounds": (c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "sn_voxel_desc_sized": (c_int, [_P, _I, _P, _I, _I, _I, _P, _P, _P, _P]),
    "sn_voxel_finalize_sized": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "sn_voxel_scatter": (c_int, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "sn_voxel_finalize": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "sn_voxel_occupancy": (c_int, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "sn_voxel_occupancy_fused": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P,
                                         _P]),
    "sn_voxel_occupancy_fused_bank": (c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P,
                                              _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "sn_voxel_occupancy_sized": (c_int, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P,
                                         _P, _P, _P]),
    "sn_voxel_occupancy_sized_bank": (c_int, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P,
                                              _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P]),
    "sn_voxel_prepare": (c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "sn_gather_points": (c_int, [_P, _I, _I, _P, _P, _I, _P, _I, _I, _I, ctypes.c_double, _P, _P]),
    "sn_grid_to_points": (c_int, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "sn_conv_corr": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sn_conv_corr_t": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "sn_conv_corr_blocks": (c_int, [_I, _I, _I, _I]),
    "sn_conv_corr_ws_bytes": (ctypes.c_size_t, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "sn_conv_corr_ws": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, ctypes.c_size_t, _P, _P]),
    "sn_geneo_bank_lambdas": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "sn_geneo_bank_bwd": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "sn_geneo_backward": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P]),
    "sn_loss_forward": (c_int, [_P, _I, _P, _I, _I, ctypes.c_int64, _P, _P, _I, _I] + [ctypes.c_double] * 6
                        + [_P, _P, _P, _P, _P]),
    "sn_param_penalty": (c_int, [_P, _P, _I, ctypes.c_float, _I, _P, _P, _P]),
    "sn_loss_backward": (c_int, [_P, _I, _P, _I, _I, ctypes.c_int64, _P, _I, _P, _P, _P, _P]),
    "sn_conv_fused_prep_bytes": (ctypes.c_size_t, [_I, _I, _I]),
    "sn_conv_fused_prep": (c_int, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "sn_conv_fused_prepared": (c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "sn_conv_fused_v": (c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P]),
    "sn_loss_forward_m": (c_int, [_P, _I, _P, _I, _I, ctypes.c_int64, _P, _P, _I, _I] + [ctypes.c_double] * 6
                          + [_P, _P, _P, _P, _P, _P]),
    "sn_loss_backward_u": (c_int, [_P, _I, _P, _I, _I, ctypes.c_int64, _P, _I, _P, _P, _I, _P, _P]),
    "sn_criterion_forward": (c_int, [_P, _I, _P, _I, _I, ctypes.c_int64, _P, _P, _I, _I] + [ctypes.c_double] * 6
                             + [_P, _P, _P, _P, _P, _P, _P, _I, ctypes.c_float, _I, _P, _P, _P, _P]),
    "sn_criterion_backward": (c_int, [_P, _I, _P, _I, _I, ctypes.c_int64, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P]),
}
SN_CONV_PREP_BYTES = 16384
SN_LOSS_WMSE, SN_LOSS_FOCAL_TVERSKY, SN_LOSS_DICE, SN_LOSS_WBCE = 1, 2, 4, 8
SN_LOSS_MAX_BINS = 16
SN_OCC_PARTS = 16
SN_BBOX_PARTS = 32
OCC_MAX_WORDS = 16 * 1024


class HipLibraryError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Loads libscenenet_hip.so (built by `make` / __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `make` or __graft_entry__.build()). "
            "There is no CPU fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sn_last_error()
        raise HipLibraryError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def device_status() -> Tuple[int, int, int]:
    """(code, workgroup, detail) of the current device's sticky status (sn_device_status): code 0 = healthy; 1 a dependency
    spin of the z-walk gave up, 2 a launch made with `assume_served` was declined by its bank's guard, 3 a tile kernel's
    hand-over spin gave up.  While a code is latched every launching entry of the library raises.  A host memory read: no
    synchronisation -- a kernel that is still running may latch later."""
    code = ctypes.c_int(0)
    detail = (ctypes.c_int * 2)(0, 0)
    _check_plain(load().sn_device_status(ctypes.byref(code), detail), "sn_device_status")
    return int(code.value), int(detail[0]), int(detail[1])


def device_status_clear() -> None:
    """Synchronises the device and re-arms its sticky status (sn_device_status_clear)."""
    _check_plain(load().sn_device_status_clear(), "sn_device_status_clear")


def _check_plain(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sn_last_error()
        raise HipLibraryError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def set_option(name: str, value: int) -> None:
    """Process-wide library option (include/scenenet_hip.h: sn_set_option)."""
    _check(load().sn_set_option(name.encode(), int(value)), "sn_set_option")


def get_option(name: str) -> int:
    return int(load().sn_get_option(name.encode()))


def conv_i8_path_counts() -> Tuple[int, int, int]:
    """(served by the folded int8 kernel, declined by it: bank not symmetric, sent to the fp32 kernel by its guard) --
    launches of this process on the current device; synchronises (sn_conv_i8_path_counts)."""
    buf = (ctypes.c_ulonglong * 3)()
    _check(load().sn_conv_i8_path_counts(ctypes.cast(buf, ctypes.c_void_p)), "sn_conv_i8_path_counts")
    return int(buf[0]), int(buf[1]), int(buf[2])


def conv_i8_spin_timeouts() -> int:
    """Waves of the int8 kernels that ever gave up a bounded LDS hand-over spin (must be 0); synchronises."""
    buf = (ctypes.c_ulonglong * 1)()
    _check(load().sn_conv_i8_spin_timeouts(ctypes.cast(buf, ctypes.c_void_p)), "sn_conv_i8_spin_timeouts")
    return int(buf[0])


def launch_timing_events(start: "torch.cuda.Event", stop: "torch.cuda.Event") -> None:
    """sn_launch_timing_events: the next z-walk launch of this thread writes its own start / stop timestamps into the two
    events (both created with enable_timing=True and RECORDED ONCE before -- torch creates the hipEvent_t at the first
    record, and elapsed_time() wants both marked as recorded)."""
    a, b = int(start.cuda_event), int(stop.cuda_event)
    if not a or not b:
        raise HipLibraryError("launch_timing_events: record each event once first (torch creates the handle lazily)")
    _check_plain(load().sn_launch_timing_events(ctypes.c_void_p(a), ctypes.c_void_p(b)), "sn_launch_timing_events")


def _ptr(t: Optional[torch.Tensor], dtype: Optional[torch.dtype] = None, name: str = "tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    if not t.is_contiguous():
        raise HipLibraryError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise HipLibraryError(f"{name} must be {dtype} (got {t.dtype})")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """HIP stream handle of torch's current stream on the current device (what every launch of the C ABI goes to).
    The raw accessor costs ~0.3 us; torch.cuda.current_stream() builds a Stream object (~12 us per call measured on
    the GPU box, six calls per step: the fused forward step is 0.11 ms of GPU work)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _common_device(tensors) -> Optional[torch.device]:
    """The one HIP device every tensor argument of a C-ABI call lives on (None when there is no device tensor).
    Raises on a mix: the library takes raw pointers and ONE stream, so operands on different devices would be
    dereferenced from the wrong device."""
    dev = None
    for t in tensors:
        if t is None or not getattr(t, "is_cuda", False):
            continue   # CPU tensors are rejected with their own message by _ptr
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise HipLibraryError(f"tensor arguments live on different devices ({dev} and {t.device}); "
                                  "a call of the C ABI runs on one device and one stream")
    return dev


_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _on_tensor_device(fn):
    """Runs a wrapper with the HIP device of its tensor arguments current, so `_stream()` (torch's current stream on
    the CURRENT device) and the launch itself match the pointers -- `model.to('cuda:1')` while cuda:0 is current would
    otherwise launch on device 0 with device-1 pointers.  Tensors on different devices are refused (_common_device).
    Cheap on the common path (the eager forward step is host bound at ~0.1 ms): one `get_device()` per tensor argument,
    no objects built."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        idx = -1
        for a in args:
            if isinstance(a, torch.Tensor):
                d = a.get_device()
                if d >= 0:
                    if idx < 0:
                        idx = d
                    elif d != idx:
                        _common_device([t for t in args if isinstance(t, torch.Tensor)])   # raises with the message
        if kwargs:
            for a in kwargs.values():
                if isinstance(a, torch.Tensor):
                    d = a.get_device()
                    if d >= 0:
                        if idx < 0:
                            idx = d
                        elif d != idx:
                            _common_device([t for t in list(args) + list(kwargs.values()) if isinstance(t, torch.Tensor)])
        if idx < 0 or idx == (_get_device() if _get_device is not None else torch.cuda.current_device()):
            return fn(*args, **kwargs)
        with torch.cuda.device(idx):
            return fn(*args, **kwargs)
    return wrapper


# torch.bool (one byte, 0/1) is the binary-occupancy dtype: sn_conv_bank takes it on the int8 matrix cores
# torch.bfloat16: STORAGE of activations and their gradients in the training path (sn_conv_fused out, sn_loss_* pred /
# grad, sn_conv_corr_t gradients); never an input grid
_DT = {torch.float32: SN_F32, torch.float64: SN_F64, torch.uint8: SN_U8, torch.bool: SN_OCC8, torch.bfloat16: SN_BF16}
_DT_OUT = {torch.float32: SN_F32, torch.float64: SN_F64, torch.uint8: SN_U8, torch.bool: SN_U8,
           torch.bfloat16: SN_BF16}


# --------------------------------------------------------------------------- #
@_on_tensor_device
def geneo_bank(params: torch.Tensor, kinds: torch.Tensor, kernel_size: Sequence[int],
               status: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """params [G, SN_NPARAM] f32, kinds [G] i32 -> bank [G, kz, kx, ky] f32 (sn_geneo_bank)."""
    G = params.shape[0]
    kz, kx, ky = (int(k) for k in kernel_size)
    if out is None:
        out = torch.empty((G, kz, kx, ky), dtype=torch.float32, device=params.device)
    rc = load().sn_geneo_bank(_ptr(params, torch.float32, "params"), _ptr(kinds, torch.int32, "kinds"), G, kz, kx, ky,
                              _ptr(out, torch.float32, "bank"), _ptr(status, torch.int32, "status"), _stream())
    _check(rc, "sn_geneo_bank")
    return out


@_on_tensor_device
def geneo_bank_lambdas(params: torch.Tensor, kinds: torch.Tensor, kernel_size: Sequence[int], lambdas: torch.Tensor,
                       order: torch.Tensor, last: int):
    """sn_geneo_bank_lambdas: (bank [G,kz,kx,ky], effective coefficients [G]) in one launch; `lambdas[last]` is
    refreshed in place like sn_effective_lambdas does."""
    G = params.shape[0]
    kz, kx, ky = (int(k) for k in kernel_size)
    bank = torch.empty((G, kz, kx, ky), dtype=torch.float32, device=params.device)
    lam = torch.empty((G,), dtype=torch.float32, device=params.device)
    rc = load().sn_geneo_bank_lambdas(_ptr(params, torch.float32, "params"), _ptr(kinds, torch.int32, "kinds"), G, kz,
                                      kx, ky, _ptr(bank), None, _ptr(lambdas, torch.float32, "lambdas"),
                                      _ptr(order, torch.int32, "order"), int(last), _ptr(lam), _stream())
    _check(rc, "sn_geneo_bank_lambdas")
    return bank, lam


@_on_tensor_device
def geneo_bank_prep(params: torch.Tensor, kinds: torch.Tensor, lambdas: Optional[torch.Tensor] = None,
                    order: Optional[torch.Tensor] = None, last: int = 0, bank: Optional[torch.Tensor] = None,
                    lam_out: Optional[torch.Tensor] = None, prep: Optional[torch.Tensor] = None):
    """sn_geneo_bank_prep: the 9 x 9 x 9 bank, (optionally) the effective coefficients, and the int8 contraction's
    preparation blob in ONE launch -> (bank [G,9,9,9] f32, lam [G] f32 | None, prep uint8).  `bank`, `lam_out`, `prep`:
    caller-owned outputs to write into (persistent buffers: nothing is allocated on the step)."""
    G = params.shape[0]
    if bank is None:
        bank = torch.empty((G, 9, 9, 9), dtype=torch.float32, device=params.device)
    nbytes = SN_CONV_PREP_BYTES * ((G + 15) // 16)
    if prep is None:
        prep = torch.empty(nbytes, dtype=torch.uint8, device=params.device)
    if lambdas is not None and lam_out is None:
        lam_out = torch.empty(G, dtype=torch.float32, device=params.device)
    rc = load().sn_geneo_bank_prep(_ptr(params, torch.float32, "params"), _ptr(kinds, torch.int32, "kinds"), G, 9, 9, 9,
                                   _ptr(bank, torch.float32, "bank"), None, _ptr(lambdas, torch.float32, "lambdas"),
                                   _ptr(order, torch.int32, "order"), int(last), _ptr(lam_out, torch.float32, "lam_out"),
                                   _ptr(prep, torch.uint8, "prep"), _stream())
    _check(rc, "sn_geneo_bank_prep")
    return bank, (lam_out if lambdas is not None else None), prep


def prep_verdicts(prep: torch.Tensor) -> torch.Tensor:
    """view of the walk's verdict words (int32, one per group of 16 kernels) inside a preparation blob"""
    off = int(load().sn_conv_prep_verdict_offset())
    return prep.view(-1, SN_CONV_PREP_BYTES)[:, off:off + 4].view(torch.int32).reshape(-1)


class PreparedVerdict:
    """Learns, WITHOUT synchronising, whether the launches of one (weights, coefficients, tolerance, outputs) combination
    are served by the z-walk, so that later launches of the same combination can leave the fallback launch out
    (sn_conv_bank_prepared_served).  After a launch made with the fallback in place, `note(prep, key)` enqueues a 4-byte
    copy of the verdict words into pinned host memory and records an event; `served(key)` is True once that copy has
    completed and read 0 -- until then (and for every new key: an optimiser step changes it) the caller keeps the fallback.
    The words start every preparation at -1 (conv_prep.h): a 0 can only have been written by a walk that ran on THIS bank,
    and a -1 that comes back (the noted call was not the walk's) teaches nothing.  A caller whose knowledge is stale anyway
    -- the one thing the key cannot see is a raw-pointer write -- gets NaN outputs and the sticky device status from the
    launch itself (sn_conv_bank_prepared_served), never an unwritten buffer."""

    def __init__(self):
        self._key = None
        self._state = 0          # 0 nothing known, 1 read-back in flight, 2 served, 3 not served
        self._host = None
        self._event = None

    def __reduce__(self):
        # copied / pickled with a module: a fresh object that knows nothing (the event and the pinned buffer stay behind)
        return (PreparedVerdict, ())

    def served(self, key) -> bool:
        if torch.cuda.is_current_stream_capturing():
            return False   # a captured graph outlives these parameters: it keeps the fallback launch
        if key != self._key:
            self._key, self._state = key, 0
            return False
        if self._state == 1 and self._event.query():
            # -1: the preparation's sentinel -- no walk has written a verdict for this bank (the call that was noted went to
            # other kernels: a shape the walk does not serve): nothing learnt, the next call is noted again
            if int(self._host.min()) < 0:
                self._state = 0
            else:
                self._state = 2 if int(self._host.max()) == 0 else 3
        return self._state == 2

    def note(self, prep: torch.Tensor, key) -> None:
        self.note_words(prep_verdicts(prep), key)

    def note_words(self, words: torch.Tensor, key) -> None:
        """`words`: int32 device tensor of verdict words (all 0 = served) written by the launch just enqueued"""
        if key != self._key or self._state != 0 or torch.cuda.is_current_stream_capturing():
            return
        if self._host is None or self._host.numel() != words.numel():
            self._host = torch.empty(words.numel(), dtype=torch.int32, pin_memory=True)
        self._host.copy_(words, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._state = 1


@_on_tensor_device
def conv_bank_prep(bank: torch.Tensor, prep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The per-bank work of the int8 contraction, once (sn_conv_bank_prep): symmetry verdict, 24-bit fixed-point weights,
    error bounds and digit table of `bank` [G,kz,kx,ky] f32 into a caller-owned blob (uint8, SN_CONV_PREP_BYTES per group
    of 16 kernels), which conv_bank(..., prep=) then uses.  One blob serves the launches of one stream at a time."""
    G, kz, kx, ky = bank.shape
    nbytes = SN_CONV_PREP_BYTES * ((G + 15) // 16)
    if prep is None:
        prep = torch.empty(nbytes, dtype=torch.uint8, device=bank.device)
    elif prep.dtype != torch.uint8 or prep.numel() < nbytes:
        raise HipLibraryError(f"prep must be uint8 with at least {nbytes} bytes")
    rc = load().sn_conv_bank_prep(_ptr(bank, torch.float32, "bank"), G, kz, kx, ky, _ptr(prep, torch.uint8, "prep"),
                                  _stream())
    _check(rc, "sn_conv_bank_prep")
    return prep


@_on_tensor_device
def conv_bank(x: torch.Tensor, bank: torch.Tensor, lambdas: Optional[torch.Tensor], want_act: bool = False,
              want_out: bool = True, out_dtype: Optional[torch.dtype] = None, prep: Optional[torch.Tensor] = None,
              assume_served: bool = False):
    """x [B,1,Z,X,Y] (f32|f64|u8|bool), bank [G,kz,kx,ky] f32, lambdas [G] f32 (effective) ->
    (act [B,G,Z,X,Y] | None, out [B,1,Z,X,Y] | None) of out_dtype (sn_conv_bank; with `prep` = conv_bank_prep(bank):
    sn_conv_bank_prepared -- same results bit for bit, the per-bank work not repeated; `assume_served`: the caller has
    read the blob's verdict as 0 for these weights / coefficients / tolerance / outputs: sn_conv_bank_prepared_served, no
    fallback launch -- see PreparedVerdict)."""
    if x.dim() != 5 or x.shape[1] != 1:
        raise HipLibraryError(f"x must be [B,1,Z,X,Y] (got {tuple(x.shape)})")
    if x.dtype not in _DT or x.dtype == torch.bfloat16:
        raise HipLibraryError(f"x dtype {x.dtype} unsupported (f32, f64, u8, bool)")
    B, _, Z, X, Y = x.shape
    G, kz, kx, ky = bank.shape
    if out_dtype is None:
        out_dtype = x.dtype if x.dtype in (torch.float32, torch.float64) else torch.float32
    act = torch.empty((B, G, Z, X, Y), dtype=out_dtype, device=x.device) if want_act else None
    out = torch.empty((B, 1, Z, X, Y), dtype=out_dtype, device=x.device) if want_out else None
    if prep is not None:
        if prep.dtype != torch.uint8 or prep.numel() < SN_CONV_PREP_BYTES * ((G + 15) // 16):
            raise HipLibraryError("prep: not a blob of conv_bank_prep for this bank")
        fn = load().sn_conv_bank_prepared_served if assume_served else load().sn_conv_bank_prepared
        rc = fn(_ptr(x, None, "x"), _DT[x.dtype], _ptr(bank, torch.float32, "bank"),
                _ptr(lambdas, torch.float32, "lambdas"), _ptr(prep, torch.uint8, "prep"),
                B, Z, X, Y, G, kz, kx, ky, _ptr(act, None, "act"), _ptr(out, None, "out"),
                _DT_OUT[out_dtype], _stream())
        _check(rc, "sn_conv_bank_prepared")
        return act, out
    rc = load().sn_conv_bank(_ptr(x, None, "x"), _DT[x.dtype], _ptr(bank, torch.float32, "bank"),
                             _ptr(lambdas, torch.float32, "lambdas"), B, Z, X, Y, G, kz, kx, ky,
                             _ptr(act, None, "act"), _ptr(out, None, "out"), _DT_OUT[out_dtype], _stream())
    _check(rc, "sn_conv_bank")
    return act, out


def conv_fused_supported(x: torch.Tensor, kernel_size: Sequence[int]) -> bool:
    """Shapes sn_conv_fused serves: binary occupancy [B,1,Z,X,Y], and whatever the library's own plan accepts
    (sn_conv_fused_supported: Y % 4 == 0, a 16-y strip's window within 32 bytes, tables + halo in LDS)."""
    if x.dtype != torch.bool or x.dim() != 5:
        return False
    kz, kx, ky = (int(k) for k in kernel_size)
    B, _, Z, X, Y = x.shape
    return bool(load().sn_conv_fused_supported(int(B), int(Z), int(X), int(Y), kz, kx, ky))


@_on_tensor_device
def conv_fused(x: torch.Tensor, bank: torch.Tensor, lambdas: torch.Tensor,
               out_dtype: torch.dtype = torch.float32, verdict: Optional[torch.Tensor] = None,
               assume_served: bool = False, prep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu(tanh(conv3d(x, sum_g lambda_g K_g))) [B,1,Z,X,Y] (sn_conv_fused): the forward output through linearity.
    verdict: a caller-owned int32 [1] device tensor that receives the guard's verdict (sn_conv_fused_v); assume_served:
    the caller has read it as 0 for these weights / coefficients / tolerance -- the gated fp32 launches are left out."""
    B, _, Z, X, Y = x.shape
    G, kz, kx, ky = bank.shape
    out = torch.empty((B, 1, Z, X, Y), dtype=out_dtype, device=x.device)
    if prep is not None:   # the tables come from a blob (conv_fused_prep): same bits
        rc = load().sn_conv_fused_prepared(_ptr(x, None, "x"), _DT[x.dtype], _ptr(bank, torch.float32, "bank"),
                                           _ptr(lambdas, torch.float32, "lambdas"), _ptr(prep, torch.uint8, "prep"), B, Z,
                                           X, Y, G, kz, kx, ky, _ptr(out), _DT_OUT[out_dtype], int(bool(assume_served)),
                                           _stream())
        _check(rc, "sn_conv_fused_prepared")
        return out
    if verdict is None:
        rc = load().sn_conv_fused(_ptr(x, None, "x"), _DT[x.dtype], _ptr(bank, torch.float32, "bank"),
                                  _ptr(lambdas, torch.float32, "lambdas"), B, Z, X, Y, G, kz, kx, ky, _ptr(out),
                                  _DT_OUT[out_dtype], _stream())
    else:
        rc = load().sn_conv_fused_v(_ptr(x, None, "x"), _DT[x.dtype], _ptr(bank, torch.float32, "bank"),
                                    _ptr(lambdas, torch.float32, "lambdas"), B, Z, X, Y, G, kz, kx, ky, _ptr(out),
                                    _DT_OUT[out_dtype], _ptr(verdict, torch.int32, "verdict"), int(bool(assume_served)),
                                    _stream())
    _check(rc, "sn_conv_fused")
    return out


def conv_fused_prep_bytes(kernel_size: Sequence[int]) -> int:
    kz, kx, ky = (int(k) for k in kernel_size)
    return int(load().sn_conv_fused_prep_bytes(kz, kx, ky))


@_on_tensor_device
def conv_fused_prep(bank: torch.Tensor, lambdas: torch.Tensor, blob: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sn_conv_fused_prep: the fused forward's per-(bank, coefficients) tables into a blob (uint8) conv_fused(prep=) uses."""
    G, kz, kx, ky = bank.shape
    nbytes = conv_fused_prep_bytes((kz, kx, ky))
    if nbytes == 0:
        raise HipLibraryError(f"sn_conv_fused_prep: kernel {kz}x{kx}x{ky} is outside the combined kernel")
    if blob is None:
        blob = torch.empty(nbytes, dtype=torch.uint8, device=bank.device)
    elif blob.dtype != torch.uint8 or blob.numel() < nbytes:
        raise HipLibraryError(f"blob must be uint8 with at least {nbytes} bytes")
    _check(load().sn_conv_fused_prep(_ptr(bank, torch.float32, "bank"), _ptr(lambdas, torch.float32, "lambdas"), G, kz, kx,
                                     ky, _ptr(blob, torch.uint8, "blob"), _stream()), "sn_conv_fused_prep")
    return blob


def conv_fused_prep_verdict(blob: torch.Tensor, kernel_size: Sequence[int]) -> torch.Tensor:
    """view of the guard's verdict word (int32 [1]) inside a fused-forward blob"""
    n = conv_fused_prep_bytes(kernel_size)
    return blob[n - 12:n - 8].view(torch.int32)


@_on_tensor_device
def forward_auto(x: torch.Tensor, bank: torch.Tensor, lambdas: torch.Tensor, out_dtype: Optional[torch.dtype] = None):
    """sn_forward_auto: the forward output for a float grid; binary grids (the reference's f64 {0,1} input) take the
    int8 path, anything else the fp32 contraction, decided on the device.  Returns (out, not_binary flag [1] i32)."""
    B, _, Z, X, Y = x.shape
    G, kz, kx, ky = bank.shape
    out_dtype = x.dtype if out_dtype is None else out_dtype
    if x.data_ptr() % 16:   # a sliced view (x[1:] of an odd-sized grid): the binary check reads 16-byte vectors
        x = x.clone()
    out = torch.empty((B, 1, Z, X, Y), dtype=out_dtype, device=x.device)
    occ = torch.empty((x.numel(),), dtype=torch.uint8, device=x.device)
    flag = torch.empty((1,), dtype=torch.int32, device=x.device)
    rc = load().sn_forward_auto(_ptr(x, None, "x"), _DT[x.dtype], _ptr(bank, torch.float32, "bank"),
                                _ptr(lambdas, torch.float32, "lambdas"), B, Z, X, Y, G, kz, kx, ky, _ptr(occ), _ptr(flag),
                                _ptr(out), _DT_OUT[out_dtype], _stream())
    _check(rc, "sn_forward_auto")
    return out, flag


def desc_len(nx: int, ny: int, nz: int) -> int:
    return 6 + nx + ny + nz + 3


@_on_tensor_device
def voxel_bbox(pts: torch.Tensor, offsets: torch.Tensor) -> torch.Tensor:
    B = offsets.numel() - 1
    bbox = torch.empty((B, 6), dtype=torch.float64, device=pts.device)
    rc = load().sn_voxel_bbox(_ptr(pts, torch.float64, "pts"), _ptr(offsets, torch.int64, "offsets"), B,
                              _ptr(bbox), _stream())
    _check(rc, "sn_voxel_bbox")
    return bbox


@_on_tensor_device
def voxel_prepare(pts: torch.Tensor, offsets: torch.Tensor, n_xyz: Sequence[int], regular: bool = True,
                  want_bbox: bool = False):
    """bbox + cube + linspace edge tables in two launches (sn_voxel_prepare): returns (desc, bbox | None)."""
    B = offsets.numel() - 1
    nx, ny, nz = (int(v) for v in n_xyz)
    dev = pts.device
    partial = torch.empty((B, SN_BBOX_PARTS, 6), dtype=torch.float64, device=dev)
    desc = torch.empty((B, desc_len(nx, ny, nz)), dtype=torch.float64, device=dev)
    bbox = torch.empty((B, 6), dtype=torch.float64, device=dev) if want_bbox else None
    rc = load().sn_voxel_prepare(_ptr(pts, torch.float64, "pts"), _ptr(offsets, torch.int64, "offsets"), B, nx, ny, nz,
                                 int(regular), _ptr(partial), _ptr(bbox), _ptr(desc), _stream())
    _check(rc, "sn_voxel_prepare")
    return desc, bbox


@_on_tensor_device
def voxel_desc(bbox: torch.Tensor, n_xyz: Sequence[int], regular: bool = True, from_bounds: bool = False):
    B = bbox.shape[0]
    nx, ny, nz = (int(v) for v in n_xyz)
    desc = torch.empty((B, desc_len(nx, ny, nz)), dtype=torch.float64, device=bbox.device)
    if from_bounds:
        rc = load().sn_voxel_desc_from_bounds(_ptr(bbox, torch.float64, "bounds"), B, nx, ny, nz, _ptr(desc),
                                              _stream())
    else:
        rc = load().sn_voxel_desc(_ptr(bbox, torch.float64, "bbox"), B, nx, ny, nz, int(regular), _ptr(desc),
                                  _stream())
    _check(rc, "sn_voxel_desc")
    return desc


@_on_tensor_device
def voxel_desc_sized(bbox: torch.Tensor, voxel_dims: Sequence[float], max_n_xyz: Sequence[int]):
    """sn_voxel_desc_sized: size-mode descriptors of a batch on the device.  Returns (desc [B, desc_len(max)], dims
    [B,3] i32 = (n_x, n_y, n_z) per tile, status [B] i32 = 1 where a tile needs more than the maximum)."""
    B = bbox.shape[0]
    nx, ny, nz = (int(v) for v in max_n_xyz)
    dev = bbox.device
    desc = torch.empty((B, desc_len(nx, ny, nz)), dtype=torch.float64, device=dev)
    dims = torch.empty((B, 3), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    if len(voxel_dims) != 3:
        raise HipLibraryError("voxel_dims must have 3 entries (size_x, size_y, size_z)")
    sz = (ctypes.c_double * 3)(*[float(v) for v in voxel_dims])
    rc = load().sn_voxel_desc_sized(_ptr(bbox, torch.float64, "bbox"), B, ctypes.cast(sz, c_void_p), nx, ny, nz,
                                    _ptr(desc), _ptr(dims), _ptr(status), _stream())
    _check(rc, "sn_voxel_desc_sized")
    return desc, dims, status


@_on_tensor_device
def voxel_finalize_sized(counts, towers, desc, want_density=False, want_gt=False, want_occ=True, want_gt_occ=False):
    """sn_voxel_finalize_sized: voxel_finalize over each tile's own part of the padded grids."""
    B, nz, nx, ny = counts.shape
    dev = counts.device
    shape = (B, 1, nz, nx, ny)
    colstats = torch.empty((B, 2, ny), dtype=torch.int32, device=dev) if (want_density or want_occ) else None
    density = torch.empty(shape, dtype=torch.float64, device=dev) if want_density else None
    gt = torch.empty(shape, dtype=torch.float64, device=dev) if want_gt else None
    occ = torch.empty(shape, dtype=torch.float32, device=dev) if want_occ else None
    gt_occ = torch.empty(shape, dtype=torch.float32, device=dev) if want_gt_occ else None
    rc = load().sn_voxel_finalize_sized(_ptr(counts, torch.int32, "counts"), _ptr(towers, torch.int32, "towers"), B, nx,
                                        ny, nz, _ptr(desc, torch.float64, "desc"), _ptr(colstats), _ptr(density),
                                        _ptr(gt), _ptr(occ), _ptr(gt_occ), _stream())
    _check(rc, "sn_voxel_finalize_sized")
    return density, gt, occ, gt_occ


@_on_tensor_device
def voxel_scatter(pts, labels, offsets, desc, n_xyz, keep_labels: Sequence[float] = (), want_towers: bool = False,
                  counts=None, towers=None, dropped=None):
    B = offsets.numel() - 1
    nx, ny, nz = (int(v) for v in n_xyz)
    dev = pts.device
    if counts is None:
        counts = torch.empty((B, nz, nx, ny), dtype=torch.int32, device=dev)
    if want_towers and towers is None:
        towers = torch.empty((B, nz, nx, ny), dtype=torch.int32, device=dev)
    if dropped is None:
        dropped = torch.empty((B,), dtype=torch.int32, device=dev)
    keep = (ctypes.c_double * max(1, len(keep_labels)))(*[float(k) for k in keep_labels])
    rc = load().sn_voxel_scatter(_ptr(pts, torch.float64, "pts"), _ptr(labels, torch.float64, "labels"),
                                 _ptr(offsets, torch.int64, "offsets"), B, _ptr(desc, torch.float64, "desc"),
                                 nx, ny, nz, _ptr(counts, torch.int32, "counts"),
                                 _ptr(towers, torch.int32, "towers") if want_towers else None,
                                 ctypes.cast(keep, c_void_p), len(keep_labels), _ptr(dropped), _stream())
    _check(rc, "sn_voxel_scatter")
    return counts, (towers if want_towers else None), dropped


@_on_tensor_device
def voxel_finalize(counts, towers, want_density=False, want_gt=False, want_occ=True, want_gt_occ=False):
    B, nz, nx, ny = counts.shape
    dev = counts.device
    shape = (B, 1, nz, nx, ny)
    colstats = torch.empty((B, 2, ny), dtype=torch.int32, device=dev) if (want_density or want_occ) else None
    density = torch.empty(shape, dtype=torch.float64, device=dev) if want_density else None
    gt = torch.empty(shape, dtype=torch.float64, device=dev) if want_gt else None
    occ = torch.empty(shape, dtype=torch.float32, device=dev) if want_occ else None
    gt_occ = torch.empty(shape, dtype=torch.float32, device=dev) if want_gt_occ else None
    rc = load().sn_voxel_finalize(_ptr(counts, torch.int32, "counts"), _ptr(towers, torch.int32, "towers"), B, nx, ny,
                                  nz, _ptr(colstats), _ptr(density), _ptr(gt), _ptr(occ), _ptr(gt_occ), _stream())
    _check(rc, "sn_voxel_finalize")
    return density, gt, occ, gt_occ


def occupancy_supported(n_xyz: Sequence[int], planes: int) -> bool:
    """mirrors sn_voxel_occupancy's z-slab rule: some power-of-two slab count <= SN_OCC_PARTS fits the LDS bitmap."""
    nx, ny, nz = (int(v) for v in n_xyz)
    V = nx * ny * nz
    if V % 32:
        return False
    sl = 1
    while sl <= SN_OCC_PARTS:
        if nz % sl == 0 and (V // 32) % sl == 0 and ((nz // sl) * nx * ny) % 32 == 0 \
                and (V // 32 // sl) * planes <= OCC_MAX_WORDS:
            return True
        sl *= 2
    return False


@_on_tensor_device
def voxel_occupancy(pts, labels, offsets, desc, n_xyz, keep_labels: Sequence[float] = (), want_gt_occ: bool = False,
                    out_dtype: torch.dtype = torch.uint8, exact_fallback: bool = True):
    """LDS-bitmap occupancy (sn_voxel_occupancy): returns (occ, gt_occ | None, flags, dropped), grids [B,1,nz,nx,ny]."""
    B = offsets.numel() - 1
    nx, ny, nz = (int(v) for v in n_xyz)
    V = nx * ny * nz
    dev = pts.device
    planes = 2 if want_gt_occ else 1
    bits = torch.empty((B * SN_OCC_PARTS * (planes * (V // 32) + 1),), dtype=torch.int32, device=dev)
    occ = torch.empty((B, 1, nz, nx, ny), dtype=out_dtype, device=dev)
    gt_occ = torch.empty((B, 1, nz, nx, ny), dtype=out_dtype, device=dev) if want_gt_occ else None
    flags = torch.empty((B,), dtype=torch.int32, device=dev)
    dropped = torch.empty((B,), dtype=torch.int32, device=dev)
    counts = towers = None
    if exact_fallback:  # scratch for tiles whose flag is raised (gated launch: untouched otherwise)
        counts = torch.empty((B, V), dtype=torch.int32, device=dev)
        towers = torch.empty((B, V), dtype=torch.int32, device=dev) if want_gt_occ else None
    keep = (ctypes.c_double * max(1, len(keep_labels)))(*[float(k) for k in keep_labels])
    rc = load().sn_voxel_occupancy(_ptr(pts, torch.float64, "pts"),
                                   _ptr(labels, torch.float64, "labels") if want_gt_occ else None,
                                   _ptr(offsets, torch.int64, "offsets"), B, _ptr(desc, torch.float64, "desc"),
                                   nx, ny, nz, ctypes.cast(keep, c_void_p), len(keep_labels) if want_gt_occ else 0,
                                   _ptr(bits), _ptr(occ), _ptr(gt_occ), _DT_OUT[out_dtype], _ptr(flags), _ptr(dropped),
                                   _ptr(counts), _ptr(towers), _stream())
    _check(rc, "sn_voxel_occupancy")
    return occ, gt_occ, flags, dropped


@_on_tensor_device
def voxel_occupancy_fused(pts, labels, offsets, n_xyz, regular: bool = True, keep_labels: Sequence[float] = (),
                          want_gt_occ: bool = False, out_dtype: torch.dtype = torch.uint8, exact_fallback: bool = True,
                          want_bbox: bool = False, bank_rider=None):
    """sn_voxel_occupancy_fused: bbox + descriptor + LDS-bitmap occupancy in four launches.  Returns
    (occ, gt_occ | None, flags, dropped, desc, bbox | None).  bank_rider = (params [G, SN_NPARAM] f32, kinds [G] i32,
    bank [G,9,9,9] f32 out, prep uint8 out): K2 and the int8 contraction's preparation ride in the first launch
    (sn_voxel_occupancy_fused_bank) -- the two output buffers hold what sn_geneo_bank_prep would have written."""
    B = offsets.numel() - 1
    nx, ny, nz = (int(v) for v in n_xyz)
    V = nx * ny * nz
    dev = pts.device
    planes = 2 if want_gt_occ else 1
    partial = torch.empty((B, SN_BBOX_PARTS, 6), dtype=torch.float64, device=dev)
    desc = torch.empty((B, desc_len(nx, ny, nz)), dtype=torch.float64, device=dev)
    bbox = torch.empty((B, 6), dtype=torch.float64, device=dev) if want_bbox else None
    bits = torch.empty((B * SN_OCC_PARTS * (planes * (V // 32) + 1),), dtype=torch.int32, device=dev)
    occ = torch.empty((B, 1, nz, nx, ny), dtype=out_dtype, device=dev)
    gt_occ = torch.empty((B, 1, nz, nx, ny), dtype=out_dtype, device=dev) if want_gt_occ else None
    flags = torch.empty((B,), dtype=torch.int32, device=dev)
    dropped = torch.empty((B,), dtype=torch.int32, device=dev)
    counts = towers = None
    if exact_fallback:
        counts = torch.empty((B, V), dtype=torch.int32, device=dev)
        towers = torch.empty((B, V), dtype=torch.int32, device=dev) if want_gt_occ else None
    keep = (ctypes.c_double * max(1, len(keep_labels)))(*[float(k) for k in keep_labels])
    common = (_ptr(pts, torch.float64, "pts"), _ptr(labels, torch.float64, "labels") if want_gt_occ else None,
              _ptr(offsets, torch.int64, "offsets"), B, nx, ny, nz, int(regular), ctypes.cast(keep, c_void_p),
              len(keep_labels) if want_gt_occ else 0, _ptr(partial), _ptr(bbox), _ptr(desc), _ptr(bits), _ptr(occ),
              _ptr(gt_occ), _DT_OUT[out_dtype], _ptr(flags), _ptr(dropped), _ptr(counts), _ptr(towers))
    if bank_rider is None:
        rc = load().sn_voxel_occupancy_fused(*common, _stream())
        _check(rc, "sn_voxel_occupancy_fused")
    else:
        _check(load().sn_voxel_occupancy_fused_bank(*common, *_rider_args(bank_rider), _stream()),
               "sn_voxel_occupancy_fused_bank")
    return occ, gt_occ, flags, dropped, desc, bbox


def _rider_args(bank_rider):
    """bank_rider = (params [G, SN_NPARAM] f32, kinds [G] i32, bank [G,9,9,9] f32 out, prep uint8 out) or, with the effective
    coefficients riding too, (params, kinds, bank, prep, lambdas [G] f32 (refreshed in place), order [G] i32, last, lam_out [G]
    f32 out) -> the C argument tail of the sn_voxel_occupancy_*_bank entries"""
    params, kinds, bank, prep = bank_rider[:4]
    lambdas, order, last, lam_out = bank_rider[4:] if len(bank_rider) > 4 else (None, None, 0, None)
    G = params.shape[0]
    if tuple(bank.shape) != (G, 9, 9, 9) or prep.numel() < SN_CONV_PREP_BYTES * ((G + 15) // 16):
        raise HipLibraryError("bank_rider: bank must be [G,9,9,9] f32 and prep SN_CONV_PREP_BYTES x ceil(G / 16) bytes")
    return (_ptr(params, torch.float32, "params"), _ptr(kinds, torch.int32, "kinds"), G, 9, 9, 9,
            _ptr(bank, torch.float32, "bank"), None, _ptr(lambdas, torch.float32, "lambdas"),
            _ptr(order, torch.int32, "order"), int(last), _ptr(lam_out, torch.float32, "lam_out"),
            _ptr(prep, torch.uint8, "prep"))


@_on_tensor_device
def voxel_occupancy_sized(pts, labels, offsets, size_xyz: Sequence[float], n_xyz_max, keep_labels: Sequence[float] = (),
                          want_gt_occ: bool = False, out_dtype: torch.dtype = torch.uint8, exact_fallback: bool = True,
                          bank_rider=None):
    """sn_voxel_occupancy_sized: voxel-size mode on the LDS-bitmap kernels.  Returns
    (occ, gt_occ | None, flags, dropped, desc, dims [B,3], status [B], bbox [B,6])."""
    B = offsets.numel() - 1
    nx, ny, nz = (int(v) for v in n_xyz_max)
    V = nx * ny * nz
    dev = pts.device
    planes = 2 if want_gt_occ else 1
    partial = torch.empty((B, SN_BBOX_PARTS, 6), dtype=torch.float64, device=dev)
    desc = torch.empty((B, desc_len(nx, ny, nz)), dtype=torch.float64, device=dev)
    bbox = torch.empty((B, 6), dtype=torch.float64, device=dev)
    dims = torch.empty((B, 3), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    bits = torch.empty((B * SN_OCC_PARTS * (planes * (V // 32) + 1),), dtype=torch.int32, device=dev)
    occ = torch.empty((B, 1, nz, nx, ny), dtype=out_dtype, device=dev)
    gt_occ = torch.empty((B, 1, nz, nx, ny), dtype=out_dtype, device=dev) if want_gt_occ else None
    flags = torch.empty((B,), dtype=torch.int32, device=dev)
    dropped = torch.empty((B,), dtype=torch.int32, device=dev)
    counts = towers = None
    if exact_fallback:
        counts = torch.empty((B, V), dtype=torch.int32, device=dev)
        towers = torch.empty((B, V), dtype=torch.int32, device=dev) if want_gt_occ else None
    keep = (ctypes.c_double * max(1, len(keep_labels)))(*[float(k) for k in keep_labels])
    size = (ctypes.c_double * 3)(*[float(v) for v in size_xyz])
    common = (_ptr(pts, torch.float64, "pts"), _ptr(labels, torch.float64, "labels") if want_gt_occ else None,
              _ptr(offsets, torch.int64, "offsets"), B, ctypes.cast(size, c_void_p), nx, ny, nz,
              ctypes.cast(keep, c_void_p), len(keep_labels) if want_gt_occ else 0, _ptr(partial), _ptr(bbox), _ptr(desc),
              _ptr(dims), _ptr(status), _ptr(bits), _ptr(occ), _ptr(gt_occ), _DT_OUT[out_dtype], _ptr(flags),
              _ptr(dropped), _ptr(counts), _ptr(towers))
    if bank_rider is None:
        _check(load().sn_voxel_occupancy_sized(*common, _stream()), "sn_voxel_occupancy_sized")
    else:   # (as voxel_occupancy_fused: K2 + the preparation ride in the first launch)
        _check(load().sn_voxel_occupancy_sized_bank(*common, *_rider_args(bank_rider), _stream()),
               "sn_voxel_occupancy_sized_bank")
    return occ, gt_occ, flags, dropped, desc, dims, status, bbox


@_on_tensor_device
def gather_points(grid: torch.Tensor, pts: torch.Tensor, offsets: torch.Tensor, desc: torch.Tensor,
                  fill: float = 0.0) -> torch.Tensor:
    """grid [B,C,nz,nx,ny] (f32|f64) -> per-point values [C, total] via the scatter's binning (sn_gather_points)."""
    if grid.dim() != 5 or grid.dtype not in (torch.float32, torch.float64):
        raise HipLibraryError("grid must be [B,C,nz,nx,ny] float32/float64")
    B, C, nz, nx, ny = grid.shape
    out = torch.empty((C, pts.shape[0]), dtype=grid.dtype, device=grid.device)
    rc = load().sn_gather_points(_ptr(grid, None, "grid"), _DT[grid.dtype], C, _ptr(pts, torch.float64, "pts"),
                                 _ptr(offsets, torch.int64, "offsets"), B, _ptr(desc, torch.float64, "desc"),
                                 nx, ny, nz, float(fill), _ptr(out), _stream())
    _check(rc, "sn_gather_points")
    return out


@_on_tensor_device
def grid_to_points(grid: torch.Tensor, origin=None, voxel_size=None) -> torch.Tensor:
    """grid [n0,n1,n2] (f32|f64|u8|bool) -> rows [n0*n1*n2, 4] f64 = (origin + index * voxel_size, value), C order of
    the indices (sn_grid_to_points; utils/voxelization.py:328-360)."""
    if grid.dim() != 3 or grid.dtype not in _DT:
        raise HipLibraryError("grid must be [n0,n1,n2] float32/float64/uint8/bool")
    n0, n1, n2 = grid.shape

    def host3(v, name):
        if v is None:
            return None, None
        a = (ctypes.c_double * 3)(*[float(t) for t in v])
        if len(v) != 3:
            raise HipLibraryError(f"{name} must have 3 entries")
        return a, ctypes.cast(a, ctypes.c_void_p)

    o_keep, o_ptr = host3(origin, "origin")
    s_keep, s_ptr = host3(voxel_size, "voxel_size")
    out = torch.empty((n0 * n1 * n2, 4), dtype=torch.float64, device=grid.device)
    rc = load().sn_grid_to_points(_ptr(grid, None, "grid"), _DT[grid.dtype], n0, n1, n2, o_ptr, s_ptr, _ptr(out),
                                  _stream())
    _check(rc, "sn_grid_to_points")
    return out


@_on_tensor_device
def conv_corr(x: torch.Tensor, gout: torch.Tensor, out: Optional[torch.Tensor], kernel_size: Sequence[int]):
    """C [kz,kx,ky] f32 = sum_{b,v} delta[b,v] x[b, v+t-p]  (sn_conv_corr_ws: bool input as the gather over the set
    voxels, anything else as the GEMM); gout/out [B,1,Z,X,Y] both f32 or both bf16 (bf16 activation storage: the products
    are formed and summed in fp32 either way)."""
    B, _, Z, X, Y = x.shape
    kz, kx, ky = (int(k) for k in kernel_size)
    if gout.dtype not in (torch.float32, torch.bfloat16):
        raise HipLibraryError(f"gout must be float32 or bfloat16 (got {gout.dtype})")
    if out is not None and out.dtype != gout.dtype:
        raise HipLibraryError(f"out ({out.dtype}) and gout ({gout.dtype}) must have the same dtype")
    nbytes = int(load().sn_conv_corr_ws_bytes(_DT[x.dtype], B, Z, X, Y, kz, kx, ky))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)   # (torch's allocator: 512-byte aligned)
    C = torch.empty((kz, kx, ky), dtype=torch.float32, device=x.device)
    rc = load().sn_conv_corr_ws(_ptr(x, None, "x"), _DT[x.dtype], _ptr(gout, None, "gout"), _ptr(out, None, "out"),
                                _DT[gout.dtype], B, Z, X, Y, kz, kx, ky, _ptr(ws), nbytes, _ptr(C), _stream())
    _check(rc, "sn_conv_corr_ws")
    return C


@_on_tensor_device
def geneo_bank_bwd(params: torch.Tensor, kinds: torch.Tensor, kernel_size: Sequence[int], dW: torch.Tensor):
    """dparams [G, SN_NPARAM] f32 from dW [G,kz,kx,ky] f32 (sn_geneo_bank_bwd)."""
    G = params.shape[0]
    kz, kx, ky = (int(k) for k in kernel_size)
    dparams = torch.empty((G, SN_NPARAM), dtype=torch.float32, device=params.device)
    rc = load().sn_geneo_bank_bwd(_ptr(params, torch.float32, "params"), _ptr(kinds, torch.int32, "kinds"), G, kz, kx,
                                  ky, _ptr(dW, torch.float32, "dW"), _ptr(dparams), _stream())
    _check(rc, "sn_geneo_bank_bwd")
    return dparams


@_on_tensor_device
def geneo_backward(params: torch.Tensor, kinds: torch.Tensor, kernel_size: Sequence[int], bank: torch.Tensor,
                   lambdas: torch.Tensor, corr: torch.Tensor, last: int, out: torch.Tensor) -> torch.Tensor:
    """sn_geneo_backward: writes out[: G*SN_NPARAM] (dparams) and out[G*SN_NPARAM :] (dlambdas) of the packed gradient
    vector in one launch, from the correlation `corr` [kz,kx,ky]."""
    G = params.shape[0]
    kz, kx, ky = (int(k) for k in kernel_size)
    n = G * SN_NPARAM
    if out.numel() != n + G or out.dtype != torch.float32:
        raise HipLibraryError("packed gradient must be f32 [G*SN_NPARAM + G]")
    base = _ptr(out, torch.float32, "out")
    rc = load().sn_geneo_backward(_ptr(params, torch.float32, "params"), _ptr(kinds, torch.int32, "kinds"), G, kz, kx,
                                  ky, _ptr(bank, torch.float32, "bank"), _ptr(lambdas, torch.float32, "lambdas"),
                                  _ptr(corr, torch.float32, "corr"), int(last), base, base + 4 * n, _stream())
    _check(rc, "sn_geneo_backward")
    return out


@_on_tensor_device
def effective_lambdas(lambdas: torch.Tensor, order: torch.Tensor, last: int) -> torch.Tensor:
    """sn_effective_lambdas: [G] f32 effective coefficients; `lambdas[last]` is refreshed in place."""
    G = int(lambdas.numel())
    out = torch.empty((G,), dtype=torch.float32, device=lambdas.device)
    rc = load().sn_effective_lambdas(_ptr(lambdas, torch.float32, "lambdas"), _ptr(order, torch.int32, "order"), G,
                                     int(last), _ptr(out), _stream())
    _check(rc, "sn_effective_lambdas")
    return out


# --------------------------------------------------------------------------- #
def loss_parts(n_per: int) -> int:
    """SN_LOSS_PARTS of include/scenenet_hip.h."""
    return 1 if n_per <= 16384 else min(256, (n_per + 16383) // 16384)


@_on_tensor_device
def loss_forward(pred: torch.Tensor, gt: torch.Tensor, ranges: torch.Tensor, bin_w: torch.Tensor, terms: int,
                 mse_weight: float = 1.0, tversky_alpha: float = 0.5, tversky_beta: float = 1.0,
                 focal_gamma: float = 1.0, tversky_smooth: float = 1.0, dice_smooth: float = 1.0):
    """sn_loss_forward_m on pred/gt [B, ...] (same shape).  Returns (loss [5] f64 = {total, wmse, focal tversky, dice,
    weighted bce}, stats [B, 3H+5] f64, coef f64 for loss_backward, loss32 [5] f32 = the same five rounded once)."""
    if pred.shape != gt.shape:
        raise HipLibraryError(f"pred {tuple(pred.shape)} and gt {tuple(gt.shape)} must have the same shape")
    B = int(pred.shape[0])
    n_per = pred.numel() // max(B, 1)
    H = int(ranges.numel())
    dev = pred.device
    ws = torch.empty((B * loss_parts(n_per) * (3 * H + 5),), dtype=torch.float64, device=dev)
    stats = torch.empty((B, 3 * H + 5), dtype=torch.float64, device=dev)
    loss = torch.empty((5,), dtype=torch.float64, device=dev)
    loss32 = torch.empty((5,), dtype=torch.float32, device=dev)
    coef = torch.empty((2 * SN_LOSS_MAX_BINS + 3 * B,), dtype=torch.float64, device=dev)
    rc = load().sn_loss_forward_m(_ptr(pred, None, "pred"), _DT[pred.dtype], _ptr(gt, None, "gt"), _DT[gt.dtype], B,
                                  n_per, _ptr(ranges, torch.float32, "ranges"), _ptr(bin_w, torch.float32, "bin_w"), H,
                                  int(terms), float(mse_weight), float(tversky_alpha), float(tversky_beta),
                                  float(focal_gamma), float(tversky_smooth), float(dice_smooth), _ptr(ws), _ptr(stats),
                                  _ptr(loss), _ptr(loss32), _ptr(coef), _stream())
    _check(rc, "sn_loss_forward")
    return loss, stats, coef, loss32


@_on_tensor_device
def criterion_forward(pred: torch.Tensor, gt: torch.Tensor, ranges: torch.Tensor, bin_w: torch.Tensor, terms: int,
                      P: torch.Tensor, mask: torch.Tensor, weight: float, with_sum: bool, mse_weight: float = 1.0,
                      tversky_alpha: float = 0.5, tversky_beta: float = 1.0, focal_gamma: float = 1.0,
                      tversky_smooth: float = 1.0, dice_smooth: float = 1.0):
    """sn_criterion_forward: the dense terms of loss_forward and the penalties of param_penalty over the packed parameters
    P in the dense loss's two launches.  Returns (total [1] f32 = float(dense) + penalty, stats, coef, pen_grad [N] f32)."""
    if pred.shape != gt.shape:
        raise HipLibraryError(f"pred {tuple(pred.shape)} and gt {tuple(gt.shape)} must have the same shape")
    B = int(pred.shape[0])
    n_per = pred.numel() // max(B, 1)
    H = int(ranges.numel())
    N = int(P.numel())
    dev = pred.device
    ws = torch.empty((B * loss_parts(n_per) * (3 * H + 5),), dtype=torch.float64, device=dev)
    stats = torch.empty((B, 3 * H + 5), dtype=torch.float64, device=dev)
    loss = torch.empty((5,), dtype=torch.float64, device=dev)
    coef = torch.empty((2 * SN_LOSS_MAX_BINS + 3 * B,), dtype=torch.float64, device=dev)
    f32 = torch.empty((7 + N,), dtype=torch.float32, device=dev)   # loss32 [5] | total [1] | penalty [1] | its gradient [N]
    base = f32.data_ptr()
    rc = load().sn_criterion_forward(_ptr(pred, None, "pred"), _DT[pred.dtype], _ptr(gt, None, "gt"), _DT[gt.dtype], B,
                                     n_per, _ptr(ranges, torch.float32, "ranges"), _ptr(bin_w, torch.float32, "bin_w"),
                                     H, int(terms), float(mse_weight), float(tversky_alpha), float(tversky_beta),
                                     float(focal_gamma), float(tversky_smooth), float(dice_smooth), _ptr(ws), _ptr(stats),
                                     _ptr(loss), base, _ptr(coef), _ptr(P, torch.float32, "P"),
                                     _ptr(mask, torch.int8, "mask"), N, float(weight), int(bool(with_sum)), base + 24,
                                     base + 28, base + 20, _stream())
    _check(rc, "sn_criterion_forward")
    return f32[5:6], stats, coef, f32[7:]


@_on_tensor_device
def criterion_backward(pred: torch.Tensor, gt: torch.Tensor, ranges: torch.Tensor, coef: torch.Tensor,
                       upstream: torch.Tensor, pen_grad: torch.Tensor):
    """sn_criterion_backward: (dL/dpred, penalty gradient x upstream [N] f32) in the gradient pass's one launch."""
    B = int(pred.shape[0])
    n_per = pred.numel() // max(B, 1)
    grad = torch.empty_like(pred)
    N = int(pen_grad.numel())
    pen_out = torch.empty((N,), dtype=torch.float32, device=pred.device)
    if upstream.dtype not in (torch.float64, torch.float32):
        upstream = upstream.to(torch.float64)
    rc = load().sn_criterion_backward(_ptr(pred, None, "pred"), _DT[pred.dtype], _ptr(gt, None, "gt"), _DT[gt.dtype], B,
                                      n_per, _ptr(ranges, torch.float32, "ranges"), int(ranges.numel()),
                                      _ptr(coef, torch.float64, "coef"), _ptr(upstream, None, "upstream"),
                                      _DT[upstream.dtype], _ptr(grad), _ptr(pen_grad, torch.float32, "pen_grad"), N,
                                      _ptr(pen_out), _stream())
    _check(rc, "sn_criterion_backward")
    return grad, pen_out


@_on_tensor_device
def loss_backward(pred: torch.Tensor, gt: torch.Tensor, ranges: torch.Tensor, coef: torch.Tensor,
                  upstream: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dL/dpred (pred's dtype and shape) from the coefficients of loss_forward (sn_loss_backward_u); upstream: a
    one-element f64 or f32 tensor (or None = 1)."""
    B = int(pred.shape[0])
    n_per = pred.numel() // max(B, 1)
    grad = torch.empty_like(pred)
    if upstream is not None and upstream.dtype not in (torch.float64, torch.float32):
        upstream = upstream.to(torch.float64)
    up_dt = _DT[upstream.dtype] if upstream is not None else SN_F64
    rc = load().sn_loss_backward_u(_ptr(pred, None, "pred"), _DT[pred.dtype], _ptr(gt, None, "gt"), _DT[gt.dtype], B,
                                   n_per, _ptr(ranges, torch.float32, "ranges"), int(ranges.numel()),
                                   _ptr(coef, torch.float64, "coef"), _ptr(upstream, None, "upstream"), up_dt,
                                   _ptr(grad), _stream())
    _check(rc, "sn_loss_backward")
    return grad


@_on_tensor_device
def param_penalty(P: torch.Tensor, mask: torch.Tensor, weight: float, with_sum: bool):
    """sn_param_penalty: (value [1] f32, grad [N] f32) of the GENEO_Loss penalties over the packed parameters."""
    N = int(P.numel())
    value = torch.empty((1,), dtype=torch.float32, device=P.device)
    grad = torch.empty((N,), dtype=torch.float32, device=P.device)
    rc = load().sn_param_penalty(_ptr(P, torch.float32, "P"), _ptr(mask, torch.int8, "mask"), N, float(weight),
                                 int(bool(with_sum)), _ptr(value), _ptr(grad), _stream())
    _check(rc, "sn_param_penalty")
    return value, grad

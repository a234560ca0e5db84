"""ctypes binding of libtrs_hip.so (C ABI declared in include/trs_abi.h).

The library is the only compute path of this package: there is no CPU or eager-PyTorch fallback.
Loading fails loudly (RuntimeError) when the .so has not been built, and every op raises when it
is handed a tensor that is not on a HIP device.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRS_LIB_PATH") or os.path.join(_HERE, "libtrs_hip.so")      # TRS_LIB_PATH: developer override

TRS_F32, TRS_BF16 = 0, 1
TRS_I64, TRS_I32 = 0, 1

_P, _I32, _I64, _SZ, _F32 = c_void_p, c_int32, c_int64, c_size_t, ctypes.c_float

# name -> (restype, argtypes); must list every symbol of include/trs_abi.h (tests/test_abi.py checks)
SIGNATURES = {
    "trs_version": (c_int32, []),
    "trs_last_error_string": (c_char_p, []),
    "trs_rowdot_fwd": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _P, _P]),
    "trs_rowdot_bwd_workspace_bytes": (_SZ, [_I64, _I32]),
    "trs_rowdot_bwd": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _P, _P, _SZ, _P]),
    "trs_transpose_pad": (c_int32, [_P, _I64, _I32, _I32, _I32, _P, _I32, _I32, _P]),
    "trs_cat_head_fwd": (c_int32, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P]),
    "trs_cat_head_bwd_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32]),
    "trs_cat_head_bwd": (c_int32, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _SZ, _P]),
    "trs_wgrad_finish": (c_int32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "trs_wgrad_finish_t": (c_int32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "trs_wgrad_rows_splits": (c_int32, [_I32, _I32, _I64]),
    "trs_wgrad_rows": (c_int32, [_P, _I32, _P, _I32, _I64, _I32, _I32, _I32, _I32, _P, _P]),
    "trs_copy_padded_many": (c_int32, [_P, _I32, _I32, _I64, _P]),
    "trs_cin_glue_blocks": (c_int32, [_I64]),
    "trs_cin_glue_stats": (c_int32, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_cin_glue_fwd": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "trs_cin_glue_bwd_reduce": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "trs_cin_glue_bwd_apply": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "trs_cin_glue_cf_supported": (c_int32, [_I32, _I32]),
    "trs_cin_glue_fwd_cf": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "trs_cin_glue_bwd_apply_cf": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P,
                                            _P]),
    "trs_opn_vec_fwd": (c_int32, [_P, _P, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_opn_vec_bwd_workspace_bytes": (_SZ, [_I64, _I32, _I32]),
    "trs_opn_vec_bwd": (c_int32, [_P, _P, _P, _I32, _I64, _I32, _I32, _I32, _P, _P, _P, _SZ, _P]),
    "trs_pair_mul_fwd": (c_int32, [_P, _P, _P, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_rows_mul_bias_fwd": (c_int32, [_P, _P, _P, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_rows_mul_bwd": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _P, _P, _P]),
    "trs_pair_mul_bwd": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "trs_pair_bilinear_fwd": (c_int32, [_P, _P, _I32, _P, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_pair_bilinear_bwd_data": (c_int32, [_P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "trs_pair_bilinear_fwd_mfma": (c_int32, [_P, _P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_pair_bilinear_bwd_data_mfma": (c_int32, [_P, _P, _P, _P, _P, _I32, _P, _P, _I32, _P, _I32, _I64, _I32, _I32, _I32,
                                                  _P, _P, _P, _P]),
    "trs_pair_bilinear_bwd_w_mfma_workspace_bytes": (_SZ, [_I64, _I32, _I32]),
    "trs_pair_bilinear_bwd_w_mfma": (c_int32, [_P, _P, _I32, _I64, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "trs_pair_epilogue_fwd": (c_int32, [_P, _P, _P, _I32, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_pair_epilogue_bwd": (c_int32, [_P, _P, _P, _I32, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_afm_fwd": (c_int32, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "trs_afm_fwd_dropout": (c_int32, [_P, _P, _P, _P, _P, _P, _F32, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "trs_afm_bwd_dropout": (c_int32, [_P, _P, _P, _P, _P, _F32, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P,
                                      _P, _P, _SZ, _P]),
    "trs_afm_pair_tiles": (c_int32, [_I32, _P, _I32, _P]),
    "trs_pad_cols": (c_int32, [_P, _I32, _P, _I32, _I64, _I32, _P]),
    "trs_rows_gemm_workspace_bytes": (_SZ, [_I32, _I32]),
    "trs_rows_gemm_supported": (c_int32, [_I32, _I32, _I32]),
    "trs_rows_gemm": (c_int32, [_P, _I64, _I32, _P, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "trs_afm_bwd_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32]),
    "trs_afm_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "trs_pack_columns": (c_int32, [_P, _P, _I32, _I32, _I64, _P, _I32, _P]),
    "trs_mark_timestamp": (c_int32, [_P, _I32, _P]),
    "trs_wall_clock_khz": (_I64, []),
    "trs_gather_rows": (c_int32, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _I32, _P, _P, _P]),
    "trs_gather_rows_tables": (c_int32, [_P, _P, _I32, _I32, _P, _I32, _I64, _I32, _P, _P, _P]),
    "trs_fa_gather_rows": (c_int32, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _I32, _P, _P, _P]),
    "trs_csr_workspace_bytes": (_SZ, [_I64, _I64]),
    "trs_csr_build": (c_int32, [_P, _I32, _P, _I64, _I32, _I64, _P, _P, _P, _SZ, _P, _P]),
    "trs_scatter_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32]),
    "trs_scatter_rows": (c_int32, [_P, _I64, _P, _I32, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _I64, _P, _P, _SZ, _P]),
    "trs_scatter_rows_first": (c_int32, [_P, _I64, _P, _I32, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _I64, _P, _P, _P, _P,
                                         _SZ, _P]),
    "trs_scatter_rows_update": (c_int32, [_P, _I64, _P, _I32, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _I64, _I32,
                                          ctypes.c_float, ctypes.c_float, _P, _P, _SZ, _P]),
    "trs_scatter_rows_update_adam": (c_int32, [_P, _I64, _P, _I32, _P, _P, _P, _P, _I64, _I64, _I32, _I32, _I32, _I64,
                                               ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P,
                                               _P, _SZ, _P]),
    "trs_scatter_rows_update_mapped": (c_int32, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I32, _I32, _I32, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_float, _P, _P, _P, _SZ, _P]),
    "trs_embed_fm": (c_int32, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "trs_embed_fm_fields": (c_int32, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P]),
    "trs_fm_fwd": (c_int32, [_P, _I64, _I32, _I32, _I32, _P, _P, _P]),
    "trs_fm_bwd": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_pair_dot_fwd": (c_int32, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_embed_pair_dot": (c_int32, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _I32, _P, _P, _P, _P]),
    "trs_pair_dot_bwd": (c_int32, [_P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_ffm_fwd": (c_int32, [_P, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_ffm_bwd": (c_int32, [_P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_ffm_fused_fwd": (c_int32, [_P, _I64, _I32, _I32, _P, _I32, _P, _I64, _I32, _P, _P, _P]),
    "trs_ffm_fused_bwd": (c_int32, [_P, _I64, _I32, _I32, _P, _I32, _P, _P, _P, _P, _I64, _I32, _P, _P, _P]),
    "trs_cross_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32]),
    "trs_cross_fwd": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "trs_cross_bwd": (c_int32, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _SZ, _P]),
    "trs_cin_fwd": (c_int32, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "trs_cin_bwd": (c_int32, [_P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _I32, _P]),
    "trs_cin_cl_workspace_bytes": (_SZ, [_I32, _I32, _I32]),
    "trs_cin_cl_fwd": (c_int32, [_P, _I32, _P, _I32, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "trs_cin_cl_bwd_data_workspace_bytes": (_SZ, [_I32, _I32, _I32]),
    "trs_cin_cl_bwd_data": (c_int32, [_P, _I32, _P, _I32, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32,
                                      _P, _SZ, _P]),
    "trs_cin_dw_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32]),
    "trs_cin_dw": (c_int32, [_P, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "trs_cin_cl_bwd_data_live": (c_int32, [_P, _I32, _P, _I32, _P, _I32, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _I32,
                                           _P, _SZ, _P]),
    "trs_cin_dw_live": (c_int32, [_P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "trs_relu_bwd_bias_workspace_bytes": (_SZ, [_I64, _I32]),
    "trs_relu_bwd_bias": (c_int32, [_P, _P, _I64, _I32, _I32, _P, _P, _P, _SZ, _P]),
    "trs_mlp_fused_supported": (c_int32, [_I32, _P]),
    "trs_mlp_fused_family": (c_int32, [_I32, _P, _I64, _I32]),
    "trs_mlp_fused_workspace_bytes": (_SZ, [_I32, _P]),
    "trs_mlp_fused_mask_bytes": (_SZ, [_I64]),
    "trs_mlp_pack_branch": (c_int32, [_I64, _I32, _P, _P, _P, _I32, _P, _P, _SZ, _P, _I32, _I32, _P, _SZ, _P]),
    "trs_mlp_fused_fwd": (c_int32, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _SZ, _P]),
    "trs_mlp_fused_bwd_data": (c_int32, [_P, _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _SZ, _P]),
    "trs_ctr_logit_fwd": (c_int32, [_P, _I32, _P, _I32, _P, _P, _I32, _P, _I64, _I32, _P, _P]),
    "trs_bce_logits_workspace_bytes": (_SZ, [_I64]),
    "trs_bce_logits_fwd": (c_int32, [_P, _I32, _P, _I32, _I64, _P, _P, _SZ, _P]),
    "trs_bce_logits_bwd": (c_int32, [_P, _I32, _P, _I32, _P, _I64, _P, _P]),
    "trs_bucket_workspace_bytes": (_SZ, [_I64, _I32]),
    "trs_bucket_by_owner": (c_int32, [_P, _I32, _P, _I64, _I32, _I64, _I32, _P, _P, _P, _P, _P, _SZ, _P]),
    "trs_embed_fm_sharded": (c_int32, [_P, _I64, _P, _I64, _I64, _I32, _I32, _P, _P, _I32, _I32, _I64, _I32, _P, _P, _P, _P,
                                       _P]),
    "trs_permute_grad": (c_int32, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "trs_scatter_by_pos": (c_int32, [_P, _P, _I64, _I32, _I32, _P, _P]),
    "trs_gather_by_pos": (c_int32, [_P, _P, _I64, _I32, _I32, _P, _P]),
}

_lib = None
_lock = threading.Lock()


def load() -> ctypes.CDLL:
    """dlopen the in-tree library (once).  torch is imported first so that the library binds to the
    HIP runtime torch already loaded (same SONAME) and shares its streams and allocations."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"torecsys_amd: HIP library not built ({LIB_PATH} missing). Run "
                "`python -m torecsys_amd.build` (or __graft_entry__.build()). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:  # pragma: no cover
                raise RuntimeError(f"torecsys_amd: {LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    s = load().trs_last_error_string()
    return s.decode("utf-8", "replace") if s else ""


# name -> list of (start_event, end_event): filled when a name is registered with time_kernel();
# the events are recorded on the stream the kernel is launched on (torch's current stream).
_timed = {}


_hip = None


def _hiprt():
    """The HIP runtime torch already loaded (same SONAME), for the event calls of the timing hook."""
    global _hip
    if _hip is None:
        try:
            _hip = ctypes.CDLL("libamdhip64.so.7")       # resolves to the already loaded object by SONAME
        except OSError:
            _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        _hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        _hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
    return _hip


class _HipEvent:
    """A raw HIP event from a free list: hipEventCreate costs ~0.4 ms on this runtime, so events are created once and
    recycled when kernel_times_ms() has read them (two creations per timed launch made the timing hook the largest
    host cost of a step)."""
    _free: list = []

    def __init__(self):
        if _HipEvent._free:
            self.h = _HipEvent._free.pop()
        else:
            self.h = ctypes.c_void_p()
            if _hiprt().hipEventCreate(ctypes.byref(self.h)) != 0:
                raise RuntimeError("hipEventCreate failed")

    def record(self, stream, external):
        rt = _hiprt()
        rc = rt.hipEventRecordWithFlags(self.h, stream, 1) if external else rt.hipEventRecord(self.h, stream)
        if rc != 0:
            raise RuntimeError(f"hipEventRecord{'WithFlags(external)' if external else ''} failed (hipError {rc})")

    def elapsed_time(self, other) -> float:
        ms = ctypes.c_float()
        rc = _hiprt().hipEventElapsedTime(ctypes.byref(ms), self.h, other.h)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed ({rc})")
        return float(ms.value)

    def release(self):
        if self.h is not None:
            _HipEvent._free.append(self.h)
            self.h = None


_MARK_CAP = 4096          # timestamp ring entries (two per timed launch)
_rings = {}


def _mark_ring(name: str) -> torch.Tensor:
    r = _rings.get(name)
    if r is None:
        # allocated outside any capture pool (is_current_stream_capturing() callers create it before capturing)
        raise RuntimeError(f"time_kernel({name!r}) must be enabled before the capture starts")
    return r


_every = {}              # name -> [period, launches seen]


def time_kernel(name: str, enable: bool = True, expect: int = 0, every: int = 1):
    """Bracket launches of entry point ``name`` with HIP events (bench.py's live roofline); launches captured
    into a hipGraph are bracketed with device-side timestamp marks instead (see trs_mark_timestamp).
    ``expect``: number of launches that will be timed -- their events are created NOW (hipEventCreate costs ~0.4 ms
    here, which inside a timed region is enough to push a step from GPU-bound into host/GPU lock-step).
    ``every``: bracket only every ``every``-th eager launch.  A recorded timing event holds one of the runtime's
    profiling signals until it is read; with two records per step the host ends up waiting on the oldest one and
    the step runs in host/GPU lock-step (measured: 3.3 ms instead of 1.3 ms for the DeepFM step), so bench.py
    samples the kernel instead of bracketing all of its launches."""
    if enable:
        _timed[name] = []
        _every[name] = [max(1, int(every)), 0]
        _rings[name] = torch.zeros(_MARK_CAP + 1, dtype=torch.int64, device="cuda")
        fresh = [_HipEvent() for _ in range(max(0, 2 * expect - len(_HipEvent._free)))]
        for ev in fresh:
            ev.release()
    else:
        _timed.pop(name, None)
        _rings.pop(name, None)
        _every.pop(name, None)


def kernel_times_ms(name: str):
    """Elapsed milliseconds of every recorded launch of ``name`` (synchronises), and clears the records: HIP event
    pairs of eager launches followed by the timestamp-mark pairs of graph replays."""
    torch.cuda.synchronize()
    ev = _timed.get(name, [])
    out = [a.elapsed_time(b) for a, b in ev]
    for a, b in ev:
        a.release()
        b.release()
    ev.clear()
    ring = _rings.get(name)
    if ring is not None:
        host = ring.cpu()
        n = int(host[0])
        if n > _MARK_CAP:
            n = 0       # wrapped: pairs can no longer be told apart
        khz = float(load().trs_wall_clock_khz())
        if khz > 0:
            out += [float(host[2 + 2 * j] - host[1 + 2 * j]) / khz for j in range(n // 2)]
        ring.zero_()
    return out


def call(name: str, *args):
    """Call an int-returning entry point; raise RuntimeError with the library's message on failure."""
    rec = _timed.get(name)
    if rec is not None:
        st = stream_ptr()
        if torch.cuda.is_current_stream_capturing():
            # a captured event pair keeps only the latest replay (and ROCm refuses external event records in a
            # capture): bracket the launch with two device-side timestamp marks instead; each replay appends a sample
            ring = _mark_ring(name)
            lib = load()
            lib.trs_mark_timestamp(ctypes.c_void_p(ring.data_ptr()), _MARK_CAP, st)
            rc = getattr(lib, name)(*args)
            lib.trs_mark_timestamp(ctypes.c_void_p(ring.data_ptr()), _MARK_CAP, st)
        else:
            ev = _every[name]
            ev[1] += 1
            if ev[1] % ev[0] == 0:
                a, b = _HipEvent(), _HipEvent()
                a.record(st, 0)
                rc = getattr(load(), name)(*args)
                b.record(st, 0)
                rec.append((a, b))
            else:
                rc = getattr(load(), name)(*args)
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (code {rc}): {last_error()}")


def size_query(name: str, *args) -> int:
    return int(getattr(load(), name)(*args))


def value_dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return TRS_F32
    if t.dtype == torch.bfloat16:
        return TRS_BF16
    raise TypeError(f"torecsys_amd: unsupported value dtype {t.dtype} (float32 and bfloat16 only)")


def index_dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.int64:
        return TRS_I64
    if t.dtype == torch.int32:
        return TRS_I32
    raise TypeError(f"torecsys_amd: unsupported index dtype {t.dtype}")


def current_stream_of(dev: torch.device) -> torch.cuda.Stream:
    """torch.cuda.current_stream(dev) with an explicit index (the argument-less form costs a hipGetDeviceCount)."""
    return torch.cuda.current_stream(dev.index if dev.index is not None else torch._C._cuda_getDevice())


def require_device(*tensors: torch.Tensor) -> torch.device:
    """All tensors must live on one HIP device; anything else is an error (no CPU path)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "torecsys_amd: tensors must be on a HIP ('cuda') device; this package has no CPU path "
                f"(got a tensor on {t.device})")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"torecsys_amd: tensors on different devices ({dev} vs {t.device})")
    if dev is not None and dev.index is not None and dev.index != torch._C._cuda_getDevice():
        # the library enqueues on the CURRENT device's current stream (stream_ptr()) and never switches devices:
        # tensors of another device would be touched from the wrong device's stream (a fault, or an unordered race)
        raise RuntimeError(
            f"torecsys_amd: tensors live on {dev} but the current device is cuda:{torch._C._cuda_getDevice()}; "
            f"run under `with torch.cuda.device({dev.index})` (or torch.cuda.set_device) -- one process per GPU")
    return dev


def ptr(t) -> c_void_p:
    return c_void_p(0) if t is None else c_void_p(t.data_ptr())


def stream_ptr() -> c_void_p:
    # torch.cuda.current_stream() with no device argument resolves the device through torch.cuda.is_available(), i.e.
    # hipGetDeviceCount -- 50-200 us per call on this runtime, ~20 calls per training step.  The raw-stream query
    # with an explicit device index is a sub-microsecond C call.
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))

"""ctypes binding of librange_hip.so (C ABI: include/range_hip.h).

There is NO CPU fallback: if the library is missing or no gfx950 GPU is visible the product path
raises.  Only plain pointers / sizes cross the boundary; torch is used for device memory and
streams only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

LIB_NAME = "librange_hip.so"
# RANGE_LIB_PATH: load another build of the library (tuning sweeps; see tools/topk_stream_sweep.sh)
LIB_PATH = os.environ.get("RANGE_LIB_PATH") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

KEY_DIM, VAL_DIM, OUT_DIM = 256, 1024, 1280
SH_ANALYTIC, SH_CLOSED_FORM = 0, 1
MODEL_RANGE, MODEL_RANGE_PLUS = 0, 1
MAX_TOPK = 16
PROF_ENCODER, PROF_SCAN_STATS, PROF_ATTEND, PROF_TOPK_STREAM, PROF_TOPK_MERGE = 0, 1, 2, 3, 4
COORD_DIRECT, COORD_CARTESIAN3D, COORD_WRAP = 0, 1, 2   # range_coord_features(mode)
COORD_DIMS = {COORD_DIRECT: 2, COORD_CARTESIAN3D: 3, COORD_WRAP: 4}

# every symbol include/range_hip.h declares
SYMBOLS = (
    "range_abi_version", "range_last_error", "range_build_flags", "range_source_sha256", "range_create", "range_destroy", "range_set_encoder",
    "range_set_sh_table",
    "range_set_bank", "range_bank_rows", "range_encode", "range_scan_stats", "range_merge_stats",
    "range_merge_topk", "range_attend", "range_finalize", "range_forward",
    "range_last_attend_geometry", "range_profile_enable", "range_profile_read",
    "range_attend_diag", "range_encode_raw", "range_blend", "range_topk_stream",
    "range_coord_features", "range_attend_kept", "range_kept_queries", "range_forward_host",
    "range_host_copy", "range_topk_stream_exact_count", "range_topk_stream_timed",
    "range_set_pv_mode", "range_get_pv_mode", "range_set_keys", "range_debug_raise_async_error",
    "range_scan_stats_at", "range_p1_splits", "range_check_async_error", "range_stream_read_timed",
    "range_async_error_flag", "range_topk_last",
)
PV_MODES = {"exact": 0, "bf16x3": 1}   # range_set_pv_mode


class EncoderDesc(C.Structure):
    _fields_ = [("legendre_polys", C.c_int32), ("hidden", C.c_int32),
                ("num_hidden_layers", C.c_int32), ("embed_dim", C.c_int32),
                ("sh_mode", C.c_int32)]


class RangeNativeError(RuntimeError):
    pass


_lib = None


def load_library() -> C.CDLL:
    """Load librange_hip.so (built in-tree by ./build.sh or __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RangeNativeError(
            f"{LIB_PATH} not found. Build it with ./build.sh (hipcc --offload-arch=gfx950). "
            "range_amd has no CPU fallback: the HIP extension is the product.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.range_abi_version.restype = C.c_int
    lib.range_last_error.restype = C.c_char_p
    lib.range_build_flags.restype = C.c_char_p
    lib.range_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.range_destroy.argtypes = [vp]
    lib.range_destroy.restype = None
    lib.range_set_encoder.argtypes = [vp, C.POINTER(EncoderDesc), C.POINTER(vp), C.POINTER(vp)]
    lib.range_set_sh_table.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp]
    lib.range_set_bank.argtypes = [vp, vp, vp, vp, i64, i64]
    lib.range_set_keys.argtypes = [vp, vp, i64, i64]
    lib.range_debug_raise_async_error.argtypes = [vp, vp]
    lib.range_check_async_error.argtypes = [vp]
    lib.range_bank_rows.argtypes = [vp]
    lib.range_bank_rows.restype = i64
    lib.range_encode.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.range_scan_stats.argtypes = [vp, vp, vp, i64, f32, f32, vp, C.c_int, vp, vp, i32, vp]
    lib.range_scan_stats_at.argtypes = [vp, vp, vp, i64, f32, f32, vp, i64, i64, i32, vp]
    lib.range_p1_splits.argtypes = [vp, i64]
    lib.range_p1_splits.restype = i32
    lib.range_attend_kept.argtypes = [vp, i64, vp, i64, f32, f32, f32, vp, vp, vp]
    lib.range_kept_queries.argtypes = [vp]
    lib.range_kept_queries.restype = i64
    lib.range_merge_stats.argtypes = [vp, vp, i32, i64, vp, vp]
    lib.range_merge_topk.argtypes = [vp, vp, vp, i32, i64, i32, vp, vp, vp]
    lib.range_attend.argtypes = [vp, vp, vp, i64, f32, f32, f32, vp, vp, vp]
    lib.range_finalize.argtypes = [vp, vp, i32, vp, i64, vp, vp]
    lib.range_forward.argtypes = [vp, vp, i64, i32, f32, vp, vp]
    lib.range_forward_host.argtypes = [vp, vp, i64, i32, f32, vp, vp]
    lib.range_host_copy.argtypes = [vp, vp, vp, C.c_size_t]
    lib.range_last_attend_geometry.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.range_profile_enable.argtypes = [vp, i32]
    lib.range_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(i32)]
    lib.range_attend_diag.argtypes = [vp, vp, vp, i64, f32, f32, f32, vp, vp, i64, vp]
    lib.range_encode_raw.argtypes = [vp, vp, i64, vp, vp]
    lib.range_blend.argtypes = [vp, vp, vp, f32, i64, vp, vp]
    lib.range_topk_stream.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    lib.range_topk_stream_exact_count.argtypes = [vp, C.POINTER(i64)]
    lib.range_topk_last.argtypes = [vp, i64, i32, vp, vp, vp]
    lib.range_async_error_flag.argtypes = [vp, vp, vp]
    lib.range_topk_stream_timed.argtypes = [vp, vp, i64, i32, vp, vp, i32, C.POINTER(f32), vp]
    lib.range_stream_read_timed.argtypes = [vp, i32, i32, i32, C.POINTER(f32), vp]
    lib.range_coord_features.argtypes = [vp, i32, vp, i64, vp, vp]
    lib.range_set_pv_mode.argtypes = [vp, i32]
    lib.range_get_pv_mode.argtypes = [vp]
    lib.range_get_pv_mode.restype = i32
    for name in SYMBOLS:
        getattr(lib, name)
    if lib.range_abi_version() != 9:
        raise RangeNativeError("librange_hip.so ABI version mismatch (rebuild with ./build.sh)")
    # the library must be built from THIS checkout's sources (a stale in-tree .so travels with the
    # snapshot: it is git-ignored, not gpurun-ignored); RANGE_LIB_PATH builds (tuning) are exempt
    lib.range_source_sha256.restype = C.c_char_p
    if not os.environ.get("RANGE_LIB_PATH"):
        from ._srchash import SourcesMissing, source_sha256
        built = lib.range_source_sha256().decode()
        try:
            here = source_sha256()
        except SourcesMissing as ex:
            # a deployment that ships the built library without range_amd/csrc and include/: there is
            # nothing to compare the stamp with - say so once and go on with the library as built
            import warnings
            warnings.warn(f"range_amd: kernel sources not found ({ex}); {LIB_PATH} (source stamp {built[:12]}) is used "
                          "as built, its stamp unchecked", RuntimeWarning, stacklevel=2)
            here = built
        if built != here:
            raise RangeNativeError(
                f"{LIB_PATH} was built from other sources (stamp {built[:12]}, this checkout {here[:12]}): "
                "rebuild with ./build.sh or __graft_entry__.build()")
    flags = lib.range_build_flags().decode()
    if "RANGE_EXP_" in flags and os.environ.get("RANGE_ALLOW_EXPERIMENT_BUILD") != "1":
        raise RangeNativeError(
            f"{LIB_PATH} was built with timing-experiment switches ({flags}): its results are "
            "invalid.  Rebuild with ./build.sh (RANGE_ALLOW_EXPERIMENT_BUILD=1 overrides, tuning only).")
    _lib = lib
    return lib


def _check(lib, rc: int) -> None:
    if rc != 0:
        raise RangeNativeError(f"librange_hip error {rc}: {lib.range_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class HipEngine:
    """One engine context on one GPU.  All tensor arguments are CUDA(HIP) tensors on that GPU;
    work is enqueued on torch's current stream for the device."""

    def __init__(self, device: torch.device | int | str = 0):
        self.lib = load_library()
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise RangeNativeError(f"range_amd needs a GPU device, got {dev}")
        if not torch.cuda.is_available():
            raise RangeNativeError("no GPU visible: range_amd runs only on MI355X (gfx950); "
                                   "there is no CPU fallback")
        self.device = torch.device("cuda", dev.index if dev.index is not None
                                   else torch.cuda.current_device())
        h = C.c_void_p()
        _check(self.lib, self.lib.range_create(self.device.index, C.byref(h)))
        self._h = h
        self.n_rows = 0
        self.row_offset = 0

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.range_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup ---------------------------------------------------------------------------------
    def set_encoder(self, L: int, hidden: int, num_hidden_layers: int, embed_dim: int,
                    sh_mode: int, weights: Sequence[np.ndarray],
                    biases: Sequence[np.ndarray], sh_table=None) -> None:
        """``sh_table`` (range_amd.sh_table.SHTable, analytic mode only): evaluate the spherical
        harmonics through the reference's generated polynomials instead of the exact recurrence."""
        n = num_hidden_layers + 1
        if len(weights) != n or len(biases) != n:
            raise ValueError("need num_hidden_layers+1 weight and bias arrays")
        ws = [np.ascontiguousarray(w, dtype=np.float64) for w in weights]
        bs = [np.ascontiguousarray(b, dtype=np.float64) for b in biases]
        dims_in = [L * L] + [hidden] * num_hidden_layers
        dims_out = [hidden] * num_hidden_layers + [embed_dim]
        for i in range(n):
            if ws[i].shape != (dims_out[i], dims_in[i]) or bs[i].shape != (dims_out[i],):
                raise ValueError(f"layer {i}: weight {ws[i].shape} / bias {bs[i].shape} do not "
                                 f"match ({dims_out[i]},{dims_in[i]})")
        desc = EncoderDesc(L, hidden, num_hidden_layers, embed_dim, sh_mode)
        wp = (C.c_void_p * n)(*[w.ctypes.data for w in ws])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        _check(self.lib, self.lib.range_set_encoder(self._h, C.byref(desc), wp, bp))
        if sh_table is not None:
            t = sh_table
            f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)   # noqa: E731
            i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)     # noqa: E731
            arrs = [f64(t.front), f64(t.a0), f64(t.a2), i32(t.p2), i32(t.kx), i32(t.off), i32(t.cnt)]
            coef, powr = f64(t.coef), i32(t.pow)
            if any(a.shape != (L * L,) for a in arrs) or coef.shape != powr.shape:
                raise ValueError("spherical-harmonics table does not match L")
            _check(self.lib, self.lib.range_set_sh_table(
                self._h, L, *[a.ctypes.data for a in arrs], coef.shape[0], coef.ctypes.data,
                powr.ctypes.data))

    def set_bank(self, keys: np.ndarray, values: np.ndarray, xyz: np.ndarray,
                 row_offset: int = 0) -> None:
        keys = np.ascontiguousarray(keys, dtype=np.float32)
        values = np.ascontiguousarray(values, dtype=np.float32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n = keys.shape[0]
        if keys.shape != (n, KEY_DIM) or values.shape != (n, VAL_DIM) or xyz.shape != (n, 3):
            raise ValueError(f"bank shapes {keys.shape} {values.shape} {xyz.shape}")
        _check(self.lib, self.lib.range_set_bank(self._h, keys.ctypes.data, values.ctypes.data,
                                                 xyz.ctypes.data, n, row_offset))
        self.n_rows = n
        self.row_offset = row_offset

    def set_keys(self, keys, row_offset: int = 0) -> None:
        """A keys-only bank for ``topk_stream`` (range_set_keys): ``keys`` (n,256) float32, a host
        ndarray or a tensor on this engine's GPU.  Forward / attend calls then raise."""
        if torch.is_tensor(keys):
            if keys.is_cuda:
                self._t(keys, torch.float32, (KEY_DIM,))
                n, ptr = keys.shape[0], keys.data_ptr()
                torch.cuda.current_stream(self.device).synchronize()   # (the library copies on the null stream)
            else:
                keys = keys.numpy()
        if not torch.is_tensor(keys):
            keys = np.ascontiguousarray(keys, dtype=np.float32)
            if keys.ndim != 2 or keys.shape[1] != KEY_DIM:
                raise ValueError(f"keys shape {keys.shape}")
            n, ptr = keys.shape[0], keys.ctypes.data
        _check(self.lib, self.lib.range_set_keys(self._h, ptr, n, row_offset))
        self.n_rows = n
        self.row_offset = row_offset

    def set_pv_mode(self, mode: str = "exact") -> None:
        """Arithmetic of pass 2's w @ V: "exact" (default: float32 products, what the reference
        computes) or the opt-in "bf16x3" (three bf16 planes per operand, six cross products; see
        include/range_hip.h: range_set_pv_mode)."""
        if mode not in PV_MODES:
            raise ValueError(f"pv_mode must be one of {sorted(PV_MODES)}, got {mode!r}")
        _check(self.lib, self.lib.range_set_pv_mode(self._h, PV_MODES[mode]))

    @property
    def pv_mode(self) -> str:
        m = self.lib.range_get_pv_mode(self._h)
        return {v: k for k, v in PV_MODES.items()}[m]

    def check_async_error(self) -> None:
        """Raise if a persistent launch of an earlier call on this engine gave up waiting for its
        workgroups (range_hip.h: range_check_async_error).  Costs one read of host memory; meant
        to be called BEHIND a synchronisation (``.cpu()``, an event, a collective's result on the
        host) - the failed call's rows are NaN by then and the re-issued call takes the fall-back."""
        _check(self.lib, self.lib.range_check_async_error(self._h))

    def async_error_flag(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The give-up words as a float64 DEVICE scalar in stream order (range_async_error_flag): 1.0
        when a persistent launch enqueued so far on this engine has given up (and the host has not
        looked yet).  ``out``: a one-element float64 view to write into (a slot of a send buffer)."""
        if out is None:
            out = self._empty((1,), torch.float64)
        elif out.dtype != torch.float64 or out.numel() != 1 or out.device != self.device:
            raise ValueError("out must be one float64 element on the engine's device")
        _check(self.lib, self.lib.range_async_error_flag(self._h, out.data_ptr(), self._stream()))
        return out

    def debug_fail_next_persistent_launch(self) -> None:
        """Test hook (range_debug_raise_async_error)."""
        _check(self.lib, self.lib.range_debug_raise_async_error(self._h, None))

    # -- helpers -------------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _t(self, t: torch.Tensor, dtype, shape_tail) -> torch.Tensor:
        if t.device != self.device:
            raise ValueError(f"tensor on {t.device}, engine on {self.device}")
        if t.dtype != dtype or tuple(t.shape[1:]) != tuple(shape_tail) or not t.is_contiguous():
            raise ValueError(f"expected contiguous {dtype} (*,{shape_tail}), got {t.dtype} {tuple(t.shape)}")
        return t

    def _empty(self, shape, dtype) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    # -- kernels -------------------------------------------------------------------------------
    def encode(self, lonlat: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        self._t(lonlat, torch.float64, (2,))
        B = lonlat.shape[0]
        e64 = self._empty((B, KEY_DIM), torch.float64)
        e32 = self._empty((B, KEY_DIM), torch.float32)
        xq = self._empty((B, 4), torch.float32)
        _check(self.lib, self.lib.range_encode(self._h, lonlat.data_ptr(), B, e64.data_ptr(),
                                               e32.data_ptr(), xq.data_ptr(), self._stream()))
        return e64, e32, xq

    def encode_raw(self, lonlat: torch.Tensor) -> torch.Tensor:
        """Un-normalised SatCLIP embedding (B,256) float64 (range.py:244-245)."""
        self._t(lonlat, torch.float64, (2,))
        B = lonlat.shape[0]
        out = self._empty((B, KEY_DIM), torch.float64)
        _check(self.lib, self.lib.range_encode_raw(self._h, lonlat.data_ptr(), B, out.data_ptr(),
                                                   self._stream()))
        return out

    def coord_features(self, lonlat: torch.Tensor, mode: int) -> torch.Tensor:
        """Training-free coordinate encoders (range.py:262-272): Direct / Cartesian_3D / Wrap."""
        self._t(lonlat, torch.float64, (2,))
        B = lonlat.shape[0]
        out = self._empty((B, COORD_DIMS[mode]), torch.float64)
        _check(self.lib, self.lib.range_coord_features(self._h, mode, lonlat.data_ptr(), B,
                                                       out.data_ptr(), self._stream()))
        return out

    def blend(self, G: torch.Tensor, H: torch.Tensor, beta: float) -> torch.Tensor:
        """(1-beta)*G + beta*H with the reference's float32 rounding (range.py:238)."""
        self._t(G, torch.float32, (VAL_DIM,))
        self._t(H, torch.float32, (VAL_DIM,))
        out = self._empty(tuple(G.shape), torch.float32)
        _check(self.lib, self.lib.range_blend(self._h, G.data_ptr(), H.data_ptr(), beta,
                                              G.shape[0], out.data_ptr(), self._stream()))
        return out

    def scan_stats(self, e32: torch.Tensor, xq: torch.Tensor, tau_sem: float, tau_geo: float,
                   topk: int = 0, keep_logits: bool = False):
        """Pass 1.  ``keep_logits``: keep the raw semantic logits of this call in the context for
        ``attend_kept`` (see ``kept_queries``)."""
        self._t(e32, torch.float32, (KEY_DIM,))
        self._t(xq, torch.float32, (4,))
        B = e32.shape[0]
        stats = self._empty((B, 4), torch.float32)
        tv = ti = None
        if topk:
            tv = self._empty((B, topk), torch.float32)
            ti = self._empty((B, topk), torch.int64)
        _check(self.lib, self.lib.range_scan_stats(self._h, e32.data_ptr(), xq.data_ptr(), B,
                                                   tau_sem, tau_geo, stats.data_ptr(), topk,
                                                   _ptr(tv), _ptr(ti), int(bool(keep_logits)),
                                                   self._stream()))
        return (stats, tv, ti) if topk else stats

    def scan_stats_at(self, e32: torch.Tensor, xq: torch.Tensor, tau_sem: float, tau_geo: float,
                      first_query: int, total_queries: int, n_splits: int = 0) -> torch.Tensor:
        """Pass 1 for queries [first_query, first_query + len(e32)) of a scan of ``total_queries``
        queries done in chunks; the chunks' logits are kept in ONE workspace that ``attend_kept``
        addresses as if a single ``scan_stats`` had kept them (range_hip.h: range_scan_stats_at).
        ``n_splits``: the bank splits of the launch (0: by this chunk's geometry; ``p1_splits``)."""
        self._t(e32, torch.float32, (KEY_DIM,))
        self._t(xq, torch.float32, (4,))
        B = e32.shape[0]
        stats = self._empty((B, 4), torch.float32)
        _check(self.lib, self.lib.range_scan_stats_at(self._h, e32.data_ptr(), xq.data_ptr(), B, tau_sem,
                                                      tau_geo, stats.data_ptr(), first_query, total_queries,
                                                      n_splits, self._stream()))
        return stats

    def p1_splits(self, n_queries: int) -> int:
        """The bank splits a pass-1 launch of ``n_queries`` queries chooses."""
        return int(self.lib.range_p1_splits(self._h, n_queries))

    def kept_queries(self) -> int:
        """Queries whose logits the last scan_stats(keep_logits=True) - or the scan_stats_at calls of
        the current scan so far - kept (0: none)."""
        return int(self.lib.range_kept_queries(self._h))

    def attend_kept(self, first_query: int, xq: torch.Tensor, tau_sem: float, tau_geo: float,
                    beta: float, stats: torch.Tensor) -> torch.Tensor:
        """Pass 2 for queries [first_query, first_query + len(xq)) of the last kept scan, from the
        kept logits (bit-identical to ``attend``)."""
        self._t(xq, torch.float32, (4,))
        self._t(stats, torch.float32, (4,))
        B = xq.shape[0]
        if stats.shape[0] != B:
            raise ValueError("xq and stats must have the same number of rows")
        out = self._empty((B, VAL_DIM), torch.float32)
        _check(self.lib, self.lib.range_attend_kept(self._h, first_query, xq.data_ptr(), B, tau_sem,
                                                    tau_geo, beta, stats.data_ptr(), out.data_ptr(),
                                                    self._stream()))
        return out

    def topk_stream(self, e32: torch.Tensor, k: int):
        """Small-batch top-k by the HBM-streaming kernel (see range_hip.h)."""
        self._t(e32, torch.float32, (KEY_DIM,))
        B = e32.shape[0]
        tv = self._empty((B, k), torch.float32)
        ti = self._empty((B, k), torch.int64)
        _check(self.lib, self.lib.range_topk_stream(self._h, e32.data_ptr(), B, k, tv.data_ptr(),
                                                    ti.data_ptr(), self._stream()))
        return tv, ti

    def topk_last(self, B: int, k: int):
        """Top-k of the B queries the last ``forward`` / ``forward_host`` of this engine embedded
        (range_topk_last: their e-hat is still in the workspace - no second encoder pass)."""
        tv = self._empty((B, k), torch.float32)
        ti = self._empty((B, k), torch.int64)
        _check(self.lib, self.lib.range_topk_last(self._h, B, k, tv.data_ptr(), ti.data_ptr(), self._stream()))
        return tv, ti

    def topk_stream_timed(self, e32: torch.Tensor, k: int, repeats: int = 20):
        """topk_stream with the stream kernel launched ``repeats`` times back to back between one
        pair of events: returns (values, indices, microseconds per stream-kernel launch)."""
        self._t(e32, torch.float32, (KEY_DIM,))
        B = e32.shape[0]
        tv = self._empty((B, k), torch.float32)
        ti = self._empty((B, k), torch.int64)
        us = C.c_float()
        _check(self.lib, self.lib.range_topk_stream_timed(self._h, e32.data_ptr(), B, k, tv.data_ptr(),
                                                          ti.data_ptr(), repeats, C.byref(us),
                                                          self._stream()))
        return tv, ti, us.value

    def stream_read_timed(self, f32_keys: bool = False, passes: int = 1, repeats: int = 20) -> float:
        """Microseconds per launch of a plain streaming read of the bytes ``topk_stream`` streams
        (``passes`` x the bf16 copy of the keys, or the float32 keys): its same-launch ceiling."""
        us = C.c_float()
        _check(self.lib, self.lib.range_stream_read_timed(self._h, int(bool(f32_keys)), passes, repeats,
                                                          C.byref(us), self._stream()))
        return us.value

    def topk_stream_exact_count(self) -> int:
        """Queries topk_stream recomputed by brute force (its short per-lane lists could have
        dropped a top-k member) since this engine was created."""
        n = C.c_int64()
        _check(self.lib, self.lib.range_topk_stream_exact_count(self._h, C.byref(n)))
        return n.value

    def merge_stats(self, parts: torch.Tensor) -> torch.Tensor:
        if parts.dim() != 3 or parts.shape[2] != 4:
            raise ValueError("parts must be (n_parts,B,4)")
        self._t(parts, torch.float32, parts.shape[1:])
        P, B = parts.shape[0], parts.shape[1]
        out = self._empty((B, 4), torch.float32)
        _check(self.lib, self.lib.range_merge_stats(self._h, parts.data_ptr(), P, B,
                                                    out.data_ptr(), self._stream()))
        return out

    def merge_topk(self, val_parts: torch.Tensor, idx_parts: torch.Tensor):
        P, B, k = val_parts.shape
        self._t(val_parts, torch.float32, (B, k))
        self._t(idx_parts, torch.int64, (B, k))
        ov = self._empty((B, k), torch.float32)
        oi = self._empty((B, k), torch.int64)
        _check(self.lib, self.lib.range_merge_topk(self._h, val_parts.data_ptr(),
                                                   idx_parts.data_ptr(), P, B, k, ov.data_ptr(),
                                                   oi.data_ptr(), self._stream()))
        return ov, oi

    def attend(self, e32: torch.Tensor, xq: torch.Tensor, tau_sem: float, tau_geo: float,
               beta: float, stats: torch.Tensor) -> torch.Tensor:
        self._t(e32, torch.float32, (KEY_DIM,))
        self._t(xq, torch.float32, (4,))
        self._t(stats, torch.float32, (4,))
        B = e32.shape[0]
        out = self._empty((B, VAL_DIM), torch.float32)
        _check(self.lib, self.lib.range_attend(self._h, e32.data_ptr(), xq.data_ptr(), B, tau_sem,
                                               tau_geo, beta, stats.data_ptr(), out.data_ptr(),
                                               self._stream()))
        return out

    def finalize(self, partials: torch.Tensor, e64: torch.Tensor,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if partials.dim() == 2:
            partials = partials.unsqueeze(0)
        P, B = partials.shape[0], partials.shape[1]
        self._t(partials, torch.float32, (B, VAL_DIM))
        self._t(e64, torch.float64, (KEY_DIM,))
        if out is None:
            out = self._empty((B, OUT_DIM), torch.float64)
        elif out.shape[0] != B:
            raise ValueError("out must have one row per query")
        else:
            self._t(out, torch.float64, (OUT_DIM,))
        _check(self.lib, self.lib.range_finalize(self._h, partials.data_ptr(), P, e64.data_ptr(),
                                                 B, out.data_ptr(), self._stream()))
        return out

    def forward(self, lonlat: torch.Tensor, model: int, beta: float,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._t(lonlat, torch.float64, (2,))
        B = lonlat.shape[0]
        if out is None:
            out = self._empty((B, OUT_DIM), torch.float64)
        else:
            self._t(out, torch.float64, (OUT_DIM,))
        _check(self.lib, self.lib.range_forward(self._h, lonlat.data_ptr(), B, model, beta,
                                                out.data_ptr(), self._stream()))
        return out

    def forward_host(self, lonlat: torch.Tensor, model: int, beta: float,
                     out: Optional[np.ndarray] = None) -> np.ndarray:
        """The reference's contract (range/range.py:240): the (B,1280) float64 result as a host
        ndarray - a new one (see ``_hostpool``) unless ``out`` (C-contiguous rows of a float64
        array) is given.
        Synchronous; the device->host copy and the fill of the array are pipelined per slab
        inside the library (range_forward_host)."""
        self._t(lonlat, torch.float64, (2,))
        B = lonlat.shape[0]
        if out is None:
            from ._hostpool import POOL
            out = POOL.take(B, OUT_DIM)     # memory of results the caller has dropped, else fresh
        elif out.shape != (B, OUT_DIM) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous float64 array (B,1280)")
        _check(self.lib, self.lib.range_forward_host(self._h, lonlat.data_ptr(), B, model, beta,
                                                     out.ctypes.data, self._stream()))
        return out

    def host_copy(self, dst: np.ndarray, src: np.ndarray) -> None:
        """dst[...] = src on the host with the library's copy threads (both C-contiguous, same
        size in bytes): spreads the first-touch page faults of a fresh ``dst`` over cores."""
        if dst.nbytes != src.nbytes or not dst.flags.c_contiguous or not src.flags.c_contiguous:
            raise ValueError("host_copy needs C-contiguous arrays of equal size")
        _check(self.lib, self.lib.range_host_copy(self._h, dst.ctypes.data, src.ctypes.data,
                                                  dst.nbytes))

    def profile_enable(self, on: bool = True) -> None:
        _check(self.lib, self.lib.range_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, which: int) -> Tuple[float, int]:
        """(summed device ms, launches) of kernel ``which`` (0 encoder, 1 scan_stats, 2 attend)
        since profile_enable(); measured with HIP events on the launch stream."""
        ms, n = C.c_double(), C.c_int32()
        _check(self.lib, self.lib.range_profile_read(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def attend_diag(self, e32, xq, tau_sem, tau_geo, beta, stats) -> torch.Tensor:
        """Diagnostic kernel build: (workgroups, 4 waves, 16) int64 cycle sums (see range_hip.h)."""
        B = e32.shape[0]
        n = ((B + 63) // 64) * 64 * 4 * 16
        diag = torch.zeros(n, dtype=torch.int64, device=self.device)
        _check(self.lib, self.lib.range_attend_diag(self._h, e32.data_ptr(), xq.data_ptr(), B,
                                                    tau_sem, tau_geo, beta, stats.data_ptr(),
                                                    diag.data_ptr(), n, self._stream()))
        qt, ns = self.last_geometry()
        return diag[: qt * ns * 64].reshape(qt * ns, 4, 16)

    def last_geometry(self) -> Tuple[int, int]:
        a, b = C.c_int32(), C.c_int32()
        _check(self.lib, self.lib.range_last_attend_geometry(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

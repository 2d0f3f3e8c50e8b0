"""ctypes binding of include/range_probe.h (the ridge-probe entry points of librange_hip.so).

``ProbeEngine`` is a thin, typed wrapper: every method maps to exactly one C entry point and works
on CUDA(HIP) tensors of the engine's device, enqueued on torch's current stream.  No arithmetic
happens here; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native
from ._native import RangeNativeError, _check

# every symbol include/range_probe.h declares
SYMBOLS = (
    "range_probe_create", "range_probe_destroy", "range_probe_colstats", "range_probe_scale_rows",
    "range_probe_onehot", "range_probe_gemm", "range_probe_gram", "range_probe_sum_parts",
    "range_probe_solve", "range_probe_r2_sums", "range_probe_accuracy",
)

_bound = False


def load_library() -> C.CDLL:
    global _bound
    lib = _native.load_library()
    if _bound:
        return lib
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.range_probe_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.range_probe_destroy.argtypes = [vp]
    lib.range_probe_destroy.restype = None
    lib.range_probe_colstats.argtypes = [vp, vp, i64, i32, i64, vp, vp, vp, vp]
    lib.range_probe_scale_rows.argtypes = [vp, vp, i64, i32, i64, vp, vp, vp, vp, vp, i64, vp]
    lib.range_probe_onehot.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp]
    lib.range_probe_gemm.argtypes = [vp, i32, i32, i32, i32, i32, f64, vp, i64, vp, i64, f64, vp,
                                     i64, i32, vp]
    lib.range_probe_gram.argtypes = [vp, vp, i64, vp, i64, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.range_probe_sum_parts.argtypes = [vp, vp, i32, i64, vp, vp]
    lib.range_probe_solve.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, i32, i32,
                                      i32, vp, vp, vp]
    lib.range_probe_r2_sums.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]
    lib.range_probe_accuracy.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, vp, vp, vp]
    for name in SYMBOLS:
        getattr(lib, name)
    _bound = True
    return lib


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class ProbeEngine:
    """One probe context on one GPU."""

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
        _check(self.lib, self.lib.range_probe_create(self.device.index, C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self.lib.range_probe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _chk(self, t: torch.Tensor, dtype, ndim: int) -> torch.Tensor:
        if t.device != self.device or t.dtype != dtype or t.dim() != ndim or not t.is_contiguous():
            raise ValueError(f"expected contiguous {dtype} {ndim}-d tensor on {self.device}, got "
                             f"{t.dtype} {tuple(t.shape)} on {t.device}")
        return t

    def empty(self, shape, dtype=torch.float64) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    # -- entry points --------------------------------------------------------------------------
    def colstats(self, X: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Column (min, max, sum) of X (n,d) float64."""
        self._chk(X, torch.float64, 2)
        n, d = X.shape
        mn, mx, sm = self.empty(d), self.empty(d), self.empty(d)
        _check(self.lib, self.lib.range_probe_colstats(self._h, X.data_ptr(), n, d, d, _p(mn),
                                                       _p(mx), _p(sm), self._stream()))
        return mn, mx, sm

    def scale_rows(self, X: torch.Tensor, perm: Optional[torch.Tensor] = None,
                   scale: Optional[torch.Tensor] = None, offset: Optional[torch.Tensor] = None,
                   shift: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Z[i] = (X[perm[i]] * scale + offset) - shift; returns Z (len(perm) or n, d)."""
        self._chk(X, torch.float64, 2)
        d = X.shape[1]
        n = X.shape[0] if perm is None else perm.shape[0]
        if perm is not None:
            self._chk(perm, torch.int64, 1)
        for v in (scale, offset, shift):
            if v is not None and tuple(self._chk(v, torch.float64, 1).shape) != (d,):
                raise ValueError("per-column vector of the wrong length")
        Z = self.empty((n, d))
        if n:
            _check(self.lib, self.lib.range_probe_scale_rows(
                self._h, X.data_ptr(), n, d, d, _p(perm), _p(scale), _p(offset), _p(shift),
                Z.data_ptr(), d, self._stream()))
        return Z

    def onehot(self, code: torch.Tensor, c: int, first: int, shift: torch.Tensor) -> torch.Tensor:
        """T[i,k] = (+1 if code[i] == first+k else -1) - shift[k]; returns (n,c) float64."""
        self._chk(code, torch.int32, 1)
        self._chk(shift, torch.float64, 1)
        n = code.shape[0]
        T = self.empty((n, c))
        _check(self.lib, self.lib.range_probe_onehot(self._h, code.data_ptr(), n, c, first,
                                                     shift.data_ptr(), T.data_ptr(),
                                                     self._stream()))
        return T

    def gemm(self, A: torch.Tensor, B: torch.Tensor, trans_a: bool = False, trans_b: bool = False,
             alpha: float = 1.0, beta: float = 0.0, out: Optional[torch.Tensor] = None,
             lower_only: bool = False) -> torch.Tensor:
        """out = alpha * op(A) @ op(B) + beta * out on the float64 matrix cores."""
        self._chk(A, torch.float64, 2)
        self._chk(B, torch.float64, 2)
        M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
        K2, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
        if K != K2:
            raise ValueError(f"inner dimensions differ: {K} vs {K2}")
        if out is None:
            if beta != 0.0:
                raise ValueError("beta != 0 needs out")
            out = self.empty((M, N))
        self._chk(out, torch.float64, 2)
        if tuple(out.shape) != (M, N):
            raise ValueError("out has the wrong shape")
        _check(self.lib, self.lib.range_probe_gemm(
            self._h, int(trans_a), int(trans_b), M, N, K, alpha, A.data_ptr(), A.shape[1],
            B.data_ptr(), B.shape[1], beta, out.data_ptr(), N, int(lower_only), self._stream()))
        return out

    def gram(self, Z: torch.Tensor, T: torch.Tensor, G: torch.Tensor, B: torch.Tensor,
             zsum: torch.Tensor, tsum: torch.Tensor) -> None:
        """G = Z^T Z (lower), B = Z^T T, column sums of Z and T, for one block of rows; the
        outputs are preallocated (slices of the per-fold arrays)."""
        self._chk(Z, torch.float64, 2)
        self._chk(T, torch.float64, 2)
        rows, d = Z.shape
        c = T.shape[1]
        if T.shape[0] != rows or tuple(G.shape) != (d, d) or tuple(B.shape) != (d, c) or \
                tuple(zsum.shape) != (d,) or tuple(tsum.shape) != (c,):
            raise ValueError("gram: shape mismatch")
        for t in (G, B, zsum, tsum):
            if not t.is_contiguous() or t.dtype != torch.float64 or t.device != self.device:
                raise ValueError("gram: outputs must be contiguous float64 on the engine device")
        _check(self.lib, self.lib.range_probe_gram(self._h, Z.data_ptr(), d, T.data_ptr(), c, rows,
                                                   d, c, G.data_ptr(), B.data_ptr(),
                                                   zsum.data_ptr(), tsum.data_ptr(),
                                                   self._stream()))

    def sum_parts(self, parts: torch.Tensor) -> torch.Tensor:
        """Sum over the leading axis of a contiguous float64 array."""
        if parts.dtype != torch.float64 or not parts.is_contiguous() or parts.device != self.device:
            raise ValueError("sum_parts: contiguous float64 on the engine device")
        out = self.empty(parts.shape[1:])
        _check(self.lib, self.lib.range_probe_sum_parts(self._h, parts.data_ptr(), parts.shape[0],
                                                        out.numel(), out.data_ptr(),
                                                        self._stream()))
        return out

    def solve(self, Gtot, Btot, zsum_tot, tsum_tot, ntr: Sequence[float], alphas: Sequence[float],
              Gf=None, Bf=None, zsumf=None, tsumf=None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Ridge coefficients W (groups,d,n_alpha,c) and intercepts c0 (groups,n_alpha,c)."""
        d, c = Btot.shape
        groups = len(ntr)
        n_alpha = len(alphas)
        ntr_h = np.ascontiguousarray(ntr, dtype=np.float64)
        al_h = np.ascontiguousarray(alphas, dtype=np.float64)
        W = self.empty((groups, d, n_alpha, c))
        c0 = self.empty((groups, n_alpha, c))
        if Gf is not None and (tuple(Gf.shape) != (groups, d, d) or tuple(Bf.shape) != (groups, d, c)
                               or tuple(zsumf.shape) != (groups, d)
                               or tuple(tsumf.shape) != (groups, c)):
            raise ValueError("solve: fold statistics of the wrong shape")
        _check(self.lib, self.lib.range_probe_solve(
            self._h, Gtot.data_ptr(), Btot.data_ptr(), zsum_tot.data_ptr(), tsum_tot.data_ptr(),
            _p(Gf), _p(Bf), _p(zsumf), _p(tsumf), ntr_h.ctypes.data, groups, al_h.ctypes.data,
            n_alpha, d, c, W.data_ptr(), c0.data_ptr(), self._stream()))
        return W, c0

    def r2_sums(self, P: torch.Tensor, c0: torch.Tensor, T: torch.Tensor, tsum: torch.Tensor,
                n_alpha: int) -> torch.Tensor:
        """(n_alpha, c, 2): residual and total sums of squares per alpha and target."""
        rows, c = T.shape
        out = self.empty((n_alpha, c, 2))
        _check(self.lib, self.lib.range_probe_r2_sums(self._h, P.data_ptr(), c0.data_ptr(),
                                                      T.data_ptr(), rows, c, n_alpha,
                                                      tsum.data_ptr(), out.data_ptr(),
                                                      self._stream()))
        return out

    def accuracy(self, P: torch.Tensor, c0: torch.Tensor, code: torch.Tensor, c: int, n_alpha: int,
                 n_cls: int, present: Optional[torch.Tensor]) -> torch.Tensor:
        """(n_alpha,) int64 counts of correctly classified rows."""
        self._chk(code, torch.int32, 1)
        hits = torch.zeros(n_alpha, dtype=torch.int64, device=self.device)
        _check(self.lib, self.lib.range_probe_accuracy(self._h, P.data_ptr(), c0.data_ptr(),
                                                       code.data_ptr(), code.shape[0], c, n_alpha,
                                                       n_cls, _p(present), hits.data_ptr(),
                                                       self._stream()))
        return hits

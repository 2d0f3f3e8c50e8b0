"""``LocationEncoder`` for RANGE / RANGE+ on MI355X - the drop-in for the reference's
range/range.py:69-278 (RANGE branches: __init__ :76-114, forward :208-242).

Same constructor contract (an ``argparse.Namespace`` with ``location_model_name``,
``pretrained_path``, ``device``, ``range_db``, ``beta``), same attributes other code reads
(``location_feature_dim``, ``args.temp`` / ``args.geo_temp`` / ``args.beta``), same exceptions for
the same conditions, same call: ``model(coords)`` with ``coords`` a (B,2) float64 tensor of
(lon, lat) degrees returns a host ``numpy.ndarray`` (B,1280) float64.

All arithmetic runs in hand-written HIP kernels behind librange_hip.so (range_amd/_native.py);
there is no torch-op or CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import _native
from ._hostpool import POOL
from .bank import PreparedBank
from .bankfile import load_any as load_bank
from .ckpt import EncoderParams, read_checkpoint

TEMP_RANGE = 15.0        # range/range.py:103
TEMP_RANGE_PLUS = 12.0   # range/range.py:108
TEMP_GEO = 40.0          # range/range.py:109

# training-free coordinate encoders: load_model name -> (kernel mode, the reference's banner)
_COORD_MODELS = {
    "Direct": (_native.COORD_DIRECT, "Using Direct Encoding"),          # range.py:153-156
    "Cartesian_3D": (_native.COORD_CARTESIAN3D, "Using Cartesian_3D"),   # range.py:159-162
    "Wrap": (_native.COORD_WRAP, "Using Wrap"),                          # range.py:170-173
}


def _device_of(spec) -> torch.device:
    dev = torch.device(spec)
    if dev.type != "cuda":
        raise RuntimeError(
            f"range_amd runs on MI355X GPUs only (device={spec!r}); there is no CPU path. "
            "Use the reference implementation for CPU inference.")
    return dev


_SH_TABLES = {}     # (L, source path or None) -> SHTable


def sh_table_for(enc: EncoderParams, sh_eval: Optional[str] = None, sh_source: Optional[str] = None):
    """The coefficient table of the reference's generated "analytic" spherical harmonics, or None.

    ``sh_eval``: 'reference' (default for harmonics_calculation == 'analytic': the reference's own
    expanded polynomials with their 15-digit coefficients, so that embeddings agree with the
    reference at every latitude) or 'exact' (the stable recurrence: the mathematically exact basis,
    which the reference itself only matches for |lat| <~ 45 deg).  'closed-form' checkpoints always
    use the recurrence - it is what the reference runs for them
    (spherical_harmonics_closed_form.py:8-40).  ``sh_source``: path of a generated
    ``spherical_harmonics_ylm.py`` to take the coefficients from (default: regenerate them)."""
    if sh_eval is None:
        sh_eval = "reference"
    if sh_eval not in ("reference", "exact"):
        raise ValueError(f"sh_eval must be 'reference' or 'exact', got {sh_eval!r}")
    if enc.harmonics_calculation != "analytic" or sh_eval == "exact":
        return None
    from . import sh_table
    key = (enc.legendre_polys, sh_source)
    if key not in _SH_TABLES:
        if sh_source is not None:
            with open(sh_source) as f:
                _SH_TABLES[key] = sh_table.parse_ylm_source(f.read(), enc.legendre_polys)
        else:
            _SH_TABLES[key] = sh_table.generate_table(enc.legendre_polys)
    return _SH_TABLES[key]


def make_engine(enc: EncoderParams, bank: Optional[PreparedBank], device, row_offset: int = 0,
                sh_eval: Optional[str] = None, sh_source: Optional[str] = None,
                pv_mode: Optional[str] = None):
    """``pv_mode``: None / 'exact' (the reference's float32 products) or the opt-in 'bf16x3'
    (include/range_hip.h: range_set_pv_mode)."""
    eng = _native.HipEngine(device)
    if pv_mode is not None:
        eng.set_pv_mode(pv_mode)
    mode = _native.SH_ANALYTIC if enc.harmonics_calculation == "analytic" else _native.SH_CLOSED_FORM
    eng.set_encoder(enc.legendre_polys, enc.hidden, enc.num_hidden_layers, enc.embed_dim, mode,
                    enc.weights, enc.biases, sh_table=sh_table_for(enc, sh_eval, sh_source))
    if bank is not None:
        eng.set_bank(bank.keys, bank.values, bank.xyz, row_offset)
    return eng


class _FrozenLinear(nn.Module):
    """The parameters of one SirenNet layer (``weight`` (out,in), ``bias`` (out,) float64, frozen), under
    the reference's names (satclip/location_encoder.py:121-151)."""

    def __init__(self, weight: np.ndarray, bias: np.ndarray, device):
        super().__init__()
        self.weight = nn.Parameter(torch.as_tensor(np.asarray(weight), dtype=torch.float64, device=device), requires_grad=False)
        self.bias = nn.Parameter(torch.as_tensor(np.asarray(bias), dtype=torch.float64, device=device), requires_grad=False)


class _SirenParams(nn.Module):
    def __init__(self, enc: EncoderParams, device):
        super().__init__()
        n = enc.num_hidden_layers
        self.layers = nn.ModuleList([_FrozenLinear(enc.weights[i], enc.biases[i], device) for i in range(n)])
        self.last_layer = _FrozenLinear(enc.weights[n], enc.biases[n], device)


class SatCLIPLocationModel(nn.Module):
    """``model.loc_model`` of the reference (range/range.py:83-84, 119-121: ``get_satclip(...).double()``
    = satclip/location_encoder.py:267-275 ``LocationEncoder(posenc, nnet)``): a module whose call maps
    (B,2) float64 (lon,lat) degrees to the RAW - un-normalised - (B,256) float64 SatCLIP embedding on
    the device, and whose parameters are the SirenNet's, frozen (range.py:201-203), float64, on the
    engine's GPU, under the reference's names (``nnet.layers.{i}.weight`` ...: the checkpoint's
    ``model.location.*`` keys) - so that ``next(model.parameters()).device``, ``requires_grad`` checks,
    ``state_dict()`` and ``model.to(device)`` behave as on the reference.  The parameters are a
    read-only MIRROR: the arithmetic runs in the engine (fused HIP kernel A on its own re-packed copy of
    the same weights); writing to them does not change the model."""

    def __init__(self, engine, enc: EncoderParams, chunk_size: int = 16384):
        super().__init__()
        self.nnet = _SirenParams(enc, engine.device)
        self._engine = [engine]          # (a list: not a sub-module, not part of the state dict)
        self._chunk = chunk_size
        self.eval()

    @torch.no_grad()
    def forward(self, coords: torch.Tensor) -> torch.Tensor:
        eng = self._engine[0]
        if not torch.is_tensor(coords):
            coords = torch.as_tensor(np.asarray(coords))
        if coords.dim() != 2 or coords.shape[1] != 2:
            raise ValueError(f"coords must be (B,2) (lon,lat) degrees, got {tuple(coords.shape)}")
        x = coords.to(device=eng.device, dtype=torch.float64).contiguous()
        if x.shape[0] == 0:
            return torch.empty((0, 256), dtype=torch.float64, device=x.device)
        parts = [eng.encode_raw(x[i:i + self._chunk]) for i in range(0, x.shape[0], self._chunk)]
        return parts[0] if len(parts) == 1 else torch.cat(parts)


class _CoordLocationModel(nn.Module):
    """``loc_model`` of the training-free encoders: the reference's ``DummyLocationEncoder`` (identity;
    'Direct', 'Cartesian_3D': range.py:155, 161) and ``Wrap()`` (:172) - parameter-free modules there
    too (``next(model.parameters())`` raises StopIteration on both sides)."""

    def __init__(self, engine, mode):
        super().__init__()
        self._engine, self._mode = [engine], mode

    @torch.no_grad()
    def forward(self, x):
        if self._mode != _native.COORD_WRAP:
            return x                                                      # DummyLocationEncoder
        eng = self._engine[0]
        x = torch.as_tensor(x).to(device=eng.device, dtype=torch.float64).contiguous()
        return eng.coord_features(x, _native.COORD_WRAP) if x.shape[0] else torch.empty((0, 4), dtype=torch.float64, device=x.device)


class LocationEncoder(nn.Module):
    """RANGE / RANGE+ retrieval-augmented location encoder (reference: range/range.py:69)."""

    #: queries per engine call; bounds the per-call workspace (split slabs of chunk x 4 KB)
    chunk_size = 16384
    #: topk(): batches up to this size go through the HBM-streaming kernel in one call (bf16-key
    #: prefilter + float32 re-rank; one launch up to 256 queries: ~20 us for 16 queries on
    #: range_db_large, ~41 us for 64 (BENCH_r04.json roofline_scan) - faster than pass 1 + a selection
    #: over its kept logits at every size, tools/topk_total_time.py); larger batches in chunks of this size
    topk_stream_max = 16384

    def __init__(self, args):
        super().__init__()
        self.args = args
        self.location_model_name = args.location_model_name
        if "RANGE" in self.location_model_name:                        # range.py:76
            bank = load_bank(args.range_db)                            # :78-95
            enc = read_checkpoint(args.pretrained_path)                # :82-84
            if enc.embed_dim != 256:
                raise ValueError(f"checkpoint embed_dim {enc.embed_dim} != bank key width 256")
            self.location_feature_dim = 1024 + 256                     # :86
            if self.location_model_name == "RANGE":                    # :102-105
                self.args.temp = TEMP_RANGE
                self._model_id = _native.MODEL_RANGE
                print(f"Using RANGE with temperature {self.args.temp}")
            elif self.location_model_name == "RANGE+":                 # :107-112
                self.args.geo_temp = TEMP_GEO
                self.args.temp = TEMP_RANGE_PLUS
                self._model_id = _native.MODEL_RANGE_PLUS
                print(f"Using RANGE+ with temperatures {self.args.temp} and {self.args.geo_temp}")
            else:
                raise ValueError("Unimplemented RANGE model")           # :113-114
            self.encoder_params = enc
            self.n_bank_rows = bank.n_rows
            self._device = _device_of(args.device)
            self.engine = make_engine(enc, bank, self._device, sh_eval=getattr(args, "sh_eval", None),
                                      sh_source=getattr(args, "sh_source", None),
                                      pv_mode=getattr(args, "pv_mode", None))
            self.loc_model = SatCLIPLocationModel(self.engine, enc, self.chunk_size)   # :83-84
        elif self.location_model_name == "SatCLIP":                     # range.py:117-122
            print("Using SatCLIP")
            enc = read_checkpoint(args.pretrained_path)
            if enc.embed_dim != 256:
                raise ValueError(f"checkpoint embed_dim {enc.embed_dim}: only 256 is implemented")
            self.location_feature_dim = 256
            self._model_id = None
            self.encoder_params = enc
            self._device = _device_of(args.device)
            self.engine = make_engine(enc, None, self._device, sh_eval=getattr(args, "sh_eval", None),
                                      sh_source=getattr(args, "sh_source", None))
            self.loc_model = SatCLIPLocationModel(self.engine, enc, self.chunk_size)   # :119-121
        elif self.location_model_name in _COORD_MODELS:                 # range.py:152-162, 170-173
            mode, banner = _COORD_MODELS[self.location_model_name]
            print(banner)
            self.location_feature_dim = _native.COORD_DIMS[mode]
            self._model_id = None
            self._coord_mode = mode
            self._device = _device_of(args.device)
            self.engine = _native.HipEngine(self._device)
            self.loc_model = _CoordLocationModel(self.engine, mode)
        else:
            # the reference dispatches more encoder families here (GeoCLIP, CSP, SINR, TaxaBind,
            # Theory, sphere2vec; range.py:124-198): third-party pretrained baselines, out of
            # scope for this engine
            raise NotImplementedError(f"{self.location_model_name} not implemented")
        self.loc_model.eval()                                               # range.py:201-203
        for params in self.loc_model.parameters():
            params.requires_grad = False
        self.eval()

    def _coords(self, coords) -> torch.Tensor:
        if not torch.is_tensor(coords):
            coords = torch.as_tensor(np.asarray(coords))
        if coords.dim() != 2 or coords.shape[1] != 2:
            raise ValueError(f"coords must be (B,2) (lon,lat) degrees, got {tuple(coords.shape)}")
        # the reference requires float64 input (F.linear against .double() weights);
        # float32 input is accepted here by widening
        return coords.to(device=self.engine.device, dtype=torch.float64).contiguous()

    @torch.no_grad()
    def forward(self, coords, return_device: bool = False, return_topk: Optional[int] = None):
        """coords (B,2) float64 (lon,lat) deg -> (B,1280) float64 ``numpy.ndarray`` on the host
        (range.py:222/240).  ``return_device=True`` returns the device tensor instead (no D2H).

        ``return_topk=k`` (1..16; SURVEY.md 8(b) "Call" - the reference only hints at it,
        range.py:232): the call returns ``(embeddings, values (B,k) float32, indices (B,k) int64)``,
        the k bank rows most similar to each query (cosine, semantic keys; descending, ties to the
        lower row) as device tensors - ``model.topk(coords, k)``'s result bit for bit, from the SAME
        call: the queries are encoded once and the scan runs on the e-hat the forward left in the
        engine's workspace (``range_topk_last``).  A NaN / infinite coordinate gives a NaN row (as in
        the reference; rows are independent) and an undefined top-k for that row."""
        x = self._coords(coords)
        B = x.shape[0]
        if return_topk is not None:
            k = int(return_topk)
            if self._model_id is None:
                raise ValueError("return_topk needs a bank (RANGE / RANGE+)")
            if not 1 <= k <= _native.MAX_TOPK:
                raise ValueError(f"return_topk must be in 1..{_native.MAX_TOPK}, got {return_topk}")
            return self._forward_with_topk(x, k, return_device)
        if getattr(self, "_coord_mode", None) is not None:
            # Direct / Wrap return a device tensor, Cartesian_3D a host ndarray (its rad_to_cart
            # runs in numpy, range.py:265-268)
            d = self.location_feature_dim
            out = (self.engine.coord_features(x, self._coord_mode) if B else
                   torch.empty((0, d), dtype=torch.float64, device=x.device))
            if self._coord_mode == _native.COORD_CARTESIAN3D and not return_device:
                return out.cpu().numpy()
            return out
        if self._model_id is None:
            # plain SatCLIP: the un-normalised (B,256) float64 embedding, a device tensor like
            # the reference's (range.py:244-245)
            if B == 0:
                return torch.empty((0, 256), dtype=torch.float64, device=x.device)
            return torch.cat([self.engine.encode_raw(x[i:i + self.chunk_size])
                              for i in range(0, B, self.chunk_size)])
        beta = 1.0 if self._model_id == _native.MODEL_RANGE else float(self.args.beta)
        # (B == 0: nothing to launch; the reference returns an empty (0,1280) array as well)
        if not return_device:
            # the reference's contract: a fresh host array (range.py:240), filled slab by slab
            # while the device->host copies of later slabs are in flight (range_forward_host)
            host = POOL.take(B, _native.OUT_DIM)
            for i in range(0, B, self.chunk_size):
                self.engine.forward_host(x[i:i + self.chunk_size], self._model_id, beta,
                                         out=host[i:i + self.chunk_size])
            return host
        out = torch.empty((B, _native.OUT_DIM), dtype=torch.float64, device=x.device)
        for i in range(0, B, self.chunk_size):
            self.engine.forward(x[i:i + self.chunk_size], self._model_id, beta,
                                out=out[i:i + self.chunk_size])
        return out

    def _forward_with_topk(self, x: torch.Tensor, k: int, return_device: bool):
        B = x.shape[0]
        beta = 1.0 if self._model_id == _native.MODEL_RANGE else float(self.args.beta)
        tv = torch.empty((B, k), dtype=torch.float32, device=x.device)
        ti = torch.empty((B, k), dtype=torch.int64, device=x.device)
        if return_device:
            out = torch.empty((B, _native.OUT_DIM), dtype=torch.float64, device=x.device)
        else:
            out = POOL.take(B, _native.OUT_DIM)
        for i in range(0, B, self.chunk_size):
            xc = x[i:i + self.chunk_size]
            n = xc.shape[0]
            if return_device:
                self.engine.forward(xc, self._model_id, beta, out=out[i:i + n])
            else:
                self.engine.forward_host(xc, self._model_id, beta, out=out[i:i + n])
            # (behind the forward on the same stream: the host result above is complete, the scan of
            # this chunk's e-hat runs while the caller - or the next chunk's encoder launch - goes on)
            v, j = self.engine.topk_last(n, k)
            tv[i:i + n], ti[i:i + n] = v, j
        return out, tv, ti

    @torch.no_grad()
    def sweep(self, coords, betas, return_device: bool = False):
        """RANGE+ embeddings of the same queries for several beta values (BASELINE config
        "beta sweep").  beta only enters the blend of range.py:238, so the semantic retrieval H
        and the geographic retrieval G are computed ONCE (two pass-2 launches instead of one per
        beta) and blended per beta with the reference's float32 rounding.
        Returns an array (len(betas), B, 1280) float64 (host ndarray, or device tensor)."""
        if self._model_id != _native.MODEL_RANGE_PLUS:
            raise ValueError("sweep() is defined for RANGE+ only")
        betas = [float(b) for b in betas]
        x = self._coords(coords)
        B = x.shape[0]
        out = torch.empty((len(betas), B, _native.OUT_DIM), dtype=torch.float64, device=x.device)
        eng = self.engine
        for i in range(0, B, self.chunk_size):
            e64, e32, xq = eng.encode(x[i:i + self.chunk_size])
            st = eng.scan_stats(e32, xq, TEMP_RANGE_PLUS, TEMP_GEO, keep_logits=True)
            if eng.kept_queries() == e32.shape[0]:      # both passes 2 from the kept logits
                H = eng.attend_kept(0, xq, TEMP_RANGE_PLUS, TEMP_GEO, 1.0, st)
                G = eng.attend_kept(0, xq, TEMP_RANGE_PLUS, TEMP_GEO, 0.0, st)
            else:
                H = eng.attend(e32, xq, TEMP_RANGE_PLUS, TEMP_GEO, 1.0, st)
                G = eng.attend(e32, xq, TEMP_RANGE_PLUS, TEMP_GEO, 0.0, st)
            for j, b in enumerate(betas):
                out[j, i:i + e64.shape[0]] = eng.finalize(eng.blend(G, H, b), e64)
        return out if return_device else self._to_host(out)

    def _to_host(self, t: torch.Tensor) -> np.ndarray:
        """``.cpu()`` synchronises: a persistent launch of this call that gave up (NaN rows) is known
        now - report it instead of handing the rows out (range_hip.h: range_check_async_error)."""
        h = t.cpu().numpy()
        self.engine.check_async_error()
        return h

    @torch.no_grad()
    def topk(self, coords, k: int = 16):
        """Side channel: the k bank rows most similar (cosine, semantic keys) to each query,
        descending; returns (values (B,k) float32, indices (B,k) int64) device tensors."""
        x = self._coords(coords)
        if self._model_id is None:
            raise ValueError("topk() needs a bank (RANGE / RANGE+)")
        vals, idxs = [], []
        # the HBM-streaming kernel (every wave streams its own key tiles against 16 or 32 queries
        # per pass); the forward's own top-k (pass 1 keeps the logits anyway) is scan_stats(topk=k)
        for i in range(0, x.shape[0], self.topk_stream_max):
            _, e32, _ = self.engine.encode(x[i:i + self.topk_stream_max])
            tv, ti = self.engine.topk_stream(e32, k)
            vals.append(tv)
            idxs.append(ti)
        if len(vals) == 1:
            return vals[0], idxs[0]
        if not vals:
            return (torch.empty((0, k), dtype=torch.float32, device=x.device),
                    torch.empty((0, k), dtype=torch.int64, device=x.device))
        return torch.cat(vals), torch.cat(idxs)


class ShardedLocationEncoder(nn.Module):
    """RANGE / RANGE+ over a bank ROW-SHARDED across the ranks of a ``torch.distributed`` group (one
    process per GPU, backend "nccl" = RCCL over xGMI): ``load_model(..., shards=W)`` in every rank of
    a W-process job.  The reference has no distributed code; the call surface is its
    ``LocationEncoder``'s (range/range.py:69, :206-240), so the same script runs under
    ``torchrun --nproc-per-node W`` unchanged:

      ``model(coords)``              every rank passes the SAME (B,2) batch and gets the FULL
                                     (B,1280) float64 ``numpy.ndarray`` back (rank r embeds rows
                                     [r B/W, (r+1) B/W) against all shards, one all-gather of the
                                     results follows) - any B, also B < W;
      ``model(coords, local=True)``  data-parallel callers: ``coords`` are THIS rank's own queries
                                     (any count per rank, zero allowed), the result is their rows.

    Each rank loads and uploads only its rows of the bank (a ``.rbank`` file is memory-mapped, so a
    rank reads just its slice; an ``.npz`` is read whole by every rank and sliced)."""

    is_sharded = True

    def __init__(self, args, group=None):
        super().__init__()
        import torch.distributed as dist
        from .dist import ShardedRange, make_layout, shard_rows
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("load_model(..., shards=W) needs an initialised torch.distributed job "
                               "(torchrun --nproc-per-node W; range_amd.dist.init_from_env())")
        self.args = args
        self.location_model_name = args.location_model_name
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        want = getattr(args, "shards", None)
        if isinstance(want, int) and not isinstance(want, bool) and want != self.world:
            raise ValueError(f"shards={want} but the process group has {self.world} ranks")
        if self.location_model_name == "RANGE":                          # range.py:102-105
            self.args.temp = TEMP_RANGE
            print(f"Using RANGE with temperature {self.args.temp}")
        elif self.location_model_name == "RANGE+":                       # :107-112
            self.args.geo_temp = TEMP_GEO
            self.args.temp = TEMP_RANGE_PLUS
            print(f"Using RANGE+ with temperatures {self.args.temp} and {self.args.geo_temp}")
        else:
            raise ValueError("Unimplemented RANGE model")                 # :113-114
        bank = load_bank(args.range_db)
        enc = read_checkpoint(args.pretrained_path)
        if enc.embed_dim != 256:
            raise ValueError(f"checkpoint embed_dim {enc.embed_dim} != bank key width 256")
        self.location_feature_dim = 1024 + 256                           # :86
        self.encoder_params = enc
        self.n_bank_rows = bank.n_rows
        # 2-D layout (dist.make_layout): the bank row-sharded over `row_shards` ranks (default: all of
        # them), world / row_shards such groups each serving its own queries
        self.row_shards = int(getattr(args, "row_shards", None) or self.world)
        self.shard_group, shard_index, self.query_group = make_layout(self.row_shards, group)
        if self.n_bank_rows < self.row_shards:
            raise ValueError(f"bank of {self.n_bank_rows} rows cannot be sharded over {self.row_shards} ranks")
        self.row_range = shard_rows(bank.n_rows, self.row_shards, shard_index)
        self._device = _device_of(args.device)
        self.engine = make_engine(enc, bank.rows(*self.row_range), self._device, row_offset=self.row_range[0],
                                  sh_eval=getattr(args, "sh_eval", None), sh_source=getattr(args, "sh_source", None),
                                  pv_mode=getattr(args, "pv_mode", None))
        self.sharded = ShardedRange(self.engine, self.location_model_name, args.beta, group=self.shard_group)
        self.loc_model = SatCLIPLocationModel(self.engine, enc)          # range.py:83-84 (replicated on every rank)
        self.loc_model.eval()                                            # :201-203
        for params in self.loc_model.parameters():
            params.requires_grad = False
        self.eval()

    def _coords(self, coords) -> torch.Tensor:
        if not torch.is_tensor(coords):
            coords = torch.as_tensor(np.asarray(coords))
        if coords.dim() != 2 or coords.shape[1] != 2:
            raise ValueError(f"coords must be (B,2) (lon,lat) degrees, got {tuple(coords.shape)}")
        return coords.to(device=self.engine.device, dtype=torch.float64).contiguous()

    def _own_rows(self, B: int):
        return (B * self.rank) // self.world, (B * (self.rank + 1)) // self.world

    def _gather_rows(self, own: torch.Tensor, B: int, with_flags: bool = False):
        """Every rank's rows of a full-batch result -> the full result on every rank (one padded
        all-gather: the row counts differ by at most one).  ``with_flags`` (float64 rows): one more row
        travels with every rank's share, carrying its engine's give-up flag (range_async_error_flag:
        written on the device, in stream order behind the rank's kernels); returns (rows, (W,) flags)."""
        import torch.distributed as dist
        W = self.world
        per = (B + W - 1) // W
        extra = 1 if with_flags else 0
        send = torch.zeros((per + extra,) + tuple(own.shape[1:]), dtype=own.dtype, device=own.device)
        send[:own.shape[0]] = own
        if with_flags and hasattr(self.engine, "async_error_flag"):
            self.engine.async_error_flag(out=send[per].view(-1)[0:1])
        staged = send.is_cuda and dist.get_backend(self.group) == "gloo"
        src = send.cpu() if staged else send
        allr = torch.empty((W * (per + extra),) + tuple(own.shape[1:]), dtype=own.dtype, device=src.device)
        dist.all_gather_into_tensor(allr, src, group=self.group)
        parts = []
        for r in range(W):
            n = (B * (r + 1)) // W - (B * r) // W
            parts.append(allr[r * (per + extra):r * (per + extra) + n])
        full = torch.cat(parts, dim=0)
        if not with_flags:
            return full
        return full, allr.view(W, per + 1, -1)[:, per, 0].clone()

    @torch.no_grad()
    def forward(self, coords, return_device: bool = False, local: bool = False, return_topk: Optional[int] = None):
        """``return_topk=k``: (embeddings, values (B,k) float32, GLOBAL bank rows (B,k) int64), the
        top-k from the same call (``LocationEncoder.forward``): the queries are encoded and gathered
        once, the per-shard candidates merge through ONE all-gather (``ShardedRange.forward``)."""
        x = self._coords(coords)
        k = None
        if return_topk is not None:
            k = int(return_topk)
            if not 1 <= k <= _native.MAX_TOPK:
                raise ValueError(f"return_topk must be in 1..{_native.MAX_TOPK}, got {return_topk}")
        if local:
            res = self.sharded.embed(x, topk=k)
            if not k:
                return res if return_device else self._to_host(res)
            return (res[0] if return_device else self._to_host(res[0])), res[1], res[2]
        B = x.shape[0]
        lo, hi = self._own_rows(B)
        res = self.sharded.embed(x[lo:hi], topk=k)
        own, tk = (res[0], res[1:]) if k else (res, ())
        full, flags = self._gather_rows(own, B, with_flags=True)
        tk = tuple(self._gather_rows(t, B).to(self.engine.device) for t in tk)
        emb = full.to(self.engine.device) if return_device else self._to_host(full, flags)   # range.py:240: a host ndarray
        return (emb, *tk) if k else emb

    @torch.no_grad()
    def sweep(self, coords, betas, return_device: bool = False, local: bool = False):
        """(len(betas), B, 1280) float64 for several beta values (BASELINE config "beta sweep")."""
        if self.location_model_name != "RANGE+":
            raise ValueError("sweep() is defined for RANGE+ only")
        x = self._coords(coords)
        if local:
            out = self.sharded.embed_sweep(x, betas)
            return out if return_device else self._to_host(out)
        B = x.shape[0]
        lo, hi = self._own_rows(B)
        own = self.sharded.embed_sweep(x[lo:hi], betas)                  # (nb, b_own, 1280)
        full, flags = self._gather_rows(own.permute(1, 0, 2).contiguous(), B, with_flags=True)
        full = full.permute(1, 0, 2).contiguous()
        return full.to(self.engine.device) if return_device else self._to_host(full, flags)

    def _to_host(self, t: torch.Tensor, flags: Optional[torch.Tensor] = None) -> np.ndarray:
        """``.cpu()`` synchronises: a persistent launch of THIS rank that gave up is known now and is
        reported (range_hip.h: range_check_async_error).  ANOTHER rank's give-up arrives as its FLAG
        (``_gather_rows``: the rank's error word, written on the device behind its kernels, travels with
        its rows): every rank holds the same flags, so every rank refuses the result here - together,
        without a collective of its own.  The verdict comes from the word, never from the data: a NaN
        or infinite input coordinate gives a NaN row on every path - as in the reference, whose rows
        are independent - and is handed out like any other row."""
        h = t.cpu().numpy()
        bad = [] if flags is None else [int(r) for r in np.flatnonzero(flags.cpu().numpy() != 0.0)]
        self.engine.check_async_error()
        if bad:
            raise RuntimeError(f"sharded forward: a persistent launch of rank(s) {bad} of the group gave up (their rows of this "
                               "result are NaN; each reports it on its side and runs separate launches from now on): "
                               "re-issue the call")
        return h

    @torch.no_grad()
    def topk(self, coords, k: int = 16, local: bool = False):
        """Global top-k over all shards: (values (B,k) float32, global bank rows (B,k) int64)."""
        x = self._coords(coords)
        if local:
            return self.sharded.embed_topk(x, k)
        B = x.shape[0]
        lo, hi = self._own_rows(B)
        tv, ti = self.sharded.embed_topk(x[lo:hi], k)
        return (self._gather_rows(tv, B).to(self.engine.device), self._gather_rows(ti, B).to(self.engine.device))

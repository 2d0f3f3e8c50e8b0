"""CPU oracle for the RANGE / RANGE+ retrieval-augmented forward path.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement of the reference's algorithm for the
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it, and there only as the checker / the timed CPU baseline - never as the product.  The
product path (``range_amd``) runs hand-written HIP kernels through ``librange_hip.so`` and fails
loudly when that library is missing; it has no CPU fallback.

Pinning.  The reference ships no tests, golden vectors or fixtures for this path (SURVEY.md
section 4 and 8(c): "parity unpinned" by the reference itself).  The pin is therefore made here:
``tests/golden/*.npz`` were produced by importing the reference's own Python from
``/root/reference`` in the development container (script: ``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks this restatement against every one of them (bitwise on the
retrieval half given the reference's own e-hat; <=1e-6 on e-hat inside the well-conditioned
latitude band, see ``sh_features``).

Reference lines each function follows are cited in its docstring (paths relative to
``/root/reference``).  Retrieval ops are issued through ``torch`` CPU in the same order and dtype
as the reference so that the result is bit-identical to the reference's CPU path on the same torch
build; the encoder uses numpy float64.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

TEMP_RANGE = 15.0        # range/range.py:103
TEMP_RANGE_PLUS = 12.0   # range/range.py:108
TEMP_GEO = 40.0          # range/range.py:109


# --------------------------------------------------------------------------------------------
# R1  spherical harmonics  (satclip/positional_encoding/spherical_harmonics.py:27-42)
# --------------------------------------------------------------------------------------------
def sh_angles(lonlat: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """phi = deg2rad(lon+180), theta = deg2rad(lat+90) (spherical_harmonics.py:28-32)."""
    lonlat = np.asarray(lonlat, dtype=np.float64)
    phi = (lonlat[:, 0] + 180.0) * (math.pi / 180.0)
    theta = (lonlat[:, 1] + 90.0) * (math.pi / 180.0)
    return phi, theta


def sh_features(lonlat: np.ndarray, L: int, mode: str = "analytic") -> np.ndarray:
    """Real spherical-harmonic features, (B, L*L) float64, feature index ``l*l + l + m``.

    Mathematics and conventions of the reference:

    * ``analytic`` = the generated ``spherical_harmonics_ylm.py``
      (spherical_harmonics_generate_ylms.py:19-35): m>0 -> sqrt(2) N_l^m P_l^m(cos t) cos(m phi),
      m<0 -> sqrt(2) N_l^|m| P_l^|m|(cos t) sin(|m| phi), both WITHOUT the Condon-Shortley sign
      (the generator's ``(-1)**m`` cancels sympy's), and m=0 -> sqrt((2l+1) * pi / 4) P_l(cos t)
      (operator-precedence quirk at generator line 29: pi times the orthonormal value).
    * ``closed-form`` = spherical_harmonics_closed_form.py:8-40: orthonormal m=0 (no pi) and the
      Condon-Shortley sign kept, i.e. analytic x (-1)^m for m != 0 and x 1/pi for m = 0.

    Evaluation is by the stable three-term recurrence on fully normalised associated Legendre
    functions, NOT by the reference's expanded polynomials: the latter lose all accuracy in
    float64 at high latitude (error 5e-5 at |lat|=45 deg, 0.1 at 70 deg for l=39; SURVEY.md section
    0 fact 4), so this oracle is the mathematically exact value and agrees with the reference only
    inside the well-conditioned band (|lat| <= 45 deg gives <=7e-6 on e-hat).
    """
    if mode not in ("analytic", "closed-form"):
        raise ValueError(f"unknown harmonics_calculation {mode!r}")
    phi, theta = sh_angles(lonlat)
    x = np.cos(theta)
    # |sin|: the reference forms it as sqrt((1 - x)(1 + x)) (closed_form.py:11) / (1 - x^2)^(m/2) (the
    # generated file) - identical for theta in [0, pi], and the reference's continuation beyond +-90 deg
    s = np.abs(np.sin(theta))
    B = phi.shape[0]
    Y = np.empty((B, L * L), dtype=np.float64)
    c_m = math.sqrt(1.0 / (4.0 * math.pi))
    for m in range(L):
        if m > 0:
            c_m *= math.sqrt((2.0 * m + 1.0) / (2.0 * m))
        q_mm = c_m * s ** m
        q_prev2 = None
        q_prev1 = q_mm
        if m == 0:
            scale = math.pi if mode == "analytic" else 1.0
            cm = sm = None
        else:
            scale = math.sqrt(2.0) * ((-1.0) ** m if mode == "closed-form" else 1.0)
            cm = np.cos(m * phi)
            sm = np.sin(m * phi)
        for l in range(m, L):
            if l == m:
                q = q_mm
            elif l == m + 1:
                q = math.sqrt(2.0 * m + 3.0) * x * q_prev1
            else:
                a = math.sqrt((4.0 * l * l - 1.0) / (l * l - m * m))
                b = math.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1.0) ** 2 - 1.0))
                q = a * (x * q_prev1 - b * q_prev2)
            if l > m:
                q_prev2, q_prev1 = q_prev1, q
            base = l * l + l
            if m == 0:
                Y[:, base] = scale * q
            else:
                Y[:, base + m] = scale * q * cm
                Y[:, base - m] = scale * q * sm
    return Y


@dataclass
class YlmTable:
    """The numbers of the reference's generated ``spherical_harmonics_ylm.py`` (one function per
    (l, m); generator: spherical_harmonics_generate_ylms.py:19-40), parsed into arrays indexed
    ``l*L + m`` for 0 <= m <= l < L:

        Y_l^m = front * (a0 + a2 x^2)^(p2/2) * (sum_j coef[off+j] * x^pow[off+j]) * x^kx * cos|sin(m phi),
        x = cos(theta)

    Loaded from the committed fixture ``tests/golden/ylm_table_L40.npz`` (data the reference's
    generator printed; tests/golden/make_golden_shtable.py)."""
    L: int
    front: np.ndarray
    a0: np.ndarray
    a2: np.ndarray
    p2: np.ndarray
    kx: np.ndarray
    off: np.ndarray
    cnt: np.ndarray
    coef: np.ndarray
    pow: np.ndarray


#: the committed fixture with the numbers of the reference's generated functions up to L = 40
YLM_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           "tests", "golden", "ylm_table_L40.npz")


def load_ylm_table(path: str = YLM_FIXTURE) -> YlmTable:
    z = np.load(path, allow_pickle=False)
    return YlmTable(int(z["L"]), *[np.asarray(z[k]) for k in
                                   ("front", "a0", "a2", "p2", "kx", "off", "cnt", "coef", "pow")])


def sh_features_faithful(lonlat: np.ndarray, table: YlmTable, L: Optional[int] = None) -> np.ndarray:
    """The 'analytic' features THE WAY THE REFERENCE EVALUATES THEM
    (spherical_harmonics.py:27-42 over the generated spherical_harmonics_ylm.py): one function per
    (l, m), m = -l..l - L*L calls, each a fully expanded float64 polynomial in cos(theta) whose
    powers are formed by ``torch.pow`` inside that function, every product and every sum an op of
    its own on the (B,) batch - then ``torch.stack``.  Nothing is shared between functions: the
    polynomial of (l, |m|) is evaluated again for -m, cos(theta) again in every function.  This is
    both the reference's rounding (ill-conditioned towards the poles: SURVEY.md section 0 fact 4)
    and the reference's COST (70 % of its CPU time) - ``bench.py``'s ``reference-shaped`` CPU baseline."""
    L = table.L if L is None else int(L)
    if L > table.L:
        raise ValueError(f"table holds L={table.L}, asked for {L}")
    TL = table.L
    lonlat_t = torch.from_numpy(np.ascontiguousarray(lonlat, dtype=np.float64))
    phi = torch.deg2rad(lonlat_t[:, 0] + 180)                   # spherical_harmonics.py:31
    theta = torch.deg2rad(lonlat_t[:, 1] + 90)                  # :32
    Y: List[torch.Tensor] = []
    for l in range(L):
        for m in range(-l, l + 1):                              # :35-36
            i = l * TL + abs(m)
            front, p2, kx, off, cnt = float(table.front[i]), int(table.p2[i]), int(table.kx[i]), int(table.off[i]), int(table.cnt[i])
            if p2 == 0 and cnt == 0 and kx == 0 and m == 0:
                Y.append(front * torch.ones_like(phi))          # :38-39 (Y00 is a Python float)
                continue
            ct = torch.cos(theta)
            v: object = front
            if p2:
                v = v * (float(table.a0[i]) + float(table.a2[i]) * ct ** 2) ** (p2 / 2.0)
            if cnt:
                ssum = None
                for j in range(off, off + cnt):
                    pj = int(table.pow[j])
                    t = float(table.coef[j]) * (ct ** pj) if pj else torch.full_like(ct, float(table.coef[j]))
                    ssum = t if ssum is None else ssum + t
                v = v * ssum
            # (sympy prints the factors of a product in its canonical order: cos(m*phi) / sin(m*phi)
            # in front of a bare cos(theta)**kx factor - which only functions without a polynomial have)
            if m > 0:
                v = v * torch.cos(m * phi)
            elif m < 0:
                v = v * torch.sin(-m * phi)
            if kx:
                v = v * (ct ** kx if kx > 1 else ct)
            Y.append(v if torch.is_tensor(v) else v * torch.ones_like(phi))
    return torch.stack(Y, dim=-1).numpy()                       # :42


# --------------------------------------------------------------------------------------------
# R2  SirenNet  (satclip/location_encoder.py:98-112, 114-119, 146-151)
# --------------------------------------------------------------------------------------------
def siren_forward(y: np.ndarray, weights: Dict[str, np.ndarray], w0_initial: float = 30.0,
                  w0: float = 1.0) -> np.ndarray:
    """h_i = sin(w0_i * (W_i h_{i-1} + b_i)) with w0=30 on the first layer only
    (location_encoder.py:83, 119, 147-150), last layer linear + Identity (:95-96, 112).
    float64 throughout (model_old.py:326-330, range.py:83-84); dropout is identity in eval."""
    h = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float64))
    n_hidden = len([k for k in weights if k.startswith("layers.") and k.endswith(".weight")])
    for i in range(n_hidden):
        W = torch.from_numpy(np.ascontiguousarray(weights[f"layers.{i}.weight"], dtype=np.float64))
        b = torch.from_numpy(np.ascontiguousarray(weights[f"layers.{i}.bias"], dtype=np.float64))
        h = torch.sin((w0_initial if i == 0 else w0) * torch.nn.functional.linear(h, W, b))
    W = torch.from_numpy(np.ascontiguousarray(weights["last_layer.weight"], dtype=np.float64))
    b = torch.from_numpy(np.ascontiguousarray(weights["last_layer.bias"], dtype=np.float64))
    return torch.nn.functional.linear(h, W, b).numpy()


def encode(lonlat: np.ndarray, weights: Dict[str, np.ndarray], L: int,
           mode: str = "analytic", features: Optional[np.ndarray] = None) -> np.ndarray:
    """Location encoder: SH features -> SirenNet (location_encoder.py:273-275), then the L2
    normalisation of range/range.py:212.  Returns e-hat (B, 256) float64.  ``features``: SH
    features computed elsewhere (e.g. the reference's expanded polynomials evaluated from their
    coefficient table) instead of the exact basis of ``sh_features``."""
    e = siren_forward(sh_features(lonlat, L, mode) if features is None else features, weights)
    t = torch.from_numpy(e)
    return (t / t.norm(p=2, dim=-1, keepdim=True)).numpy()


# --------------------------------------------------------------------------------------------
# L0  rad_to_cart  (range/utils/utils.py:11-16)
# --------------------------------------------------------------------------------------------
def rad_to_cart(locations: np.ndarray) -> np.ndarray:
    """Column 0 = lon, column 1 = lat, radians; dtype follows the input (utils.py:11-16)."""
    x = np.cos(locations[:, 1]) * np.cos(locations[:, 0])
    y = np.cos(locations[:, 1]) * np.sin(locations[:, 0])
    z = np.sin(locations[:, 1])
    return np.stack([x, y, z], axis=1)


def coord_features(lonlat: np.ndarray, model_name: str) -> np.ndarray:
    """The training-free encoders of range/range.py:262-272 on (B,2) float64 (lon,lat) degrees.

    'Direct' (:262-264, identity module :63-67): ``coords * math.pi/180`` = (x*pi)/180.
    'Cartesian_3D' (:265-268): rad_to_cart of that, in numpy.
    'Wrap' (positional_encoding/wrap.py:20-29): torch.deg2rad(x) = x * (pi/180 as one constant),
    then (cos lon, sin lon, cos lat, sin lat)."""
    x = np.asarray(lonlat, dtype=np.float64)
    if model_name == "Wrap":
        r = x * 0.017453292519943295
        return np.stack([np.cos(r[:, 0]), np.sin(r[:, 0]), np.cos(r[:, 1]), np.sin(r[:, 1])], axis=1)
    rad = x * math.pi / 180
    if model_name == "Direct":
        return rad
    if model_name == "Cartesian_3D":
        return rad_to_cart(rad)
    raise NotImplementedError(model_name)


# --------------------------------------------------------------------------------------------
# R4  bank preparation  (range/range.py:78-100)
# --------------------------------------------------------------------------------------------
@dataclass
class Bank:
    keys: np.ndarray      # (N,256) f32, rows L2-normalised in numpy f32 (range.py:85, 89)
    values: np.ndarray    # (N,1024) f32, NOT normalised (range.py:90)
    xyz: np.ndarray       # (N,3) f32, from f32-rounded locs (range.py:79, 93-95)

    @property
    def n_rows(self) -> int:
        return int(self.keys.shape[0])


def prep_bank(locs: np.ndarray, image_embeddings: np.ndarray,
              satclip_embeddings: np.ndarray) -> Bank:
    """range/range.py:78-95, op for op: locs are cast to float32 BEFORE the trigonometry
    (:79, 93-95), keys are normalised in numpy float32 (:85, 89), values only cast (:90)."""
    db_locs_latlon = np.asarray(locs).astype(np.float32)
    keys = np.asarray(satclip_embeddings).astype(np.float32)
    keys = keys / np.linalg.norm(keys, ord=2, axis=1, keepdims=True)
    values = np.asarray(image_embeddings).astype(np.float32)
    db_locs = db_locs_latlon * math.pi / 180
    xyz = rad_to_cart(db_locs)
    return Bank(keys=np.ascontiguousarray(keys), values=np.ascontiguousarray(values),
                xyz=np.ascontiguousarray(xyz))


def load_bank(path: str) -> Bank:
    with np.load(path, allow_pickle=False) as z:
        return prep_bank(z["locs"], z["image_embeddings"], z["satclip_embeddings"])


# --------------------------------------------------------------------------------------------
# R5-R8  retrieval, blend, pack  (range/range.py:213-240)
# --------------------------------------------------------------------------------------------
def query_xyz(lonlat: np.ndarray) -> np.ndarray:
    """range.py:225-231: float64 degrees -> radians -> rad_to_cart in float64 -> ``.float()``."""
    q = np.asarray(lonlat, dtype=np.float64) * math.pi / 180
    return torch.tensor(rad_to_cart(q)).float().numpy()


def retrieve(e_hat64: np.ndarray, lonlat: np.ndarray, bank: Bank, model_name: str = "RANGE+",
             beta: Optional[float] = 0.5) -> np.ndarray:
    """range/range.py:213-240 on CPU, same op order and dtypes.  Returns (B,1280) float64:
    columns 0:1024 = blended (RANGE+) or semantic (RANGE) retrieval, 1024:1280 = e-hat."""
    e = torch.from_numpy(np.ascontiguousarray(e_hat64, dtype=np.float64))
    K = torch.from_numpy(bank.keys)
    V = torch.from_numpy(bank.values)
    if model_name == "RANGE":
        temp = TEMP_RANGE
    elif model_name == "RANGE+":
        temp = TEMP_RANGE_PLUS
    else:
        raise ValueError("Unimplemented RANGE model")
    sim = e.float() @ K.t()                                            # :213
    sim = torch.nn.functional.softmax(sim * temp, dim=-1)              # :215
    high = sim @ V                                                     # :217
    if model_name == "RANGE":
        return np.concatenate((high, e), axis=1)                       # :222
    xq = torch.from_numpy(query_xyz(lonlat))                           # :225-229
    ang = xq @ torch.from_numpy(bank.xyz).T                            # :231
    ang = torch.nn.functional.softmax(ang * TEMP_GEO, dim=-1)          # :234
    ang_high = ang @ V                                                 # :236
    avg = (1 - beta) * ang_high + beta * high                          # :238
    return np.concatenate((avg, e), axis=1)                            # :240


def forward(lonlat: np.ndarray, weights: Dict[str, np.ndarray], L: int, bank: Bank,
            model_name: str = "RANGE+", beta: Optional[float] = 0.5,
            mode: str = "analytic", chunk: int = 2000) -> np.ndarray:
    """Whole path load_model(...)(locs) -> (B,1280) float64 (range.py:206-240), chunked over
    queries only to bound the (chunk, N) float32 intermediates."""
    lonlat = np.asarray(lonlat, dtype=np.float64)
    outs = []
    for i in range(0, lonlat.shape[0], chunk):
        ll = lonlat[i:i + chunk]
        outs.append(retrieve(encode(ll, weights, L, mode), ll, bank, model_name, beta))
    return np.concatenate(outs, axis=0) if outs else np.zeros((0, 1280), dtype=np.float64)


# --------------------------------------------------------------------------------------------
# Float64 helpers used by the parity tests (independent of the float32 op order above)
# --------------------------------------------------------------------------------------------
def logits64(e_hat64: np.ndarray, lonlat: np.ndarray, bank: Bank) -> Tuple[np.ndarray, np.ndarray]:
    """Semantic and geographic similarities in float64 from the float32 operands the reference
    feeds its matmuls (e-hat rounded to f32, :213; query xyz rounded to f32, :231)."""
    e32 = np.asarray(e_hat64, dtype=np.float64).astype(np.float32).astype(np.float64)
    s = e32 @ bank.keys.astype(np.float64).T
    g = query_xyz(lonlat).astype(np.float64) @ bank.xyz.astype(np.float64).T
    return s, g


def topk64(sim64: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Top-k of each row, descending, ties broken by lower index (stable)."""
    idx = np.argsort(-sim64, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(sim64, idx, axis=1), idx


def shard_stats64(sim64: np.ndarray, temp: float) -> Tuple[np.ndarray, np.ndarray]:
    """(max, sum exp(temp*s - max)) of the scaled logits over the given rows (for merge tests)."""
    t = sim64 * temp
    m = t.max(axis=1)
    return m, np.exp(t - m[:, None]).sum(axis=1)


def retrieve64(e_hat64: np.ndarray, lonlat: np.ndarray, bank: Bank, model_name: str = "RANGE+",
               beta: Optional[float] = 0.5) -> np.ndarray:
    """Exact-arithmetic (float64) version of ``retrieve`` from the same f32 operands: the value
    both the reference's f32 path and the HIP path approximate.  (B,1024) float64."""
    s, g = logits64(e_hat64, lonlat, bank)
    V = bank.values.astype(np.float64)
    temp = TEMP_RANGE if model_name == "RANGE" else TEMP_RANGE_PLUS

    def soft(z):
        z = z - z.max(axis=1, keepdims=True)
        p = np.exp(z)
        return p / p.sum(axis=1, keepdims=True)

    high = soft(s * temp) @ V
    if model_name == "RANGE":
        return high
    return (1 - beta) * (soft(g * TEMP_GEO) @ V) + beta * high

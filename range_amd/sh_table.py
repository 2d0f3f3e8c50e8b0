"""Coefficient tables of the reference's "analytic" spherical harmonics.

The reference evaluates every real spherical harmonic Y_l^m through a machine-generated TorchScript
function (``spherical_harmonics_ylm.py``, printed by
satclip/positional_encoding/spherical_harmonics_generate_ylms.py:19-40 with sympy): a fully
expanded polynomial in cos(theta) whose coefficients - up to 1e14 for l = 39 - are printed with 15
significant digits,

    m = 0 :  c_0 cos^l - c_1 cos^(l-2) + ...
    m != 0:  front * (1.0 - cos^2)^(|m|/2) * (c_0 cos^(l-|m|) - c_1 cos^(l-|m|-2) + ...) * cos|sin(|m| phi)

In float64 these sums cancel catastrophically towards the poles, and WHAT they cancel to is decided
by the 15-digit coefficients, not by rounding noise: evaluated one query per call instead of in a
batch the reference moves by 3e-5 at |lat| 60-75 deg, while it is 3e-3 away from the exact value.
A model trained on those features has learnt them, so the drop-in evaluates the very same sums:
this module produces the coefficient tables the HIP encoder walks in the reference's order
(``sh_eval='reference'``, the default for analytic checkpoints), either

* ``parse_ylm_source``  from the generated file of the user's own reference installation, or
* ``generate_table``    from scratch, reproducing the generator's arithmetic: exact rational
  Legendre-derivative coefficients, every printed constant rounded the way sympy's ``evalf()`` +
  ``str()`` do (float64, then 15 significant decimal digits).  tests/test_sh_table_cpu.py checks it
  against a table parsed from the file the reference's generator printed in the development
  container, coefficient by coefficient.

Table layout (numpy arrays, index ``l * L + m`` for 0 <= m <= l < L; orders -m share the table of
+m, only the azimuthal factor differs):  Y = front * (a0 + a2 x^2)^(p2/2) * x^kx * sum_j coef_j x^pow_j
with x = cos(theta); ``off``/``cnt`` locate the sum's terms, kept in printed order.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass
from decimal import ROUND_HALF_UP, Decimal, getcontext
from fractions import Fraction
from typing import Dict, List, Tuple

import numpy as np

_PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803")


@dataclass
class SHTable:
    L: int
    front: np.ndarray     # (L*L,) f64
    a0: np.ndarray        # (L*L,) f64   first factor (a0 + a2 x^2) ...
    a2: np.ndarray        # (L*L,) f64
    p2: np.ndarray        # (L*L,) i32   ... to the power p2 / 2  (0: factor absent)
    kx: np.ndarray        # (L*L,) i32   bare power of cos(theta) outside the sum
    off: np.ndarray       # (L*L,) i32
    cnt: np.ndarray       # (L*L,) i32   0: the sum is absent (= 1)
    coef: np.ndarray      # (n_terms,) f64, signed, printed order
    pow: np.ndarray       # (n_terms,) i32

    def evaluate(self, lonlat: np.ndarray) -> np.ndarray:
        """float64 numpy evaluation in the reference's operation order (products and sums as
        separate roundings, terms left to right); (B, L*L), feature index l*l + l + m.  Checker
        for the tests; the product path is the HIP kernel."""
        lonlat = np.asarray(lonlat, dtype=np.float64)
        phi = (lonlat[:, 0] + 180.0) * (math.pi / 180.0)
        x = np.cos((lonlat[:, 1] + 90.0) * (math.pi / 180.0))
        L = self.L
        xp = [np.ones_like(x)]
        for _ in range(1, L + 1):
            xp.append(xp[-1] * x)          # (the kernel uses correctly rounded powers; close enough here)
        Y = np.empty((x.shape[0], L * L))
        for l in range(L):
            for m in range(l + 1):
                i = l * L + m
                v = np.full_like(x, self.front[i])
                if self.p2[i]:
                    v = v * (self.a0[i] + self.a2[i] * (x * x)) ** (self.p2[i] / 2.0)
                if self.cnt[i]:
                    s = None
                    for j in range(self.off[i], self.off[i] + self.cnt[i]):
                        t = self.coef[j] * np.power(x, int(self.pow[j]))
                        s = t if s is None else s + t
                    v = v * s
                if self.kx[i]:
                    v = v * np.power(x, int(self.kx[i]))
                if m == 0:
                    Y[:, l * l + l] = v
                else:
                    Y[:, l * l + l + m] = v * np.cos(m * phi)
                    Y[:, l * l + l - m] = v * np.sin(m * phi)
        return Y


def _empty(L: int):
    n = L * L
    return dict(front=np.ones(n), a0=np.zeros(n), a2=np.zeros(n), p2=np.zeros(n, np.int32),
                kx=np.zeros(n, np.int32), off=np.zeros(n, np.int32), cnt=np.zeros(n, np.int32))


# ----------------------------------------------------------------------------------------------
# parser of the generated file
# ----------------------------------------------------------------------------------------------
_NUM = r"[0-9]+\.?[0-9]*(?:e[+-]?[0-9]+)?"
_TERM = re.compile(rf"\s*([+-])?\s*(?:({_NUM})\*?)?(cos\(theta\)(?:\*\*([0-9]+))?)?\s*")


def _parse_poly(s: str) -> List[Tuple[float, int]]:
    """'c0*cos(theta)**k0 - c1*cos(theta)**k1 + c2' -> [(c0,k0), (-c1,k1), (c2,0)] in order."""
    terms, pos = [], 0
    s = s.strip()
    while pos < len(s):
        mt = _TERM.match(s, pos)
        if not mt or mt.end() == pos:
            raise ValueError(f"cannot parse polynomial at {s[pos:pos + 40]!r}")
        sign, num, cosf, power = mt.groups()
        if num is None and cosf is None:
            raise ValueError(f"empty term in {s!r}")
        c = float(num) if num is not None else 1.0
        k = 0 if cosf is None else (int(power) if power else 1)
        terms.append((-c if sign == "-" else c, k))
        pos = mt.end()
    return terms


def _split_factors(expr: str) -> List[str]:
    out, depth, cur = [], 0, ""
    i = 0
    while i < len(expr):
        ch = expr[i]
        if ch == "(":
            depth += 1
        elif ch == ")":
            depth -= 1
        if ch == "*" and depth == 0 and expr[i:i + 2] != "**" and (i == 0 or expr[i - 1] != "*"):
            out.append(cur)
            cur = ""
        else:
            cur += ch
        i += 1
    out.append(cur)
    return [f.strip() for f in out]


def parse_ylm_source(text: str, L: int) -> SHTable:
    """Table from the text of a generated ``spherical_harmonics_ylm.py`` (degrees l < L)."""
    funcs: Dict[Tuple[int, int], str] = {}
    for mt in re.finditer(r"def Yl(\d+)_m(_minus_)?(\d+)\(theta, phi\):\s*\n\s*return (.+)", text):
        l, neg, m, expr = int(mt.group(1)), mt.group(2), int(mt.group(3)), mt.group(4).strip()
        funcs[(l, -m if neg else m)] = expr
    t = _empty(L)
    coef: List[float] = []
    powr: List[int] = []

    def top_level_sum(expr: str) -> bool:
        depth = 0
        for i, ch in enumerate(expr):
            if ch == "(":
                depth += 1
            elif ch == ")":
                depth -= 1
            elif depth == 0 and ch in "+-" and i > 0 and expr[i - 1] == " ":
                return True
        return False

    def parse_one(l: int, m: int, expr: str):
        d = dict(front=1.0, a0=0.0, a2=0.0, p2=0, kx=0, terms=[])
        if top_level_sum(expr) or ("phi" not in expr and "(1.0" not in expr and "*(" not in expr):
            # a bare sum (or one term) in cos(theta): the m = 0 functions
            d["terms"] = _parse_poly(expr)
            if len(d["terms"]) == 1:                      # constant or monomial
                d["front"], d["kx"] = d["terms"][0]
                d["terms"] = []
            return d, None
        trig = None
        for f in _split_factors(expr):
            mm = re.fullmatch(r"(cos|sin)\((?:(\d+)\*)?phi\)", f)
            if re.fullmatch(_NUM, f):
                d["front"] *= float(f)
            elif mm:
                trig = (mm.group(1), int(mm.group(2) or 1))
            elif re.fullmatch(r"cos\(theta\)(\*\*\d+)?", f):
                d["kx"] += int(re.fullmatch(r"cos\(theta\)(?:\*\*(\d+))?", f).group(1) or 1)
            elif f.startswith("("):
                mm = re.fullmatch(r"\((.+)\)(?:\*\*([0-9.]+))?", f)
                if not mm:
                    raise ValueError(f"Yl{l}_m{m}: factor {f!r}")
                terms, p = _parse_poly(mm.group(1)), float(mm.group(2) or 1.0)
                if sorted(k for _, k in terms) == [0, 2] and not d["p2"]:
                    cd = dict((k, c) for c, k in terms)   # the (a0 + a2 cos^2)^p factor
                    d["a0"], d["a2"], d["p2"] = cd[0], cd[2], int(round(2 * p))
                elif p == 1.0 and not d["terms"]:
                    d["terms"] = terms
                else:
                    raise ValueError(f"Yl{l}_m{m}: unsupported structure {expr!r}")
            else:
                raise ValueError(f"Yl{l}_m{m}: unknown factor {f!r} in {expr!r}")
        return d, trig

    for l in range(L):
        for m in range(l + 1):
            d, trig = parse_one(l, m, funcs[(l, m)])
            if m > 0:
                assert trig == ("cos", m), (l, m, trig)
                dn, trign = parse_one(l, -m, funcs[(l, -m)])
                assert trign == ("sin", m) and dn == d, (l, m, "the -m function differs from +m")
            i = l * L + m
            for k in ("front", "a0", "a2", "p2", "kx"):
                t[k][i] = d[k]
            t["off"][i], t["cnt"][i] = len(coef), len(d["terms"])
            coef += [c for c, _ in d["terms"]]
            powr += [k for _, k in d["terms"]]
    return SHTable(L=L, coef=np.asarray(coef, np.float64), pow=np.asarray(powr, np.int32), **t)


# ----------------------------------------------------------------------------------------------
# generator (no sympy): the arithmetic of spherical_harmonics_generate_ylms.py:19-40 + evalf/str
# ----------------------------------------------------------------------------------------------
def _dec15(x: float) -> float:
    """What str(sympy.Float) hands to TorchScript: the float's exact value rounded to 15
    significant decimal digits, ties away from zero (mpmath's to_str), parsed again."""
    if x == 0.0 or not math.isfinite(x):
        return x
    d = Decimal(x)
    return float(d.quantize(Decimal(1).scaleb(d.adjusted() - 14), rounding=ROUND_HALF_UP))


def _legendre_coeffs(l: int) -> List[Fraction]:
    """P_l(x) = sum_k c[k] x^k, exact."""
    p0, p1 = [Fraction(1)], [Fraction(0), Fraction(1)]
    if l == 0:
        return p0
    for n in range(1, l):
        nxt = [Fraction(0)] * (n + 2)
        for k, c in enumerate(p1):
            nxt[k + 1] += Fraction(2 * n + 1, n + 1) * c
        for k, c in enumerate(p0):
            nxt[k] -= Fraction(n, n + 1) * c
        p0, p1 = p1, nxt
    return p1


def _sqrt_to_float(q: Fraction, times_pi: int = 0) -> float:
    """Correctly rounded float64 of sqrt(q * pi^times_pi)."""
    getcontext().prec = 80
    v = Decimal(q.numerator) / Decimal(q.denominator)
    if times_pi > 0:
        v = v * _PI ** times_pi
    elif times_pi < 0:
        v = v / _PI ** (-times_pi)
    return float(v.sqrt())


def generate_table(L: int) -> SHTable:
    t = _empty(L)
    coef: List[float] = []
    powr: List[int] = []
    for l in range(L):
        P = _legendre_coeffs(l)
        for m in range(l + 1):
            i = l * L + m
            # d^m/dx^m P_l : exact rational coefficients, descending powers (as sympy prints them)
            D = list(P)
            for _ in range(m):
                D = [k * c for k, c in enumerate(D)][1:]
            terms = [(c, k) for k, c in reversed(list(enumerate(D))) if c != 0]
            if m == 0:
                # sqrt((2l+1)/4*pi) * P_l(cos theta)   (generator line 29: pi in the numerator);
                # evalf turns the constant and every rational into a float64, the product of the
                # two is distributed over the sum in float64, str() prints 15 digits
                # (the generator's (2*l+1)/4 is a Python float, so sympy holds sqrt(2.75*pi) as
                # Float(sqrt(2.75)) * sqrt(pi) and evaluates sqrt(pi) from the float64 pi: three
                # float64 roundings - this reproduces all 418 printed coefficients of L = 40)
                front = math.sqrt((2 * l + 1) / 4) * math.sqrt(math.pi)
                vals = [(_dec15(front * float(c)), k) for c, k in terms]
                if len(vals) == 1:
                    t["front"][i], t["kx"][i] = vals[0]
                else:
                    t["off"][i], t["cnt"][i] = len(coef), len(vals)
                    coef += [c for c, _ in vals]
                    powr += [k for _, k in vals]
                continue
            # (-1)^m sqrt(2) sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) * P_l^m, P_l^m = (-1)^m (1-x^2)^(m/2) D
            n2 = Fraction(2 * (2 * l + 1), 4) * Fraction(math.factorial(l - m), math.factorial(l + m))
            norm = _sqrt_to_float(n2, -1)
            t["a0"][i], t["a2"][i], t["p2"][i] = 1.0, -1.0, m
            if len(terms) == 1:                           # monomial: its coefficient joins the front
                c, k = terms[0]
                t["front"][i] = _dec15(norm * float(c))
                t["kx"][i] = k
            else:
                t["front"][i] = _dec15(norm)
                t["off"][i], t["cnt"][i] = len(coef), len(terms)
                coef += [_dec15(float(c)) for c, _ in terms]
                powr += [k for _, k in terms]
    return SHTable(L=L, coef=np.asarray(coef, np.float64), pow=np.asarray(powr, np.int32), **t)

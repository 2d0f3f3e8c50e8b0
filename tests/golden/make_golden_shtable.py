#!/usr/bin/env python3
"""Fixture for the reference-faithful spherical harmonics (development container only).

Runs the REFERENCE's generator (spherical_harmonics_generate_ylms.py, exec'd with L = 40 as in
make_golden.py; needs sympy) and the reference's SphericalHarmonics module on a pole-to-pole set of
queries, and stores

  * the coefficient table PARSED from the generated text (range_amd.sh_table.parse_ylm_source):
    numbers the reference printed, not its source - tests/test_sh_table_cpu.py checks the
    from-scratch generator (range_amd.sh_table.generate_table) against it coefficient by coefficient;
  * the reference's own (B, 1600) float64 SH features of 24 queries, batch evaluation
    (spherical_harmonics.py:27-42), the values the table evaluation has to reproduce.
"""
from __future__ import annotations

import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from range_amd import sh_table  # noqa: E402


def main():
    tmp = os.path.join(tempfile.gettempdir(), "range_golden_tmp")
    os.makedirs(tmp, exist_ok=True)
    path = mg.regenerate_ylm(tmp, 40)
    t = sh_table.parse_ylm_source(open(path).read(), 40)
    le = mg.import_level1(tmp)
    posenc = le.get_positional_encoding("sphericalharmonics", legendre_polys=40,
                                        harmonics_calculation="analytic").double()
    rng = np.random.default_rng(5)
    q = np.stack([rng.uniform(-180, 180, 24), np.linspace(-89.5, 89.5, 24)], axis=1)
    with torch.no_grad():
        feats = posenc(torch.from_numpy(q)).numpy()
    np.savez_compressed(os.path.join(HERE, "ylm_table_L40.npz"), L=40, front=t.front, a0=t.a0, a2=t.a2,
                        p2=t.p2, kx=t.kx, off=t.off, cnt=t.cnt, coef=t.coef, pow=t.pow,
                        lonlat=q, sh_features=feats)
    print("wrote ylm_table_L40.npz", t.coef.shape, feats.shape)


if __name__ == "__main__":
    main()

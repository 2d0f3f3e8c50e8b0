"""Pin the CPU oracle (oracle/range_oracle.py) against golden vectors produced by running the
reference's own Python (tests/golden/make_golden.py).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import range_oracle as O
from tools import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ENC = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "enc_*.npz")))
E2E = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "e2e_*.npz")))


def test_manifest_lists_every_fixture():
    man = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    assert sorted(man["cases"]) == sorted(ENC + E2E)
    # error behaviour of the reference loader (range/load_model.py:31-34, range.py:113-114,199-200)
    assert man["errors"] == {"no_pretrained_path": "ValueError", "no_db_path": "AssertionError",
                             "bad_range_name": "ValueError", "unknown_model": "NotImplementedError"}


@pytest.mark.parametrize("tag", ENC)
def test_sh_and_encoder_vs_reference(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    q, L, mode = z["lonlat"], int(z["L"]), str(z["mode"])
    Y = O.sh_features(q, L, mode)
    rows = z["sh_rows"]
    d = np.abs(Y[rows] - z["sh_features"]).max(axis=1)
    lat = np.abs(q[rows, 1])
    w = synth.make_encoder_weights(L, int(z["hidden"]), 256, int(z["num_hidden_layers"]),
                                   int(z["seed"]))
    e = O.siren_forward(Y, w)
    de = np.abs(e - z["embedding"]).max(axis=1)
    latq = np.abs(q[:, 1])
    if mode == "closed-form" or L <= 16:
        # the reference's recurrence (and its low-degree polynomials) are well conditioned
        # everywhere: the restatement must agree to rounding at every latitude
        tol = 5e-12 if mode == "closed-form" else 5e-9   # analytic L=16 is ~1e-10 at 75 deg
        assert d.max() < tol
        assert de.max() < tol
    else:
        # analytic L=40: the reference's expanded polynomials are ill-conditioned in float64
        # away from the equator; gate on the |lat|<=45 band
        assert d[lat <= 30].max() < 1e-7
        assert d[lat <= 45].max() < 2e-4
        assert de[latq <= 30].max() < 1e-8
        assert de[latq <= 45].max() < 2e-5
        # and the analytic convention itself is exact: compare against closed-form x factor
        Yc = O.sh_features(q, L, "closed-form")
        for l in range(L):
            for m in range(-l, l + 1):
                f = np.pi if m == 0 else (-1.0) ** m
                np.testing.assert_allclose(Y[:, l * l + l + m], f * Yc[:, l * l + l + m],
                                           rtol=1e-13, atol=1e-15)


def test_sh_matches_scipy_orthonormal():
    """Independent check of the recurrence against scipy's associated Legendre functions."""
    from scipy.special import lpmv, gammaln
    q = synth.make_queries(64, seed=3, lat_max=89.9)
    L = 24
    Y = O.sh_features(q, L, "closed-form")
    phi, theta = O.sh_angles(q)
    x = np.cos(theta)
    for l in (0, 1, 2, 7, 15, 23):
        for m in range(-l, l + 1):
            am = abs(m)
            N = np.sqrt((2 * l + 1) / (4 * np.pi) * np.exp(gammaln(l - am + 1) - gammaln(l + am + 1)))
            P = lpmv(am, l, x)  # includes the Condon-Shortley phase, like the closed form
            if m == 0:
                ref = N * P
            elif m > 0:
                ref = np.sqrt(2) * N * np.cos(m * phi) * P
            else:
                ref = np.sqrt(2) * N * np.sin(am * phi) * P
            np.testing.assert_allclose(Y[:, l * l + l + m], ref, rtol=1e-10, atol=1e-12)


def _bank_and_weights(z):
    locs, vals, keys = synth.make_bank(int(z["bank_rows"]), int(z["bank_seed"]))
    bank = O.prep_bank(locs, vals, keys)
    w = synth.make_encoder_weights(int(z["L"]), int(z["hidden"]), 256,
                                   int(z["num_hidden_layers"]), int(z["weight_seed"]))
    return bank, w


@pytest.mark.parametrize("tag", E2E)
def test_retrieval_bitwise_vs_reference(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    bank, w = _bank_and_weights(z)
    q = z["lonlat"]
    e_ref = z["range"][:, 1024:]
    out = O.retrieve(e_ref, q, bank, "RANGE", None)
    assert out.dtype == np.float64 and out.shape == (q.shape[0], 1280)
    assert np.array_equal(out, z["range"])
    for beta in (0.0, 0.25, 0.5, 0.75, 1.0):
        assert np.array_equal(O.retrieve(e_ref, q, bank, "RANGE+", beta),
                              z[f"rangeplus_beta{beta}"]), beta


@pytest.mark.parametrize("tag", E2E)
def test_full_forward_vs_reference(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    bank, w = _bank_and_weights(z)
    q = z["lonlat"]
    for name, beta, key in (("RANGE", None, "range"), ("RANGE+", 0.5, "rangeplus_beta0.5"),
                            ("RANGE+", 0.0, "rangeplus_beta0.0")):
        out = O.forward(q, w, int(z["L"]), bank, name, beta)
        # |lat|<=45 queries: north_star tolerance 1e-4; measured ~5e-7
        np.testing.assert_allclose(out, z[key], rtol=0, atol=2e-5)


@pytest.mark.parametrize("tag", E2E)
def test_topk_vs_reference(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"))
    bank, _ = _bank_and_weights(z)
    s, _ = O.logits64(z["range"][:, 1024:], z["lonlat"], bank)
    tv, ti = O.topk64(s, 16)
    assert np.array_equal(ti, z["sem_topk_idx"])
    np.testing.assert_allclose(tv, z["sem_topk_val"], rtol=0, atol=1e-6)


def test_retrieve64_close_to_f32_path():
    locs, vals, keys = synth.make_bank(777, 5)
    bank = O.prep_bank(locs, vals, keys)
    w = synth.make_encoder_weights(10, 64, 256, 2, 5)
    q = synth.make_queries(19, seed=1)
    e = O.encode(q, w, 10)
    for name, beta in (("RANGE", None), ("RANGE+", 0.3)):
        a = O.retrieve(e, q, bank, name, beta)[:, :1024]
        b = O.retrieve64(e, q, bank, name, beta)
        np.testing.assert_allclose(a, b, rtol=0, atol=5e-6)


def test_bank_prep_dtypes_and_norms():
    locs, vals, keys = synth.make_bank(100, 1)
    bank = O.prep_bank(locs, vals, keys)
    assert bank.keys.dtype == bank.values.dtype == bank.xyz.dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(bank.keys, axis=1), 1.0, atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(bank.xyz, axis=1), 1.0, atol=1e-6)
    assert np.array_equal(bank.values, vals)


# Latitude bands of the analytic SatCLIP-L40 encoder.  The oracle (and the HIP encoder, which equals
# it to 2e-12) evaluates the spherical harmonics exactly; the reference evaluates expanded
# polynomials whose 15-digit coefficients reach 1e14 and loses accuracy towards the poles.  Per
# band: the largest |oracle - reference| allowed on the L2-normalised embedding (measured values
# in parentheses), next to the reference's OWN spread between two ways of calling it.
LAT_BANDS = ((0.0, 30.0, 5e-9), (30.0, 45.0, 5e-6), (45.0, 60.0, 1e-3), (60.0, 75.0, 1.5e-2), (75.0, 90.1, 5e-2))


def test_encoder_over_all_latitudes_vs_reference():
    z = np.load(os.path.join(GOLDEN, "latitude_L40_H512_n2.npz"))
    q = z["lonlat"]
    assert np.abs(q[:, 1]).max() > 89.0 and (np.abs(q[:, 1]) < 1.0).any()      # pole to pole
    w = synth.make_encoder_weights(int(z["L"]), int(z["hidden"]), 256, int(z["num_hidden_layers"]), int(z["seed"]))
    e = O.encode(q, w, int(z["L"]), str(z["mode"]))
    d = np.abs(e - z["embedding"]).max(axis=1)
    al = np.abs(q[:, 1])
    for lo, hi, tol in LAT_BANDS:
        m = (al >= lo) & (al < hi)
        assert m.sum() >= 15 and d[m].max() < tol, (lo, hi, d[m].max())
    # the north-star tolerance (1e-4) holds on |lat| <= 45 with a wide margin, and fails beyond 60:
    # that is the reference's conditioning, not the engine's (see the closed-form fixtures, 1e-8)
    assert d[al <= 45].max() < 1e-5 and d[al > 60].max() > 1e-4
    # the reference does not reproduce itself either: its spread grows the same way
    assert z["self_spread"][al > 60].max() > 1e-6 > z["self_spread"][al <= 30].max()


# ---- the reference-SHAPED evaluation of the 'analytic' basis (oracle.sh_features_faithful): one float64
#      torch expression per (l, m) over the numbers the reference's generator printed
YLM = os.path.join(GOLDEN, "ylm_table_L40.npz")


def test_faithful_sh_features_equal_the_reference_bitwise():
    """Pole to pole, every one of the 1600 functions of L = 40: the oracle's reference-shaped
    evaluation IS the reference's (spherical_harmonics.py:27-42 over the generated file), bit for
    bit - also where those polynomials have lost every digit to cancellation."""
    z = np.load(YLM)
    tab = O.load_ylm_table(YLM)
    Y = O.sh_features_faithful(z["lonlat"], tab)
    assert np.abs(z["lonlat"][:, 1]).max() > 89.0
    assert Y.dtype == np.float64 and np.array_equal(Y, z["sh_features"])
    # lower L: the functions of degree l < L do not depend on L
    for tag in ("enc_analytic_L10_H64_n2", "enc_analytic_L16_H128_n3", "enc_analytic_L40_H512_n2"):
        g = np.load(os.path.join(GOLDEN, tag + ".npz"))
        # (the whole batch, as the reference evaluated it: torch's vectorised pow rounds the last
        # elements of a batch - its scalar tail - differently from the ones in full SIMD groups, the
        # reference's own "batch vs single" spread of SURVEY.md section 8(c))
        Yg = O.sh_features_faithful(g["lonlat"], tab, int(g["L"]))[g["sh_rows"]]
        assert np.array_equal(Yg, g["sh_features"]), tag


@pytest.mark.parametrize("tag", ["enc_analytic_L10_H64_n2", "enc_analytic_L16_H128_n3", "enc_analytic_L40_H256_n2",
                                 "enc_analytic_L40_H512_n2", "latitude_L40_H512_n2"])
def test_faithful_encoder_matches_the_reference_at_every_latitude(tag):
    """With the reference's own features the restated SirenNet + normalisation reproduce the
    reference's embedding to float64 rounding at EVERY latitude (the exact basis of ``sh_features``
    agrees only inside |lat| <= 45: LAT_BANDS above)."""
    g = np.load(os.path.join(GOLDEN, tag + ".npz"))
    L, H = int(g["L"]), int(g["hidden"])
    w = synth.make_encoder_weights(L, H, 256, int(g["num_hidden_layers"]), int(g["seed"]))
    q = g["lonlat"]
    feats = O.sh_features_faithful(q, O.load_ylm_table(YLM), L)
    if tag.startswith("enc_"):        # (these fixtures hold the SirenNet output before the normalisation)
        e = O.siren_forward(feats, w)
        assert float(np.abs(e - g["embedding"]).max()) < 1e-12 * max(1.0, float(np.abs(g["embedding"]).max()))
    else:
        e = O.encode(q, w, L, features=feats)
        assert np.abs(q[:, 1]).max() > 89.0 and float(np.abs(e - g["embedding"]).max()) < 1e-14

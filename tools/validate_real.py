#!/usr/bin/env python3
"""One-command check of REAL artefacts against this engine (none exist offline: SURVEY.md fact 5):

    python tools/validate_real.py <satclip-*.ckpt> <range_db_*.npz | .rbank> [spherical_harmonics_ylm.py]

1. reads the checkpoint like satclip/load.py:3-18 does and prints what the engine needs to know:
   legendre_polys L, capacity H, hidden layers, embed_dim, harmonics_calculation - and refuses
   loudly (exit 2) on a shape the kernels do not cover, naming the limit and where it lives;
2. reads the bank (range/range.py:78-95 preparation), prints N and the key / value / location
   statistics the kernels rely on (unit keys after normalisation, finite values, unit xyz);
3. if a generated spherical_harmonics_ylm.py is given, parses ITS polynomials (they are what a model
   trained with the reference has seen) and reports how they compare with the regenerated table;
4. on a GPU box: loads the model through load_model(...) and runs 256 queries pole to pole through
   the HIP path against the CPU oracle (oracle/: test infrastructure) - e-hat, retrieval in float64
   and in the reference's float32 op order, top-16 indices - RANGE+ and RANGE.  Exit 0 only if all
   of it is inside the tolerances the tests use.
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np


def fail(msg, code=2):
    print(f"validate_real: UNSUPPORTED - {msg}", file=sys.stderr)
    sys.exit(code)


def main():
    if len(sys.argv) < 3:
        print(__doc__)
        sys.exit(1)
    ckpt, db = sys.argv[1], sys.argv[2]
    ylm = sys.argv[3] if len(sys.argv) > 3 else None
    import torch
    from range_amd.bankfile import load_any
    from range_amd.ckpt import read_checkpoint
    from range_amd import sh_table

    # ---- 1. checkpoint
    try:
        enc = read_checkpoint(ckpt)
    except (KeyError, NotImplementedError, ValueError) as ex:
        fail(f"checkpoint {ckpt}: {type(ex).__name__}: {ex}")
    L, H, NL, E = enc.legendre_polys, enc.hidden, enc.num_hidden_layers, enc.embed_dim
    print(f"checkpoint: legendre_polys L={L} ({L * L} SH features), capacity H={H}, hidden layers={NL}, "
          f"embed_dim={E}, harmonics_calculation={enc.harmonics_calculation!r}")
    for i, (w, b) in enumerate(zip(enc.weights, enc.biases)):
        print(f"  layer {i}: weight {tuple(w.shape)} {w.dtype}, |w| max {np.abs(w).max():.3e}, bias {tuple(b.shape)}")
    if not 1 <= L <= 64:
        fail(f"legendre_polys={L}: the encoder kernel covers 1..64 (range_amd/csrc/range_hip.hip: range_set_encoder)")
    if not 1 <= H <= 1024:
        fail(f"capacity H={H}: the encoder kernel covers hidden widths up to 1024 (the hidden activations of a 16-query "
             "workgroup live in LDS: 16 x H float64 <= 128 KB; range_amd/csrc/range_hip.hip: range_set_encoder)")
    Hk = (H + 63) // 64 * 64 if H <= 512 else (768 if H <= 768 else 1024)
    if Hk != H:
        print(f"  capacity H={H} runs zero-padded as the kernel width {Hk} (same result bit for bit, {Hk / H:.2f}x the first-layer work)")
    if E != 256:
        fail(f"embed_dim={E}: the bank keys are 256 wide (range/range.py:85-86) and so are the kernels")
    if not 1 <= NL <= 7:
        fail(f"num_hidden_layers={NL}: 1..7 supported (ENC_MAX_LAYERS)")

    # ---- 2. bank
    bank = load_any(db)
    N = bank.n_rows
    kn = np.linalg.norm(bank.keys.astype(np.float64), axis=1)
    xn = np.linalg.norm(bank.xyz.astype(np.float64), axis=1)
    print(f"bank: N={N} rows; keys {bank.keys.shape} unit to {np.abs(kn - 1).max():.1e}; values {bank.values.shape} "
          f"range [{bank.values.min():.3g}, {bank.values.max():.3g}]; xyz unit to {np.abs(xn - 1).max():.1e}")
    if not (np.isfinite(bank.keys).all() and np.isfinite(bank.values).all() and np.isfinite(bank.xyz).all()):
        fail("the bank holds non-finite numbers")
    if N >= 2 ** 31:
        fail("more than 2^31 bank rows")

    # ---- 3. the user's generated polynomials
    if ylm is not None:
        if enc.harmonics_calculation != "analytic":
            print("  (the checkpoint is 'closed-form': the generated polynomials are not used)")
        else:
            theirs = sh_table.parse_ylm_source(open(ylm).read(), L)
            ours = sh_table.generate_table(L)
            same_c = int(np.sum(theirs.coef == ours.coef)) if theirs.coef.shape == ours.coef.shape else -1
            same_f = int(np.sum(theirs.front == ours.front))
            print(f"generated polynomials: {theirs.coef.shape[0]} coefficients; bit-identical to the regenerated table: "
                  f"{same_c} coefficients, {same_f} of {ours.front.shape[0]} leading constants "
                  "(pass sh_source= to load_model to use the file's own)")

    # ---- 4. HIP vs oracle
    if not torch.cuda.is_available():
        print("no GPU visible: steps 1-3 only")
        return
    from oracle import range_oracle as O     # checker (test infrastructure)
    from range_amd import load_model
    from tools import synth
    q = synth.make_queries(256, seed=1, lat_max=90.0)
    x = torch.from_numpy(q).to("cuda:0")
    obank = O.Bank(bank.keys, bank.values, bank.xyz)
    sd = torch.load(ckpt, map_location="cpu", weights_only=False)["state_dict"]
    pre = "model.location.nnet."
    w = {k[len(pre):]: v.double().numpy() for k, v in sd.items() if k.startswith(pre)}
    worst = 0.0
    for name in ("RANGE+", "RANGE"):
        kw = dict(sh_source=ylm) if ylm and enc.harmonics_calculation == "analytic" else {}
        m = load_model(name, pretrained_path=ckpt, device="cuda:0", db_path=db, beta=0.5, **kw)
        out = m(x)
        assert isinstance(out, np.ndarray) and out.shape == (256, 1280) and out.dtype == np.float64
        e = out[:, 1024:]
        # e-hat against the oracle fed with the same SH polynomials (CPU evaluation of the table)
        from range_amd.range import sh_table_for
        tab = sh_table_for(enc, None, kw.get("sh_source"))
        if tab is None:
            e_ref = O.encode(q, w, L, enc.harmonics_calculation)
        elif ylm or L > 40:       # the user's own polynomials: their table, walked on the CPU
            e_ref = O.encode(q, w, L, features=tab.evaluate(q))
        else:                     # the reference-shaped evaluation of the oracle (bitwise the reference's features)
            e_ref = O.encode(q, w, L, features=O.sh_features_faithful(q, O.load_ylm_table(), L))
        band = np.abs(q[:, 1]) <= 45
        d_e, d_e_all = float(np.abs(e - e_ref)[band].max()), float(np.abs(e - e_ref).max())
        d64 = float(np.abs(out[:, :1024] - O.retrieve64(e, q, obank, name, 0.5)).max())
        d32 = float(np.abs(out - O.retrieve(e, q, obank, name, 0.5)).max())
        tv, ti = m.topk(x, 16)
        rv, ri = O.topk64(O.logits64(e, q, obank)[0], min(16, N))
        bad = int((ti.cpu().numpy()[:, :ri.shape[1]] != ri).any(axis=1).sum())
        print(f"{name}: e-hat vs oracle {d_e:.1e} (|lat|<=45; {d_e_all:.1e} pole to pole), retrieval vs float64 oracle {d64:.1e}, "
              f"vs the reference's float32 op order {d32:.1e}, top-16 rows differing in {bad} of 256 queries")
        worst = max(worst, d64 / 2e-5, d32 / 1e-4, d_e / 1e-6, bad / 3.0)
    if worst > 1.0:
        fail("HIP path and oracle disagree beyond the test tolerances (2e-5 / 1e-4 / 1e-6 / 2 near-tie queries)", code=3)
    print("validate_real: OK")


if __name__ == "__main__":
    main()

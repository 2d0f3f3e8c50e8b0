"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, loader /
checkpoint / bank readers mirror the reference's behaviour and errors.  No GPU compute."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import range_oracle as O
from range_amd import _native
from tools import synth
from range_amd.bank import load_bank, prepare_bank
from range_amd.ckpt import read_checkpoint
from range_amd.load_model import load_model

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "range_hip.h")).read()
    declared = set(re.findall(r"\b(range_[a-z0-9_]+)\s*\(", header))
    declared -= {"range_ctx", "range_stream_t"}
    assert declared == set(_native.SYMBOLS)
    lib = _native.load_library()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.range_abi_version() == 9
    assert lib.range_last_error() is not None
    # the ridge-probe header, same library
    from range_amd import _probe_native
    header = open(os.path.join(REPO, "include", "range_probe.h")).read()
    declared = set(re.findall(r"\b(range_probe_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_probe_native.SYMBOLS)
    lib = _probe_native.load_library()
    for name in declared:
        assert hasattr(lib, name), name


def test_loader_errors_match_reference(tmp_path):
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=10, hidden=64, seed=5)
    db = synth.write_bank(str(tmp_path / "db.npz"), 50, 1)
    with pytest.raises(ValueError):                      # load_model.py:31-32
        load_model("RANGE+", db_path=db)
    with pytest.raises(AssertionError):                  # load_model.py:34
        load_model("RANGE+", pretrained_path=ck)
    with pytest.raises(ValueError, match="Unimplemented RANGE model"):   # range.py:113-114
        load_model("RANGE++", pretrained_path=ck, device="cuda", db_path=db)
    with pytest.raises(NotImplementedError):             # range.py:199-200
        load_model("NoSuchModel", pretrained_path=ck)
    with pytest.raises(NotImplementedError):             # the 10 unrelated baseline encoders
        load_model("GeoCLIP", pretrained_path=ck)
    if not torch.cuda.is_available():
        # the product has no CPU path and must say so loudly
        with pytest.raises(RuntimeError):
            load_model("RANGE+", pretrained_path=ck, device="cpu", db_path=db)
        with pytest.raises(RuntimeError):
            load_model("RANGE+", pretrained_path=ck, device="cuda", db_path=db)


def test_checkpoint_reader(tmp_path):
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=16, hidden=128, num_hidden_layers=3,
                                seed=6, harmonics_calculation="closed-form")
    p = read_checkpoint(ck)
    assert (p.legendre_polys, p.hidden, p.num_hidden_layers, p.embed_dim) == (16, 128, 3, 256)
    assert p.harmonics_calculation == "closed-form"
    w = synth.make_encoder_weights(16, 128, 256, 3, 6)
    assert len(p.weights) == 4 and np.array_equal(p.weights[0], w["layers.0.weight"])
    assert np.array_equal(p.biases[3], w["last_layer.bias"]) and p.weights[3].dtype == np.float64
    # the three hyper-parameters the reference pops unconditionally (satclip/load.py:5-7)
    c = synth.make_checkpoint(L=10, hidden=64)
    del c["hyper_parameters"]["eval_downstream"]
    torch.save(c, str(tmp_path / "bad.ckpt"))
    with pytest.raises(KeyError):
        read_checkpoint(str(tmp_path / "bad.ckpt"))
    c = synth.make_checkpoint(L=10, hidden=64)
    c["hyper_parameters"]["le_type"] = "grid"
    torch.save(c, str(tmp_path / "grid.ckpt"))
    with pytest.raises(NotImplementedError):
        read_checkpoint(str(tmp_path / "grid.ckpt"))


def test_bank_prep_matches_oracle_bitwise(tmp_path):
    db = synth.write_bank(str(tmp_path / "db.npz"), 321, 4)
    b = load_bank(db)
    o = O.load_bank(db)
    assert np.array_equal(b.keys, o.keys) and np.array_equal(b.values, o.values)
    assert np.array_equal(b.xyz, o.xyz)
    assert b.keys.dtype == b.values.dtype == b.xyz.dtype == np.float32
    with pytest.raises(ValueError):
        prepare_bank(np.zeros((3, 2)), np.zeros((3, 1024)), np.zeros((3, 128)))
    sub = b.rows(10, 20)
    assert sub.n_rows == 10 and np.array_equal(sub.keys, b.keys[10:20])


def test_degenerate_bank_rows_are_refused_at_load_where_the_reference_turns_every_query_nan(tmp_path):
    """Round-6 decision, pinned: a zero-norm key row (0/0 in the reference's own normalisation,
    range.py:89), or a NaN / infinite embedding or location, makes the REFERENCE return NaN for every
    query of every batch (shown here on the oracle, which restates range.py:213-240 op for op); the
    product refuses the bank at load, naming the row.  Query coordinates are a different matter: a NaN
    / infinite coordinate gives a NaN row and leaves the other rows alone, on both sides
    (tests/test_gpu_round6.py)."""
    locs, vals, keys = synth.make_bank(64, 4)
    keys = keys.copy()
    keys[17] = 0.0
    with np.errstate(invalid="ignore", divide="ignore"):
        ob = O.prep_bank(locs, vals, keys)
    assert np.isnan(ob.keys[17]).all() and np.isfinite(np.delete(ob.keys, 17, axis=0)).all()
    q = synth.make_queries(5, seed=1)
    e = np.random.default_rng(0).standard_normal((5, 256))
    e /= np.linalg.norm(e, axis=1, keepdims=True)
    for name in ("RANGE", "RANGE+"):
        ref = O.retrieve(e, q, ob, name, 0.5)
        assert np.isnan(ref[:, :1024]).all()          # every query, every retrieved column
    with pytest.raises(ValueError, match=r"bank row 17 .*satclip_embeddings.*NaN for every query"):
        prepare_bank(locs, vals, keys)
    np.savez(str(tmp_path / "bad.npz"), locs=locs, image_embeddings=vals, satclip_embeddings=keys)
    with pytest.raises(ValueError, match="bank row 17"):
        load_bank(str(tmp_path / "bad.npz"))
    # non-finite values / locations: the same verdict, by section
    locs, vals, keys = synth.make_bank(64, 4)
    v2 = vals.copy()
    v2[3, 100] = np.inf
    with pytest.raises(ValueError, match=r"bank row 3 .*image_embeddings"):
        prepare_bank(locs, v2, keys)
    l2 = locs.copy()
    l2[40, 1] = np.nan
    with pytest.raises(ValueError, match=r"bank row 40 .*locs"):
        prepare_bank(l2, vals, keys)
    prepare_bank(locs, vals, keys)                    # (the clean bank loads)


def test_library_carries_this_checkouts_source_hash():
    """build.sh embeds the SHA-256 of range_amd/csrc/* and include/*.h in the library: the file carries
    it as a literal (read without loading), the loaded library returns the same, and both equal this
    checkout's - what __graft_entry__.build() and the binding's refusal of a stale library rest on."""
    import ctypes
    from range_amd._srchash import library_stamp, source_sha256
    here = source_sha256()
    assert len(here) == 64 and library_stamp(_native.LIB_PATH) == here
    lib = ctypes.CDLL(_native.LIB_PATH)
    lib.range_source_sha256.restype = ctypes.c_char_p
    assert lib.range_source_sha256().decode() == here
    assert library_stamp(__file__) is None                      # (a file without a stamp)


def test_engine_requires_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.RangeNativeError):
        _native.HipEngine("cuda:0")
    with pytest.raises(_native.RangeNativeError):
        _native.HipEngine("cpu")


def test_no_foreign_m0_writes(tmp_path):
    """The LDS-DMA groups in attend_kernels.h set M0 in one asm statement and rely on it in the
    next three (include comment at dma_group_begin).  That is only sound while hipcc itself never
    writes M0 in those kernels: check the generated gfx950 assembly."""
    import shutil
    import subprocess
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = tmp_path / "dev.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                    "-o", str(out), os.path.join(REPO, "range_amd", "csrc", "range_hip.hip")],
                   check=True, cwd=REPO, capture_output=True)
    text = out.read_text()
    kernels = re.split(r"\n(?=_ZN9range_hip\w+:)", text)
    checked = 0
    for k in kernels:
        name = k.split(":", 1)[0]
        if "attend_kernel" not in name and "scan_stats_kernel" not in name:
            continue
        checked += 1
        in_asm = False
        for line in k.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm and re.search(r"\bm0\b", line.split(";")[0]):
                raise AssertionError(f"{name}: compiler-generated M0 access: {line.strip()}")
    assert checked >= 6
    # second check on the same assembly: no instruction right behind an inline-asm MFMA writes one
    # of its source registers (WAR hazard measured on gfx950, see attend_kernels.h mfma_v)
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_mfma_war", os.path.join(REPO, "tools", "check_mfma_war.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = mod.check(str(out), ["attend_kernel", "scan_stats_kernel", "attend_bf16x3_kernel"])
    assert not bad, bad[:5]
    # third check: the stream top-k kernels load their first query operand by inline asm into
    # accumulator registers and wait for it by hand (topk_stream.h: TopkQLoad / topk_qwait) - hipcc
    # must not read, copy or overwrite those registers between a load and the wait that covers it,
    # the kernels must not use scratch memory, and everything workgroups hand each other goes through
    # global (not flat) agent-scope accesses
    n_stream = 0
    for k in kernels:
        name = k.split(":", 1)[0]
        if "topk_stream" not in name:
            continue
        n_stream += 1
        body = k.splitlines()
        loads = [i for i, l in enumerate(body)
                 if re.search(r"global_load_dwordx4 a\[\d+:\d+\], v\[\d+:\d+\], off offset:(0x[0-9a-f]+|\d+)\s*$", l)]
        waits = [i for i, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\((32|48)\)", l)]
        assert len(loads) in (16, 32) and len(waits) == len(loads) // 16, (name, len(loads), waits)
        for grp, w in enumerate(waits):
            regs = set()
            for i in loads[16 * grp:16 * grp + 16]:
                lo, hi = map(int, re.search(r"a\[(\d+):(\d+)\]", body[i]).groups())
                regs.update(range(lo, hi + 1))
            for line in body[loads[16 * grp]:w]:
                if "global_load_dwordx4 a[" in line:
                    continue
                code = line.split(";")[0]
                used = {int(x) for x in re.findall(r"\ba(\d+)\b", code)}
                for lo, hi in re.findall(r"a\[(\d+):(\d+)\]", code):
                    used.update(range(int(lo), int(hi) + 1))
                assert not (used & regs), f"{name}: query register touched before its wait: {line.strip()}"
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", k), f"{name} uses scratch memory"
        assert not re.search(r"\n\s*flat_(load|store|atomic)", k), f"{name}: flat memory access"
    assert n_stream == 4
    # fourth check: the one-pass kernels (attend_small.h).  Where MFMAs are inline asm (two query
    # tiles, both heads: 256 pinned accumulator registers) hipcc does not know their results need
    # wait states: nothing but the asm MFMAs themselves may touch their destination registers inside
    # the loop (a first build parked accumulator tiles in vector registers around the logit chain);
    # and a spill inside the loop would be a vector-memory operation among the hand-counted ones
    n_small = 0
    for k in kernels:
        name = k.split(":", 1)[0]
        if "attend_small_kernel" not in name:
            continue
        n_small += 1
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", k), f"{name} uses scratch memory"
        body = [l.split(";")[0].strip() for l in k.splitlines()]
        body = [l for l in body if l and not l.startswith(".") and not l.startswith("#")]
        asm_dst = set()
        for i, l in enumerate(body[:-1]):
            m = re.match(r"v_mfma\S+ ([av])\[(\d+):(\d+)\]", l)
            if m and body[i + 1] == "s_nop 1":                     # (the asm forms carry a trailing s_nop 1)
                asm_dst.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        for l in body:
            if l.startswith("v_accvgpr") and "a[" not in l:
                regs = {("a", int(x)) for x in re.findall(r"\ba(\d+)\b", l)}
                # (zero-initialisation in front of the loop and the read-out behind it are apart from
                # every MFMA by construction: writes of 0 / reads after the final vmcnt(0) wait)
                if regs & asm_dst and "v_accvgpr_read" in l:
                    idx = body.index(l)
                    last_mfma = max(i for i, b in enumerate(body) if b.startswith("v_mfma"))
                    assert idx > last_mfma, f"{name}: accumulator of an asm MFMA read inside the loop: {l}"
    assert n_small == 4
    # fifth check: the persistent encoder launch and the batch top-k's GEMM passes must not use scratch
    # memory (round 5: per-part copies of the argument struct inside encoder_tile_kernel's loops put 456 B
    # of it on the stack - the small-batch encoder went from 58 to 74 us before anyone looked)
    n_noscratch = 0
    for k in kernels:
        name = k.split(":", 1)[0]
        if "encoder_tile_kernel" in name or "topk_gemm_kernel" in name or "attend_stored_kernel" in name:
            n_noscratch += 1
            assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", k), f"{name} uses scratch memory"
    assert n_noscratch >= 8


def test_bankfile_roundtrip_and_shards(tmp_path):
    from range_amd.bankfile import convert_npz, is_bankfile, load_any, load_bankfile
    db = synth.write_bank(str(tmp_path / "db.npz"), 1003, 9)
    ref = load_bank(db)
    rb = convert_npz(db, str(tmp_path / "db.rbank"))
    assert is_bankfile(rb) and not is_bankfile(db)
    b = load_bankfile(rb, verify=True)
    for x, y in ((b.keys, ref.keys), (b.values, ref.values), (b.xyz, ref.xyz)):
        assert x.dtype == np.float32 and np.array_equal(np.asarray(x), y)
    sh = load_bankfile(rb, rows=(250, 777))
    assert sh.n_rows == 527 and np.array_equal(np.asarray(sh.values), ref.values[250:777])
    assert np.array_equal(np.asarray(load_any(rb, (5, 9)).xyz), np.asarray(load_any(db, (5, 9)).xyz))
    with pytest.raises(ValueError):
        load_bankfile(rb, rows=(0, 5000))
    # a flipped byte is caught by the checksum
    raw = bytearray(open(rb, "rb").read())
    raw[4096 + 17] ^= 0xFF
    open(str(tmp_path / "bad.rbank"), "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        load_bankfile(str(tmp_path / "bad.rbank"), verify=True)


def test_a_sharded_rank_touches_only_its_slice_of_the_bank_file(tmp_path):
    """What the .rbank is for, pinned: a rank of W maps the file and touches 1/W of each
    section (+ page rounding) - measured as the pages of the mapped file in the process's RSS (RssFile) by
    tools/load_time.py's probe in a fresh process - where the reference's .npz is read, cast and
    normalised WHOLE by every rank (profiles/r06/load_time.log: 3.3 s and 1.6 GB per rank at N = 100 000)."""
    import json
    import subprocess
    import sys
    from range_amd.bankfile import write_bankfile
    N, W = 20_000, 8
    locs, vals, keys = synth.make_bank(N, 2024)
    path = write_bankfile(str(tmp_path / "db.rbank"), prepare_bank(locs, vals, keys))
    size_mb = os.path.getsize(path) / 2 ** 20          # (MiB, like /proc's RssFile)
    res = {}
    for rank, world in ((5, W), (0, 1)):
        p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "load_time.py"), "--probe", path, str(rank), str(world)],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        res[world] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    r = res[W]
    assert r["rows_loaded"] == N // W and r["host_s"] < 0.25                 # (an mmap and three slices: no read, no cast)
    # (+ 6 MiB: the pages of numpy's / python's own shared objects first touched under the measurement: a
    # constant 3.6-4 MiB at N = 20 000 and at N = 100 000)
    assert r["rss_file_mb"] <= size_mb / W + 6.0, (r["rss_file_mb"], size_mb / W)
    assert res[1]["rss_file_mb"] >= 0.9 * size_mb                            # (the whole bank: every page, once)


def test_checkpoint_with_uninstalled_helper_classes(tmp_path):
    """A Lightning checkpoint may pickle helper types of packages that are absent here
    (e.g. lightning.fabric.utilities.data.AttributeDict); the reader must still get the
    hyper-parameters and tensors out."""
    import sys
    import types
    mod = types.ModuleType("fakelightning_pkg")
    class AttributeDict(dict):
        pass
    AttributeDict.__module__ = "fakelightning_pkg"
    AttributeDict.__qualname__ = "AttributeDict"
    mod.AttributeDict = AttributeDict
    sys.modules["fakelightning_pkg"] = mod
    try:
        c = synth.make_checkpoint(L=10, hidden=64)
        c["hyper_parameters"] = AttributeDict(c["hyper_parameters"])
        c["callbacks"] = {"note": AttributeDict(a=1)}
        p = str(tmp_path / "lightning_like.ckpt")
        torch.save(c, p)
    finally:
        del sys.modules["fakelightning_pkg"]
    enc = read_checkpoint(p)
    assert (enc.legendre_polys, enc.hidden, enc.embed_dim) == (10, 64, 256)
    assert np.array_equal(enc.weights[0], synth.make_encoder_weights(10, 64)["layers.0.weight"])


def test_host_result_pool_recycles_only_unreferenced_memory():
    """range_amd/_hostpool.py: the memory of a dropped result is handed out again, but never
    while a view of that result is alive (it would be overwritten under the caller).  Ownership is
    the interpreter's own reachability of a per-result guard object - no reference counts are
    read - so the adversarial cases hold by construction: a view that outlives the result in
    another thread, a torch tensor over it, a view parked in a reference cycle until the garbage
    collector runs (what a delayed collector - PyPy, a free-threaded build - does to every object)."""
    import gc
    import threading
    from range_amd._hostpool import HostResultPool, _size_class
    P = HostResultPool(enabled=True)
    cap = _size_class(64 * 1280 * 8)
    a = P.take(64, 1280)
    assert a.dtype == np.float64 and a.shape == (64, 1280) and a.flags.c_contiguous and a.flags.writeable
    addr = a.ctypes.data
    del a
    assert P._free_bytes == cap
    b = P.take(64, 1280)
    assert b.ctypes.data == addr and P._free_bytes == 0        # recycled
    b[:] = 3.0
    v = b[2:4]                                                 # a view outlives the result
    del b
    assert P._free_bytes == 0                                  # not recycled ...
    c = P.take(64, 1280)
    c[:] = 5.0
    assert (v == 3.0).all()                                    # ... so the view keeps its data
    t = c.T
    del c
    assert P._free_bytes == 0
    del t, v
    assert P._free_bytes == 2 * cap
    # a torch tensor over a dropped result
    d = P.take(64, 1280)
    d[:] = 7.0
    tt = torch.from_numpy(d)[5]
    del d
    e = P.take(64, 1280); e[:] = 9.0
    f = P.take(64, 1280); f[:] = 11.0
    assert float(tt.sum()) == 7.0 * 1280
    del tt, e, f
    # a view handed to another thread that is still running when the result is dropped
    g = P.take(64, 1280)
    g[:] = 13.0
    seen, go = [], threading.Event()

    def worker(view):
        go.wait()
        seen.append(float(view.sum()))

    th = threading.Thread(target=worker, args=(g[10:20].reshape(-1),))
    th.start()
    del g
    others = [P.take(64, 1280) for _ in range(3)]
    for o in others:
        o[:] = -1.0
    go.set()
    th.join()
    assert seen == [13.0 * 10 * 1280]
    del others, th
    # a view in a reference cycle: unreachable, but only the garbage collector can tell
    gc.collect()
    gc.disable()
    try:
        free0 = P._free_bytes
        h = P.take(32, 1280)
        cyc = {"view": h[3:5]}
        cyc["self"] = cyc
        del h, cyc
        assert P._free_bytes == free0                                      # (a fresh block) still out of the pool
        gc.collect()
        assert P._free_bytes == free0 + _size_class(32 * 1280 * 8)         # in it once collected
    finally:
        gc.enable()
    # bounded: at most max_free_per_size blocks of one capacity are kept; capacities come from a
    # bounded set of size classes whatever the batch sizes
    P2 = HostResultPool(enabled=True)
    arrs = [P2.take(8, 1280) for _ in range(5)]
    del arrs
    assert P2._free_bytes == P2.max_free_per_size * _size_class(8 * 1280 * 8)
    assert len({_size_class(n * 10240) for n in range(1, 20000)}) < 120
    assert all(_size_class(n) >= n and _size_class(n) <= max(4096, n * 1.125 + 1) for n in (1, 70_000, 10 ** 6, 10 ** 8 + 7))
    # RANGE_HOST_POOL=0: plain fresh arrays
    off = HostResultPool(enabled=False)
    x = off.take(4, 1280)
    del x
    assert off._free_bytes == 0


@pytest.mark.parametrize("flags", [["-fsanitize=address,undefined", "-fno-sanitize-recover=all"],
                                   ["-fsanitize=thread"]], ids=["asan+ubsan", "tsan"])
def test_host_code_under_sanitizers(flags, tmp_path):
    """The host-side arithmetic and threading of the C-ABI library (range_amd/csrc/host_plan.h,
    host_copy.h - the headers librange_hip.so itself is built from) compiled with g++ under
    sanitizers and run on the CPU (sanitizers on the CPU build only)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    src = os.path.join(REPO, "tests", "native", "host_sanitize.cpp")
    subprocess.run(["g++", "-std=c++17", "-g", "-O1", "-pthread", *flags, src, "-o", exe], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "host_sanitize ok" in p.stdout, p.stdout + p.stderr


def test_bench_self_launch_fails_loudly_without_gpus():
    """``python bench.py --gpus 2`` from a plain shell spawns its rank processes (never touching
    a GPU in the parent) and exits NON-ZERO when the ranks fail - here because there is no GPU."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("this is the no-GPU case")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=REPO)
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr + p.stdout
    assert '{"metric"' not in p.stdout

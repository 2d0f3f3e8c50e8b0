"""Prepared bank file (``.rbank``): the ``range_db_*.npz`` bank converted ONCE into the arrays
the engine uploads - float32 unit keys, float32 values, float32 unit xyz - page-aligned so that
loading is an ``mmap`` and a row shard is a contiguous slice of each section.

The reference re-derives these on every ``LocationEncoder.__init__`` (range/range.py:78-95:
``np.load`` of float64 arrays, casts, a float32 normalisation, trigonometry).  The conversion
applies exactly that preparation (``range_amd.bank.prepare_bank``), so the file content is
bit-identical to what loading the .npz produces.

Layout (little endian):
    0     8   magic  b"RBANK\\x00\\x01\\x00"
    8     8   n_rows (u64)
    16    4   key_dim (u32) = 256      20  4  val_dim (u32) = 1024
    24    8   offset of keys   (u64, multiple of 4096)
    32    8   offset of values (u64)   40  8  offset of xyz (u64)
    48    16  md5 of the three sections
    4096  ..  keys (n,256) f32 | values (n,1024) f32 | xyz (n,3) f32, each 4096-aligned
"""
from __future__ import annotations

import hashlib
import struct
from typing import Optional, Tuple

import numpy as np

from .bank import PreparedBank, load_bank as _load_npz, prepare_bank

MAGIC = b"RBANK\x00\x01\x00"
ALIGN = 4096
_HEADER = struct.Struct("<8sQIIQQQ16s")


def _align(x: int) -> int:
    return (x + ALIGN - 1) // ALIGN * ALIGN


def write_bankfile(path: str, bank: PreparedBank) -> str:
    n = bank.n_rows
    k_off = ALIGN
    v_off = _align(k_off + n * 256 * 4)
    x_off = _align(v_off + n * 1024 * 4)
    md5 = hashlib.md5()
    for a in (bank.keys, bank.values, bank.xyz):
        md5.update(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    with open(path, "wb") as f:
        f.write(_HEADER.pack(MAGIC, n, 256, 1024, k_off, v_off, x_off, md5.digest()))
        for off, a in ((k_off, bank.keys), (v_off, bank.values), (x_off, bank.xyz)):
            f.seek(off)
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        f.truncate(_align(x_off + n * 3 * 4))
    return path


def convert_npz(npz_path: str, out_path: str) -> str:
    """range_db_*.npz (generate_db.py:212-214 schema) -> prepared .rbank."""
    return write_bankfile(out_path, _load_npz(npz_path))


def is_bankfile(path: str) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(8) == MAGIC
    except OSError:
        return False


def load_bankfile(path: str, rows: Optional[Tuple[int, int]] = None, verify: bool = False) -> PreparedBank:
    """mmap the file; ``rows=(start, stop)`` maps only that row shard.  Arrays are read-only
    memory maps - the engine copies them to the GPU straight from the page cache."""
    with open(path, "rb") as f:
        magic, n, kd, vd, k_off, v_off, x_off, digest = _HEADER.unpack(f.read(_HEADER.size))
    if magic != MAGIC or kd != 256 or vd != 1024:
        raise ValueError(f"{path}: not a range_amd bank file")
    lo, hi = (0, n) if rows is None else rows
    if not (0 <= lo <= hi <= n):
        raise ValueError(f"rows {rows} outside [0,{n}]")
    mm = lambda off, cols: np.memmap(path, dtype=np.float32, mode="r", offset=off + lo * cols * 4,
                                     shape=(hi - lo, cols))
    bank = PreparedBank(mm(k_off, 256), mm(v_off, 1024), mm(x_off, 3))
    if verify:
        if rows is not None:
            raise ValueError("verify needs the whole file")
        md5 = hashlib.md5()
        for a in (bank.keys, bank.values, bank.xyz):
            md5.update(np.ascontiguousarray(a).tobytes())
        if md5.digest() != digest:
            raise ValueError(f"{path}: checksum mismatch")
    return bank


def load_any(path: str, rows: Optional[Tuple[int, int]] = None) -> PreparedBank:
    """.rbank (mmap) or the reference's .npz (prepared on the fly)."""
    if is_bankfile(path):
        return load_bankfile(path, rows)
    bank = _load_npz(path)
    return bank if rows is None else bank.rows(*rows)


if __name__ == "__main__":      # python -m range_amd.bankfile range_db_large.npz range_db_large.rbank
    import sys

    if len(sys.argv) != 3:
        raise SystemExit("usage: python -m range_amd.bankfile <range_db_*.npz> <out.rbank>")
    print(convert_npz(sys.argv[1], sys.argv[2]))

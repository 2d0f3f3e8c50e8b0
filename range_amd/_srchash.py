"""SHA-256 over the sources of librange_hip.so (range_amd/csrc/*, include/*.h: file names and
contents, sorted; and build.sh, which holds the compiler flags).  build.sh embeds it in the library (``range_source_sha256()``, and as the literal
``RANGE_SRC_SHA256=<hex>`` in the file); ``__graft_entry__.build()`` rebuilds when the in-tree
library's stamp is not this checkout's, ``range_amd._native`` refuses a library built from other
sources, ``bench.py`` reports counter traffic only for the sources it was measured on.
No imports beyond the standard library: build.sh runs this file directly."""
import hashlib
import os

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARKER = b"RANGE_SRC_SHA256="


class SourcesMissing(FileNotFoundError):
    """The checkout's kernel sources are not there (a deployment that ships the built library only)."""


def source_files():
    out = []
    for d in (os.path.join(_REPO, "range_amd", "csrc"), os.path.join(_REPO, "include")):
        try:
            names = sorted(os.listdir(d))
        except OSError as ex:
            raise SourcesMissing(f"{d}: {ex.strerror or ex}") from ex
        for name in names:
            if name.endswith((".h", ".hip", ".cpp")):
                out.append(os.path.join(d, name))
    if not out:
        raise SourcesMissing(f"no kernel sources under {_REPO}/range_amd/csrc")
    build = os.path.join(_REPO, "build.sh")            # (the compiler flags live there)
    if os.path.exists(build):
        out.append(build)
    return out


def source_sha256() -> str:
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


def library_stamp(lib_path: str):
    """The source hash a built library carries (read from the file, without loading it), or None."""
    try:
        blob = open(lib_path, "rb").read()
    except OSError:
        return None
    i = blob.find(MARKER)
    if i < 0:
        return None
    hx = blob[i + len(MARKER):i + len(MARKER) + 64]
    try:
        return hx.decode("ascii") if len(hx) == 64 and int(hx, 16) >= 0 else None
    except ValueError:
        return None


if __name__ == "__main__":
    print(source_sha256())

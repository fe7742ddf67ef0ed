"""Digest of the sources libbgamd.so is compiled from (csrc/* + include/bgamd.h).  build() compiles it into the
library (-DBGAMD_SRC_HASH), bgamd_source_hash() returns it, and _capi.load() refuses a library whose digest differs
from the sources next to it -- a stale .so can therefore not pass for the committed code.  No imports beyond the
standard library: __graft_entry__.build() loads this file by path before anything else exists."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(_HERE)
ROOT = os.path.dirname(PKG)
MARKER = b"BGAMD_SRC_HASH="


def source_files():
    csrc = os.path.join(PKG, "csrc")
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))] + [os.path.join(ROOT, "include", "bgamd.h")]


def source_hash() -> str:
    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_hash(path: str):
    """The digest compiled into a built library, read from the file (no dlopen), or None."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(MARKER)
    if i < 0:
        return None
    j = blob.find(b"\0", i)
    return blob[i + len(MARKER):j].decode("ascii", "replace")

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
    """The tracked sources of the library and nothing else: csrc/*.h, csrc/*.hip and the ABI header.  (Compiler temporaries such
    as csrc/*.hipfb are git-ignored but may sit in the directory: a digest that took them in could not be reproduced from a
    checkout of the commit -- tests/test_abi_cpu.py checks exactly that against `git archive HEAD`.)"""
    csrc = os.path.join(PKG, "csrc")
    names = sorted(f for f in os.listdir(csrc) if f.endswith((".h", ".hip")))
    return [os.path.join(csrc, f) for f in names] + [os.path.join(ROOT, "include", "bgamd.h")]


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

"""Identity of the kernel build: a hash over EVERY source of the library (diffsheg_amd/csrc/*.hip, *.h, the Makefile with its
compile flags) and the public header.  rocprofv3 --pmc traffic files under profiles/ carry it, and bench.py attaches a committed
traffic figure to its roofline block only when it was collected on the same build.  Host-side files are included on purpose:
they choose grids, instantiations and per-launch shapes, so a change there can change what one launch does."""
from __future__ import annotations

import hashlib
import os

_ROOT = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_ROOT, "csrc")


def kernel_sources() -> list[str]:
    """Paths hashed into the build id, in a fixed order."""
    files = sorted(f for f in os.listdir(_CSRC) if f.endswith((".hip", ".h")) or f == "Makefile")
    return [os.path.join(_CSRC, f) for f in files] + [os.path.join(os.path.dirname(_ROOT), "include", "diffsheg_hip.h")]


def kernel_build_id() -> str:
    h = hashlib.sha256()
    for f in kernel_sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]

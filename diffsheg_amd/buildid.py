"""Identity of the kernel build: a hash over the HIP sources.  rocprofv3 --pmc traffic files under profiles/ carry it, and
bench.py attaches a committed traffic figure to its roofline block only when it was collected on the same kernel build."""
from __future__ import annotations

import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_build_id() -> str:
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_CSRC, "*.h*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]

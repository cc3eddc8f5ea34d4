"""Identity of the kernel build: a hash over the DEVICE-code sources (kernels and the headers they include).  rocprofv3 --pmc
traffic files under profiles/ carry it, and bench.py attaches a committed traffic figure to its roofline block only when it was
collected on the same kernel build.  Host-side orchestration (denoiser.hip, sampler.hip, capi.hip and their headers) launches
these kernels but does not change what one launch does, so it is not part of the identity."""
from __future__ import annotations

import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
_KERNEL_SOURCES = ("attention.hip", "dsh_common.h", "dsh_kernels.h", "gemm.hip", "rowops.hip", "sampler_kernels.hip", "tl2.hip",
                   "tl_common.h", "tl_linear.hip")


def kernel_build_id() -> str:
    h = hashlib.sha256()
    for f in _KERNEL_SOURCES:
        h.update(open(os.path.join(_CSRC, f), "rb").read())
    return h.hexdigest()[:16]

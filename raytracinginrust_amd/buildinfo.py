"""Identity of the kernel sources in this tree: profiles/ snapshots carry it, and bench.py only quotes counters from a profile
that was taken on the same kernels."""
from __future__ import annotations

import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
KERNEL_SOURCES = ("rt_kernel.hip", "rt_ir.h", "rt_rng.h", "rt_launch.h", "Makefile")


def kernel_source_id() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_id())

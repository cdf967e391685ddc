"""Identity of the kernel sources AND of the launch code in this tree (the launch shape — queue entries, LDS node cache, chunk size,
workgroups per CU — lives in rt_host.cpp and drives the PMC counters as much as the kernels do): profiles/ snapshots carry it, and
bench.py only quotes counters from a profile that was taken on the same build."""
from __future__ import annotations

import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
KERNEL_SOURCES = ("rt_kernel.hip", "rt_ir.h", "rt_rng.h", "rt_launch.h", "rt_host.cpp", "rt_scene.h", "rt_flatten.cpp", "Makefile")   # (rt_flatten.cpp since round 5: the filter tree it builds is what the box steps walk)


def kernel_source_id() -> str:
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_id())

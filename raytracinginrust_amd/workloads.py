"""BASELINE.json's configurations as named workloads (scene, frame, spp, depth)."""
from __future__ import annotations

from dataclasses import dataclass

from . import scenes


@dataclass(frozen=True)
class Workload:
    key: str
    scene: str
    W: int
    H: int
    spp: int
    max_depth: int
    note: str

    @property
    def samples(self) -> int:
        return self.W * self.H * self.spp

    def describe(self) -> str:
        return f"{self.scene} {self.W}x{self.H} spp={self.spp} depth={self.max_depth}"


WORKLOADS = {
    "C1": Workload("C1", "random", 400, 225, 64, 8, "random_scene (In-One-Weekend spheres), the reference's CPU-runnable case"),
    "C2": Workload("C2", "cornell", 800, 800, 1024, 50, "Cornell Box (rect + cube + DiffuseLight): the configuration BASELINE's metric is quoted on"),
    "C3": Workload("C3", "final", 800, 800, 4096, 50, "final scene (BVH, Perlin noise, volumes, motion blur); depth unspecified in BASELINE.json -> 50"),
    "C4": Workload("C4", "teapot", 1920, 1080, 2048, 50, "teapot.obj triangle mesh in the cornell_test room, tiles over 8 GPUs"),
    "C5": Workload("C5", "cornell", 3840, 2160, 8192, 50, "Cornell Box 4K strong-scaling case"),
}

# Algorithmic bytes per sample (SURVEY.md §8(d) event x record-size model), measured by the CPU oracle's event counters on each
# workload's own pixel grid under the default seed (tests/sweeps/measure_bytes_per_sample.py, round 2: C1 at its 64 spp, C2-C4 at
# 16 spp, C5 at 4 spp; the run is recorded in BASELINE.md).  ONE committed value per workload: bench.py's `roofline.achieved` uses it
# at every N, and never re-measures it.  (C5 is C2's scene on a 16:9 frame: the side columns look past the box, so fewer bounces.)
BYTES_PER_SAMPLE = {"C1": 5158.1, "C2": 1473.4, "C3": 2988.0, "C4": 2160.2, "C5": 1039.8}


def build(w: Workload, backend, earth=None):
    aspect = w.W / w.H
    if w.scene == "cornell":
        return scenes.cornell_box(backend, aspect_ratio=aspect)
    if w.scene == "random":
        return scenes.random_scene(backend, aspect_ratio=aspect)
    if w.scene == "final":
        return scenes.final_scene(backend, *earth, aspect_ratio=aspect)
    if w.scene == "teapot":
        return scenes.cornell_test(backend, scenes.asset_path("teapot.obj"), aspect_ratio=aspect)
    raise KeyError(w.scene)

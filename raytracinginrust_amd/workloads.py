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


# f64 operations per sample by kind: what the reference's arithmetic executes per camera path, counted by the op-counting build of
# the CPU oracle (oracle/orc_opcount.h: `double` replaced by a counting stand-in; same samples) on each workload's own pixel grid under
# the default seed (tests/sweeps/measure_ops_per_sample.py, round 3: C1 at 16 spp, C2/C3 at 4, C4 at 2, C5 at 1; kinds that never occur are left out).  ONE committed table per workload; bench.py prices it against the
# f64 VALU issue peak at every N and never re-measures it.
F64_OPS_PER_SAMPLE = {
    "C1": {"add": 973.23, "mul": 992.4, "div": 440.41, "sqrt": 25.23, "cmp": 819.77, "minmax": 795.85, "sin": 3.1, "cos": 1.16, "atan2": 2.51, "acos": 2.51, "negabs": 12.27, "cvt": 1501.51},
    "C2": {"add": 276.99, "mul": 247.13, "div": 95.28, "sqrt": 12.6, "cmp": 155.5, "minmax": 1.64, "sin": 0.82, "cos": 0.82, "negabs": 3.81, "cvt": 315.15},
    "C3": {"add": 952.16, "mul": 1032.77, "div": 250.74, "sqrt": 71.61, "cmp": 427.21, "minmax": 308.08, "sin": 0.79, "cos": 0.65, "atan2": 6.97, "acos": 6.97, "log": 2.57, "floor": 5.82, "negabs": 36.92, "cvt": 909.83},
    "C4": {"add": 522.01, "mul": 532.74, "div": 202.38, "sqrt": 14.54, "cmp": 328.95, "minmax": 244.48, "sin": 0.97, "cos": 0.97, "negabs": 2.94, "cvt": 683.31},
    "C5": {"add": 195.44, "mul": 167.35, "div": 62.42, "sqrt": 7.66, "cmp": 111.29, "minmax": 0.93, "sin": 0.46, "cos": 0.46, "negabs": 2.15, "cvt": 196.37},
}

# What one operation of each kind costs in f64 VALU issue slots, in units of ONE full-rate f64 instruction.  Everything comes from ONE
# committed measurement, profiles/r04_ubench.csv (tools/ubench on the MI355X: 4 waves on every SIMD, 16 independent operations per lane
# and loop trip, launches of ~20 ms, cost = kernel time / operations one SIMD issued x the shader clock measured in the same launch):
#   v_add_f64 4.69 and v_mul_f64 5.16 SIMD cycles per wave-instruction at 2.15-2.37 GHz under that load -> UNIT = their mean, 4.92 cycles
#   (the chip's specification says 4: 16 f64 lanes per clock; `F64_VALU_PEAK_OPS` below keeps the specification, and
#   `F64_VALU_MEASURED_ISSUE_OPS` states what the instruction stream actually sustains);
#   weight(op) = (cycles of the op's row - cycles of the add / mul the row's kernel also runs) / UNIT:
#   divide 59.19 -> 12.0, sqrt 89.29 - 4.69 -> 17.2, sin 249.1 - 4.7 -> 49.6, cos 250.3 - 4.7 -> 49.9, tan 350.3 - 9.9 -> 69.1,
#   atan 183.9 - 4.7 -> 36.4, atan2 209.7 - 4.7 -> 41.6, acos 110.9 - 5.2 -> 21.5, log 363.4 - 4.7 -> 72.8, log2 328.4 - 4.7 -> 65.7,
#   pow 744.2 - 9.9 -> 149.1, floor 12.6 - 9.9 -> 0.56, fmax / fmin (NaN-quieting pair of v_max, 16.67 - 4.69 for two) -> 1.22,
#   compare + select (17.97 - 4.69 for v_cmp and two v_cndmask) -> counted as 1.0 per comparison;
# sign flips / |x| are source modifiers and int<->f64 conversions are not f64 arithmetic: 0.  The path has no fused multiply-adds
# (-ffp-contract=off, as rustc), so one issue slot carries ONE flop per lane.  tests/test_host_logic.py recomputes this table from
# the CSV.  (Rounds 1-3 used stated figures for the libm rows — sin / cos 50, atan2 60, acos 50, log 45 — and 12.4 / 17.8 for
# division / square root; with the measured table the weighted
# counts per sample move from 9981 / 2169 / 8064 / 4494 / 1431 (C1 ... C5) to 9846 / 2123 / 7731 / 4457 / 1402.)
UBENCH_CSV = "profiles/r04_ubench.csv"
UBENCH_UNIT_ROWS = ("add_f64", "mul_f64")
# row of the CSV -> (operation kind, f64 adds and multiplies the row's kernel runs beside the operation)
UBENCH_ROWS = {"div_f64": ("div", 0, 0), "sqrt_f64": ("sqrt", 1, 0), "sin_f64": ("sin", 1, 0), "cos_f64": ("cos", 1, 0), "tan_f64": ("tan", 1, 1),
               "atan_f64": ("atan", 1, 0), "atan2_f64": ("atan2", 1, 0), "acos_f64": ("acos", 0, 1), "log_f64": ("log", 1, 0), "log2_f64": ("log2", 1, 0),
               "pow_f64": ("pow", 1, 1), "floor_f64": ("floor", 1, 1)}


def weights_from_ubench(rows: dict) -> dict:
    """VALU_OP_WEIGHTS from {row name: SIMD cycles per wave-instruction} of a tools/ubench run (see the comment above)."""
    add, mul = rows["add_f64"], rows["mul_f64"]
    unit = 0.5 * (add + mul)
    w = {"add": 1.0, "mul": 1.0, "cmp": 1.0, "negabs": 0.0, "cvt": 0.0, "minmax": round((rows["minmax_f64"] - add) / 2.0 / unit, 2)}
    for row, (kind, n_add, n_mul) in UBENCH_ROWS.items():
        w[kind] = round((rows[row] - n_add * add - n_mul * mul) / unit, 2 if kind == "floor" else 1)
    return w


VALU_OP_WEIGHTS = {"add": 1.0, "mul": 1.0, "cmp": 1.0, "minmax": 1.22, "floor": 0.56, "div": 12.0, "sqrt": 17.2,
                   "sin": 49.6, "cos": 49.9, "tan": 69.1, "atan": 36.4, "atan2": 41.6, "acos": 21.5, "log": 72.8, "log2": 65.7, "pow": 149.1,
                   "negabs": 0.0, "cvt": 0.0}
# The peak the fraction is quoted against: 256 CUs x 4 SIMDs x 16 f64 lanes per clock x 2.4 GHz = 39.3e12 lane-operations per second,
# the chip's SPECIFIED f64 vector rate (78.6 TFLOP/s counts an FMA as two) for a path without FMAs — i.e. 4 cycles per wave-instruction.
F64_VALU_PEAK_OPS = 39.3e12
# What the same SIMDs sustain on a stream of independent f64 adds and multiplies (profiles/r04_ubench.csv): 64 lanes / 4.92 cycles x
# 2.26 GHz (the clock under that load) x 1024 SIMDs = 30.1e12 lane-operations per second.  bench.py prints the fraction of this too.
F64_VALU_MEASURED_ISSUE_OPS = 64.0 / 4.92 * 2.26e9 * 1024


def valu_ops(per_kind: dict) -> float:
    """f64 VALU issue-equivalents per sample of a per-kind operation table."""
    return sum(VALU_OP_WEIGHTS[k] * v for k, v in per_kind.items())


def flops(per_kind: dict) -> float:
    """Plain f64 operation count per sample (every arithmetic operation, division, square root and libm call counted once)."""
    return sum(v for k, v in per_kind.items() if k not in ("negabs", "cvt"))


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

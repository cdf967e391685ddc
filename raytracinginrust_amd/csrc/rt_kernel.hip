// csrc/rt_kernel.hip — the per-pixel sample loop of 4meame/RayTracingInRust as one persistent HIP kernel
// for gfx950 (MI355X).  Replaces src/main.rs:772-833 (+ everything ray_color reaches).
//
// Shape of the kernel (not a translation of the reference's rayon loop):
//   * work unit = one camera path (pixel, sample).  A wavefront (64 lanes) pulls work chunks from a global queue (one atomic
//     per chunk: whole pixels for the bulk of the frame, 256-sample pieces for its tail) and generates the chunk's camera
//     paths 64 at a time, every lane busy, into a per-wave LDS queue.  Whenever lanes' paths end (miss, light, absorbed,
//     depth), `__ballot` + `mbcnt` prefix ranks let every dead lane pop the next queued path ("compaction by
//     regeneration"): the wave never drains between pixels, lanes stay full until the queue is empty.
//   * the reference's recursion `e + w * ray_color(child)` (src/main.rs:41-120) is linear, so it runs as an in-kernel
//     bounce loop carrying the throughput `beta`; radiance is added when the path terminates.
//   * top-level objects (HittableList push order) are walked with wave-uniform indices through the constant address
//     space -> scalar loads, no VGPR cost; only (t, object, primitive) of the closest hit is kept and the hit record is
//     rebuilt once per bounce with per-lane gathers.  BVH traversal is per lane and, in the reference's left-then-right order,
//     stackless (every node carries the link to where the recursion goes next); BVH kernels run one workgroup per CU and stage
//     the tree (or its top levels) in LDS.
//   * per-pixel sums stay in registers (one accumulator per lane); when a lane moves to another pixel the partial sums of
//     one pixel are combined by a masked wave butterfly and added to the frame with one hardware f64 atomic per channel.
//   * two loop shapes: list scenes run the bounce loop in lock-step (every lane's closest-hit search costs the same); mesh scenes
//     whose BVH stands beside other objects run a resumable search with persistent traversal (trace_resumable: lanes keep their
//     place inside the BVH while the rest of the wave shades / regenerates) — same samples, different scheduling.
//   * no MFMA: there is no dense contraction anywhere on this path.
//
// Arithmetic follows the reference expression by expression (cited inline) and is compiled with
// -ffp-contract=off, so in f64 every +,-,*,/ and sqrt rounds exactly like the CPU; only libm-class
// functions (sin, cos, atan2, acos, log) may differ from a host libm in the last ulp.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rt_ir.h"
#include "rt_rng.h"
#include "rt_launch.h"

// The kernels are built as two translation units from this one source (csrc/Makefile): RT_TU == 1 holds the lean (list-scene)
// kernels, RT_TU == 2 all the others; RT_TU == 0 (tools/mkvariant.sh) keeps everything together.  The split lets a code-generation
// choice differ between them: the shared-denominator Vec3 division below pays in the 168-VGPR kernels (*measured* +3.8 % teapot
// room, +1.7 % final scene) and costs the 128-VGPR lean kernel four more spilled registers (-1.2 % on the Cornell box).
#ifndef RT_TU
#define RT_TU 0
#endif
// Minimum resident waves per SIMD the register allocation is held to (512 VGPRs / waves, in steps of 8), per kernel family.
// *Measured* (round 2, A/B in one process, with -disable-machine-licm): the lock-step BVH kernels are faster at 4 waves with a few
// dozen spilled registers than at 3 without (random spheres +6.6 %, final scene +10 %); the persistent-traversal mesh kernel is not
// (131 spills at 4 waves: teapot room -26 %; round 5, filtered walk, 94 spills: -14 %); the lean kernel fits 5 since round 4.
#ifndef RT_WAVES_LEAN
#define RT_WAVES_LEAN 5
#endif
#ifndef RT_WAVES_BVH
#define RT_WAVES_BVH 4
#endif
#ifndef RT_WAVES_PERSIST
#define RT_WAVES_PERSIST 3
#endif
#ifndef RT_WAVES_PBR
#define RT_WAVES_PBR 3
#endif

namespace rt {

// Settled by A/B measurements (docs/history.md names the logs; tools/mkvariant.sh patches a copy of this file for a re-measurement):
static constexpr int BOX_STEPS = 6;             // box steps per box-or-leaf vote, lock-step BVH loops (round 5, filtered walk, against 3: 2 -1 %, 4 +1 %, 6 +2 %)
static constexpr int BOX_STEPS_PERSIST = 8;     // ... persistent loop (round 5, against 4: 3 -2 %, 6 +1.5 %, 8 +2 %)
static constexpr uint32_t WW_NUM = 5u, WW_DEN = 8u;       // leaf step once 5/8 of the lanes still in a walk hold a pending leaf (3/8 until round 5)
static constexpr uint32_t SPEC_NUM = 5u, SPEC_DEN = 8u;   // ... in the walk-ahead form (one-BVH worlds)
// The kernels are two translation units (RT_TU, above) so that these two code-generation choices can differ between them:
static constexpr bool SHARED_DIV3 = RT_TU == 2;  // shared-denominator Vec3 division: +3.8 % teapot room, +1.7 % final scene; -1.2 % Cornell box
static constexpr bool MERGE_ARMS = RT_TU == 1;   // shade_hit: Lambertian's two sampling arms and Metal share their common instruction runs (round 4,
                                                 // samples bit-identical): Cornell box +2.5 %; in the BVH kernels -3 ... -7 % (register allocation)

// ------------------------------------------------------------------ small vector algebra (src/vec.rs)
template <typename T> struct V3 { T x, y, z; };
#define DEV __device__ __forceinline__
template <typename T> DEV V3<T> mk(T x, T y, T z) { V3<T> v; v.x = x; v.y = y; v.z = z; return v; }
template <typename T> DEV V3<T> operator+(V3<T> a, V3<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> DEV V3<T> operator-(V3<T> a, V3<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> DEV V3<T> operator*(V3<T> a, V3<T> b) { return mk<T>(a.x * b.x, a.y * b.y, a.z * b.z); }
template <typename T> DEV V3<T> operator*(V3<T> a, T s) { return mk<T>(a.x * s, a.y * s, a.z * s); }
template <typename T> DEV V3<T> operator*(T s, V3<T> a) { return mk<T>(s * a.x, s * a.y, s * a.z); }
// Vec3 / f64 (vec.rs:186-190) = three IEEE divisions by one denominator.  The compiler expands each `/` into the same
// sequence (v_div_scale x2, v_rcp, two Newton steps, v_div_fmas, v_div_fixup); the denominator's part of it — scaling,
// reciprocal, refinement — is identical for the three whenever v_div_scale leaves the denominator at the same value for all
// three numerators (it rescales only at the ends of the exponent range).  Then it is computed once and each quotient
// finishes with its own numerator: the same instructions on the same values, bit for bit the three separate divisions.
// Otherwise (checked) the three plain divisions run.
DEV V3<double> operator/(V3<double> a, double d) {
    if (!SHARED_DIV3) return mk<double>(a.x / d, a.y / d, a.z / d);
    bool fx, fy, fz, unused;
    const double d0 = __builtin_amdgcn_div_scale(a.x, d, false, &unused);
    const double d1 = __builtin_amdgcn_div_scale(a.y, d, false, &unused);
    const double d2 = __builtin_amdgcn_div_scale(a.z, d, false, &unused);
    if (__double_as_longlong(d0) == __double_as_longlong(d1) && __double_as_longlong(d0) == __double_as_longlong(d2)) {
        const double nd = -d0;
        double r = __builtin_amdgcn_rcp(d0);
        r = __builtin_fma(r, __builtin_fma(nd, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(nd, r, 1.0), r);
        const double nx = __builtin_amdgcn_div_scale(a.x, d, true, &fx);
        const double ny = __builtin_amdgcn_div_scale(a.y, d, true, &fy);
        const double nz = __builtin_amdgcn_div_scale(a.z, d, true, &fz);
        const double qx = nx * r, qy = ny * r, qz = nz * r;
        V3<double> o;
        o.x = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nd, qx, nx), r, qx, fx), d, a.x);
        o.y = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nd, qy, ny), r, qy, fy), d, a.y);
        o.z = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nd, qz, nz), r, qz, fz), d, a.z);
        return o;
    }
    return mk<double>(a.x / d, a.y / d, a.z / d);
}
DEV V3<float> operator/(V3<float> a, float s) { return mk<float>(a.x / s, a.y / s, a.z / s); }
template <typename T> DEV T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                  // vec.rs:38-40
template <typename T> DEV V3<T> cross(V3<T> a, V3<T> b) { return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }   // vec.rs:46-54
DEV double rsqrt_(double x) { return ::sqrt(x); }
DEV float rsqrt_(float x) { return ::sqrtf(x); }
template <typename T> DEV T length(V3<T> a) { return rsqrt_(dot(a, a)); }                                           // vec.rs:42-44

template <typename T> DEV V3<T> normalized(V3<T> a) { return a / length(a); }                                       // vec.rs:56-58
template <typename T> DEV T sq_of_len(V3<T> a) { T l = length(a); return l * l; }                                   // `.length().powi(2)`
template <typename T> DEV T get(V3<T> v, uint32_t k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
template <typename T> DEV V3<T> ld3(const T* p) { return mk<T>(p[0], p[1], p[2]); }

// Scene tables are immutable for the whole launch: read them through the constant address space, so that a
// wave-uniform index becomes a scalar load (s_load, no VGPRs, scalar cache) and a per-lane index a plain gather.
#define CAS __attribute__((address_space(4)))
template <typename F> DEV F cl(const F* p) { return *(const CAS F*)p; }
template <typename T> DEV V3<T> cl3(const T* p) { return mk<T>(cl(p), cl(p + 1), cl(p + 2)); }
// Launch parameters that only the camera / work-queue code reads (camera, frame size, tiling, chunking, seed: ~60 scalar registers):
// read from the kernel-argument segment where they are used, through a laundered pointer, instead of living in — and being spilled
// from — scalar registers through the whole bounce loop.  (The kernel's only argument is the KParams block, at offset 0.)
#define COLD_K const KParams<T>* K = cold_params<T>()
#define PK(f) cl(&K->f)
template <typename T> DEV const KParams<T>* cold_params() {
    const KParams<T>* k = (const KParams<T>*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return k;
}
// Whole-record fetch of a 16-byte-aligned POD record as 16-byte (then 8 / 4-byte) pieces through the constant address space.
template <typename R> DEV R ld_record(const R* p) {
    static_assert(alignof(R) >= 16 && sizeof(R) % 4 == 0, "record must be 16-byte aligned");
    R r;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    constexpr size_t N = sizeof(R);
    const char* src = (const char*)p; char* dst = (char*)&r;
    size_t off = 0;
#pragma unroll
    for (; off + 16 <= N; off += 16) *(u4*)(dst + off) = *(const CAS u4*)(src + off);
    if (off + 8 <= N) { *(u2*)(dst + off) = *(const CAS u2*)(src + off); off += 8; }
    if (off + 4 <= N) { *(uint32_t*)(dst + off) = *(const CAS uint32_t*)(src + off); }
    return r;
}
// (whole-record fetches: a wave-uniform index gives a few s_load_dwordx4, a per-lane one — BVH leaves, the hit record, the material
// — a few global_load_dwordx4 instead of one vector load per field: per-lane vector loads are what the BVH kernels are short of)
template <typename T> DEV DRect<T> ld_rect(const DRect<T>* p) { return ld_record(p); }
template <typename T> DEV DSphere<T> ld_sphere(const DSphere<T>* p) { return ld_record(p); }
template <typename T> DEV DMSphere<T> ld_msphere(const DMSphere<T>* p) { return ld_record(p); }
template <typename T> DEV DTri<T> ld_tri(const DTri<T>* p) { return ld_record(p); }
template <typename T> DEV DOp<T> ld_op(const DOp<T>* p) { return ld_record(p); }
DEV DObject ld_obj(const DObject* p) { return ld_record(p); }
template <typename T> DEV DBvhNode<T> ld_node(const DBvhNode<T>* p) { return ld_record(p); }
template <typename T> DEV DMaterial<T> ld_mat(const DMaterial<T>* p) { DMaterial<T> r = ld_record(p); r.kind &= MAT_KIND_MASK; return r; }
template <typename T> DEV DPbr<T> ld_pbr(const DPbr<T>* p) { DPbr<T> r; r.metallic = cl(&p->metallic); r.subsurface = cl(&p->subsurface); r.specular = cl(&p->specular); r.roughness = cl(&p->roughness); r.specular_tint = cl(&p->specular_tint); r.anisotropic = cl(&p->anisotropic); r.sheen = cl(&p->sheen); r.sheen_tint = cl(&p->sheen_tint); r.clearcoat = cl(&p->clearcoat); r.clearcoat_gloss = cl(&p->clearcoat_gloss); return r; }
template <typename T> DEV DTexture<T> ld_tex(const DTexture<T>* p) { return ld_record(p); }
DEV DLight ld_light(const DLight* p) { DLight r; r.kind = cl(&p->kind); r.index = cl(&p->index); return r; }

DEV double m_sin(double x) { return ::sin(x); }   DEV float m_sin(float x) { return ::sinf(x); }
DEV double m_cos(double x) { return ::cos(x); }   DEV float m_cos(float x) { return ::cosf(x); }
DEV void m_sincos(double x, double& sn, double& cs) { ::sincos(x, &sn, &cs); }   // same reduction + polynomials as sin() and cos()
DEV void m_sincos(float x, float& sn, float& cs) { ::sincosf(x, &sn, &cs); }

// sin and cos of phi = 2*pi*r1 in [0, 2*pi) (pdf.rs:12-15, sphere.rs:31-33).  The general-purpose device sincos spends most
// of its ~150 instructions on argument reduction for arbitrary magnitudes; here the range is known, so: k = nearest
// integer to phi * 2/pi (0..4), r = phi - k*pi/2 with a two-word pi/2 (Cody-Waite, exact products via fma), then the
// classic minimax kernels on [-pi/4, pi/4] (coefficients of the fdlibm __kernel_sin/__kernel_cos polynomials).  Measured
// on the host against long double over 2e7 arguments: max error 0.78 ulp, 96.9 % of results bit-equal to glibc — the same
// class as the device libm (neither is correctly rounded); f64 only.
DEV void sincos_0_2pi(double phi, double& sn, double& cs) {
    const double two_over_pi = 6.36619772367581382433e-01;
    const double pio2_hi = 1.57079632673412561417e+00;     // first 33 bits of pi/2
    const double pio2_lo = 6.07710050630396597660e-11;     // next 33 bits (k * pio2_lo is exact for k <= 4)
    const double pio2_lo2 = 2.02226624879595063154e-21;    // pi/2 - (pio2_hi + pio2_lo)
    double kf = __builtin_rint(phi * two_over_pi);
    int k = (int)kf;
    double r = __builtin_fma(-kf, pio2_hi, phi);            // exact: kf <= 4 and pio2_hi has 33 significant bits
    double w = kf * pio2_lo;
    double r1 = r - w;
    double c = (r - r1) - w;                                // rounding error of r - w
    c = __builtin_fma(-kf, pio2_lo2, c);
    double x = r1 + c;                                      // reduced argument, |x| <= pi/4 (+ ulp)
    double y = c - (x - r1);                                // its tail
    double z = x * x;
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double v = z * x;
    double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
    double s_ = x - ((z * (0.5 * y - v * rs) - y) - v * S1);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
    double hz = 0.5 * z;
    double wv = 1.0 - hz;
    double c_ = wv + (((1.0 - wv) - hz) + (z * rc - x * y));
    // quadrant
    const bool swap = (k & 1) != 0;
    double ss = swap ? c_ : s_;
    double cc = swap ? s_ : c_;
    sn = (k & 2) ? -ss : ss;
    cs = ((k + 1) & 2) ? -cc : cc;
}
DEV void sincos_0_2pi(float phi, float& sn, float& cs) { ::sincosf(phi, &sn, &cs); }
DEV double m_log(double x) { return ::log(x); }   DEV float m_log(float x) { return ::logf(x); }
DEV double m_acos(double x) { return ::acos(x); } DEV float m_acos(float x) { return ::acosf(x); }
DEV double m_atan2(double y, double x) { return ::atan2(y, x); } DEV float m_atan2(float y, float x) { return ::atan2f(y, x); }
DEV double m_pow(double x, double y) { return ::pow(x, y); } DEV float m_pow(float x, float y) { return ::powf(x, y); }
DEV double m_log2(double x) { return ::log2(x); } DEV float m_log2(float x) { return ::log2f(x); }
DEV double m_tan(double x) { return ::tan(x); } DEV float m_tan(float x) { return ::tanf(x); }
DEV double m_atan(double x) { return ::atan(x); } DEV float m_atan(float x) { return ::atanf(x); }
DEV double m_floor(double x) { return ::floor(x); } DEV float m_floor(float x) { return ::floorf(x); }
DEV double m_abs(double x) { return ::fabs(x); }  DEV float m_abs(float x) { return ::fabsf(x); }
DEV double m_max(double a, double b) { return ::fmax(a, b); } DEV float m_max(float a, float b) { return ::fmaxf(a, b); }   // f64::max: NaN-ignoring
DEV double m_min(double a, double b) { return ::fmin(a, b); } DEV float m_min(float a, float b) { return ::fminf(a, b); }

// min / max of two numbers known not to be NaN: the bare instruction (fmin / fmax would first quiet each operand with a v_max x, x)
DEV double min_nn(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV double max_nn(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float min_nn(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float max_nn(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DEV float max3_nn(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DEV float min3_nn(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DEV float med3_nn(float a, float b, float c) { float r; asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
DEV float amax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// f() N times, as straight-line code (a `#pragma unroll` on a loop whose body holds a lambda with LDS reads and ballots is refused by the
// optimizer for some instantiations — "loop not unrolled", 108 warnings per build until round 6 — although it does unroll the others)
template <int N, typename F> DEV void repeat(F&& f) { if constexpr (N > 0) { f(); repeat<N - 1>(f); } }
template <typename T> struct Lim;
template <> struct Lim<double> { static DEV double max() { return 1.7976931348623157e308; } static DEV double inf() { return __longlong_as_double(0x7FF0000000000000LL); } };
template <> struct Lim<float> { static DEV float max() { return 3.402823466e38f; } static DEV float inf() { return __uint_as_float(0x7F800000u); } };
#define PI_T T(3.14159265358979323846264338327950288)
// t_min of world.hit (main.rs:48).  The reference's 0.00001 is below one f32 ulp at Cornell scale (ulp(555) = 6e-5),
// so the f32 throughput variant uses the 0.001 the reference's own comment (main.rs:47) names; f64 keeps 0.00001.
template <typename T> struct TMin;
template <> struct TMin<double> { static DEV double v() { return 0.00001; } };
template <> struct TMin<float> { static DEV float v() { return 0.001f; } };

template <typename T> struct RayT { V3<T> o, d; T tm; };
template <typename T> DEV V3<T> ray_at(const RayT<T>& r, T t) { return r.o + t * r.d; }                             // ray.rs:26-28

template <typename T> struct Rec {                                                                                 // hit.rs:9-24
    V3<T> p, n; T t, u, v; bool front; uint32_t mat;
};
template <typename T> DEV void set_face_normal(Rec<T>& rec, V3<T> dir, V3<T> outward) {                            // hit.rs:34-41
    rec.front = dot(dir, outward) < T(0);
    rec.n = rec.front ? outward : T(-1.0) * outward;
}

// ------------------------------------------------------------------ primitive tests (closest-hit search keeps only t)
DEV void plane_axes(uint32_t plane, uint32_t& ki, uint32_t& ai, uint32_t& bi) {                                    // rect.rs:26-32
    ki = 2u - plane; ai = (plane == 2u) ? 1u : 0u; bi = (plane == 0u) ? 1u : 2u;
}
template <typename T> DEV bool rect_test_axes(const DRect<T>& r, T ok, T dk, T oa, T da, T ob, T db, T t_min, T t_max, T& t_out) {   // rect.rs:49-60
    T t = (r.k - ok) / dk;
    if (t < t_min || t > t_max) return false;
    T a = oa + t * da;
    T b = ob + t * db;
    if (a < r.a0 || a > r.a1 || b < r.b0 || b > r.b1) return false;
    t_out = t;
    return true;
}
// The axis triple (k,a,b) of rect.rs:26-32 is chosen by a branch on `plane` rather than by per-component selects:
// at the top level `plane` is wave-uniform (a scalar branch); in BVH leaves (cube faces) it is uniform in practice.
template <typename T> DEV bool rect_test(const DRect<T>& r, const RayT<T>& ray, T t_min, T t_max, T& t_out) {
    if (r.plane == 2u) return rect_test_axes(r, ray.o.x, ray.d.x, ray.o.y, ray.d.y, ray.o.z, ray.d.z, t_min, t_max, t_out);   // YZ: k=x a=y b=z
    if (r.plane == 1u) return rect_test_axes(r, ray.o.y, ray.d.y, ray.o.x, ray.d.x, ray.o.z, ray.d.z, t_min, t_max, t_out);   // XZ: k=y a=x b=z
    return rect_test_axes(r, ray.o.z, ray.d.z, ray.o.x, ray.d.x, ray.o.y, ray.d.y, t_min, t_max, t_out);                      // XY: k=z a=x b=y
}
template <typename T> DEV bool sphere_test(V3<T> center, T radius, const RayT<T>& ray, T t_min, T t_max, T& t_out) {   // sphere.rs:56-74
    V3<T> oc = ray.o - center;
    T a = sq_of_len(ray.d);
    T half_b = dot(oc, ray.d);
    T c = sq_of_len(oc) - radius * radius;
    T discriminant = half_b * half_b - a * c;
    if (discriminant < T(0)) return false;
    T sqrt_d = rsqrt_(discriminant);
    T root = (-half_b - sqrt_d) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrt_d) / a;
        if (root < t_min || root > t_max) return false;
    }
    t_out = root;
    return true;
}
template <typename T> DEV V3<T> msphere_center(const DMSphere<T>& s, T time) {                                      // sphere.rs:144-146
    return ld3(s.c0) + (time - s.t0) / (s.t1 - s.t0) * (ld3(s.c1) - ld3(s.c0));
}
template <typename T> DEV bool tri_test(const DTri<T>& tr, const RayT<T>& ray, T t_min, T t_max, T& t_out, T& b1o, T& b2o) {   // tri.rs:24-41
    V3<T> s = ray.o - ld3(tr.v0);
    V3<T> e1 = ld3(tr.e1), e2 = ld3(tr.e2);
    V3<T> s1 = cross(ray.d, e2);
    V3<T> s2 = cross(s, e1);
    T s1_e1 = dot(s1, e1);
    const V3<T> q = mk<T>(dot(s2, e2), dot(s1, s), dot(s2, ray.d)) / s1_e1;      // tri.rs:33-35: three quotients, one denominator
    const T t = q.x, b1 = q.y, b2 = q.z;
    if (t < t_min || t > t_max) return false;
    if (b1 < T(0) || b2 < T(0) || (T(1.0) - b1 - b2) < T(0)) return false;
    t_out = t; b1o = b1; b2o = b2;
    return true;
}

// ------------------------------------------------------------------ Cube::hit: one exact rect test instead of six
// Cube::hit is HittableList::hit over the six faces (cube.rs:14-36): six times t = (k - o_k) / d_k, the range test, the bounds test.
// Its result is the accepted face with the smallest t.  The six plane distances are first computed APPROXIMATELY (the exact f64
// numerators k - o_k, converted to f32, times v_rcp_f32 of d_k: relative error <= 2^-21.5) and, when every comparison that decides the
// outcome is clear of that error, the answer is known without a division: the ray enters the box at s1 = max of the three per-axis
// nearer planes and leaves at e1 = min of the farther ones; if s1 > e1 no face is hit at all; otherwise only the entry and the exit
// face pass their bounds tests, and the winner is the entry face if its t lies in [t_min, closest], else the exit face if its t does.
// That ONE face then gets the reference's exact test — the same division, the same `o + t d` and comparisons (rect.rs:49-60), so the t
// that is kept is the reference's — and the other five divisions are never made.
// CLEAR means: with mu = RHO Tm + A (Tm = the largest magnitude among s1, e1 and the runners-up s2, e2), s1 - s2 > mu, e2 - e1 > mu and
// |e1 - s1| > mu; and s1 (or e1) lies inside or outside [t_min, closest] by more than RHO (|v| + |bound|).  Why that is enough:
//   * a face's bounds test on axis a compares p_a = fl(o_a + fl(t d_a)) with the box: in exact arithmetic that is t against slab a's
//     two plane distances, and fl() moves p_a by at most 2^-52 (|o_a| + |t d_a|), i.e. t's side of a plane distance v is decided
//     correctly whenever |t - v| > 2^-52 (|t| + |v| + M_a / |d_a|); A = 2^-40 max_a M_a |1/d_a| and RHO = 2^-18 >= 8 * 2^-21.5 cover that
//     and both approximation errors, for the pairs tested directly and — RHO being 8x what a pair needs — for the third-ranked
//     planes too (|v| <= 7 Tm: mu is already the pair's margin; |v| > 7 Tm: the gap is > 6/7 |v|, the margin < 1/2 |v|);
//   * so for a clear lane the ORDER of the six exact distances is the approximate one, every face other than entry / exit fails a
//     bounds test exactly, and entry / exit pass theirs iff s1 < e1; the range tests are decided the same way; the list's running
//     closest only ever drops to the exit face's t before the entry face is tried, which is larger: the entry face still wins.
// Anything not clear — a ray through an edge, a grazing ray, a thin box, a zero or denormal direction component (inf / NaN fail
// every comparison), t within a few ppm of t_min or closest — makes the WAVE take the six exact tests (range_hit's loop) instead.
// (KParams::rect_m >= every |coordinate| of every rect — a scene-wide M_a — and is 0 when some Cube has min > max: no fast path then.)
// WAVE: the kernels' form — false when some lane of the wave is not clear (the caller then runs the six exact tests for everybody;
// nothing was changed), the exact test only when some lane has a face to test.  !WAVE (the device known-answer test): per lane,
// `clear_out` says whether this lane's outcome was clear.  face_out in cube.rs:17-24 order.  One function, plain scalars: with the box
// or the classification in a struct the compiler selects among their fields through an ADDRESS, i.e. puts them in scratch memory.
// ROOM form (round 6; list-scene kernels): the same test for a box of which only SOME faces exist — bare AARects of a list that are exact faces
// of one axis-aligned box (the Cornell room's five walls; rt_flatten.cpp: form_room).  `map` holds, per face in cube.rs:17-24 order, three
// bits: the face's place in the room's run of rect records, or 7 = no such face.  The argument above is about the six plane distances and
// the box, not about which faces exist: for a clear lane only the entry and the exit face of the FULL box could pass their bounds tests,
// so of the faces that exist the list accepts the entry face if it exists and lies in range, else the exit face if it exists and lies in
// range, else nothing — an absent face is classed "out of range", whatever its distance.  face_out is then the record's place in the run.
// map == 0: a Cube, all six faces, face_out in cube.rs order (the code above, unchanged).
// (SITE: which call site family an instantiation serves — 0 everything, 1 the mesh kernels' world list.  The compiler's interprocedural
// constant propagation folds `map == 0` INTO these functions where every caller of the translation unit passes it; a caller with a real map
// would undo that for all of them and shift the other kernels' register allocation: its own copy keeps the others' machine code as it was.)
template <bool WAVE, int SITE = 0>
DEV bool cube_fast(float rect_m, double mnx, double mxx, double mny, double mxy, double mnz, double mxz, const RayT<double>& ray, double t_min, double t_max,
                   double& t_out, uint32_t& face_out, bool& any, bool& clear_out, uint32_t map = 0u) {
    const float rx = __builtin_amdgcn_rcpf((float)ray.d.x), ry = __builtin_amdgcn_rcpf((float)ray.d.y), rz = __builtin_amdgcn_rcpf((float)ray.d.z);
    // the six plane distances, approximately: the exact numerator k - o_k of rect.rs:50, rounded to f32, times the approximate 1 / d_k
    const float ax0 = (float)(mnx - ray.o.x) * rx, ax1 = (float)(mxx - ray.o.x) * rx;
    const float ay0 = (float)(mny - ray.o.y) * ry, ay1 = (float)(mxy - ray.o.y) * ry;
    const float az0 = (float)(mnz - ray.o.z) * rz, az1 = (float)(mxz - ray.o.z) * rz;
    const float nrx = min_nn(ax0, ax1), frx = max_nn(ax0, ax1), nry = min_nn(ay0, ay1), fry = max_nn(ay0, ay1), nrz = min_nn(az0, az1), frz = max_nn(az0, az1);
    const float s1 = max3_nn(nrx, nry, nrz), s2 = med3_nn(nrx, nry, nrz), e1 = min3_nn(frx, fry, frz), e2 = med3_nn(frx, fry, frz);
    const float RHO = 0x1p-18f;
    const float A = (0x1p-40f * rect_m) * amax3(rx, ry, rz);
    const float tm = __builtin_fmaxf(amax3(s1, e1, s2), __builtin_fabsf(e2));
    const float mu = __builtin_fmaf(RHO, tm, A);
    const bool order_clear = rect_m > 0.0f && tm < 1.0e36f && s1 - s2 > mu && e2 - e1 > mu;      // (1e36: far inside the +-1e37 that [t_min, t_max] is clamped to below)
    const bool through = e1 - s1 > mu, past = s1 - e1 > mu;              // enters before it leaves: the box is hit / leaves first: it is missed
    // a plane distance against [t_min, t_max], by more than the approximations can be off
    const float lo = __builtin_fmaxf((float)t_min, -1.0e37f), hi = __builtin_fminf((float)t_max, 1.0e37f);
    const float gl = RHO * __builtin_fabsf(lo), gh = RHO * __builtin_fabsf(hi);
    const float g1 = RHO * __builtin_fabsf(s1), g2 = RHO * __builtin_fabsf(e1);
    bool en_in = s1 - lo > g1 + gl && hi - s1 > g1 + gh, en_out = lo - s1 > g1 + gl || s1 - hi > g1 + gh;
    bool ex_in = e1 - lo > g2 + gl && hi - e1 > g2 + gh, ex_out = lo - e1 > g2 + gl || e1 - hi > g2 + gh;
    if (map != 0u) {                                                     // (wave-uniform) a room: which of the two faces exist
        const bool enx = s1 == nrx, eny = !enx && s1 == nry, exx = e1 == frx, exy = !exx && e1 == fry;
        // (the sign of d from its f32 reciprocal: the same sign wherever d is not a zero, and the axis of a zero component is never the
        // entry or exit axis of a clear lane — its plane distances are infinite)
        const float den_en = enx ? rx : (eny ? ry : rz), den_ex = exx ? rx : (exy ? ry : rz);
        // the entry plane of an axis is min's where d > 0 and max's where d < 0, the exit plane the other one; (XY, XZ, YZ) x (max, min)
        const uint32_t sh_en = (enx ? 12u : (eny ? 6u : 0u)) + (den_en < 0.0f ? 0u : 3u), sh_ex = (exx ? 12u : (exy ? 6u : 0u)) + (den_ex < 0.0f ? 3u : 0u);
        const bool has_en = ((map >> sh_en) & 7u) != 7u, has_ex = ((map >> sh_ex) & 7u) != 7u;
        // (*measured*, profiles/r06_room_ab.log: the same as lane-mask arithmetic over six wave-uniform "face exists" flags — seven vector
        // compares, the rest on the scalar unit — is 1.9 % SLOWER: the kernel has no scalar registers to spare either)
        en_in = en_in && has_en; en_out = en_out || !has_en;
        ex_in = ex_in && has_ex; ex_out = ex_out || !has_ex;
    }
    const bool clear = order_clear && (past || (through && (en_in || (en_out && (ex_in || ex_out)))));
    clear_out = clear;
    if (WAVE ? __ballot(!clear) != 0ull : !clear) return false;
    const bool cand = through && (en_in || ex_in);
    if (WAVE && __ballot(cand) == 0ull) return true;                      // nobody's ray has a face to test
    if (cand) {
        // which face: the axis of the chosen distance; of that axis' two planes the entry one is min's where d > 0, max's where d < 0
        const bool use_exit = !en_in;
        const bool on_x = use_exit ? e1 == frx : s1 == nrx, on_y = !on_x && (use_exit ? e1 == fry : s1 == nry), on_z = !on_x && !on_y;
        const double den = on_x ? ray.d.x : (on_y ? ray.d.y : ray.d.z), org = on_x ? ray.o.x : (on_y ? ray.o.y : ray.o.z);
        const bool hi_side = (den < 0.0) != use_exit;
        const double k = hi_side ? (on_x ? mxx : (on_y ? mxy : mxz)) : (on_x ? mnx : (on_y ? mny : mnz));
        const double t = (k - org) / den;                                 // rect.rs:50
        if (!(t < t_min || t > t_max)) {                                  // rect.rs:51-53
            const double px = ray.o.x + t * ray.d.x, py = ray.o.y + t * ray.d.y, pz = ray.o.z + t * ray.d.z;          // rect.rs:54-55, the two axes that are not the face's
            const bool out_x = px < mnx || px > mxx, out_y = py < mny || py > mxy, out_z = pz < mnz || pz > mxz;      // rect.rs:56-58
            if (!((out_x && !on_x) || (out_y && !on_y) || (out_z && !on_z))) {
                t_out = t; any = true;
                face_out = (on_x ? 4u : (on_y ? 2u : 0u)) + (hi_side ? 0u : 1u);                                      // (XY, XZ, YZ) x (max, min)
                if (map != 0u) face_out = (map >> (3u * face_out)) & 7u;                                              // a room: the record's place in its run
            }
        }
    }
    return true;
}
// Which instantiations carry the fast path (a Cube is served by the six exact tests elsewhere — same samples): the list-scene kernels and
// the all-features BVH kernels.  *Measured* (round 5, profiles/r05_cube_in_bvh_kernels_ab.log): final scene +4 % with it (400 ground boxes
// as BVH leaves); the mesh kernels and the one-BVH-world kernel, whose scenes have no Cube, lose 1 % to its registers: left out there.
template <typename T, uint32_t FEATS> struct CubeFast { static constexpr bool on = sizeof(T) == 8u && ((FEATS & F_BVH) == 0u || ((FEATS & F_SPHERES) != 0u && (FEATS & F_SPEC) == 0u)); };
// `room` (DObject::is_cube of a room object: 2 | map << 8; 0 for a Cube): the run holds `count` wall records in the list's order and, behind
// them, two records that only carry the box — laid out like a Cube's first two faces.
// The mesh kernels (BVH of triangles beside list objects: the teapot room, C4) carry the fast path at ONE site — the world list's objects —
// for rooms (round 6: their five walls stand next to each other in the list, nothing between them, so the room stands exactly where they
// stood: no tie rule, no second list) and bare Cubes; their BVH leaves hold triangles and keep the plain code (*measured* round 5: the
// fast path compiled into the leaf tests costs these kernels 1 % and serves nothing there).
// (the persistent-traversal ones: *measured* teapot room +1.6 %; the lock-step mesh kernels -0.7 % with it — they serve a room by its walls' exact tests)
template <typename T, uint32_t FEATS> struct RoomSite { static constexpr bool on = sizeof(T) == 8u && (FEATS & ~(uint32_t)F_NEAR_FIRST) == (uint32_t)(F_BVH | F_TRIS | F_PERSIST); };
template <int SITE = 0>
DEV bool cube_hit(const KParams<double>& P, uint32_t first, uint32_t count, uint32_t room, const RayT<double>& ray, double t_min, double t_max, double& t_out, uint32_t& prim_out, bool& any) {
    const uint32_t box_at = room != 0u ? first + count : first;
    const DRect<double> f0 = ld_rect(P.rects + box_at);                 // XY face at z = max.z: a = x range, b = y range (cube.rs:17)
    const double mnz = cl(&P.rects[box_at + 1u].k);                     // XY face at z = min.z (cube.rs:18)
    uint32_t face = 0u; bool hit = false, clear;
    if (!cube_fast<true, SITE>(P.rect_m, f0.a0, f0.a1, f0.b0, f0.b1, mnz, f0.k, ray, t_min, t_max, t_out, face, hit, clear, room >> 8)) return false;
    if (hit) { any = true; prim_out = (G_RECT << 28) | (first + face); }
    return true;
}

// closest accepted hit of a typed primitive range under HittableList semantics (hit.rs:59-71): each item is
// offered [t_min, closest_so_far]; a later item with t <= closest replaces an earlier one.
template <typename T, uint32_t FEATS>
DEV bool range_hit(const KParams<T>& P, uint32_t kind, uint32_t first, uint32_t count, const RayT<T>& ray, T t_min, T t_max, T& t_out, uint32_t& prim_out, bool is_cube = false, uint32_t room = 0u) {
    bool any = false;
    T closest = t_max;
    if constexpr (CubeFast<T, FEATS>::on) {
        if (kind == G_RECT && (is_cube || room != 0u) && cube_hit(P, first, count, room, ray, t_min, t_max, closest, prim_out, any)) { t_out = closest; return any; }
    }
    if (kind == G_RECT) {
        // software-pipelined record fetch: record i+1 is requested before record i is tested (the table carries one
        // padding record), so a scalar load's latency overlaps a whole rect test instead of stalling each iteration
        DRect<T> cur = ld_rect(P.rects + first);
        for (uint32_t i = first; i < first + count; i++) {
            // touch the current record first (its load was issued one iteration ago), THEN request the next one: the
            // scalar-load wait lands before the new request instead of behind it
            asm volatile("" :: "s"(cur.plane));
            __builtin_amdgcn_sched_barrier(0);
            const DRect<T> nxt = ld_rect(P.rects + i + 1);
            __builtin_amdgcn_sched_barrier(0);
            T t;
#ifdef RT_DIAG_RECTS    // (its own build: the two atomics per test distort RT_DIAG's timings) how often a rect test could be skipped for the whole
                        // wave by a filter on t: stats[14] tests, [15] with no lane in range
            if (P.stats) {
                const T tk = cur.plane == 2u ? (cur.k - ray.o.x) / ray.d.x : (cur.plane == 1u ? (cur.k - ray.o.y) / ray.d.y : (cur.k - ray.o.z) / ray.d.z);
                const bool in_range = !(tk < t_min || tk > closest);
                const unsigned long long m = __ballot(in_range);
                const unsigned long long ex = __ballot(true);
                if (__builtin_amdgcn_mbcnt_hi((uint32_t)(ex >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex, 0u)) == 0u) { atomicAdd(&P.stats[14], 1ull); if (m == 0ull) atomicAdd(&P.stats[15], 1ull); }
            }
#endif
            if (rect_test(cur, ray, t_min, closest, t)) { closest = t; prim_out = (G_RECT << 28) | i; any = true; }
            cur = nxt;
        }
    } else if ((FEATS & F_SPHERES) && kind == G_SPHERE) {
        for (uint32_t i = first; i < first + count; i++) {
            T t; const DSphere<T> s = ld_sphere(P.spheres + i);
            if (sphere_test(ld3(s.c), s.r, ray, t_min, closest, t)) { closest = t; prim_out = (G_SPHERE << 28) | i; any = true; }
        }
    } else if ((FEATS & F_SPHERES) && kind == G_MSPHERE) {
        for (uint32_t i = first; i < first + count; i++) {
            T t; const DMSphere<T> s = ld_msphere(P.mspheres + i);
            if (sphere_test(msphere_center(s, ray.tm), s.r, ray, t_min, closest, t)) { closest = t; prim_out = (G_MSPHERE << 28) | i; any = true; }
        }
    } else if ((FEATS & F_TRIS) && kind == G_TRI) {
        for (uint32_t i = first; i < first + count; i++) {
            T t, b1, b2;
            if (tri_test(ld_tri(P.tris + i), ray, t_min, closest, t, b1, b2)) { closest = t; prim_out = (G_TRI << 28) | i; any = true; }
        }
    }
    t_out = closest;
    return any;
}

// BVH::hit, bvh.rs:77-91: bbox test, then left subtree, then right subtree with t_max shrunk to the left hit.
// The recursion's t_max at any node equals min(original t_max, closest hit found earlier in DFS order), so one running
// `closest` over the same sequence of nodes is the same search.  In the reference's fixed order that sequence needs no stack:
// every node carries a skip link (rt_ir.h) — the node the recursion reaches next when this node's subtree is finished or culled.
// (Nearer-first order, opt-in, depends on the ray and keeps an explicit stack in LDS, one dword column per lane.)
// AABB::hit (aabb.rs:19-36) recomputes 1/d per node; the value is the same every time, so it is hoisted.
// In the reference's left-then-right order a later leaf with t <= t_max replaces an earlier one (hit.rs:62, bvh.rs:81-84).
// When children are visited nearer-first (RT_NEAR_FIRST_BVH) the same winner is kept by letting an exact tie go to the leaf
// that comes later in preorder (= later in the reference's DFS).
template <typename T> DEV bool bvh_accept(bool near_first, T t, T closest, uint32_t leaf, uint32_t best_leaf) {
    return !near_first || t < closest || !(t == closest) || leaf >= best_leaf;
}

// AABB::hit, aabb.rs:19-36, for one node.  `inv` is 1/d (aabb.rs:21 recomputes it per node; same value every time).
// EXACT form: per axis t0 = (min - o) * inv, t1 = (max - o) * inv, swapped when inv < 0, t_in = t0.max(t_in), t_out = t1.min(t_out)
// (f64::max / min ignore a NaN operand), miss as soon as t_out <= t_in.
// TAME form (same answer, fewer instructions), used when no NaN can arise and every box has min <= max:
//   * min <= max and monotonic rounding give (min - o) <= (max - o), so t0 <= t1 for inv > 0 and t0 >= t1 for inv < 0: the swapped
//     pair is (min(t0, t1), max(t0, t1));
//   * t_in only grows and t_out only shrinks along the axes, so "t_out <= t_in at some axis" is "t_out <= t_in after the last";
//   * without NaNs !(t_out <= t_in) is t_out > t_in.
// No NaN arises when o and inv are finite and |o|, |box| < 1e300 (no inf - inf, no 0 * inf); a ray with a zero direction component
// (inv = inf) or a scene with an inverted / non-finite box takes the exact form (ray_is_tame, KParams::bvh_tame).
template <typename T> DEV bool box_inside_exact(const DBvhNode<T>& nd, V3<T> o, V3<T> inv, T t_min, T closest) {
    bool inside = true;
    T t_in = t_min, t_o = closest;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        T inv_d = a == 0 ? inv.x : (a == 1 ? inv.y : inv.z);
        T org = a == 0 ? o.x : (a == 1 ? o.y : o.z);
        T t0 = (nd.mn[a] - org) * inv_d;
        T t1 = (nd.mx[a] - org) * inv_d;
        if (inv_d < T(0)) { T tmp = t0; t0 = t1; t1 = tmp; }
        t_in = m_max(t_in, t0);
        t_o = m_min(t_o, t1);
        if (t_o <= t_in) inside = false;    // aabb.rs:31-33 returns here; later axes cannot un-fail it
    }
    return inside;
}
template <typename T> DEV bool box_inside_tame(const DBvhNode<T>& nd, V3<T> o, V3<T> inv, T t_min, T closest) {
    const T ax = (nd.mn[0] - o.x) * inv.x, bx = (nd.mx[0] - o.x) * inv.x;
    const T ay = (nd.mn[1] - o.y) * inv.y, by = (nd.mx[1] - o.y) * inv.y;
    const T az = (nd.mn[2] - o.z) * inv.z, bz = (nd.mx[2] - o.z) * inv.z;
    const T t_in = max_nn(max_nn(max_nn(min_nn(ax, bx), t_min), min_nn(ay, by)), min_nn(az, bz));
    const T t_o = min_nn(min_nn(min_nn(max_nn(ax, bx), closest), max_nn(ay, by)), max_nn(az, bz));
    return t_o > t_in;
}
template <typename T> DEV bool finite_(T x) { return x - x == T(0); }
template <typename T> DEV bool ray_is_tame(V3<T> o, V3<T> inv) {
    const T big = T(sizeof(T) == 8 ? 1e300 : 1e30);
    return finite_(inv.x) && finite_(inv.y) && finite_(inv.z) && m_abs(o.x) < big && m_abs(o.y) < big && m_abs(o.z) < big;
}
// one BVH node through the constant address space with a 32-bit byte offset (scalar base + per-lane offset addressing)
template <typename T> DEV DBvhNode<T> ld_node_at(const DBvhNode<T>* base, uint32_t node) {
    return ld_record((const DBvhNode<T>*)((const char*)base + (size_t)(node * (uint32_t)sizeof(DBvhNode<T>))));
}
// The workgroup's dynamic LDS.  Layout (pathtrace_kernel): [n_cached BVH nodes][per wave: camera-path queue][per wave: BVH stack].
extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
// A traversal step's node.  Incoherent rays make every lane fetch a different 64-byte node: per-lane vector loads of that shape
// are what bounds the traversal (*measured*, round 2: doubling the node loads — same cache lines — cost the random-spheres frame
// +64 %, while removing a fifth of the box test's arithmetic gained 2 %).  Node ids are handed out by depth (rt_flatten.cpp), so
// the ids below n_cached are the top levels of every tree, where most visits go: each workgroup holds them in LDS (one copy per
// CU: the BVH kernels run one workgroup per CU) and a visit there is four ds_read_b128 instead of four global loads.
template <typename T> DEV DBvhNode<T> fetch_node(const KParams<T>& P, uint32_t node) {
    if (node < P.n_cached) return *(const DBvhNode<T>*)(lds_raw + node * (uint32_t)sizeof(DBvhNode<T>));
    return ld_node_at(P.bvh, node);
}

// ------------------------------------------------------------------ the filtered walk (f64 kernels, reference traversal order)
// BVH::hit (bvh.rs:77-91) tests a leaf's primitives iff the leaf's OWN box passes AABB::hit with the closest hit at that moment:
//   (=>) the recursion tests the leaf's box itself;  (<=) a parent's bounds are the min / max of its children's (aabb.rs:40-51), and
//   subtraction, multiplication by 1/d and min / max are monotonic in floating point, so a child's slab interval lies inside every
//   ancestor's — exactly, not approximately; and a box test made earlier saw a closest hit that was no smaller.  So when the leaf's box
//   passes with the present closest hit, every ancestor passed on the way down: the recursion is standing at this leaf with this t_max.
// The recursion is therefore an ORDERED SCAN OF THE LEAVES — own box, then primitives, the closest hit running — and the inner boxes only
// decide how much of the scan can be skipped; nothing about a skipped subtree reaches the result.  (Round 3 shipped a first use of this:
// speculative box steps past an untested leaf; round 5 takes it to its end.)  So the box steps need not run the reference's arithmetic:
// any test that never fails where the exact test of a leaf below would pass visits the same leaves with the same closest hits, and the
// exact f64 test (aabb.rs:19-36) is run once, on the leaf's own box, in the leaf step.  The box steps here are that: f32 slab tests on
// outward-rounded boxes with an error margin, 18 f32 instructions and two 16-byte LDS reads per node instead of ~45 f64 / select
// instructions and four reads.  The same freedom lets the flattener take near-duplicate inner nodes out (rt_flatten.cpp: contraction).
//
// CONSERVATIVE, proof.  Ray (o, inv = 1/d as the kernel computes it, both f64), a leaf box B, a node X whose f64 box contains B, X32 its
// outward-rounded f32 box (contains B too).  For axis j let N_j(.) / F_j(.) be the REAL near / far slab values min / max((mn_j - o_j) inv_j,
// (mx_j - o_j) inv_j): N_j(X32) <= N_j(B), F_j(X32) >= F_j(B).  The exact test computes near_j, far_j = fl64(fl64(m - o_j) inv_j), each
// within 2.01 * 2^-53 (|m| + |o_j|) |inv_j| of the real value, and passes iff min(c, far_x, far_y, far_z) > max(t_min, near_x, near_y, near_z).
// The filter computes v = fma32(m32, inv32_j, s) with inv32 = rn32(inv), s = rn32(rn32(-fl64(o_j inv_j)) -/+ E_j):
//   |m32 inv32 - m32 inv| <= u |m32| |inv|,  |rn32(-fl64(o inv)) + o inv| <= 1.01 u |o| |inv|,  the rounding of s <= 1.01 u (|o| |inv| + E_j),
//   the fma's own rounding <= u |v| <= 1.01 u (|m32| + |o|) |inv|      (u = 2^-24; ranges below keep everything normal)
// so v is within 3.1 u (M + |o_j|) |inv_j| + 1.01 u E_j of (m32 - o_j) inv_j -/+ E_j, where M >= every |coordinate| of every X32.  With
// E_j = 6 u (M + |o_j|) |inv_j| (computed in f32: >= 5.99 u (..)) the near value is <= N_j(X32) <= N_j(B) <= near_j + tiny and the far value
// >= F_j(X32) >= far_j - tiny, "tiny" (2^-52 (..)) being covered by the slack left in E_j.  With c32 >= c and tmin32 <= t_min (rounded
// away and bumped by an ulp), min(c32, far'..) >= min(c, far..) and max(tmin32, near'..) <= max(t_min, near..): exact pass => filter pass.
// Ranges (else the wave takes the exact walk): o, inv finite, |o_j| <= 2^40, 2^-40 <= |inv_j| <= 2^40, every box finite, min <= max,
// M <= 2^40 (host: KParams::filter_m, >= 1): products stay below 2^82 and E_j above 2^-62, so no overflow, and an underflow (< 2^-126)
// anywhere is far inside E_j.  A NaN closest hit (never seen) converts to a quiet NaN, which v_min_f32 ignores: conservative.
// The f32 kernels (RT_F32, T = float) walk the same filter tree; their LEAF test is the exact form in f32 on the node's coordinates rounded
// to NEAREST (the outward-rounded filter box contains those too): near_j = fl32(fl32(m - o_j) inv_j) is within 2.01 u (|m| + |o_j|) |inv_j|
// of the real value — u, not 2^-53, so it no longer hides in the slack — and the filter's own terms are as above with inv32 = inv and
// nx = fl32(-(o_j inv_j)) (one rounding, where the f64 kernels have a product and a conversion): 3.1 u + 2.01 u = 5.11 u of (M + |o_j|) |inv_j|
// plus 1.01 u E_j.  E_j = 7 u (..) there (>= 6.99 u as computed): 1.8 u of slack.  rt_debug_aabb_hit checks both precisions.
struct BoxFilter { float ix, iy, iz, ax, bx, ay, by, az, bz, tmin, c; bool ok; };
DEV float up32(double x) { x = x < -3.0e38 ? -3.0e38 : x; const float f = (float)x; return __builtin_fmaf(__builtin_fabsf(f), 0x1p-23f, f); }      // >= x
DEV float down32(double x) { x = x > 3.0e38 ? 3.0e38 : x; const float f = (float)x; return __builtin_fmaf(__builtin_fabsf(f), -0x1p-23f, f); }   // <= x
DEV float up32(float x) { return x; }
DEV float down32(float x) { return x; }
template <typename T> DEV BoxFilter make_filter(float M, V3<T> o, V3<T> inv, T t_min, T closest) {
    BoxFilter F;
    F.ix = (float)inv.x; F.iy = (float)inv.y; F.iz = (float)inv.z;
    const float jx = __builtin_fabsf(F.ix), jy = __builtin_fabsf(F.iy), jz = __builtin_fabsf(F.iz);
    const float ox = __builtin_fabsf((float)o.x), oy = __builtin_fabsf((float)o.y), oz = __builtin_fabsf((float)o.z);
    F.ok = M > 0.0f && max3_nn(jx, jy, jz) <= 0x1p40f && min3_nn(jx, jy, jz) >= 0x1p-40f && max3_nn(ox, oy, oz) <= 0x1p40f;
    const float K = (sizeof(T) == 8u ? 6.0f : 7.0f) * 0x1p-24f;      // (f32 kernels: the leaf's exact test has an error of its own, see the proof)
    const float ex = __builtin_copysignf((M + ox) * jx * K, F.ix), ey = __builtin_copysignf((M + oy) * jy * K, F.iy), ez = __builtin_copysignf((M + oz) * jz * K, F.iz);
    const float nx = (float)(-(o.x * inv.x)), ny = (float)(-(o.y * inv.y)), nz = (float)(-(o.z * inv.z));
    // the min plane is the near one where inv > 0: it gets -E, the max plane +E; the other way round where inv < 0 (the sign rides on E)
    F.ax = nx - ex; F.bx = nx + ex; F.ay = ny - ey; F.by = ny + ey; F.az = nz - ez; F.bz = nz + ez;
    F.tmin = down32(t_min); F.c = up32(closest);
    asm volatile("" : "+v"(F.tmin));      // a register, not a literal re-materialised in every box step
    return F;
}
DEV bool filter_pass(const DFNode& nd, const BoxFilter& F) {
    const float a0 = __builtin_fmaf(nd.b[0], F.ix, F.ax), b0 = __builtin_fmaf(nd.b[1], F.ix, F.bx);
    const float a1 = __builtin_fmaf(nd.b[2], F.iy, F.ay), b1 = __builtin_fmaf(nd.b[3], F.iy, F.by);
    const float a2 = __builtin_fmaf(nd.b[4], F.iz, F.az), b2 = __builtin_fmaf(nd.b[5], F.iz, F.bz);
    const float t_in = max_nn(max3_nn(min_nn(a0, b0), min_nn(a1, b1), min_nn(a2, b2)), F.tmin);
    const float t_o = min_nn(min3_nn(max_nn(a0, b0), max_nn(a1, b1), max_nn(a2, b2)), F.c);
    return !(t_o < t_in);
}
// Where a walk stands is one word per lane, a STATE: a node (box step next) | a leaf's id with FNODE_LEAF set (its box passed the
// filter: leaf step next) | ST_DONE.  When the whole tree is in LDS (KParams::n_cached == n_bvh: every BASELINE scene) the staged copy's
// links are LDS ADDRESSES (pathtrace_kernel rewrites them while it stages) and so is a node state: a box step is then two ds_read_b128
// at the state itself, no shift, no base, no cached-or-not test.  Otherwise states are node ids and fetch_fnode picks LDS or memory.
static const uint32_t ST_DONE = 0xFFFFFFFFu;
typedef __attribute__((address_space(3))) const unsigned char* lds_cptr;
DEV uint32_t lds_base() { return (uint32_t)(size_t)(lds_cptr)lds_raw; }
template <typename T> DEV bool all_in_lds(const KParams<T>& P) { return P.n_cached >= P.n_bvh; }
template <typename T> DEV uint32_t state_of(const KParams<T>& P, uint32_t id) { return id == ST_DONE ? ST_DONE : (all_in_lds(P) ? lds_base() + id * (uint32_t)sizeof(DFNode) : id); }
template <typename T> DEV uint32_t id_of(const KParams<T>& P, uint32_t state) { return all_in_lds(P) ? (state - lds_base()) / (uint32_t)sizeof(DFNode) : state; }
DEV bool st_walking(uint32_t st) { return st < FNODE_LEAF; }
DEV bool st_pending(uint32_t st) { return (int32_t)st >= (int32_t)FNODE_LEAF; }          // (ST_DONE is negative)
template <bool ALL, typename T> DEV DFNode fetch_fnode(const KParams<T>& P, uint32_t st) {
    if (ALL) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const u4* lds_u4;
        DFNode nd;
        *(u4*)&nd = *(lds_u4)(size_t)st; *((u4*)&nd + 1) = *((lds_u4)(size_t)st + 1);
        return nd;
    }
    if (st < P.n_cached) return *(const DFNode*)(lds_raw + st * (uint32_t)sizeof(DFNode));
    return ld_record((const DFNode*)((const char*)P.bvh_f + (size_t)(st * (uint32_t)sizeof(DFNode))));
}
// the state a walk moves to once leaf `id` is finished (the leaf's skip link, in the form states have in this launch)
template <typename T> DEV uint32_t fnode_skip(const KParams<T>& P, uint32_t id) {
    if (id < P.n_cached) return *(const uint32_t*)(lds_raw + id * (uint32_t)sizeof(DFNode) + 24u);
    return cl((const uint32_t*)((const char*)P.bvh_f + (size_t)(id * (uint32_t)sizeof(DFNode)) + 24u));
}
// AABB::hit's exact form on an f64 node, as a state transition (untamed waves only: a ray or a scene outside the filter's ranges)
template <typename T> DEV uint32_t exact_step(const KParams<T>& P, uint32_t st, V3<T> o, V3<T> inv, T t_min, T closest) {
    const uint32_t id = id_of(P, st);
    const DBvhNode<T> nd = ld_node_at(P.bvh, id);
    if (!box_inside_exact(nd, o, inv, t_min, closest)) return state_of(P, nd.skip);
    return (nd.a & BVH_LEAF) ? (id | FNODE_LEAF) : state_of(P, nd.c);
}
// BVH leaves that are not bare primitives (F_NESTED kernels): any Hittable is a legal child of BVH::new (bvh.rs:18-31 needs only its
// bounding_box) — a list, a wrapped object, a ConstantMedium, another BVH.  Such a leaf holds a run of SUB-OBJECTS (rt_ir.h: DObject::nest)
// and its leaf step is HittableList::hit (hit.rs:59-71) over them through the same object_hit the world list uses, one nesting level down.
struct HitId { uint32_t obj, prim; };   // prim: GeomKind << 28 | index, or PRIM_MEDIUM; obj: the object (top-level, or the sub-object of a G_OBJ leaf)
static const uint32_t PRIM_MEDIUM = 0xFFFFFFFFu, NO_SUB = 0xFFFFFFFFu;
template <uint32_t FEATS> struct Nested { static constexpr bool on = (FEATS & F_NESTED) != 0u; };
template <typename T, uint32_t FEATS, int NEST>
DEV void object_hit(const KParams<T>& P, uint32_t oi, const DObject& ob, const RayT<T>& ray, T t_min, Rng& rng, T& closest, HitId& id, bool& any, uint32_t* stack);
template <typename T, uint32_t FEATS, int NEST>
DEV bool subobjects_hit(const KParams<T>& P, uint32_t first, uint32_t count, const RayT<T>& ray, T t_min, T t_max, Rng& rng, T& t_out, uint32_t& prim_out, uint32_t& sub_out) {
    T closest = t_max; bool any = false;
    HitId id; id.obj = 0u; id.prim = 0u;
    if constexpr (Nested<FEATS>::on && NEST <= RT_MAX_NEST) {
        for (uint32_t i = first; i < first + count; i++) object_hit<T, FEATS, NEST + 1>(P, i, ld_obj(P.objects + i), ray, t_min, rng, closest, id, any, nullptr);
    }
    t_out = closest; prim_out = id.prim; sub_out = id.obj;
    return any;
}
// which instantiations walk this way (the host sizes the LDS node cache by the same rule: rt_launch.h filtered_walk)
template <typename T, uint32_t FEATS> struct Filt { static constexpr bool on = (FEATS & F_BVH) != 0u && (FEATS & F_NEAR_FIRST) == 0u; };

// Box steps and leaf steps are chosen by vote as in bvh_hit_ww.  SPEC (worlds that are one BVH: every lane walks): a lane does not wait
// with one pending leaf, it walks on with its closest hit as it is and waits with two — every leaf is tested against its own box with
// the closest hit of THAT moment in the leaf step anyway, so walking ahead with a stale (larger) bound only visits more.
template <typename T, uint32_t FEATS, bool SPEC, int NEST>
DEV bool bvh_hit_filt(const KParams<T>& P, uint32_t root, const RayT<T>& ray, T t_min, T t_max, T& t_out, uint32_t& prim_out, Rng& rng, uint32_t& sub_out) {
    const V3<T> inv = mk<T>(T(1.0) / ray.d.x, T(1.0) / ray.d.y, T(1.0) / ray.d.z);
    T closest = t_max;
    bool any = false;
    BoxFilter F = make_filter(P.filter_m, ray.o, inv, t_min, closest);
    const bool tame = P.bvh_tame != 0u && __ballot(!(ray_is_tame(ray.o, inv) && F.ok)) == 0ull;   // wave-uniform: every lane of this search
    uint32_t node = state_of(P, root), p1 = ST_DONE;       // SPEC: p1 = the older pending leaf (id); `node` may hold a second one
    auto box_steps = [&](auto all) {
        constexpr bool ALL = decltype(all)::value;
        for (;;) {
            const bool want_box = st_walking(node);
            const uint32_t n_box = (uint32_t)__popcll(__ballot(want_box)), n_leaf = (uint32_t)__popcll(__ballot(SPEC ? p1 != ST_DONE : st_pending(node)));
            const uint32_t n_act = SPEC ? (uint32_t)__popcll(__ballot(node != ST_DONE || p1 != ST_DONE)) : n_box + n_leaf;
            if (n_box == 0u || n_leaf * (SPEC ? SPEC_DEN : WW_DEN) >= n_act * (SPEC ? SPEC_NUM : WW_NUM)) break;
            auto box_step = [&]() {
                const DFNode nd = fetch_fnode<ALL>(P, node);
                const bool pass = filter_pass(nd, F);
                const bool ahead = SPEC && pass && (nd.info & FNODE_LEAF) != 0u && p1 == ST_DONE;      // first pending leaf: remember it, walk on
                if (ahead) p1 = nd.info & ~FNODE_LEAF;
                node = (pass && !ahead) ? nd.info : nd.skip;
            };
            if (want_box) box_step();
            repeat<BOX_STEPS - 1>([&]() { if (st_walking(node)) box_step(); });
        }
    };
    for (;;) {
        if (tame) { if (all_in_lds(P)) box_steps(std::true_type()); else box_steps(std::false_type()); }
        else while (__ballot(st_walking(node)) != 0ull) { if (st_walking(node)) node = exact_step(P, node, ray.o, inv, t_min, closest); }   // to every lane's next leaf
        const uint32_t leaf = (SPEC && p1 != ST_DONE) ? p1 : (st_pending(node) ? node & ~FNODE_LEAF : ST_DONE);
        if (leaf != ST_DONE) {
            const DBvhNode<T> lf = ld_node_at(P.bvh, leaf);
            T t; uint32_t prim;
            if constexpr (Nested<FEATS>::on) {
                // (untamed waves — a Rotate child's box is all of space, rotate.rs:40-57 — come here from exact_step, which has run AABB::hit on the leaf's own box)
                uint32_t sub = NO_SUB;
                if (!tame || box_inside_tame(lf, ray.o, inv, t_min, closest)) {                     // aabb.rs:19-36 on the leaf's own box
                    const uint32_t lk = (lf.a >> 28) & 7u;
                    const bool hit = lk == G_OBJ ? subobjects_hit<T, FEATS, NEST>(P, lf.a & 0x0FFFFFFFu, lf.b, ray, t_min, closest, rng, t, prim, sub)
                                                 : range_hit<T, FEATS>(P, lk, lf.a & 0x0FFFFFFFu, lf.b, ray, t_min, closest, t, prim, lf.b == 6u);
                    if (hit) { closest = t; prim_out = prim; sub_out = lk == G_OBJ ? sub : NO_SUB; any = true; F.c = up32(closest); }
                }
            } else
            if ((!tame || box_inside_tame(lf, ray.o, inv, t_min, closest)) &&                       // aabb.rs:19-36 on the leaf's own box
                range_hit<T, FEATS>(P, (lf.a >> 28) & 7u, lf.a & 0x0FFFFFFFu, lf.b, ray, t_min, closest, t, prim, lf.b == 6u)) { closest = t; prim_out = prim; any = true; F.c = up32(closest); }
            if (SPEC && p1 != ST_DONE) { p1 = ST_DONE; if (st_pending(node)) { p1 = node & ~FNODE_LEAF; node = fnode_skip(P, p1); } }   // the second one moves up; the walk goes on behind it
            else node = fnode_skip(P, leaf);
        }
        if (__ballot(node != ST_DONE || (SPEC && p1 != ST_DONE)) == 0ull) break;
    }
    t_out = closest;
    return any;
}

// `root` is the node a lane's walk starts at.
template <typename T, uint32_t FEATS>
DEV bool bvh_hit_ww(const KParams<T>& P, uint32_t root, const RayT<T>& ray, T t_min, T t_max, T& t_out, uint32_t& prim_out, uint32_t* stack) {
    V3<T> inv = mk<T>(T(1.0) / ray.d.x, T(1.0) / ray.d.y, T(1.0) / ray.d.z);
    T closest = t_max;
    bool any = false;
    uint32_t node = root;
    uint32_t sp = 0;
    const bool near_first = (FEATS & F_NEAR_FIRST) != 0u;   // RT_NEAR_FIRST_BVH (opt-in), compile-time: no cost in the default mode
    uint32_t best_leaf = 0;
    // "while-while" traversal with a vote: box steps (bbox test, next node) and leaf steps (primitive tests) are separate, so
    // that the expensive primitive tests never run with one or two lanes active.  A lane's own sequence of box tests, leaf tests
    // and t_max updates is exactly the recursion's (bbox, left, right): a lane that reaches a leaf keeps it pending and waits.
    // *Measured* against "every lane waits until all hold a leaf": final scene +8 %, random spheres +15 %; against one node (box and
    // leaf) per iteration on the teapot mesh: +8 %.
    // Box steps go on until the lanes holding a leaf are at least WW_NUM/WW_DEN (3/8: the best of 1/8 .. 1) of the lanes still
    // walking, or nobody can step; then the pending leaves are tested together.
    uint32_t leaf_a = 0, leaf_b = 0, leaf_node = 0;
    bool have_leaf = false;
    const uint32_t DONE = 0xFFFFFFFFu;
    const bool tame = P.bvh_tame != 0u && __ballot(!ray_is_tame(ray.o, inv)) == 0ull;   // wave-uniform: every lane of this search
    for (;;) {
        for (;;) {
            const bool want_box = node != DONE && !have_leaf;
            const uint32_t n_box = (uint32_t)__popcll(__ballot(want_box)), n_leaf = (uint32_t)__popcll(__ballot(have_leaf));
            if (n_box == 0u || n_leaf * WW_DEN >= (n_box + n_leaf) * WW_NUM) break;
            auto box_step = [&]() {
                const DBvhNode<T> nd = fetch_node(P, node);
                const bool inside = tame ? box_inside_tame(nd, ray.o, inv, t_min, closest) : box_inside_exact(nd, ray.o, inv, t_min, closest);
                if (!near_first) {
                    // reference order (left, then right): a threaded preorder walk — into the left child on a hit of an inner node,
                    // otherwise along the node's skip link (the node the recursion would reach next).  No stack.
                    if (inside && (nd.a & BVH_LEAF)) { have_leaf = true; leaf_a = nd.a; leaf_b = nd.b; leaf_node = nd.c; }
                    node = (inside && !(nd.a & BVH_LEAF)) ? nd.c : nd.skip;
                } else if (inside && !(nd.a & BVH_LEAF)) {
                    const bool right_first = get(ray.d, nd.a) < T(0);
                    stack[sp * 64u] = right_first ? nd.c : nd.b;           // the farther child waits
                    sp++;
                    node = right_first ? nd.b : nd.c;
                } else {
                    if (inside) { have_leaf = true; leaf_a = nd.a; leaf_b = nd.b; leaf_node = nd.c; }
                    if (sp == 0) node = DONE;
                    else { sp--; node = stack[sp * 64u]; }
                }
            };
            if (want_box) box_step();
            // further box steps under the same vote (the vote is ~30 scalar instructions and two ballots: *measured* with two steps
            // per vote random spheres +8 %, final scene +4 %); a lane that reached a leaf or the end sits them out
#pragma unroll
            for (int k = 1; k < BOX_STEPS; k++) if (node != DONE && !have_leaf) box_step();
        }
        if (have_leaf) {
            T t; uint32_t prim;
            if (range_hit<T, FEATS>(P, (leaf_a >> 28) & 7u, leaf_a & 0x0FFFFFFFu, leaf_b, ray, t_min, closest, t, prim) &&
                bvh_accept(near_first, t, closest, leaf_node, best_leaf)) { closest = t; prim_out = prim; any = true; best_leaf = leaf_node; }
            have_leaf = false;
        }
        if (__ballot(node != DONE) == 0ull) break;          // every lane of the wave is finished
    }
    t_out = closest;
    return any;
}

template <typename T, uint32_t FEATS, int NEST>
DEV bool bvh_hit(const KParams<T>& P, uint32_t root, const RayT<T>& ray, T t_min, T t_max, T& t_out, uint32_t& prim_out, uint32_t* stack, Rng& rng, uint32_t& sub_out) {
    if constexpr (Filt<T, FEATS>::on) return bvh_hit_filt<T, FEATS, (FEATS & F_SPEC) != 0u, NEST>(P, root, ray, t_min, t_max, t_out, prim_out, rng, sub_out);
    return bvh_hit_ww<T, FEATS>(P, root, ray, t_min, t_max, t_out, prim_out, stack);
}

// ------------------------------------------------------------------ wrapper chain (translate.rs, rotate.rs, hit.rs FlipNormal)
template <typename T> DEV void rot_fwd(uint32_t axis, T sn, T cs, V3<T>& v) {       // rotate.rs:82-86
    T a, b;
    if (axis == 1u) { a = v.x; b = v.z; v.x = cs * a - sn * b; v.z = sn * a + cs * b; }
    else if (axis == 0u) { a = v.y; b = v.z; v.y = cs * a - sn * b; v.z = sn * a + cs * b; }
    else { a = v.x; b = v.y; v.x = cs * a - sn * b; v.y = sn * a + cs * b; }
}
template <typename T> DEV void rot_back(uint32_t axis, T sn, T cs, V3<T>& v) {      // rotate.rs:95-99
    T a, b;
    if (axis == 1u) { a = v.x; b = v.z; v.x = cs * a + sn * b; v.z = (-sn) * a + cs * b; }
    else if (axis == 0u) { a = v.y; b = v.z; v.y = cs * a + sn * b; v.z = (-sn) * a + cs * b; }
    else { a = v.x; b = v.y; v.x = cs * a + sn * b; v.y = (-sn) * a + cs * b; }
}
template <typename T> DEV void op_fwd(const DOp<T>& op, RayT<T>& r) {
    if (op.kind == OP_TRANSLATE) { r.o = r.o - mk<T>(op.x, op.y, op.z); }           // translate.rs:23
    else if (op.kind == OP_ROTATE) { rot_fwd(op.axis, op.x, op.y, r.o); rot_fwd(op.axis, op.x, op.y, r.d); }
}

// ------------------------------------------------------------------ world.hit: closest hit over the top-level list
// (sub: the sub-object a hit inside a BVH belongs to when the leaf was a G_OBJ one — F_NESTED kernels; NO_SUB otherwise)
template <typename T, uint32_t FEATS, int NEST = 0>
DEV bool geom_hit(const KParams<T>& P, const DObject& ob, const RayT<T>& r, T t_min, T t_max, T& t, uint32_t& prim, uint32_t* stack, Rng& rng, uint32_t& sub, bool rooms = false) {
    if ((FEATS & F_BVH) && ob.geom_kind == G_BVH) {
        if constexpr (NEST <= RT_MAX_NEST) return bvh_hit<T, FEATS, NEST>(P, ob.geom_first, r, t_min, t_max, t, prim, stack, rng, sub);
        else return false;              // (the flattener refuses BVHs nested deeper)
    }
    // (a room — is_cube & 2 — exists in list scenes only and has no wrappers: one call site knows about it, the others' code is unchanged)
    return range_hit<T, FEATS>(P, ob.geom_kind, ob.geom_first, ob.geom_count, r, t_min, t_max, t, prim, ob.is_cube != 0u, rooms && (ob.is_cube & 2u) ? ob.is_cube : 0u);
}
// One object under HittableList::hit in an F_NESTED kernel — a top-level object (NEST 0) or a sub-object of a BVH leaf (NEST >= 1; `ray` is
// then the ray as the enclosing BVH received it: the first n_outer ops of the object's chain are already in it).  Same arithmetic as
// object_hit's general form; additionally a ConstantMedium may stand anywhere in the chain (medium.rs:27-61 measures the free flight with
// the ray IT receives: the ops outside it applied, those between it and its boundary not) and a hit inside a BVH names its sub-object.
template <typename T, uint32_t FEATS, int NEST>
DEV void object_hit_nested(const KParams<T>& P, uint32_t oi, const DObject& ob, const RayT<T>& ray, T t_min, Rng& rng, T& closest, HitId& id, bool& any) {
    const uint32_t n_outer = ob.nest & 0xFFu, med_at = (ob.nest >> 8) & 0xFFu;
    RayT<T> r = ray;
    uint32_t sub = NO_SUB;
    if (!(FEATS & F_MEDIUM) || ob.medium < 0) {
        for (uint32_t k = n_outer; k < ob.n_ops; k++) op_fwd(ld_op(P.ops + ob.first_op + k), r);
        T t; uint32_t prim;
        if (geom_hit<T, FEATS, NEST>(P, ob, r, t_min, closest, t, prim, nullptr, rng, sub)) { closest = t; id.obj = sub != NO_SUB ? sub : oi; id.prim = prim; any = true; }
        return;
    }
    for (uint32_t k = n_outer; k < med_at; k++) op_fwd(ld_op(P.ops + ob.first_op + k), r);        // the ray ConstantMedium::hit receives
    const T len = length(r.d);                                                                     // medium.rs:40
    for (uint32_t k = med_at; k < ob.n_ops; k++) op_fwd(ld_op(P.ops + ob.first_op + k), r);        // ... and its boundary
    T t1, t2; uint32_t p1, p2;
    if (geom_hit<T, FEATS, NEST>(P, ob, r, -Lim<T>::max(), Lim<T>::max(), t1, p1, nullptr, rng, sub)) {          // medium.rs:29
        if (geom_hit<T, FEATS, NEST>(P, ob, r, t1 + T(0.0001), Lim<T>::max(), t2, p2, nullptr, rng, sub)) {      // medium.rs:30
            if (t1 < t_min) t1 = t_min;
            if (t2 > closest) t2 = closest;
            if (t1 < t2) {
                T distance_inside_boundary = (t2 - t1) * len;
                T hit_distance = cl(&P.media[ob.medium].neg_inv_density) * m_log(rng_u01(rng, T(0)));
                if (hit_distance < distance_inside_boundary) {
                    closest = t1 + hit_distance / len;
                    id.obj = oi; id.prim = PRIM_MEDIUM; any = true;
                }
            }
        }
    }
}

// One object of the top-level list under HittableList::hit (hit.rs:59-71): offered [t_min, closest], a hit replaces the
// running (closest, id).  ConstantMedium objects draw from the path's RNG (medium.rs:44).
template <typename T, uint32_t FEATS, int NEST = 0>
DEV void object_hit(const KParams<T>& P, uint32_t oi, const DObject& ob, const RayT<T>& ray, T t_min, Rng& rng, T& closest, HitId& id, bool& any, uint32_t* stack) {
    if constexpr (Nested<FEATS>::on) { object_hit_nested<T, FEATS, NEST>(P, oi, ob, ray, t_min, rng, closest, id, any); return; }
    uint32_t no_sub = NO_SUB;          // (only F_NESTED kernels have sub-objects)
    if (FEATS == 0u && ob.n_ops == 2u) {        // Translate(RotateY(..)), the reference's instance idiom (main.rs:300-309): straight-line code
        const DOp<T> o0 = ld_op(P.ops + ob.first_op), o1 = ld_op(P.ops + ob.first_op + 1u);
        if (o0.kind == OP_TRANSLATE && o1.kind == OP_ROTATE && o1.axis == 1u) {
            RayT<T> r;
            const V3<T> q = ray.o - mk<T>(o0.x, o0.y, o0.z);                          // translate.rs:23
            r.o.x = o1.y * q.x - o1.x * q.z; r.o.y = q.y; r.o.z = o1.x * q.x + o1.y * q.z;           // rotate.rs:82-86
            r.d.x = o1.y * ray.d.x - o1.x * ray.d.z; r.d.y = ray.d.y; r.d.z = o1.x * ray.d.x + o1.y * ray.d.z;
            r.tm = ray.tm;
            T t; uint32_t prim;
            if (geom_hit<T, FEATS>(P, ob, r, t_min, closest, t, prim, stack, rng, no_sub)) { closest = t; id.obj = oi; id.prim = prim; any = true; }
            return;
        }
    }
    if (FEATS == 0u && (ob.n_ops == 0u || (ob.nest & 0x10000u) != 0u)) {        // no wrapper, or FlipNormals only (hit.rs:113-119: they change the record, not the ray; rt_flatten.cpp marks such chains):
                                                                                // test the path's own ray (no copy of it into the registers the wrappers rewrite).  Round 6: the Cornell light, +0.9 %
        T t; uint32_t prim;
        if (geom_hit<T, FEATS>(P, ob, ray, t_min, closest, t, prim, stack, rng, no_sub, true)) {
            // A room stands where the LAST of its walls stood in the list (rt_flatten.cpp: form_room), so an object that stood between two
            // walls is searched before all of them.  HittableList::hit's result is the hit with the smallest t, the LATER item on an exact
            // tie (hit.rs:62-68: `t <= closest` accepts); every t is compared as before, only a tie between a wall and an object that came
            // AFTER it in the list must still go to that object: first_op holds, five bits per wall, the index of the first such object.
            // (All of which is about plane distances that are numbers: a wave with a ray that could make a NaN one searches the list as
            // the reference has it instead — world_hit.)
            bool keep = true;
            if (ob.is_cube & 2u) keep = !(any && t == closest && id.obj >= ((ob.first_op >> (5u * ((prim & 0x0FFFFFFFu) - ob.geom_first))) & 31u));
            if (keep) { closest = t; id.obj = oi; id.prim = prim; any = true; }
        }
        return;
    }
    RayT<T> r = ray;
    for (uint32_t k = 0; k < ob.n_ops; k++) op_fwd(ld_op(P.ops + ob.first_op + k), r);
    if (!(FEATS & F_MEDIUM) || ob.medium < 0) {
        T t; uint32_t prim;
        if constexpr (RoomSite<T, FEATS>::on) {
            // (the mesh kernels' one site of the Cube fast path: a room or a bare Cube in the world list; an unclear wave takes the plain tests below)
            if (ob.geom_kind == G_RECT && ob.is_cube != 0u) {
                bool a = false;
                t = closest;
                if (cube_hit<1>(P, ob.geom_first, ob.geom_count, (ob.is_cube & 2u) ? ob.is_cube : 0u, r, t_min, closest, t, prim, a)) {
                    if (a) { closest = t; id.obj = oi; id.prim = prim; any = true; }
                    return;
                }
            }
        }
        if (geom_hit<T, FEATS>(P, ob, r, t_min, closest, t, prim, stack, rng, no_sub)) { closest = t; id.obj = oi; id.prim = prim; any = true; }
    } else {
        // ConstantMedium::hit, medium.rs:27-61
        T t1, t2; uint32_t p1, p2;
        if (geom_hit<T, FEATS>(P, ob, r, -Lim<T>::max(), Lim<T>::max(), t1, p1, stack, rng, no_sub)) {
            if (geom_hit<T, FEATS>(P, ob, r, t1 + T(0.0001), Lim<T>::max(), t2, p2, stack, rng, no_sub)) {
                if (t1 < t_min) t1 = t_min;
                if (t2 > closest) t2 = closest;
                if (t1 < t2) {
                    T len = length(ray.d);
                    T distance_inside_boundary = (t2 - t1) * len;
                    T hit_distance = cl(&P.media[ob.medium].neg_inv_density) * m_log(rng_u01(rng, T(0)));
                    if (hit_distance < distance_inside_boundary) {
                        closest = t1 + hit_distance / len;
                        id.obj = oi; id.prim = PRIM_MEDIUM; any = true;
                    }
                }
            }
        }
    }
}

// world.hit of a LIST scene (the lean kernels, FEATS 0).  With a room in the list (rt_flatten.cpp form_room) the list [0, n_objects) searches
// things that stood between two walls before walls they stood behind — the same search whenever every plane distance is a number.  A NaN one
// (0 / 0: a zero direction component on a ray that starts ON a plane; or a ray that is not finite) passes `t < t_min || t > t_max` and
// then lets every later item pass `t <= closest`: HittableList::hit's result then depends on the ORDER of the items (hit.rs:59-71).
// Everything between the first and the last wall sees the path's own ray (the flattener's condition), so a plane distance (k - o_k) / d_k
// can only be NaN when a direction component is a zero or the ray is not finite.  So, BEFORE the search: a wave in which some lane's
// direction has a zero, denormal or non-finite component, or a non-finite origin, searches objects[n_objects, n_objects + n_objects_alt)
// — the list as the reference has it — instead.  Six v_cmp_class_f64 per bounce (*measured*, profiles/r06_room_ab.log: the room form
// +2.3 % without this, +1.6 % with it; testing the origins only in waves that have met such a ray before — a wave flag kept scalar by
// readfirstlane — is no faster and needs an argument about origins; a re-search after the room has found such a lane costs all of the
// gain: the outer loop spills thirteen more scalar registers).  Objects before the first and after the last wall are searched in the
// same order by both lists, so whatever THEY return — NaN hits of rotated boxes included — is the reference's either way.
template <typename T> DEV bool risky_component(T x, bool origin);
template <> DEV bool risky_component<double>(double x, bool origin) { return origin ? __builtin_amdgcn_class(x, 0x207) : __builtin_amdgcn_class(x, 0x2F7); }     // NaN, inf (| zero, denormal)
template <> DEV bool risky_component<float>(float x, bool origin) { return origin ? __builtin_amdgcn_classf(x, 0x207) : __builtin_amdgcn_classf(x, 0x2F7); }
template <typename T>
DEV bool world_hit_list(const KParams<T>& P, const RayT<T>& ray, T t_min, Rng& rng, T& t_hit, HitId& id) {
    T closest = Lim<T>::inf();
    bool any = false;
    uint32_t oi = 0u, oi_last = P.n_objects;
    if (P.n_objects_alt != 0u) {
        const bool risk = risky_component(ray.d.x, false) || risky_component(ray.d.y, false) || risky_component(ray.d.z, false) ||
                          risky_component(ray.o.x, true) || risky_component(ray.o.y, true) || risky_component(ray.o.z, true);
        if (__ballot(risk) != 0ull) { oi = oi_last; oi_last += P.n_objects_alt; }
    }
    for (; oi < oi_last; oi++) object_hit<T, 0u>(P, oi, ld_obj(P.objects + oi), ray, t_min, rng, closest, id, any, nullptr);      // wave-uniform: scalar loads
    t_hit = closest;
    return any;
}

template <typename T, uint32_t FEATS>
DEV bool world_hit(const KParams<T>& P, const RayT<T>& ray, T t_min, Rng& rng, T& t_hit, HitId& id, uint32_t* stack) {
    T closest = Lim<T>::inf();
    bool any = false;
#ifdef RT_DIAG_OBJ      // diagnostic build only: where world.hit's time goes by object class, and how many lanes enter each BVH object:
                        // stats[3] wave-cycles in BVH objects without wrappers, [4] in BVH objects behind wrappers, [5] in all other objects,
                        // [6] / [7] lanes whose ray passes the root box of the former / latter, [8] calls, [12] lanes in the calls
    unsigned long long dg[3] = {0, 0, 0}, dl[2] = {0, 0};
#endif
    for (uint32_t oi = 0; oi < P.n_objects; oi++) {          // wave-uniform: scalar loads
        const DObject ob = ld_obj(P.objects + oi);
#ifdef RT_DIAG_OBJ
        __builtin_amdgcn_sched_barrier(0); const unsigned long long t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0);
        if ((FEATS & F_BVH) && ob.geom_kind == G_BVH) {
            RayT<T> r = ray;
            for (uint32_t k = 0; k < ob.n_ops; k++) op_fwd(ld_op(P.ops + ob.first_op + k), r);
            const V3<T> inv = mk<T>(T(1.0) / r.d.x, T(1.0) / r.d.y, T(1.0) / r.d.z);
            dl[ob.n_ops ? 1 : 0] += (unsigned long long)__popcll(__ballot(box_inside_exact(ld_node_at(P.bvh, ob.geom_first), r.o, inv, t_min, closest)));
        }
#endif
        object_hit<T, FEATS>(P, oi, ob, ray, t_min, rng, closest, id, any, stack);
#ifdef RT_DIAG_OBJ
        __builtin_amdgcn_sched_barrier(0); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0);
        dg[ob.geom_kind == G_BVH ? (ob.n_ops ? 1 : 0) : 2] += t1 - t0;
#endif
    }
#ifdef RT_DIAG_OBJ
    {
        const unsigned long long ex = __ballot(true);
        if (P.stats && __builtin_amdgcn_mbcnt_hi((uint32_t)(ex >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ex, 0u)) == 0u) {
            atomicAdd(&P.stats[3], dg[0]); atomicAdd(&P.stats[4], dg[1]); atomicAdd(&P.stats[5], dg[2]);
            atomicAdd(&P.stats[6], dl[0]); atomicAdd(&P.stats[7], dl[1]); atomicAdd(&P.stats[8], 1ull); atomicAdd(&P.stats[12], (unsigned long long)__popcll(ex));
        }
    }
#endif
    t_hit = closest;
    return any;
}

// get_sphere_uv, sphere.rs:11-25
template <typename T> DEV void sphere_uv(V3<T> p, T& u, T& v) {
    T phi = m_atan2(-p.z, p.x) + PI_T;
    T theta = m_acos(-p.y);
    u = phi / (T(2.0) * PI_T);
    v = theta / PI_T;
}

// (u, v) are read by ImageTexture alone; the flag rides on the material's kind word (rt_ir.h)
template <typename T> DEV bool mat_reads_uv(const KParams<T>& P, uint32_t mat) { return (cl(&P.materials[mat].kind) & MAT_NEEDS_UV) != 0u; }

// Rebuild the full HitRecord of the winning (object, primitive, t): the same arithmetic the reference runs
// eagerly inside every `hit`, run once.  Per-lane gathers: lanes may hold different objects.
template <typename T, uint32_t FEATS>
DEV void finalize_hit(const KParams<T>& P, const RayT<T>& ray, T t, HitId id, bool want_uv, Rec<T>& rec) {
    const DObject ob = ld_obj(P.objects + id.obj);
    rec.t = t; rec.u = T(0); rec.v = T(0);
    if ((FEATS & F_MEDIUM) && id.prim == PRIM_MEDIUM) {                               // medium.rs:45-55
        if constexpr (Nested<FEATS>::on) {
            // the medium may stand under wrappers (med_at of the chain's ops): its record is made with the ray it received and then goes
            // back up through them like any other record (translate.rs:24-28, rotate.rs:88-104, hit.rs:113-119)
            const uint32_t med_at = (ob.nest >> 8) & 0xFFu;
            RayT<T> rm = ray;
            for (uint32_t k = 0; k < med_at; k++) op_fwd(ld_op(P.ops + ob.first_op + k), rm);
            rec.p = ray_at(rm, t);
            rec.n = mk<T>(T(1.0), T(0), T(0));
            rec.front = false;
            rec.mat = cl(&P.media[ob.medium].mat);
            for (int k = (int)med_at - 1; k >= 0; k--) {
                const DOp<T> op = ld_op(P.ops + ob.first_op + (uint32_t)k);
                if (op.kind == OP_TRANSLATE) rec.p = rec.p + mk<T>(op.x, op.y, op.z);
                else if (op.kind == OP_ROTATE) {
                    RayT<T> rr = ray;                                                     // the ray this Rotate handed to its child
                    for (int q = 0; q <= k; q++) op_fwd(ld_op(P.ops + ob.first_op + (uint32_t)q), rr);
                    rot_back(op.axis, op.x, op.y, rec.p);
                    V3<T> nw = rec.n;
                    rot_back(op.axis, op.x, op.y, nw);
                    set_face_normal(rec, rr.d, nw);
                } else rec.front = !rec.front;
            }
            return;
        }
        rec.p = ray_at(ray, t);
        rec.n = mk<T>(T(1.0), T(0), T(0));
        rec.front = false;
        rec.mat = cl(&P.media[ob.medium].mat);
        return;
    }
    RayT<T> r = ray;
    bool fused = false;
    DOp<T> f0, f1;
    if (FEATS == 0u && ob.n_ops == 2u) {
        f0 = ld_op(P.ops + ob.first_op); f1 = ld_op(P.ops + ob.first_op + 1u);
        fused = f0.kind == OP_TRANSLATE && f1.kind == OP_ROTATE && f1.axis == 1u;
    }
    if (fused) {
        const V3<T> q = ray.o - mk<T>(f0.x, f0.y, f0.z);
        r.o.x = f1.y * q.x - f1.x * q.z; r.o.y = q.y; r.o.z = f1.x * q.x + f1.y * q.z;
        r.d.x = f1.y * ray.d.x - f1.x * ray.d.z; r.d.z = f1.x * ray.d.x + f1.y * ray.d.z;
    } else
    for (uint32_t k = 0; k < ob.n_ops; k++) op_fwd(ld_op(P.ops + ob.first_op + k), r);
    const uint32_t kind = id.prim >> 28, idx = id.prim & 0x0FFFFFFFu;
    if (kind == G_RECT) {                                                             // rect.rs:61-79
        const DRect<T> rc = ld_rect(P.rects + idx);
        uint32_t ki, ai, bi; plane_axes(rc.plane, ki, ai, bi);
        if ((FEATS & F_TEXTURES) && want_uv && mat_reads_uv(P, rc.mat)) {
            T a = get(r.o, ai) + t * get(r.d, ai);
            T b = get(r.o, bi) + t * get(r.d, bi);
            rec.u = (a - rc.a0) / (rc.a1 - rc.a0);
            rec.v = (b - rc.b0) / (rc.b1 - rc.b0);
        }
        rec.p = ray_at(r, t);
        V3<T> normal = mk<T>(ki == 0u ? T(1.0) : T(0), ki == 1u ? T(1.0) : T(0), ki == 2u ? T(1.0) : T(0));
        set_face_normal(rec, r.d, normal);
        rec.mat = rc.mat;
    } else if ((FEATS & F_SPHERES) && (kind == G_SPHERE || kind == G_MSPHERE)) {      // sphere.rs:76-94, :170-188
        V3<T> center; T radius;
        if (kind == G_SPHERE) { const DSphere<T> s = ld_sphere(P.spheres + idx); center = ld3(s.c); radius = s.r; rec.mat = s.mat; }
        else { const DMSphere<T> s = ld_msphere(P.mspheres + idx); center = msphere_center(s, r.tm); radius = s.r; rec.mat = s.mat; }
        rec.p = ray_at(r, t);
        V3<T> outward = (rec.p - center) / radius;
        set_face_normal(rec, r.d, outward);
        if ((FEATS & F_TEXTURES) && want_uv && mat_reads_uv(P, rec.mat)) sphere_uv(outward, rec.u, rec.v);
    } else if ((FEATS & F_TRIS) && kind == G_TRI) {                                   // tri.rs:42-56
        const DTri<T> tr = ld_tri(P.tris + idx);
        V3<T> e1 = ld3(tr.e1), e2 = ld3(tr.e2);
        if ((FEATS & F_TEXTURES) && want_uv && mat_reads_uv(P, tr.mat)) {
            V3<T> s = r.o - ld3(tr.v0);
            V3<T> s1 = cross(r.d, e2);
            V3<T> s2 = cross(s, e1);
            T s1_e1 = dot(s1, e1);
            rec.u = dot(s1, s) / s1_e1;
            rec.v = dot(s2, r.d) / s1_e1;
        }
        rec.p = ray_at(r, t);
        V3<T> normal = normalized(cross(e1, e2));
        set_face_normal(rec, r.d, normal);
        rec.mat = tr.mat;
    }
    // unwind the wrapper chain innermost -> outermost
    if (fused) {
        rot_back(1u, f1.x, f1.y, rec.p);                                              // rotate.rs:90-104
        V3<T> nw = rec.n;
        rot_back(1u, f1.x, f1.y, nw);
        set_face_normal(rec, r.d, nw);
        rec.p = rec.p + mk<T>(f0.x, f0.y, f0.z);                                      // translate.rs:26
        return;
    }
    for (int k = (int)ob.n_ops - 1; k >= 0; k--) {
        const DOp<T> op = ld_op(P.ops + ob.first_op + (uint32_t)k);
        if (op.kind == OP_TRANSLATE) {
            rec.p = rec.p + mk<T>(op.x, op.y, op.z);                                  // translate.rs:26
        } else if (op.kind == OP_ROTATE) {                                            // rotate.rs:90-104
            RayT<T> rr = r;                                                           // the ray this Rotate handed to its child:
            if (k != (int)ob.n_ops - 1) {                                             // = r when the Rotate is innermost (usual)
                rr = ray;
                for (int q = 0; q <= k; q++) op_fwd(ld_op(P.ops + ob.first_op + (uint32_t)q), rr);
            }
            rot_back(op.axis, op.x, op.y, rec.p);
            V3<T> nw = rec.n;
            rot_back(op.axis, op.x, op.y, nw);
            set_face_normal(rec, rr.d, nw);                                           // object-space ray vs world-space normal, as the reference does
        } else {
            rec.front = !rec.front;                                                   // FlipNormal, hit.rs:113-119
        }
    }
}

// ------------------------------------------------------------------ textures (texture.rs, perlin.rs)
DEV uint32_t sat_index(double x) { return x > 0.0 ? (x >= 18446744073709551616.0 ? 0xFFFFFFFFu : (uint32_t)((unsigned long long)x & 0xFFFFFFFFull)) : 0u; }   // low 32 bits of `x as usize`
DEV uint32_t sat_index(float x) { return x > 0.0f ? (x >= 18446744073709551616.0f ? 0xFFFFFFFFu : (uint32_t)((unsigned long long)x & 0xFFFFFFFFull)) : 0u; }
DEV unsigned long long sat_u64(double x) { return x > 0.0 ? (x >= 18446744073709551616.0 ? ~0ull : (unsigned long long)x) : 0ull; }
DEV unsigned long long sat_u64(float x) { return x > 0.0f ? (x >= 18446744073709551616.0f ? ~0ull : (unsigned long long)x) : 0ull; }

template <typename T> DEV T perlin_noise(const DPerlin<T>& pn, V3<T> p, T scale) {    // perlin.rs:77-109 + perlin_interp :39-56
    T fx = m_floor(scale * p.x), fy = m_floor(scale * p.y), fz = m_floor(scale * p.z);
    T u = scale * p.x - fx, v = scale * p.y - fy, w = scale * p.z - fz;
    u = u * u * (T(3.0) - T(2.0) * u);
    v = v * v * (T(3.0) - T(2.0) * v);
    w = w * w * (T(3.0) - T(2.0) * w);
    uint32_t i = sat_index(fx), j = sat_index(fy), k = sat_index(fz);
    T uu = u * u * (T(3.0) - T(2.0) * u);
    T vv = v * v * (T(3.0) - T(2.0) * v);
    T ww = w * w * (T(3.0) - T(2.0) * w);
    T accum = T(0);
#pragma unroll
    for (uint32_t di = 0; di < 2; di++)
#pragma unroll
        for (uint32_t dj = 0; dj < 2; dj++)
#pragma unroll
            for (uint32_t dk = 0; dk < 2; dk++) {
                uint32_t h = (uint32_t)cl(&pn.perm_x[(i + di) & 255u]) ^ (uint32_t)cl(&pn.perm_y[(j + dj) & 255u]) ^ (uint32_t)cl(&pn.perm_z[(k + dk) & 255u]);
                V3<T> c = cl3(&pn.rd_vec[h * 3u]);
                V3<T> weight = mk<T>(u - T(di), v - T(dj), w - T(dk));
                T fu = di ? uu : (T(1.0) - uu), fv = dj ? vv : (T(1.0) - vv), fw = dk ? ww : (T(1.0) - ww);
                accum += fu * fv * fw * dot(c, weight);
            }
    return accum;
}
template <typename T> DEV T perlin_turb(const DPerlin<T>& pn, V3<T> p, T scale) {     // perlin.rs:111-120, depth 7 (texture.rs:77)
    T accum = T(0);
    V3<T> temp_p = p;
    T weight = T(1.0);
    for (int d = 0; d < 7; d++) {
        accum += weight * perlin_noise(pn, temp_p, scale);
        weight *= T(0.5);
        temp_p = temp_p * T(2.0);
    }
    return m_abs(accum);
}
template <typename T> DEV T clamp_(T x, T lo, T hi) { return x < lo ? lo : (x > hi ? hi : x); }   // f64::clamp (NaN stays NaN)

template <typename T, uint32_t FEATS>
DEV V3<T> tex_eval(const KParams<T>& P, uint32_t id, T u, T v, V3<T> p) {             // Texture::mapping, texture.rs:5-7
    for (;;) {
        const DTexture<T> tx = ld_tex(P.textures + id);
        if (!(FEATS & F_TEXTURES) || tx.kind == T_CONSTANT) return ld3(tx.color);     // texture.rs:23-27
        if (tx.kind == T_CHECK) {                                                     // texture.rs:45-54
            T sines = m_sin(T(10.0) * p.x) * m_sin(T(10.0) * p.y) * m_sin(T(10.0) * p.z);
            id = (sines < T(0)) ? tx.a : tx.b;
            continue;
        }
        if (tx.kind == T_NOISE) {                                                     // texture.rs:71-79
            T s = T(0.5) * (T(1.0) + m_sin(tx.scale * p.z + T(10.0) * perlin_turb(P.perlins[tx.a], p, tx.scale)));
            return mk<T>(T(1.0) * s, T(1.0) * s, T(1.0) * s);
        }
        // ImageTexture, texture.rs:99-120
        unsigned long long w = tx.b, h = tx.c;
        unsigned long long i = sat_u64(clamp_(u, T(0), T(1.0)) * T(w));
        unsigned long long j = sat_u64(clamp_(T(1.0) - v, T(0), T(1.0)) * T(h));
        if (i > w - 1) i = w - 1;
        if (j > h - 1) j = h - 1;
        const uint8_t* px = P.image_bytes + tx.a + 3ull * i + 3ull * w * j;
        return mk<T>(T(cl(px)) / T(255.0), T(cl(px + 1)) / T(255.0), T(cl(px + 2)) / T(255.0));
    }
}

// ------------------------------------------------------------------ ONB / PDFs / lights (onb.rs, pdf.rs, hit.rs:90-96)
template <typename T> struct Onb { V3<T> u, v, w; };
template <typename T> DEV Onb<T> onb_from_w(V3<T> n) {                               // onb.rs:8-20
    Onb<T> o;
    o.w = normalized(n);
    V3<T> a = (m_abs(o.w.x) > T(0.9)) ? mk<T>(T(0), T(1.0), T(0)) : mk<T>(T(1.0), T(0), T(0));
    o.v = normalized(cross(o.w, a));
    o.u = cross(o.w, o.v);
    return o;
}
template <typename T> DEV V3<T> onb_local(const Onb<T>& o, V3<T> a) { return a.x * o.u + a.y * o.v + a.z * o.w; }   // onb.rs:34-36

template <typename T> DEV V3<T> random_cosine_direction(Rng& rng) {                   // pdf.rs:8-18
    T r1 = rng_u01(rng, T(0));
    T r2 = rng_u01(rng, T(0));
    T z = rsqrt_(T(1.0) - r2);
    T phi = T(2.0) * PI_T * r1;
    T sn, cs; sincos_0_2pi(phi, sn, cs);
    T x = cs * rsqrt_(r2);
    T y = sn * rsqrt_(r2);
    return mk<T>(x, y, z);
}
template <typename T> DEV V3<T> random_in_unit_sphere(Rng& rng) {                     // vec.rs:78-85
    for (;;) {
        T a = rng_range(rng, T(-1.0), T(1.0)), b = rng_range(rng, T(-1.0), T(1.0)), c = rng_range(rng, T(-1.0), T(1.0));
        V3<T> v = mk<T>(a, b, c);
        // vec.rs:81 tests `v.length() < 1.0`; for a correctly rounded sqrt, sqrt(x) < 1 <=> x < 1, so the sqrt is skipped
        if (dot(v, v) < T(1.0)) return v;
    }
}

template <typename T, uint32_t FEATS> DEV T light_pdf_value(const KParams<T>& P, DLight L, V3<T> o, V3<T> v) {
    RayT<T> r; r.o = o; r.d = v; r.tm = T(0);
    if (L.kind == L_RECT) {                                                           // rect.rs:91-101
        const DRect<T> rc = ld_rect(P.rects + L.index);
        T t;
        if (rect_test(rc, r, T(0.001), Lim<T>::inf(), t)) {
            uint32_t ki, ai, bi; plane_axes(rc.plane, ki, ai, bi);
            V3<T> axis = mk<T>(ki == 0u ? T(1.0) : T(0), ki == 1u ? T(1.0) : T(0), ki == 2u ? T(1.0) : T(0));
            V3<T> n = (dot(v, axis) < T(0)) ? axis : T(-1.0) * axis;
            T area = (rc.a1 - rc.a0) * (rc.b1 - rc.b0);
            T distance_squared = (t * t) * sq_of_len(v);
            T cosine = m_abs(dot(v, n)) / length(v);
            return (cosine != T(0)) ? distance_squared / (cosine * area) : T(0);
        }
        return T(0);
    }
    if ((FEATS & F_SPHERES) && L.kind == L_SPHERE) {                                  // sphere.rs:104-112
        const DSphere<T> s = ld_sphere(P.spheres + L.index);
        T t;
        if (sphere_test(ld3(s.c), s.r, r, T(0.001), Lim<T>::max(), t)) {
            T cos_theta_max = rsqrt_(T(1.0) - s.r * s.r / sq_of_len(ld3(s.c) - o));
            T solid_angle = T(2.0) * PI_T * (T(1.0) - cos_theta_max);
            return T(1.0) / solid_angle;
        }
        return T(0);
    }
    return T(0);                                                                      // Hittable::pdf_value default, hit.rs:29
}
template <typename T, uint32_t FEATS> DEV V3<T> light_random(const KParams<T>& P, DLight L, V3<T> o, Rng& rng) {
    if (L.kind == L_RECT) {                                                           // rect.rs:103-111
        const DRect<T> rc = ld_rect(P.rects + L.index);
        uint32_t ki, ai, bi; plane_axes(rc.plane, ki, ai, bi);
        T ra = rng_range(rng, rc.a0, rc.a1);
        T rb = rng_range(rng, rc.b0, rc.b1);
        V3<T> pt;
        pt.x = ki == 0u ? rc.k : (ai == 0u ? ra : rb);
        pt.y = ki == 1u ? rc.k : (ai == 1u ? ra : rb);
        pt.z = ki == 2u ? rc.k : rb;
        return pt - o;
    }
    if ((FEATS & F_SPHERES) && L.kind == L_SPHERE) {                                  // sphere.rs:114-119, :27-36
        const DSphere<T> s = ld_sphere(P.spheres + L.index);
        V3<T> direction = ld3(s.c) - o;
        T distance_squared = sq_of_len(direction);
        Onb<T> uvw = onb_from_w(direction);
        T r1 = rng_u01(rng, T(0));
        T r2 = rng_u01(rng, T(0));
        T z = T(1.0) + r2 * (rsqrt_(T(1.0) - s.r * s.r / distance_squared) - T(1.0));
        T phi = T(2.0) * PI_T * r1;
        T sn, cs; sincos_0_2pi(phi, sn, cs);
        T x = cs * rsqrt_(T(1.0) - z * z);
        T y = sn * rsqrt_(T(1.0) - z * z);
        return onb_local(uvw, mk<T>(x, y, z));
    }
    return mk<T>(T(1.0), T(0), T(0));                                                 // Hittable::random default, hit.rs:30
}

// ------------------------------------------------------------------ principled ("Disney") material: mat.rs:10-52,133-195; pdf.rs:20-60,97-130,151-160
template <typename T> DEV T mixf(T a, T b, T t) { return a * (T(1.0) - t) + b * t; }                                                       // mat.rs:50-52
template <typename T> DEV V3<T> mixv(V3<T> a, V3<T> b, T t) { return mk<T>(a.x * (T(1.0) - t) + b.x * t, a.y * (T(1.0) - t) + b.y * t, a.z * (T(1.0) - t) + b.z * t); }   // vec.rs:60-68
template <typename T> DEV T schlick_fresnel(T u) { T m = clamp_(T(1.0) - u, T(0), T(1.0)); T m2 = m * m; return m2 * m2 * m; }            // mat.rs:10-14
template <typename T> DEV T GTR_1(T n_dot_h, T a) {                                                                                        // mat.rs:16-24
    if (a >= T(1.0)) return T(1.0) / PI_T;
    T a2 = a * a;
    T t = T(1.0) + (a2 - T(1.0)) * n_dot_h * n_dot_h;
    return (a2 - T(1.0)) / (PI_T * m_log2(a2) * t);
}
template <typename T> DEV T GTR_2_aniso(T n_dot_h, T h_dot_x, T h_dot_y, T ax, T ay) {                                                     // mat.rs:32-34
    T p = h_dot_x / ax, q = h_dot_y / ay;
    T s = p * p + q * q + n_dot_h * n_dot_h;
    return T(1.0) / (PI_T * ax * ay * (s * s));
}
template <typename T> DEV T smithG_GGX(T n_dot_v, T alphaG) { T a = alphaG * alphaG, b = n_dot_v * n_dot_v; return T(1.0) / (n_dot_v + rsqrt_(a + b - a * b)); }   // mat.rs:36-40
template <typename T> DEV T smithG_GGX_aniso(T n_dot_v, T v_dot_x, T v_dot_y, T ax, T ay) {                                                // mat.rs:42-44
    T p = v_dot_x * ax, q = v_dot_y * ay;
    return T(1.0) / (n_dot_v + rsqrt_(p * p + q * q + n_dot_v * n_dot_v));
}
template <typename T> DEV V3<T> reflect_(V3<T> v, V3<T> n) { return v + ((-dot(v, n)) * T(2.0) * n); }                                      // vec.rs:112-114

// Material::brdf for PBR, mat.rs:133-195
template <typename T> DEV V3<T> pbr_brdf(const DPbr<T>& m, V3<T> base, V3<T> r_in_dir, V3<T> r_out_dir, V3<T> normal) {
    V3<T> l = normalized(r_in_dir) * T(-1.0);
    V3<T> v = normalized(r_out_dir);
    Onb<T> onb = onb_from_w(normal);
    V3<T> n = onb.w, x = onb.u, y = onb.v;
    T n_dot_v = dot(n, v);
    T n_dot_l = dot(n, l);
    if (n_dot_l < T(0) || n_dot_v < T(0)) return mk<T>(T(0), T(0), T(0));
    V3<T> h = normalized(l + v);
    T n_dot_h = dot(n, h);
    T l_dot_h = dot(l, h);
    V3<T> cd_lin = mk<T>(m_pow(base.x, T(2.2)), m_pow(base.y, T(2.2)), m_pow(base.z, T(2.2)));                     // mon_to_lin, mat.rs:46-48
    T cd_lum = T(0.3) * cd_lin.x + T(0.6) * cd_lin.y + T(0.1) * cd_lin.z;
    V3<T> one = mk<T>(T(1.0), T(1.0), T(1.0));
    V3<T> c_tint = (cd_lum > T(0)) ? cd_lin / cd_lum : one;
    V3<T> c_spec0 = mixv(mixv(one, c_tint, m.specular_tint) * T(0.08) * m.specular, cd_lin, m.metallic);
    V3<T> c_sheen = mixv(one, c_tint, m.sheen_tint);
    T fresnel_l = schlick_fresnel(n_dot_l);
    T fresnel_v = schlick_fresnel(n_dot_v);
    T fresnel_diffuse_90 = T(0.5) + T(2.0) * l_dot_h * l_dot_h * m.roughness;
    T fresnel_diffuse = mixf(T(1.0), fresnel_diffuse_90, fresnel_l) * mixf(T(1.0), fresnel_diffuse_90, fresnel_v);
    T fss90 = l_dot_h * l_dot_h * m.roughness;
    T fss = mixf(T(1.0), fss90, fresnel_l) * mixf(T(1.0), fss90, fresnel_v);
    T subface_scatter = T(1.25) * (fss * (T(1.0) / (n_dot_l + n_dot_v) - T(0.5)) + T(0.5));
    T aspect = rsqrt_(T(1.0) - m.anisotropic * T(0.9));
    T ax = m_max(m.roughness * m.roughness / aspect, T(0.001));
    T ay = m_max(m.roughness * m.roughness * aspect, T(0.001));
    T d_specular = GTR_2_aniso(n_dot_h, dot(h, x), dot(h, y), ax, ay);
    T fresnel_h = schlick_fresnel(l_dot_h);
    V3<T> f_specular = mixv(c_spec0, one, fresnel_h);
    T g_specular = smithG_GGX_aniso(n_dot_l, dot(l, x), dot(l, y), ax, ay) * smithG_GGX_aniso(n_dot_v, dot(v, x), dot(v, y), ax, ay);
    V3<T> fresnel_sheen = (fresnel_h * m.sheen) * c_sheen;
    T d_reflect = GTR_1(n_dot_h, mixf(T(0.1), T(0.001), m.clearcoat_gloss));
    T f_reflect = mixf(T(0.04), T(1.0), fresnel_h);
    T g_reflect = smithG_GGX(n_dot_l, T(0.25)) * smithG_GGX(n_dot_v, T(0.25));
    return (((T(1.0) / PI_T) * mixf(fresnel_diffuse, subface_scatter, m.subsurface)) * cd_lin + fresnel_sheen) * (T(1.0) - m.metallic)
           + (g_specular * f_specular) * d_specular + (((mk<T>(T(0.25), T(0.25), T(0.25)) * m.clearcoat) * g_reflect) * f_reflect) * d_reflect;
}
// PDF::BRDF value, pdf.rs:97-130
template <typename T> DEV T brdf_pdf_value(const DPbr<T>& m, const Onb<T>& uvw, V3<T> r_in, V3<T> r_out) {
    T cosine = dot(normalized(r_out), uvw.w);
    if (cosine <= T(0)) return T(0);
    T diffuse_pdf = cosine / PI_T;
    V3<T> l = normalized(r_in) * T(-1.0);
    V3<T> v = normalized(r_out);
    T n_dot_l = dot(uvw.w, l);
    V3<T> h = normalized(l + v);
    T n_dot_h = dot(uvw.w, h);
    if (n_dot_h <= T(0)) return T(0);
    T aspect = rsqrt_(T(1.0) - m.anisotropic * T(0.9));
    T ax = m_max(m.roughness * m.roughness / aspect, T(0.001));
    T ay = m_max(m.roughness * m.roughness * aspect, T(0.001));
    T specular_pdf = GTR_2_aniso(n_dot_h, dot(h, uvw.u), dot(h, uvw.v), ax, ay) * m_abs(n_dot_h) * T(0.25) / n_dot_l;
    T clearcoat_pdf = GTR_1(n_dot_h, mixf(T(0.1), T(0.001), m.clearcoat_gloss)) * m_abs(n_dot_h) * T(0.25) / n_dot_l;
    return (diffuse_pdf + specular_pdf + clearcoat_pdf) / T(3.0);
}
// PDF::BRDF generate, pdf.rs:151-160 with GTR_1_direction :24-36 and GTR_2_aniso_direction :38-60
template <typename T> DEV V3<T> brdf_pdf_generate(const DPbr<T>& m, const Onb<T>& uvw, V3<T> r_in, Rng& rng) {
    T pick = rng_range(rng, T(0), T(1.0));
    if (pick < T(0.333)) return onb_local(uvw, random_cosine_direction<T>(rng));
    T r1 = rng_range(rng, T(0), T(1.0));
    T r2 = rng_range(rng, T(0), T(1.0));
    T sin_theta, cos_theta, phi;
    if (pick < T(0.666)) {
        T a = mixf(T(0.1), T(0.001), m.clearcoat_gloss);
        T a2 = a * a;
        cos_theta = rsqrt_(m_max(T(0.001), (T(1.0) - m_pow(a2, T(1.0) - r1)) / (T(1.0) - a2)));
        sin_theta = rsqrt_(m_max(T(0.001), T(1.0) - cos_theta * cos_theta));
        phi = PI_T * T(2.0) * r2;
    } else {
        T aspect = rsqrt_(T(1.0) - m.anisotropic * T(0.9));
        T ax = m_max(m.roughness * m.roughness / aspect, T(0.001));
        T ay = m_max(m.roughness * m.roughness * aspect, T(0.001));
        phi = m_atan(ay / ax * m_tan(T(2.0) * PI_T * r2 + T(0.5) * PI_T));
        if (r2 > T(0.5)) phi += PI_T;
        T sin_phi = m_sin(phi), cos_phi = m_cos(phi);
        T ax_2 = ax * ax, ay_2 = ay * ay;
        T a2 = T(1.0) / (cos_phi * cos_phi / ax_2 + sin_phi * sin_phi / ay_2);
        T tan_theta_2 = a2 * r1 / (T(1.0) - r1);
        cos_theta = T(1.0) / rsqrt_(T(1.0) + tan_theta_2);
        sin_theta = rsqrt_(m_max(T(0.001), T(1.0) - cos_theta * cos_theta));
    }
    V3<T> wh = mk<T>(sin_theta * m_cos(phi), sin_theta * m_sin(phi), cos_theta);      // spherical_direction, pdf.rs:20-22
    return onb_local(uvw, reflect_(r_in, wh));
}

// A lane's partial sum of one pixel: three f64 in registers (LDS instead was measured and is slower: docs/history.md).
struct AccReg { double v[3]; DEV double get(int k) const { return v[k]; } DEV void set(int k, double x) { v[k] = x; } };

// ------------------------------------------------------------------ launch geometry per kernel family
// List scenes: 256-thread workgroups, several per CU.  BVH scenes: ONE workgroup per CU holding every wave the register budget
// allows (4 per SIMD = 1024 threads; 3 = 768 for the persistent-traversal and the principled-material kernels), so that the CU's
// 160 KB of LDS holds one copy of the top of the BVH beside the waves' stacks and queues.
template <uint32_t FEATS> struct Shape {
    static constexpr uint32_t WAVES_PER_SIMD = FEATS == 0u ? RT_WAVES_LEAN : ((FEATS & F_PBR) ? RT_WAVES_PBR : ((FEATS & F_PERSIST) ? RT_WAVES_PERSIST : RT_WAVES_BVH));
    static constexpr bool ONE_PER_CU = (FEATS & F_BVH) != 0u;
    static constexpr uint32_t WAVES = ONE_PER_CU ? 4u * WAVES_PER_SIMD : 4u;
    static constexpr uint32_t THREADS = 64u * WAVES;
    static constexpr uint32_t QN_MIN = ONE_PER_CU ? 16u : 64u;    // camera-path queue entries per wave: at least this, up to 64 (KParams::queue_entries)
    typedef AccReg Acc;
};

// ------------------------------------------------------------------ wave helpers
DEV uint32_t lane_rank(unsigned long long mask) {   // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
DEV double wave_sum(double x) {                      // fixed butterfly: deterministic for a given set of inputs
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Lanes with `need` set hold a partial per-pixel sum (acc) for local pixel acc_px.  All partials of one pixel are
// combined by a masked butterfly and added to out[] by one lane with one f64 atomic per channel (a handful per pixel
// per frame: this is the kernel's only global write traffic).
// PER_LANE (the persistent-traversal kernels, round 6): every lane adds its own partial sum with three atomics instead — few lanes change
// pixel together in an advance pass, and the butterfly costs the same for two lanes as for sixty-four.  *Measured*
// (profiles/r06_flush_per_lane_ab.log): teapot room +6.7 %; Cornell box +0.3 %, random spheres -1.4 %, final scene -3 ... -7 % — so only there.
template <typename T, typename A, bool PER_LANE>
DEV void flush_acc(const KParams<T>& P, bool need, uint32_t acc_px, const A& acc, uint32_t lane, uint32_t& n_flush) {
    if constexpr (PER_LANE) {
        const unsigned long long mm = __ballot(need);
        if (mm == 0ull) return;
        if (need) {
            COLD_K;
            double* o = PK(out) + (size_t)acc_px * 3u;
            unsafeAtomicAdd(o + 0, acc.get(0)); unsafeAtomicAdd(o + 1, acc.get(1)); unsafeAtomicAdd(o + 2, acc.get(2));
        }
        n_flush += (uint32_t)__popcll(mm);                // lane flushes, three atomics each
        return;
    }
    unsigned long long m = __ballot(need);
    while (m) {
        COLD_K;
        double* const out = PK(out);
        uint32_t leader = (uint32_t)__builtin_ctzll(m);
        uint32_t px = (uint32_t)__builtin_amdgcn_readlane((int)acc_px, (int)leader);
        bool mine = need && acc_px == px;
        double s0 = wave_sum(mine ? acc.get(0) : 0.0);
        double s1 = wave_sum(mine ? acc.get(1) : 0.0);
        double s2 = wave_sum(mine ? acc.get(2) : 0.0);
        if (lane == leader) {
            double* o = out + (size_t)px * 3u;        // hardware f64 atomics: a pixel's samples may be split over several waves
            unsafeAtomicAdd(o + 0, s0); unsafeAtomicAdd(o + 1, s1); unsafeAtomicAdd(o + 2, s2);
        }
        n_flush++;                                    // wave-uniform (a scalar register)
        need = need && !mine;
        m = __ballot(need);
    }
}

// ------------------------------------------------------------------ wave-uniform work cursor + regeneration queue
// per-wave regeneration queue of QN camera paths (sized for f64): 80 B per entry.  List scenes generate 64 at a time (every lane
// busy: the camera code is 6 % of their bounce); BVH scenes, where it is noise, keep 16 so that LDS goes to BVH nodes instead.
static constexpr uint32_t regen_bytes(uint32_t qn) { return 7u * qn * 8u + 6u * qn * 4u; }
static const uint32_t NONE_PX = 0xFFFFFFFFu;
// RT_DIAG (diagnostic build only, never shipped): per-section wave-cycle shares via s_memtime, written to stats[3..8].
#ifdef RT_DIAG
#define DIAG_DECL unsigned long long dg_t = 0, dg_sum[6] = {0, 0, 0, 0, 0, 0};
#define DIAG_T0() do { __builtin_amdgcn_sched_barrier(0); dg_t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DIAG_ADD(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t1_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); dg_sum[k] += t1_ - dg_t; dg_t = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DIAG_DECL
#define DIAG_T0() do {} while (0)
#define DIAG_ADD(k) do {} while (0)
#endif

// Everything here is wave-uniform (lives in SGPRs).  Work cursor: samples [cur_s, s_hi) of local pixel cur_px, then pixels
// up to end_px (one dequeued chunk); cur_gp / cur_i / cur_j are the cursor pixel's global index and image coordinates,
// recomputed once per pixel, not per sample.  Regeneration queue: entries [q_head, q_head + q_count) of this wave's LDS
// queue hold camera paths that were generated 64 at a time with every lane busy; lanes whose path ended pop one instead
// of running the camera code themselves at ~1/3 lane occupancy.
struct WaveWork {
    uint32_t cur_px, end_px, cur_s, s_lo, s_hi;
    uint32_t cur_gp, cur_i, cur_j;
    uint32_t q_head, q_count;
    bool queue_done;
};
template <typename T> DEV void locate(const KParams<T>& P, WaveWork& w) {
    COLD_K;   // local pixel -> global output-order pixel (tile t = rank + q * world)
    uint32_t q = w.cur_px / PK(tile_px), kk = w.cur_px - q * PK(tile_px);
    w.cur_gp = (PK(rank) + q * PK(world)) * PK(tile_px) + kk;
    uint32_t row = w.cur_gp / PK(W);
    w.cur_i = w.cur_gp - row * PK(W);
    w.cur_j = PK(H) - 1u - row;           // row 0 is j = H-1, main.rs:772
}
// Point the cursor at work chunk c (KParams: whole pixels first, then pieces of chunk_spp samples of one pixel).
template <typename T> DEV void take_chunk(const KParams<T>& P, WaveWork& w, uint32_t c) {
    COLD_K;
    const uint32_t n_local_px = PK(n_local_tiles) * PK(tile_px);
    if (c < PK(n_coarse_px)) {                      // a whole pixel
        w.cur_px = c; w.end_px = c + 1u; w.s_lo = 0u; w.s_hi = PK(spp);
    } else {
        const uint32_t c2 = c - PK(n_coarse_px);
        const uint32_t cp = c2 / PK(chunks_per_px), sub = c2 - cp * PK(chunks_per_px);
        w.cur_px = PK(n_coarse_px) + cp * PK(chunk_px);
        w.end_px = w.cur_px + PK(chunk_px); if (w.end_px > n_local_px) w.end_px = n_local_px;
        w.s_lo = sub * PK(chunk_spp);
        w.s_hi = w.s_lo + PK(chunk_spp); if (w.s_hi > PK(spp)) w.s_hi = PK(spp);
    }
    w.cur_s = w.s_lo;
    locate(P, w);
}
// Refill the queue: the next (up to) 64 samples of the cursor, all lanes generating (main.rs:813-820).  False when the
// global work queue is exhausted and nothing was generated.
template <typename T>
DEV bool refill_queue(const KParams<T>& P, WaveWork& w, uint32_t lane, T* q_real, uint32_t* q_u32) {
    COLD_K;
    const uint32_t QN = PK(queue_entries);          // 16, 32 or 64 (wave-uniform: the host gives BVH kernels what LDS the node cache leaves)
    const uint32_t n_px = PK(W) * PK(H);
    bool have = false;
    uint32_t g_px = 0, g_s = 0, g_gp = 0, g_i = 0, g_j = 0;
    uint32_t n_gen = 0;
    while (n_gen < QN) {
        if (w.cur_px == w.end_px) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(PK(queue), 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c >= PK(n_chunks)) { w.queue_done = true; break; }
            take_chunk(P, w, c);
        }
        if (w.cur_gp >= n_px || w.s_lo >= w.s_hi) { w.cur_px++; w.cur_s = w.s_lo; if (w.cur_px != w.end_px) locate(P, w); continue; }   // padding pixel / empty range
        uint32_t avail = w.s_hi - w.cur_s;
        uint32_t room = QN - n_gen;
        uint32_t take = room < avail ? room : avail;
        if (lane >= n_gen && lane < n_gen + take) { have = true; g_px = w.cur_px; g_s = w.cur_s + (lane - n_gen); g_gp = w.cur_gp; g_i = w.cur_i; g_j = w.cur_j; }
        n_gen += take;
        w.cur_s += take;
        if (w.cur_s == w.s_hi) { w.cur_px++; w.cur_s = w.s_lo; if (w.cur_px != w.end_px) locate(P, w); }
    }
    if (n_gen == 0) return false;
    if (have) {
        Rng g = rng_for_path(PK(seed), g_gp, g_s);
        T random_u = rng_u01(g, T(0));
        T random_v = rng_u01(g, T(0));
        T u = (T(g_i) + random_u) / T(PK(W) - 1u);
        T v = (T(g_j) + random_v) / T(PK(H) - 1u);
        // Camera::get_ray, camera.rs:51-59 (random_in_unit_disk, vec.rs:96-105)
        T da, db;
        for (;;) {
            da = rng_range(g, T(-1.0), T(1.0));
            db = rng_range(g, T(-1.0), T(1.0));
            V3<T> pd = mk<T>(da, db, T(0));
            if (dot(pd, pd) < T(1.0)) break;       // `p.length() < 1.0` (vec.rs:101): sqrt(x) < 1 <=> x < 1
        }
        V3<T> rd = PK(cam.lens_radius) * mk<T>(da, db, T(0));
        V3<T> offset = cl3(K->cam.cu) * rd.x + cl3(K->cam.cv) * rd.y;
        T time = PK(cam.time0) + rng_u01(g, T(0)) * (PK(cam.time1) - PK(cam.time0));
        V3<T> go = cl3(K->cam.origin) + offset;
        V3<T> gd = cl3(K->cam.lower_left_corner) + u * cl3(K->cam.horizontal) + v * cl3(K->cam.vertical) - (cl3(K->cam.origin) + offset);
        q_real[0u * QN + lane] = go.x; q_real[1u * QN + lane] = go.y; q_real[2u * QN + lane] = go.z;
        q_real[3u * QN + lane] = gd.x; q_real[4u * QN + lane] = gd.y; q_real[5u * QN + lane] = gd.z;
        q_real[6u * QN + lane] = time;
        q_u32[0u * QN + lane] = g.s0; q_u32[1u * QN + lane] = g.s1; q_u32[2u * QN + lane] = g.s2; q_u32[3u * QN + lane] = g.s3;
        q_u32[4u * QN + lane] = g_px; q_u32[5u * QN + lane] = g_s;
    }
    w.q_head = 0; w.q_count = n_gen;
    __builtin_amdgcn_wave_barrier();          // one wave: LDS writes above are ordered before the reads below
    return true;
}
// Lanes with `dead` set take the next camera path from the wave's queue ("compaction by regeneration").  True for the
// lanes that got one (ray, rng, new_px, path_s are then the new path's).
template <typename T>
DEV bool take_new_paths(const KParams<T>& P, WaveWork& w, uint32_t lane, T* q_real, uint32_t* q_u32, bool dead,
                        RayT<T>& ray, Rng& rng, uint32_t& new_px, uint32_t& path_s) {
    const uint32_t QN = P.queue_entries;
    bool got_new = false;
    for (;;) {
        unsigned long long want = __ballot(dead && !got_new);
        if (want == 0) break;
        if (w.q_count == 0) {
            if (w.queue_done) break;
            if (!refill_queue(P, w, lane, q_real, q_u32)) break;
        }
        uint32_t n_want = (uint32_t)__popcll(want);
        uint32_t take = n_want < w.q_count ? n_want : w.q_count;
        uint32_t rank = lane_rank(want);
        if (dead && !got_new && rank < take) {
            const uint32_t e = w.q_head + rank;
            got_new = true;
            ray.o = mk<T>(q_real[0u * QN + e], q_real[1u * QN + e], q_real[2u * QN + e]);
            ray.d = mk<T>(q_real[3u * QN + e], q_real[4u * QN + e], q_real[5u * QN + e]);
            ray.tm = q_real[6u * QN + e];
            rng.s0 = q_u32[0u * QN + e]; rng.s1 = q_u32[1u * QN + e]; rng.s2 = q_u32[2u * QN + e]; rng.s3 = q_u32[3u * QN + e];
            new_px = q_u32[4u * QN + e]; path_s = q_u32[5u * QN + e];
        }
        w.q_head += take; w.q_count -= take;
    }
    return got_new;
}

// The colour of a Lambertian / DiffuseLight material.  The lean kernels (constant textures only) take it from the material record they
// hold — rt_flatten.cpp copies a constant texture's colour there — instead of gathering the texture record behind it (*measured* +1.4 %
// on the Cornell box; no gain in the BVH kernels, which keep the texture walk).
template <typename T, uint32_t FEATS> DEV V3<T> const_or_tex(const KParams<T>& P, const DMaterial<T>& mt, const Rec<T>& rec) {
    if (FEATS == 0u) return ld3(mt.albedo);
    return tex_eval<T, FEATS>(P, mt.tex, rec.u, rec.v, rec.p);
}
// U01 / R(a,b) of rt_rng.h from a u64 that was drawn earlier (shade_hit draws for several arms at once)
DEV double rng_u01_of(uint64_t u, double) { const uint64_t v = u >> 11; return ((double)(uint32_t)(v >> 32) * 4294967296.0 + (double)(uint32_t)v) * 0x1.0p-53; }
DEV double rng_range_of(uint64_t u, double a, double b) {
    double res = (__longlong_as_double((long long)(0x3FF0000000000000ULL | (u >> 12))) - 1.0) * (b - a) + a;
    if (!(res < b)) { uint64_t bb = (uint64_t)__double_as_longlong(b); if (b > 0.0) bb -= 1; else if (b < 0.0) bb += 1; else bb = 0x8000000000000001ULL; res = __longlong_as_double((long long)bb); }
    return res;
}
// ------------------------------------------------------------------ one level of ray_color after the hit: main.rs:50-116
// In: the hit record.  In/out: ray (becomes the scattered ray), beta, rng, depth_left.  Out: done (the path ends here) and
// e, the terminal radiance to be multiplied by beta.
template <typename T, uint32_t FEATS>
DEV void shade_hit(const KParams<T>& P, const Rec<T>& rec, RayT<T>& ray, V3<T>& beta, Rng& rng, uint32_t& depth_left, bool& done, V3<T>& e) {
    const DMaterial<T> mt = ld_mat(P.materials + rec.mat);
    // Lambertian (its two sampling arms) and Metal in one instruction stream where they run the same instructions on their own
    // values — per lane the operations and the draw order are exactly those of the separate arms below (bit-identical samples):
    // (1) the first normalisation (onb.rs:10 / vec.rs:112-114 + mat.rs:285), (2) the next two u64 draws (pdf.rs:10-11 /
    // rect.rs:104-105 / the first try of vec.rs:80); the cosine pdf is computed once for the scenes with and without lights.
    constexpr bool MERGE = MERGE_ARMS && sizeof(T) == 8u;               // (the f32 kernels draw one u32 per value)
    const bool lam = mt.kind == M_LAMBERTIAN, met = mt.kind == M_METAL;
    if (MERGE && (lam || met)) { if constexpr (MERGE) {
        V3<T> unit0 = rec.n;
        if (met) unit0 = ray.d + ((-dot(ray.d, rec.n)) * T(2.0) * rec.n);  // reflect, vec.rs:112-114
        unit0 = normalized(unit0);
        V3<T> attenuation = unit0;
        Onb<T> uvw; uvw.w = unit0; uvw.u = unit0; uvw.v = unit0;
        bool to_light = false;
        DLight L; L.kind = 0xFFFFFFFFu; L.index = 0u;
        if (lam) {                                                          // mat.rs:225-249, main.rs:92-98
            attenuation = const_or_tex<T, FEATS>(P, mt, rec);
            V3<T> a = (m_abs(uvw.w.x) > T(0.9)) ? mk<T>(T(0), T(1.0), T(0)) : mk<T>(T(1.0), T(0), T(0));   // onb.rs:8-20
            uvw.v = normalized(cross(uvw.w, a));
            uvw.u = cross(uvw.w, uvw.v);
            if (P.n_lights != 0u && rng_bool(rng)) {                        // pdf.rs:167-173 (no lights: deviation D2, cosine only)
                to_light = true;
                L = ld_light(P.lights + rng_index(rng, P.n_lights));        // hit.rs:94-96
            }
        }
        const bool two = !to_light || L.kind == L_RECT;
        uint64_t b1 = 0, b2 = 0;
        if (two) { b1 = rng_u64(rng); b2 = rng_u64(rng); }
        if (lam) {
            V3<T> dir;
            if (!to_light) {                                                // random_cosine_direction, pdf.rs:8-18
                T r1 = rng_u01_of(b1, T(0)), r2 = rng_u01_of(b2, T(0));
                T z = rsqrt_(T(1.0) - r2);
                T phi = T(2.0) * PI_T * r1;
                T sn, cs; sincos_0_2pi(phi, sn, cs);
                dir = onb_local(uvw, mk<T>(cs * rsqrt_(r2), sn * rsqrt_(r2), z));
            } else if (L.kind == L_RECT) {                                  // rect.rs:103-111
                const DRect<T> rc = ld_rect(P.rects + L.index);
                uint32_t ki, ai, bi; plane_axes(rc.plane, ki, ai, bi);
                T ra = rng_range_of(b1, rc.a0, rc.a1);
                T rb = rng_range_of(b2, rc.b0, rc.b1);
                V3<T> pt;
                pt.x = ki == 0u ? rc.k : (ai == 0u ? ra : rb);
                pt.y = ki == 1u ? rc.k : (ai == 1u ? ra : rb);
                pt.z = ki == 2u ? rc.k : rb;
                dir = pt - rec.p;
            } else {
                dir = light_random<T, FEATS>(P, L, rec.p, rng);
            }
            T cosine = dot(normalized(dir), uvw.w);                         // pdf.rs:131-139
            T pdf_value = (cosine > T(0)) ? cosine / PI_T : T(0);
            if (P.n_lights != 0u) {
                T lsum = T(0);                                              // hit.rs:90-92
                for (uint32_t li = 0; li < P.n_lights; li++) lsum += light_pdf_value<T, FEATS>(P, ld_light(P.lights + li), rec.p, dir);
                T lpdf = P.n_lights == 1u ? lsum : lsum / T(P.n_lights);    // x / 1.0 is x
                pdf_value = T(0.5) * lpdf + T(0.5) * pdf_value;             // pdf.rs:143-145
            }
            T sc = m_max(dot(rec.n, normalized(dir)), T(0)) / PI_T;         // scattering_pdf, mat.rs:246-249
            beta = (beta * (attenuation * sc)) / pdf_value;                 // main.rs:97 (emitted is the literal zero here)
            ray.o = rec.p; ray.d = dir;                                     // time unchanged
        } else {                                                            // mat.rs:280-293
            V3<T> fz;                                                       // random_in_unit_sphere, vec.rs:78-85: the first try's a and b are drawn above
            fz.x = rng_range_of(b1, T(-1.0), T(1.0)); fz.y = rng_range_of(b2, T(-1.0), T(1.0)); fz.z = rng_range(rng, T(-1.0), T(1.0));
            while (!(dot(fz, fz) < T(1.0))) { T a = rng_range(rng, T(-1.0), T(1.0)), b = rng_range(rng, T(-1.0), T(1.0)), c = rng_range(rng, T(-1.0), T(1.0)); fz = mk<T>(a, b, c); }
            V3<T> sd = unit0 + mt.param * fz;
            if (dot(sd, rec.n) > T(0)) { beta = ld3(mt.albedo) * beta; ray.o = rec.p; ray.d = sd; }   // main.rs:89-91
            else done = true;                                               // None -> emitted = 0, main.rs:108-110
        }
    } } else if (lam) {                                                     // mat.rs:225-249, main.rs:92-98
        V3<T> attenuation = tex_eval<T, FEATS>(P, mt.tex, rec.u, rec.v, rec.p);
        Onb<T> uvw = onb_from_w(rec.n);                                     // PDF::cosine_pdf, pdf.rs:81-85
        V3<T> dir; T pdf_value;
        if (P.n_lights == 0u) {                                             // stated deviation D2 (reference panics)
            dir = onb_local(uvw, random_cosine_direction<T>(rng));
            T cosine = dot(normalized(dir), uvw.w);
            pdf_value = (cosine > T(0)) ? cosine / PI_T : T(0);
        } else {
            if (rng_bool(rng)) {                                            // pdf.rs:167-173
                uint32_t li = rng_index(rng, P.n_lights);                   // hit.rs:94-96
                dir = light_random<T, FEATS>(P, ld_light(P.lights + li), rec.p, rng);
            } else {
                dir = onb_local(uvw, random_cosine_direction<T>(rng));
            }
            T lsum = T(0);                                                  // hit.rs:90-92
            for (uint32_t li = 0; li < P.n_lights; li++) lsum += light_pdf_value<T, FEATS>(P, ld_light(P.lights + li), rec.p, dir);
            T lpdf = lsum / T(P.n_lights);
            T cosine = dot(normalized(dir), uvw.w);                         // pdf.rs:131-139
            T cpdf = (cosine > T(0)) ? cosine / PI_T : T(0);
            pdf_value = T(0.5) * lpdf + T(0.5) * cpdf;                      // pdf.rs:143-145
        }
        T sc = m_max(dot(rec.n, normalized(dir)), T(0)) / PI_T;             // scattering_pdf, mat.rs:246-249
        beta = (beta * (attenuation * sc)) / pdf_value;                     // main.rs:97 (emitted is the literal zero here)
        ray.o = rec.p; ray.d = dir;                                         // time unchanged
    } else if (met) {                                                       // mat.rs:280-293
        V3<T> dn = ray.d + ((-dot(ray.d, rec.n)) * T(2.0) * rec.n);         // reflect, vec.rs:112-114
        V3<T> reflected = normalized(dn);
        V3<T> fz = random_in_unit_sphere<T>(rng);
        V3<T> sd = reflected + mt.param * fz;
        if (dot(sd, rec.n) > T(0)) { beta = ld3(mt.albedo) * beta; ray.o = rec.p; ray.d = sd; }   // main.rs:89-91
        else done = true;                                                   // None -> emitted = 0, main.rs:108-110
    } else if ((FEATS & F_DIELECTRIC) && mt.kind == M_DIELECTRIC) {         // mat.rs:343-374
        T refraction_ratio = rec.front ? T(1.0) / mt.param : mt.param;
        V3<T> unit_direction = normalized(ray.d);
        T cos_theta = m_min(dot(T(-1.0) * unit_direction, rec.n), T(1.0));
        T sin_theta = rsqrt_(T(1.0) - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > T(1.0);
        T q = (T(1.0) - refraction_ratio) / (T(1.0) + refraction_ratio);    // reflectance, mat.rs:309-313
        T r0 = q * q;
        T m1 = T(1.0) - cos_theta, m2 = m1 * m1;
        T refl = r0 + (T(1.0) - r0) * (m1 * (m2 * m2));
        bool will_reflect = rng_u01(rng, T(0)) < refl;
        V3<T> direction;
        if (cannot_refract || will_reflect) {
            direction = unit_direction + ((-dot(unit_direction, rec.n)) * T(2.0) * rec.n);
        } else {                                                            // refract, vec.rs:116-121
            T ct = m_min(dot(T(-1.0) * unit_direction, rec.n), T(1.0));
            V3<T> r_out_perp = refraction_ratio * (unit_direction + ct * rec.n);
            T l = length(r_out_perp);
            V3<T> r_out_para = (T(-1.0) * rsqrt_(m_abs(T(1.0) - l * l))) * rec.n;
            direction = r_out_perp + r_out_para;
        }
        ray.o = rec.p; ray.d = direction;                                   // attenuation (1,1,1): beta unchanged
    } else if ((FEATS & F_PBR) && mt.kind == M_PBR) {                       // mat.rs:118-131 + main.rs:99-105 (Microfacet arm)
        const DPbr<T> pm = ld_pbr(P.pbr + (uint32_t)mt.albedo[0]);
        Onb<T> uvw = onb_from_w(rec.n);                                     // PDF::brdf_pdf, pdf.rs:70-79
        V3<T> dir; T pdf_value;
        if (P.n_lights == 0u) {                                             // stated deviation D2
            dir = brdf_pdf_generate(pm, uvw, ray.d, rng);
            pdf_value = brdf_pdf_value(pm, uvw, ray.d, dir);
        } else {
            if (rng_bool(rng)) {
                uint32_t li = rng_index(rng, P.n_lights);
                dir = light_random<T, FEATS>(P, ld_light(P.lights + li), rec.p, rng);
            } else {
                dir = brdf_pdf_generate(pm, uvw, ray.d, rng);
            }
            T lsum = T(0);
            for (uint32_t li = 0; li < P.n_lights; li++) lsum += light_pdf_value<T, FEATS>(P, ld_light(P.lights + li), rec.p, dir);
            pdf_value = T(0.5) * (lsum / T(P.n_lights)) + T(0.5) * brdf_pdf_value(pm, uvw, ray.d, dir);
        }
        V3<T> base = tex_eval<T, FEATS>(P, mt.tex, rec.u, rec.v, rec.p);
        V3<T> f = pbr_brdf(pm, base, ray.d, dir, rec.n);
        beta = (beta * f) / pdf_value;                                      // main.rs:104
        ray.o = rec.p; ray.d = dir;
    } else if (mt.kind == M_DIFFUSE_LIGHT) {                                // mat.rs:395-401; no scatter -> main.rs:108-110
        if (rec.front) e = const_or_tex<T, FEATS>(P, mt, rec);
        done = true;
    } else if ((FEATS & F_MEDIUM) && (P.flags & 4u) && mt.kind == M_ISOTROPIC) {   // RT_ISOTROPIC_SCATTER (opt-in, non-reference):
        V3<T> attenuation = tex_eval<T, FEATS>(P, mt.tex, rec.u, rec.v, rec.p);     // the old Isotropic::scatter, mat.rs:418-421
        V3<T> sd = random_in_unit_sphere<T>(rng);
        beta = attenuation * beta; ray.o = rec.p; ray.d = sd;
    } else {                                                                // Isotropic: no scatter_mc_method (mat.rs:417-422) -> absorbs
        done = true;
    }
    if (!done) {
        depth_left--;
        if (depth_left == 0) done = true;                                   // the child call returns 0 at main.rs:42-45
        if ((P.flags & 2u) && beta.x == T(0) && beta.y == T(0) && beta.z == T(0)) done = true;   // RT_STOP_ON_ZERO (opt-in)
    }
}

// A finished path hands in beta * e (kept even when e = 0, so that inf * 0 = NaN poisons a sample exactly when the
// reference's arithmetic does).
template <typename T, typename A>
DEV void add_radiance(const KParams<T>& P, V3<T> L, A& acc, uint32_t& n_nonfinite, uint32_t path_px, uint32_t path_s) {
    double l0 = (double)L.x, l1 = (double)L.y, l2 = (double)L.z;
    acc.set(0, acc.get(0) + l0); acc.set(1, acc.get(1) + l1); acc.set(2, acc.get(2) + l2);
    if (!(l0 - l0 == 0.0 && l1 - l1 == 0.0 && l2 - l2 == 0.0)) n_nonfinite++;
    if (P.samples_out) {
        double* so = P.samples_out + ((size_t)path_px * P.spp + path_s) * 3u;
        so[0] = l0; so[1] = l1; so[2] = l2;
    }
}
DEV unsigned long long* stats_row(unsigned long long* stats) {     // one of the RT_STATS_ROWS copies of the counter block (rt_launch.h)
    // row from the CU / shader-array / shader-engine bits of HW_REG_HW_ID (bits 8..15): read here, at the very end of the
    // kernel, so that nothing (such as the block index) has to stay alive through the main loop for it
    const uint32_t where = (uint32_t)__builtin_amdgcn_s_getreg((8 - 1) << 11 | 8 << 6 | 4);
    return stats ? stats + (where % RT_STATS_ROWS) * RT_STATS_SLOTS : stats;
}
DEV void write_stats(unsigned long long* stats, uint32_t lane, uint32_t n_nonfinite, unsigned long long n_iters, unsigned long long n_active, uint32_t n_flush) {
    if (!stats) return;
    if (n_nonfinite) atomicAdd(&stats[0], (unsigned long long)n_nonfinite);
    if (lane == 0 && n_flush) atomicAdd(&stats[11], (unsigned long long)n_flush);
    if (lane == 0) { atomicAdd(&stats[1], n_iters); atomicAdd(&stats[2], n_active); }      // (both wave-uniform: kept in scalar registers)
}

// ------------------------------------------------------------------ list scenes: lock-step bounce loop
// Every iteration: dead lanes regenerate, then all 64 lanes run one level of ray_color together (closest hit over the
// wave-uniform object list, hit record, material).  Used when the scene has no BVH: every lane's closest-hit search costs
// the same, so lock-step wastes nothing.
template <typename T, uint32_t FEATS>
DEV void trace_lockstep(const KParams<T>& P, uint32_t lane, T* q_real, uint32_t* q_u32, uint32_t* stack) {
    WaveWork w; w.cur_px = w.end_px = w.cur_s = w.s_lo = w.s_hi = w.cur_gp = w.cur_i = w.cur_j = w.q_head = w.q_count = 0; w.queue_done = false;
    // per-lane path state
    bool alive = false;
    RayT<T> ray; ray.o = mk<T>(T(0), T(0), T(0)); ray.d = ray.o; ray.tm = T(0);
    V3<T> beta = mk<T>(T(0), T(0), T(0));
    uint32_t depth_left = 0, path_s = 0;                 // (the path's pixel is the accumulator's: acc_px)
#define path_px acc_px
    Rng rng; rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    // per-lane accumulator for one local pixel
    uint32_t acc_px = NONE_PX;
    typename Shape<FEATS>::Acc acc;
    acc.set(0, 0.0); acc.set(1, 0.0); acc.set(2, 0.0);
    uint32_t n_nonfinite = 0, n_flush = 0;
    unsigned long long n_iters = 0, n_active = 0;
    DIAG_DECL

    for (;;) {
        DIAG_T0();
        // ---- lanes whose path has ended take the next camera path from the wave's queue
        uint32_t new_px = 0;
        const bool got_new = take_new_paths(P, w, lane, q_real, q_u32, !alive, ray, rng, new_px, path_s);
        if (__ballot(alive || got_new) == 0) break;     // queue empty and every path finished
        DIAG_ADD(0);

        // ---- lanes moving on to another pixel hand in their partial sum
        flush_acc<T, typename Shape<FEATS>::Acc, false>(P, got_new && acc_px != NONE_PX && acc_px != new_px, acc_px, acc, lane, n_flush);

        if (got_new) {
            if (acc_px != new_px) { acc_px = new_px; acc.set(0, 0.0); acc.set(1, 0.0); acc.set(2, 0.0); }
            beta = mk<T>(T(1.0), T(1.0), T(1.0));
            depth_left = P.max_depth;
            alive = true;
        }

        n_iters++;
        n_active += (unsigned long long)__popcll(__ballot(alive));
        DIAG_ADD(1);

        // ---- one level of ray_color (main.rs:41-120) for every live lane
        if (alive) {
            bool done = false;
            V3<T> e = mk<T>(T(0), T(0), T(0));          // terminal radiance of this path (times beta)
            if (depth_left == 0) {
                done = true;                            // main.rs:42-45
            } else {
                T t_hit; HitId id; id.obj = 0; id.prim = 0;
                bool any_hit;                                                                            // main.rs:48
                if constexpr (FEATS == 0u) any_hit = world_hit_list<T>(P, ray, TMin<T>::v(), rng, t_hit, id);
                else any_hit = world_hit<T, FEATS>(P, ray, TMin<T>::v(), rng, t_hit, id, stack);
                DIAG_ADD(2);
                if (!any_hit) {
                    e = ld3(P.background); done = true;                                     // main.rs:118
                } else {
                    Rec<T> rec;
                    finalize_hit<T, FEATS>(P, ray, t_hit, id, true, rec);
                    DIAG_ADD(3);
#ifdef RT_TRACE_PATH    // debugging build only (tools/fuzz_probe.py): the hits of one path, level by level, for comparison with the oracle's orc_trace_path
                    if (P.trace_out && path_px == P.trace_px && path_s == P.trace_s && depth_left <= P.max_depth) {
                        double* o = P.trace_out + 16u * (P.max_depth - depth_left);
                        o[0] = (double)rec.t; o[1] = (double)rec.p.x; o[2] = (double)rec.p.y; o[3] = (double)rec.p.z;
                        o[4] = (double)rec.n.x; o[5] = (double)rec.n.y; o[6] = (double)rec.n.z; o[7] = rec.front ? 1.0 : 0.0;
                        o[8] = (double)id.obj; o[9] = (double)(id.prim >> 28); o[10] = (double)(id.prim & 0x0FFFFFFFu); o[11] = (double)rec.mat;
                        o[12] = (double)ray.d.x; o[13] = (double)ray.d.y; o[14] = (double)ray.d.z; o[15] = 1.0;
                    }
#endif
                    shade_hit<T, FEATS>(P, rec, ray, beta, rng, depth_left, done, e);
                }
            }
            DIAG_ADD(4);
            if (done) {
                add_radiance(P, beta * e, acc, n_nonfinite, path_px, path_s);
                alive = false;
            }
            DIAG_ADD(5);
        }
    }
    // ---- the queue is empty: hand in what is left
    flush_acc<T, typename Shape<FEATS>::Acc, false>(P, acc_px != NONE_PX, acc_px, acc, lane, n_flush);
    unsigned long long* st; { COLD_K; st = stats_row(PK(stats)); }
    unsigned long long live = n_active;
    write_stats(st, lane, n_nonfinite, n_iters, live, n_flush);
#ifdef RT_DIAG
    if (st && lane == 0) for (int k = 0; k < 6; k++) atomicAdd(&st[3 + k], dg_sum[k]);
#endif
}

#undef path_px
// ------------------------------------------------------------------ BVH scenes: resumable closest-hit search, persistent traversal
// With a BVH in the scene the cost of `world.hit` differs wildly between lanes (a ray that misses the root box is done
// after one node, its neighbour walks a hundred), and in lock-step every lane waits for the slowest one: *measured* VALU
// lane utilisation 16 % (teapot room) to 27 % (final scene), ~90 % of the time spent in traversal.  Here a lane's
// closest-hit search is a small state machine instead:
//   PH_NEW   needs a camera path            PH_OBJ   walking the top-level list, at object my_oi
//   PH_BVH   inside the BVH of object my_oi  PH_SHADE list finished: hit record + material next
// and the wave alternates between two kinds of passes.  An *advance* pass is one lock-step iteration for the lanes that
// are not inside a BVH (regenerate; walk the list up to the next BVH object or to its end; shade).  A *traversal* pass
// steps the BVH lanes, one node per step, and stops as soon as fewer than `trav_lo` of them are still walking: the
// finished ones go through an advance pass (which also brings fresh rays to the BVH) while the stragglers simply keep
// their node/stack and resume in the next traversal pass together with the newcomers.  Traversal therefore runs with
// between trav_lo and 64 lanes busy instead of "whoever is slowest".  Per lane the order of operations (objects in push
// order, bbox / left / right, t_max shrinking, RNG draws) is exactly the lock-step one, so samples are bit-identical.
enum : uint32_t { PH_NEW = 0u, PH_OBJ = 1u, PH_BVH = 2u, PH_SHADE = 3u };

template <typename T, uint32_t FEATS>
DEV void trace_resumable(const KParams<T>& P, uint32_t lane, T* q_real, uint32_t* q_u32, uint32_t* stack) {
    WaveWork w; w.cur_px = w.end_px = w.cur_s = w.s_lo = w.s_hi = w.cur_gp = w.cur_i = w.cur_j = w.q_head = w.q_count = 0; w.queue_done = false;
    const bool near_first = (FEATS & F_NEAR_FIRST) != 0u;
    // per-lane path state
    uint32_t phase = PH_NEW;
    RayT<T> ray; ray.o = mk<T>(T(0), T(0), T(0)); ray.d = ray.o; ray.tm = T(0);
    V3<T> beta = mk<T>(T(0), T(0), T(0));
    uint32_t depth_left = 0, path_px = 0, path_s = 0;
    Rng rng; rng.s0 = rng.s1 = rng.s2 = rng.s3 = 0;
    // closest-hit search of the current level (HittableList::hit, hit.rs:59-71)
    uint32_t my_oi = 0;
    T closest = Lim<T>::inf();
    HitId id; id.obj = 0; id.prim = 0;
    bool any_hit = false;
    // traversal state while phase == PH_BVH (the node stack is the lane's LDS column)
    const uint32_t BVH_DONE = 0xFFFFFFFFu;
    uint32_t tv_node = 0, tv_sp = 0, tv_prim = 0, tv_best = 0, tv_leaf_a = 0, tv_leaf_b = 0, tv_leaf_node = 0;
    T tv_closest = T(0);
    bool tv_any = false, tv_have_leaf = false;
    // per-lane accumulator for one local pixel
    uint32_t acc_px = NONE_PX;
    typename Shape<FEATS>::Acc acc;
    acc.set(0, 0.0); acc.set(1, 0.0); acc.set(2, 0.0);
    uint32_t n_nonfinite = 0, n_flush = 0;
    unsigned long long n_iters = 0, n_active = 0, n_steps = 0, n_step_lanes = 0, n_leaf_steps = 0, n_leaf_lanes = 0;   // steps: box steps
    DIAG_DECL
    DIAG_T0();

    for (;;) {
        const uint32_t n_bvh = (uint32_t)__popcll(__ballot(phase == PH_BVH));
        const bool work_left = !(w.queue_done && w.q_count == 0u);
        const uint32_t n_adv = (uint32_t)__popcll(__ballot(phase == PH_OBJ || phase == PH_SHADE || (phase == PH_NEW && work_left)));
        if (n_bvh == 0u && n_adv == 0u) break;

        if (n_bvh >= P.trav_hi || n_adv == 0u) {
            // ================= traversal pass
            // object-space ray of every traversing lane: recomputed on entry (the wrapper chain is short and wave-uniform
            // per object) rather than kept alive through the advance passes
            RayT<T> r = ray;
            for (uint32_t oi = 0; oi < P.n_objects; oi++) {
                const bool here = phase == PH_BVH && my_oi == oi;
                if (__ballot(here) == 0) continue;
                const DObject ob = ld_obj(P.objects + oi);
                if (here) for (uint32_t k = 0; k < ob.n_ops; k++) op_fwd(ld_op(P.ops + ob.first_op + k), r);
            }
            const V3<T> inv = mk<T>(T(1.0) / r.d.x, T(1.0) / r.d.y, T(1.0) / r.d.z);      // AABB::hit's 1/d (aabb.rs:21), same value at every node
            if constexpr (Filt<T, FEATS>::on) {
                // ---- the filtered walk (bvh_hit_filt), resumable: tv_node is the lane's state word, across passes too
                const T t_min = TMin<T>::v();
                const bool act = phase == PH_BVH;
                BoxFilter F = make_filter(P.filter_m, r.o, inv, t_min, tv_closest);
                const bool tame = P.bvh_tame != 0u && __ballot(act && !(ray_is_tame(r.o, inv) && F.ok)) == 0ull;   // wave-uniform, this pass
                uint32_t stop_below = n_bvh * 3u / 4u;              // entered below trav_hi (nothing else to do): until a quarter has finished
                if (stop_below > P.trav_lo) stop_below = P.trav_lo;
                if (stop_below < 1u) stop_below = 1u;
                bool few = false;
                auto box_steps = [&](auto all) {
                    constexpr bool ALL = decltype(all)::value;
                    for (;;) {
                        const bool want_box = act && st_walking(tv_node);
                        const uint32_t n_box = (uint32_t)__popcll(__ballot(want_box)), n_leaf = (uint32_t)__popcll(__ballot(act && st_pending(tv_node)));
                        few = n_box + n_leaf < stop_below;
                        if (few || n_box == 0u || n_leaf * 64u >= P.trav_leaf * (n_box + n_leaf)) break;
                        n_steps++; n_step_lanes += n_box;
                        auto box_step = [&]() { const DFNode nd = fetch_fnode<ALL>(P, tv_node); tv_node = filter_pass(nd, F) ? nd.info : nd.skip; };
                        if (want_box) box_step();
                        repeat<BOX_STEPS_PERSIST - 1>([&]() {                     // more box steps under the same vote
                            const bool more = act && st_walking(tv_node);
                            n_steps++; n_step_lanes += (unsigned long long)__popcll(__ballot(more));
                            if (more) box_step();
                        });
                    }
                };
                for (;;) {
                    if (tame) { if (all_in_lds(P)) box_steps(std::true_type()); else box_steps(std::false_type()); }
                    else {
                        while (__ballot(act && st_walking(tv_node)) != 0ull) { if (act && st_walking(tv_node)) tv_node = exact_step(P, tv_node, r.o, inv, t_min, tv_closest); }
                        few = (uint32_t)__popcll(__ballot(act && st_pending(tv_node))) < stop_below;
                    }
                    if (few) break;                               // (pending leaves wait for the next traversal pass)
                    DIAG_ADD(0);
                    n_leaf_steps++; n_leaf_lanes += (unsigned long long)__popcll(__ballot(act && st_pending(tv_node)));
                    if (act && st_pending(tv_node)) {
                        const uint32_t leaf = tv_node & ~FNODE_LEAF;
                        const DBvhNode<T> lf = ld_node_at(P.bvh, leaf);
                        T t; uint32_t prim;
                        if ((!tame || box_inside_tame(lf, r.o, inv, t_min, tv_closest)) &&           // aabb.rs:19-36 on the leaf's own box
                            range_hit<T, FEATS>(P, (lf.a >> 28) & 7u, lf.a & 0x0FFFFFFFu, lf.b, r, t_min, tv_closest, t, prim, lf.b == 6u)) { tv_closest = t; tv_prim = prim; tv_any = true; F.c = up32(tv_closest); }
                        tv_node = fnode_skip(P, leaf);
                    }
                    DIAG_ADD(5);
                }
                if (act && tv_node == ST_DONE) {
                    // ---- this BVH is done: its result joins the list search (hit.rs:62-69), the lane moves to the next object
                    if (tv_any) { closest = tv_closest; id.obj = my_oi; id.prim = tv_prim; any_hit = true; }
                    my_oi++;
                    phase = PH_OBJ;
                }
                DIAG_ADD(0);
                continue;
            }
            const T t_min = TMin<T>::v();
            const bool tame = P.bvh_tame != 0u && __ballot(phase == PH_BVH && !ray_is_tame(r.o, inv)) == 0ull;   // wave-uniform, this pass
            uint32_t stop_below = n_bvh * 3u / 4u;                  // entered below trav_hi (nothing else to do): until a quarter has finished
            if (stop_below > P.trav_lo) stop_below = P.trav_lo;
            if (stop_below < 1u) stop_below = 1u;
            // Two kinds of step, chosen by vote so that each runs with many lanes: a box step (lanes that hold a node: bbox test,
            // next node; a lane that reaches a leaf keeps it pending and waits) and a leaf step (lanes with a pending leaf:
            // primitive tests).  Per lane the order bbox, left, right and the shrinking t_max are those of BVH::hit (bvh.rs:77-91) —
            // a lane never walks on before its pending leaf has been tested.  The box steps are an inner loop of their own with
            // nothing else in it (same shape as bvh_hit_ww's); a lane whose search ends just idles until the pass is over.
            const bool act = phase == PH_BVH;
            for (;;) {
                bool few = false;
                for (;;) {
                    const bool want_box = act && !tv_have_leaf && tv_node != BVH_DONE;
                    const uint32_t n_box = (uint32_t)__popcll(__ballot(want_box)), n_leaf = (uint32_t)__popcll(__ballot(act && tv_have_leaf));
                    few = n_box + n_leaf < stop_below;
                    if (few || n_box == 0u || n_leaf * 64u >= P.trav_leaf * (n_box + n_leaf)) break;
                    n_steps++; n_step_lanes += n_box;
                    auto box_step = [&]() {
                        const DBvhNode<T> nd = fetch_node(P, tv_node);
                        const bool inside = tame ? box_inside_tame(nd, r.o, inv, t_min, tv_closest) : box_inside_exact(nd, r.o, inv, t_min, tv_closest);
                        if (!near_first) {                                    // threaded preorder walk (bvh_hit_ww)
                            // a leaf whose box is hit stays the lane's node until the leaf step has tested it: its record is read again
                            // there (one more LDS read per leaf) instead of riding through the box steps in three registers
                            if (inside && (nd.a & BVH_LEAF)) tv_have_leaf = true;
                            else tv_node = inside ? nd.c : nd.skip;
                        } else if (inside && !(nd.a & BVH_LEAF)) {
                            const bool right_first = get(r.d, nd.a) < T(0);
                            stack[tv_sp * 64u] = right_first ? nd.c : nd.b;       // the farther child waits
                            tv_sp++;
                            tv_node = right_first ? nd.b : nd.c;
                        } else {
                            if (inside) { tv_have_leaf = true; tv_leaf_a = nd.a; tv_leaf_b = nd.b; tv_leaf_node = nd.c; }
                            // this node is finished (culled, or a leaf now pending): the next one comes off the stack
                            if (tv_sp == 0u) tv_node = BVH_DONE;
                            else { tv_sp--; tv_node = stack[tv_sp * 64u]; }
                        }
                    };
                    if (want_box) box_step();
#pragma unroll
                    for (int k = 1; k < BOX_STEPS_PERSIST; k++) {              // more box steps under the same vote (bvh_hit_ww)
                        const bool more = act && !tv_have_leaf && tv_node != BVH_DONE;
                        n_steps++; n_step_lanes += (unsigned long long)__popcll(__ballot(more));
                        if (more) box_step();
                    }
                }
                if (few) break;                                   // (pending leaves wait for the next traversal pass)
                DIAG_ADD(0);
                n_leaf_steps++; n_leaf_lanes += (unsigned long long)__popcll(__ballot(act && tv_have_leaf));
                if (act && tv_have_leaf) {
                    T t; uint32_t prim;
                    if (!near_first) {
                        const DBvhNode<T> lf = fetch_node(P, tv_node);
                        if (range_hit<T, FEATS>(P, (lf.a >> 28) & 7u, lf.a & 0x0FFFFFFFu, lf.b, r, t_min, tv_closest, t, prim)) { tv_closest = t; tv_prim = prim; tv_any = true; }
                        tv_node = lf.skip;
                    } else if (range_hit<T, FEATS>(P, (tv_leaf_a >> 28) & 7u, tv_leaf_a & 0x0FFFFFFFu, tv_leaf_b, r, t_min, tv_closest, t, prim) &&
                        bvh_accept(near_first, t, tv_closest, tv_leaf_node, tv_best)) { tv_closest = t; tv_prim = prim; tv_any = true; tv_best = tv_leaf_node; }
                    tv_have_leaf = false;
                }
                DIAG_ADD(5);
            }
            if (act && !tv_have_leaf && tv_node == BVH_DONE) {
                // ---- this BVH is done: its result joins the list search (hit.rs:62-69), the lane moves to the next object
                if (tv_any) { closest = tv_closest; id.obj = my_oi; id.prim = tv_prim; any_hit = true; }
                my_oi++;
                phase = PH_OBJ;
            }
            DIAG_ADD(0);
            continue;
        }

        // ================= advance pass
        n_iters++;
        n_active += (unsigned long long)__popcll(__ballot(phase == PH_OBJ || phase == PH_SHADE));
        // ---- lanes whose list search is complete: the rest of the level (main.rs:50-118), then the child level starts
        if (phase == PH_OBJ && my_oi >= P.n_objects) phase = PH_SHADE;
        if (phase == PH_SHADE) {
            bool done = false;
            V3<T> e = mk<T>(T(0), T(0), T(0));          // terminal radiance of this path (times beta)
            if (!any_hit) {
                e = ld3(P.background); done = true;                                         // main.rs:118
            } else {
                Rec<T> rec;
                finalize_hit<T, FEATS>(P, ray, closest, id, true, rec);
                shade_hit<T, FEATS>(P, rec, ray, beta, rng, depth_left, done, e);
            }
            if (done) {
                add_radiance(P, beta * e, acc, n_nonfinite, path_px, path_s);
                phase = PH_NEW;
            } else {
                phase = PH_OBJ; my_oi = 0; closest = Lim<T>::inf(); any_hit = false;
            }
        }
        DIAG_ADD(1);
        // ---- lanes whose path has ended take the next camera path from the wave's queue
        uint32_t new_px = 0;
        const bool got_new = take_new_paths(P, w, lane, q_real, q_u32, phase == PH_NEW, ray, rng, new_px, path_s);
        DIAG_ADD(2);
        // ---- lanes moving on to another pixel hand in their partial sum
        flush_acc<T, typename Shape<FEATS>::Acc, true>(P, got_new && acc_px != NONE_PX && acc_px != new_px, acc_px, acc, lane, n_flush);
        if (got_new) {
            if (acc_px != new_px) { acc_px = new_px; acc.set(0, 0.0); acc.set(1, 0.0); acc.set(2, 0.0); }
            path_px = new_px;
            beta = mk<T>(T(1.0), T(1.0), T(1.0));
            depth_left = P.max_depth;
            phase = PH_OBJ; my_oi = 0; closest = Lim<T>::inf(); any_hit = false;
            if (depth_left == 0u) {                     // main.rs:42-45: the sample is beta * 0
                add_radiance(P, beta * mk<T>(T(0), T(0), T(0)), acc, n_nonfinite, path_px, path_s);
                phase = PH_NEW;
            }
        }
        DIAG_ADD(3);
        // ---- world.hit (main.rs:48), resumable: objects in push order from my_oi up to the next BVH object or the end
        for (uint32_t oi = 0; oi < P.n_objects; oi++) {
            const bool here = phase == PH_OBJ && my_oi == oi;
            if (__ballot(here) == 0) continue;
            const DObject ob = ld_obj(P.objects + oi);
            if (here) {
                if (ob.geom_kind == G_BVH && (!(FEATS & F_MEDIUM) || ob.medium < 0)) {
                    phase = PH_BVH; tv_node = Filt<T, FEATS>::on ? state_of(P, ob.geom_first) : ob.geom_first; tv_sp = 0; tv_closest = closest; tv_any = false; tv_best = 0; tv_have_leaf = false;
                } else {
                    object_hit<T, FEATS>(P, oi, ob, ray, TMin<T>::v(), rng, closest, id, any_hit, stack);
                    my_oi = oi + 1u;
                }
            }
        }
        DIAG_ADD(4);
    }
    // ---- the queue is empty: hand in what is left
    flush_acc<T, typename Shape<FEATS>::Acc, true>(P, acc_px != NONE_PX, acc_px, acc, lane, n_flush);
    unsigned long long* st; { COLD_K; st = stats_row(PK(stats)); }
    write_stats(st, lane, n_nonfinite, n_iters, n_active, n_flush);
    if (st) {
        if (lane == 0) { atomicAdd(&st[9], n_steps + n_leaf_steps); atomicAdd(&st[10], n_step_lanes + n_leaf_lanes); atomicAdd(&st[12], n_leaf_steps); atomicAdd(&st[13], n_leaf_lanes); }
#ifdef RT_DIAG
        if (lane == 0) for (int k = 0; k < 6; k++) atomicAdd(&st[3 + k], dg_sum[k]);
#endif
    }
}

// ------------------------------------------------------------------ the kernel
template <typename T, uint32_t FEATS>
__global__ void __launch_bounds__(Shape<FEATS>::THREADS, Shape<FEATS>::WAVES_PER_SIMD) pathtrace_kernel(const KParams<T> P) {
    // dynamic LDS: [n_cached BVH nodes] [WAVES][regen_bytes(queue_entries)] camera-path queues [WAVES][stack_depth][64] BVH stacks
    typedef Shape<FEATS> S;
    const uint32_t lane = threadIdx.x & 63u, wave_in_block = threadIdx.x >> 6;
    const uint32_t nodes_bytes = P.n_cached * (uint32_t)(Filt<T, FEATS>::on ? sizeof(DFNode) : sizeof(DBvhNode<T>));
    if ((FEATS & F_BVH) && P.n_cached != 0u) {
        // the top of the BVH, once per workgroup: 16-byte pieces, consecutive threads consecutive pieces; the only barrier of the kernel.
        // (Round 3: one 16-byte slot of padding per node, so that the k-th pieces of different nodes spread over all sixteen slots of the
        // LDS bank row instead of four, was measured and is not faster — random spheres -1.4 %, final scene -1.5 %, teapot room -6 %, whose
        // tree then no longer fits: bank conflicts are not what a step waits for.)
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4* src = Filt<T, FEATS>::on ? (const u4*)P.bvh_f : (const u4*)P.bvh; u4* dst = (u4*)lds_raw;
        if (Filt<T, FEATS>::on && all_in_lds(P)) {
            // the whole tree is here: links become LDS addresses (a node state IS where its record lies: bvh_hit_filt); a leaf's info stays its id
            const uint32_t base = lds_base();
            for (uint32_t i = threadIdx.x; i < nodes_bytes / 16u; i += S::THREADS) {
                u4 w = src[i];
                if (i & 1u) {                                     // second half of a DFNode: max.z-side bounds, skip, info
                    if (w.z != ST_DONE) w.z = base + w.z * (uint32_t)sizeof(DFNode);
                    if (!(w.w & FNODE_LEAF)) w.w = base + w.w * (uint32_t)sizeof(DFNode);
                }
                dst[i] = w;
            }
        } else
        for (uint32_t i = threadIdx.x; i < nodes_bytes / 16u; i += S::THREADS) dst[i] = src[i];
        __syncthreads();
    }
    const uint32_t QN = P.queue_entries;
    unsigned char* regen = lds_raw + nodes_bytes + wave_in_block * regen_bytes(QN);
    T* q_real = (T*)regen;                                         // [7][QN]: o.x o.y o.z d.x d.y d.z time
    uint32_t* q_u32 = (uint32_t*)(regen + 7u * QN * sizeof(T));    // [6][QN]: rng s0..s3, local pixel, sample
    uint32_t* stack = (uint32_t*)(lds_raw + nodes_bytes + S::WAVES * regen_bytes(QN)) + wave_in_block * (P.stack_depth * 64u) + lane;
    if (FEATS & F_PERSIST) trace_resumable<T, FEATS>(P, lane, q_real, q_u32, stack);
    else trace_lockstep<T, FEATS>(P, lane, q_real, q_u32, stack);
}

// ------------------------------------------------------------------ launch
template <typename T, uint32_t FEATS>
static hipError_t allow_lds(size_t shmem) {            // more than the default 64 KB of dynamic LDS needs to be asked for
    if (shmem <= 65536u) return hipSuccess;
    return hipFuncSetAttribute((const void*)pathtrace_kernel<T, FEATS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}
template <typename T, uint32_t FEATS>
static hipError_t launch_one(const KParams<T>& P, uint32_t n_blocks, size_t shmem, hipStream_t stream) {
    hipError_t e = allow_lds<T, FEATS>(shmem);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pathtrace_kernel<T, FEATS>), dim3(n_blocks), dim3(Shape<FEATS>::THREADS), shmem, stream, P);
    return hipGetLastError();
}
template <typename T, uint32_t FEATS>
static int occupancy_one(size_t shmem) {
    int nb = 0;
    if (allow_lds<T, FEATS>(shmem) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pathtrace_kernel<T, FEATS>, (int)Shape<FEATS>::THREADS, shmem) != hipSuccess) return 0;
    if (Shape<FEATS>::ONE_PER_CU && nb > 1) nb = 1;       // the register budget is set for exactly one such workgroup per CU
    return nb;
}
template <uint32_t FEATS> static LaunchShape shape_one() { LaunchShape g; g.threads = Shape<FEATS>::THREADS; g.queue_entries = Shape<FEATS>::QN_MIN; g.one_per_cu = Shape<FEATS>::ONE_PER_CU; return g; }

// Instantiations per arithmetic type, leanest first: rects + instances + Lambertian/Metal/DiffuseLight (everything the
// Cornell box needs; 5 waves/SIMD), the same plus BVH + triangles (mesh scenes such as the teapot room; 3 waves/SIMD),
// everything but the principled material (3 waves/SIMD with some spilling: measured 6 % faster on the final scene than
// 2 waves/SIMD without), and everything (2 waves/SIMD); the BVH ones also with near-first traversal.
static const uint32_t FEATS_LEAN = 0u;
[[maybe_unused]] static const uint32_t FEATS_MESH = F_BVH | F_TRIS;
[[maybe_unused]] static const uint32_t FEATS_NO_PBR = F_ALL & ~F_PBR;

// lean kernels: their own translation unit in the product build
template <typename T> hipError_t launch_lean(const KParams<T>& P, uint32_t n_blocks, size_t shmem, hipStream_t stream);
template <typename T> int occupancy_lean(size_t shmem);
static LaunchShape shape_lean() { return shape_one<FEATS_LEAN>(); }
#if RT_TU != 2
template <typename T> hipError_t launch_lean(const KParams<T>& P, uint32_t n_blocks, size_t shmem, hipStream_t stream) { return launch_one<T, FEATS_LEAN>(P, n_blocks, shmem, stream); }
template <typename T> int occupancy_lean(size_t shmem) { return occupancy_one<T, FEATS_LEAN>(shmem); }
template hipError_t launch_lean<double>(const KParams<double>&, uint32_t, size_t, hipStream_t);
template hipError_t launch_lean<float>(const KParams<float>&, uint32_t, size_t, hipStream_t);
template int occupancy_lean<double>(size_t);
template int occupancy_lean<float>(size_t);
// Known-answer access to the list-scene kernels' closest-hit search (rt_debug_list_hit): world.hit (main.rs:48) + the hit record for
// given rays — world_hit<double, 0> and finalize_hit<double, 0>, the code the frames run, one ray per lane, waves of 64 (the search
// votes).  out[12 i ..] = hit (0 / 1), t, position[3], normal[3], front_face, object, primitive, material.
__global__ void list_hit_kat_kernel(const KParams<double> P, uint32_t n, const double* rays, const double* tlim, double* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = i < n ? i : n - 1u;                           // (a partial last wave repeats the last ray: every lane takes part in the votes)
    RayT<double> ray; ray.o = mk<double>(rays[j * 6], rays[j * 6 + 1], rays[j * 6 + 2]); ray.d = mk<double>(rays[j * 6 + 3], rays[j * 6 + 4], rays[j * 6 + 5]); ray.tm = 0.0;
    Rng rng = rng_for_path(0ull, 0u, 0u);
    double t_hit; HitId id; id.obj = 0u; id.prim = 0u;
    const bool any = world_hit_list<double>(P, ray, tlim[j], rng, t_hit, id);
    Rec<double> rec; rec.p = mk<double>(0.0, 0.0, 0.0); rec.n = rec.p; rec.t = 0.0; rec.u = rec.v = 0.0; rec.front = false; rec.mat = 0u;
    if (any) finalize_hit<double, 0u>(P, ray, t_hit, id, true, rec);
    if (i < n) {
        double* o = out + 12ull * i;
        o[0] = any ? 1.0 : 0.0; o[1] = any ? rec.t : 0.0; o[2] = rec.p.x; o[3] = rec.p.y; o[4] = rec.p.z; o[5] = rec.n.x; o[6] = rec.n.y; o[7] = rec.n.z;
        o[8] = rec.front ? 1.0 : 0.0; o[9] = any ? (double)id.obj : -1.0; o[10] = any ? (double)(id.prim & 0x0FFFFFFFu) : -1.0; o[11] = any ? (double)rec.mat : -1.0;
    }
}
hipError_t launch_list_hit_kat(const KParams<double>& P, uint32_t n, const double* d_rays, const double* d_tlim, double* d_out, hipStream_t stream) {
    hipLaunchKernelGGL(list_hit_kat_kernel, dim3((n + 63u) / 64u), dim3(64), 0, stream, P, n, d_rays, d_tlim, d_out);
    return hipGetLastError();
}
#endif

#if defined(RT_KRES_ONLY)      // tools/kres.py: one instantiation only (compile-time exploration, never the product build)
template __global__ void pathtrace_kernel<double, RT_KRES_ONLY>(const KParams<double>);
#elif RT_TU != 1
template <typename T, typename F, typename L> static auto dispatch(uint32_t scene_feats, uint32_t flags, L&& lean, F&& f) {
    const bool nf = (flags & 8u) && (scene_feats & F_BVH);          // RT_NEAR_FIRST_BVH
    const bool ps = (flags & 16u) && (scene_feats & F_BVH);         // RT_PERSISTENT_BVH
    if ((scene_feats & ~FEATS_LEAN) == 0u) return lean();
    if (scene_feats & F_NESTED) return f(std::integral_constant<uint32_t, F_ALL | F_NESTED>());      // any Hittable as a BVH leaf: one instantiation (reference order, lock-step)
    if ((scene_feats & ~FEATS_MESH) == 0u) {
        if (ps) return nf ? f(std::integral_constant<uint32_t, FEATS_MESH | F_PERSIST | F_NEAR_FIRST>()) : f(std::integral_constant<uint32_t, FEATS_MESH | F_PERSIST>());
        return nf ? f(std::integral_constant<uint32_t, FEATS_MESH | F_NEAR_FIRST>()) : f(std::integral_constant<uint32_t, FEATS_MESH>());
    }
    if ((scene_feats & ~FEATS_NO_PBR) == 0u) {
        if ((flags & 1024u) && !nf && !ps) return f(std::integral_constant<uint32_t, FEATS_NO_PBR | F_SPEC>());      // RT_SPECULATE_BVH
        if (ps && !nf) return f(std::integral_constant<uint32_t, FEATS_NO_PBR | F_PERSIST>());
        return nf ? f(std::integral_constant<uint32_t, FEATS_NO_PBR | F_NEAR_FIRST>()) : f(std::integral_constant<uint32_t, FEATS_NO_PBR>());
    }
    return nf ? f(std::integral_constant<uint32_t, F_ALL | F_NEAR_FIRST>()) : f(std::integral_constant<uint32_t, F_ALL>());
}
template <typename T> hipError_t launch_pathtrace(const KParams<T>& P, uint32_t scene_feats, uint32_t n_blocks, size_t shmem, hipStream_t stream) {
    return dispatch<T>(scene_feats, P.flags, [&]() { return launch_lean<T>(P, n_blocks, shmem, stream); },
                       [&](auto feats) { return launch_one<T, decltype(feats)::value>(P, n_blocks, shmem, stream); });
}
template <typename T> int pathtrace_blocks_per_cu(uint32_t scene_feats, uint32_t flags, size_t shmem) {
    return dispatch<T>(scene_feats, flags, [&]() { return occupancy_lean<T>(shmem); },
                       [&](auto feats) { return occupancy_one<T, decltype(feats)::value>(shmem); });
}
LaunchShape pathtrace_shape(uint32_t scene_feats, uint32_t flags) {
    return dispatch<double>(scene_feats, flags, [&]() { return shape_lean(); }, [&](auto feats) { return shape_one<decltype(feats)::value>(); });
}
// the FEATS template argument of the instantiation that serves (scene_feats, flags): the kernel's name is pathtrace_kernel<T, that>
uint32_t pathtrace_feats(uint32_t scene_feats, uint32_t flags) {
    return dispatch<double>(scene_feats, flags, [&]() { return FEATS_LEAN; }, [&](auto feats) { return (uint32_t) decltype(feats)::value; });
}

template hipError_t launch_pathtrace<double>(const KParams<double>&, uint32_t, uint32_t, size_t, hipStream_t);
template hipError_t launch_pathtrace<float>(const KParams<float>&, uint32_t, uint32_t, size_t, hipStream_t);
template int pathtrace_blocks_per_cu<double>(uint32_t, uint32_t, size_t);
template int pathtrace_blocks_per_cu<float>(uint32_t, uint32_t, size_t);
#endif

} // namespace rt

#if RT_TU != 1
// ------------------------------------------------------------------ known-answer access to AABB::hit on the device (tests only)
namespace rt {
__global__ void aabb_kat_kernel(uint32_t n, const double* boxes, const double* rays, const double* tlim, int* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DBvhNode<double> nd;
    for (int k = 0; k < 3; k++) { nd.mn[k] = boxes[i * 6 + k]; nd.mx[k] = boxes[i * 6 + 3 + k]; }
    nd.a = nd.b = nd.c = nd.skip = 0;
    const V3<double> o = mk<double>(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]);
    const V3<double> d = mk<double>(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    const V3<double> inv = mk<double>(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    const bool exact = box_inside_exact(nd, o, inv, tlim[i * 2], tlim[i * 2 + 1]);
    const bool tame_ray = ray_is_tame(o, inv);
    const bool tame = box_inside_tame(nd, o, inv, tlim[i * 2], tlim[i * 2 + 1]);
    // the filtered walk's box step on this very box (outward-rounded to f32, M = its own largest |coordinate|: the tightest margin
    // any node containing it can have): must pass wherever the exact test does, for rays inside the filter's ranges
    DFNode fn; float m = 1.0f;
    for (int k = 0; k < 3; k++) {
        float lo = (float)nd.mn[k], hi = (float)nd.mx[k];
        if ((double)lo > nd.mn[k]) lo = ::nextafterf(lo, -__builtin_inff());
        if ((double)hi < nd.mx[k]) hi = ::nextafterf(hi, __builtin_inff());
        fn.b[2 * k] = lo; fn.b[2 * k + 1] = hi;
        m = __builtin_fmaxf(m, __builtin_fmaxf(__builtin_fabsf(lo), __builtin_fabsf(hi)));
    }
    fn.skip = fn.info = 0u;
    const bool box_ok = m <= 0x1p40f;                  // (a NaN / infinite coordinate fails this: rt_flatten.cpp make_filter_nodes)
    const BoxFilter F = make_filter(box_ok ? m : 0.0f, o, inv, tlim[i * 2], tlim[i * 2 + 1]);
    // the same for the f32 kernels: the exact form in f32 on the box rounded to nearest (what their leaf steps run) against the filter they build
    DBvhNode<float> nf;
    for (int k = 0; k < 3; k++) { nf.mn[k] = (float)nd.mn[k]; nf.mx[k] = (float)nd.mx[k]; }
    nf.a = nf.b = nf.c = nf.skip = 0;
    const V3<float> of = mk<float>((float)o.x, (float)o.y, (float)o.z), df = mk<float>((float)d.x, (float)d.y, (float)d.z);
    const V3<float> invf = mk<float>(1.0f / df.x, 1.0f / df.y, 1.0f / df.z);
    const float tmin_f = (float)tlim[i * 2], tmax_f = (float)tlim[i * 2 + 1];
    const bool exact_f = box_inside_exact(nf, of, invf, tmin_f, tmax_f);
    const bool tame_f = ray_is_tame(of, invf);
    const BoxFilter Ff = make_filter(box_ok ? m : 0.0f, of, invf, tmin_f, tmax_f);
    out[i] = (exact ? 1 : 0) | (tame ? 2 : 0) | (tame_ray ? 4 : 0) | (F.ok ? 8 : 0) | (filter_pass(fn, F) ? 16 : 0)
             | (exact_f ? 32 : 0) | ((tame_f && Ff.ok) ? 64 : 0) | (filter_pass(fn, Ff) ? 128 : 0);
}
}
// out[i]: bit 0 = AABB::hit by the exact form, bit 1 = by the NaN-free form, bit 2 = the ray qualifies for the NaN-free form,
// bit 3 = ray and box are inside the f32 filter's ranges, bit 4 = the filter (rt_kernel.hip: filter_pass) lets the box through;
// bits 5, 6, 7 = the same three for the f32 kernels (the exact form in f32 on the inputs rounded to nearest, their ranges, their filter).
// boxes: n x (min[3], max[3]); rays: n x (origin[3], direction[3]); tlim: n x (t_min, t_max).  Host pointers.
extern "C" int rt_debug_aabb_hit(uint32_t n, const double* boxes, const double* rays, const double* tlim, int* out) {
    if (n == 0) return 0;
    double *db = nullptr, *dr = nullptr, *dt = nullptr; int* dout = nullptr;
    int rc = -1;
    if (hipMalloc(&db, n * 48ull) == hipSuccess && hipMalloc(&dr, n * 48ull) == hipSuccess && hipMalloc(&dt, n * 16ull) == hipSuccess &&
        hipMalloc(&dout, n * sizeof(int)) == hipSuccess &&
        hipMemcpy(db, boxes, n * 48ull, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dr, rays, n * 48ull, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(dt, tlim, n * 16ull, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(rt::aabb_kat_kernel, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, n, db, dr, dt, dout);
        if (hipGetLastError() == hipSuccess && hipMemcpy(out, dout, n * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    (void)hipFree(db); (void)hipFree(dr); (void)hipFree(dt); (void)hipFree(dout);
    return rc;
}
// Known-answer access to the Cube fast path: per case the six exact rect tests in cube.rs order (the reference) and, beside them,
// what cube_fast says.  out[4 i ..]: t of the six tests (NaN: no face accepted), its face (0..5, -1), t of the fast
// path (NaN: clear and no hit; only meaningful when clear), 8 * clear + face + 1 of the fast path (face + 1 = 0: no hit).
namespace rt {
__global__ void cube_kat_kernel(uint32_t n, float rect_m, const double* boxes, const double* rays, const double* tlim, double* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    struct { double mnx, mny, mnz, mxx, mxy, mxz; } b; b.mnx = boxes[i * 6]; b.mny = boxes[i * 6 + 1]; b.mnz = boxes[i * 6 + 2]; b.mxx = boxes[i * 6 + 3]; b.mxy = boxes[i * 6 + 4]; b.mxz = boxes[i * 6 + 5];
    RayT<double> ray; ray.o = mk<double>(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]); ray.d = mk<double>(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]); ray.tm = 0.0;
    const double t_min = tlim[i * 2], t_max = tlim[i * 2 + 1];
    // the reference: Cube::new's six AARects (cube.rs:17-24) under HittableList::hit
    DRect<double> f[6] = {{b.mnx, b.mxx, b.mny, b.mxy, b.mxz, 0u, 0u}, {b.mnx, b.mxx, b.mny, b.mxy, b.mnz, 0u, 0u}, {b.mnx, b.mxx, b.mnz, b.mxz, b.mxy, 1u, 0u},
                          {b.mnx, b.mxx, b.mnz, b.mxz, b.mny, 1u, 0u}, {b.mny, b.mxy, b.mnz, b.mxz, b.mxx, 2u, 0u}, {b.mny, b.mxy, b.mnz, b.mxz, b.mnx, 2u, 0u}};
    double closest = t_max, t_ref = __builtin_nan(""); int face_ref = -1;
    for (int k = 0; k < 6; k++) { double t; if (rect_test(f[k], ray, t_min, closest, t)) { closest = t; t_ref = t; face_ref = k; } }
    double t_fast = __builtin_nan(""); uint32_t face = 0xFFFFFFFFu; bool hit = false, clear = false;
    { double t; uint32_t fc = 0u; (void)cube_fast<false>(rect_m, b.mnx, b.mxx, b.mny, b.mxy, b.mnz, b.mxz, ray, t_min, t_max, t, fc, hit, clear); if (hit) { t_fast = t; face = fc; } }
    out[i * 4] = t_ref; out[i * 4 + 1] = (double)face_ref; out[i * 4 + 2] = t_fast; out[i * 4 + 3] = (double)((clear ? 8 : 0) + (int)(face + 1u));
}
}
// The same for the ROOM form: masks[i] says which of the six faces exist (bit f = face f in cube.rs:17-24 order); the reference side is
// HittableList::hit over the rects of those faces, the fast side cube_fast with a face map (a face's place in the map = its own index).
namespace rt {
__global__ void room_kat_kernel(uint32_t n, float rect_m, const double* boxes, const double* rays, const double* tlim, const uint32_t* masks, double* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    struct { double mnx, mny, mnz, mxx, mxy, mxz; } b; b.mnx = boxes[i * 6]; b.mny = boxes[i * 6 + 1]; b.mnz = boxes[i * 6 + 2]; b.mxx = boxes[i * 6 + 3]; b.mxy = boxes[i * 6 + 4]; b.mxz = boxes[i * 6 + 5];
    RayT<double> ray; ray.o = mk<double>(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]); ray.d = mk<double>(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]); ray.tm = 0.0;
    const double t_min = tlim[i * 2], t_max = tlim[i * 2 + 1];
    const uint32_t mask = masks[i] & 0x3Fu;
    DRect<double> f[6] = {{b.mnx, b.mxx, b.mny, b.mxy, b.mxz, 0u, 0u}, {b.mnx, b.mxx, b.mny, b.mxy, b.mnz, 0u, 0u}, {b.mnx, b.mxx, b.mnz, b.mxz, b.mxy, 1u, 0u},
                          {b.mnx, b.mxx, b.mnz, b.mxz, b.mny, 1u, 0u}, {b.mny, b.mxy, b.mnz, b.mxz, b.mxx, 2u, 0u}, {b.mny, b.mxy, b.mnz, b.mxz, b.mnx, 2u, 0u}};
    double closest = t_max, t_ref = __builtin_nan(""); int face_ref = -1;
    uint32_t map = 0u;
    for (int k = 0; k < 6; k++) {
        const bool has = ((mask >> k) & 1u) != 0u;
        map |= (has ? (uint32_t)k : 7u) << (3 * k);
        double t; if (has && rect_test(f[k], ray, t_min, closest, t)) { closest = t; t_ref = t; face_ref = k; }
    }
    map |= 1u << 18;        // (never 0: a Cube with all six faces is a valid room too — the word the flattener makes has its flag bits above the map as well)
    double t_fast = __builtin_nan(""); uint32_t face = 0xFFFFFFFFu; bool hit = false, clear = false;
    { double t; uint32_t fc = 0u; (void)cube_fast<false>(rect_m, b.mnx, b.mxx, b.mny, b.mxy, b.mnz, b.mxz, ray, t_min, t_max, t, fc, hit, clear, map); if (hit) { t_fast = t; face = fc; } }
    out[i * 4] = t_ref; out[i * 4 + 1] = (double)face_ref; out[i * 4 + 2] = t_fast; out[i * 4 + 3] = (double)((clear ? 8 : 0) + (int)(face + 1u));
}
}
extern "C" int rt_debug_room_hit(uint32_t n, double rect_m, const double* boxes, const double* rays, const double* tlim, const uint32_t* masks, double* out) {
    if (n == 0) return 0;
    double *db = nullptr, *dr = nullptr, *dt = nullptr, *dout = nullptr; uint32_t* dm = nullptr;
    int rc = -1;
    if (hipMalloc(&db, n * 48ull) == hipSuccess && hipMalloc(&dr, n * 48ull) == hipSuccess && hipMalloc(&dt, n * 16ull) == hipSuccess &&
        hipMalloc(&dout, n * 32ull) == hipSuccess && hipMalloc(&dm, n * 4ull) == hipSuccess &&
        hipMemcpy(db, boxes, n * 48ull, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dr, rays, n * 48ull, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(dt, tlim, n * 16ull, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dm, masks, n * 4ull, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(rt::room_kat_kernel, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, n, (float)rect_m, db, dr, dt, dm, dout);
        if (hipGetLastError() == hipSuccess && hipMemcpy(out, dout, n * 32ull, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    (void)hipFree(db); (void)hipFree(dr); (void)hipFree(dt); (void)hipFree(dout); (void)hipFree(dm);
    return rc;
}
extern "C" int rt_debug_cube_hit(uint32_t n, double rect_m, const double* boxes, const double* rays, const double* tlim, double* out) {
    if (n == 0) return 0;
    double *db = nullptr, *dr = nullptr, *dt = nullptr, *dout = nullptr;
    int rc = -1;
    if (hipMalloc(&db, n * 48ull) == hipSuccess && hipMalloc(&dr, n * 48ull) == hipSuccess && hipMalloc(&dt, n * 16ull) == hipSuccess &&
        hipMalloc(&dout, n * 32ull) == hipSuccess &&
        hipMemcpy(db, boxes, n * 48ull, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dr, rays, n * 48ull, hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(dt, tlim, n * 16ull, hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(rt::cube_kat_kernel, dim3((n + 255u) / 256u), dim3(256), 0, nullptr, n, (float)rect_m, db, dr, dt, dout);
        if (hipGetLastError() == hipSuccess && hipMemcpy(out, dout, n * 32ull, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    }
    (void)hipFree(db); (void)hipFree(dr); (void)hipFree(dt); (void)hipFree(dout);
    return rc;
}
#endif

// csrc/rt_rng.h — the seeded random stream of the MI355X path tracer (host + device).
//
// The reference draws from rand 0.8.5 `thread_rng()` everywhere (src/main.rs:813, src/vec.rs:71,97,
// src/camera.rs:56, src/pdf.rs:9,168, src/mat.rs:356, src/hit.rs:95, src/rect.rs:104,
// src/medium.rs:28, src/perlin.rs:5,22): OS-seeded, so no two runs agree.  A GPU needs one
// independent, reproducible stream per camera path.  Spec (also in DESIGN.md §RNG):
//
//   generator  xoshiro128++, 32-bit output, four u32 of state per lane (adds / xors / rotates only:
//              no integer multiplies on the per-draw path, which are quarter-rate on CDNA)
//   keying     id = (pixel << 32) | sample; z = seed + 2*id*G (G = 0x9E3779B97F4A7C15);
//              (s0,s1) = lo/hi of mix64(z + G), (s2,s3) = lo/hi of mix64(z + 2G)   [SplitMix64 finaliser]
//              pixel = output-order pixel index (row 0 = top); pixel 0xFFFFFFFF = host streams.
//              The factor 2 shifts bit 31 of `pixel` out of the 64-bit word, so pixels p and p + 2^31 share a stream and the
//              host streams' key equals that of pixel 0x7FFFFFFF: frames are therefore limited to W*H <= 0x7FFFFFFF pixels
//              (indices <= 0x7FFFFFFE; rt_render* refuse larger frames), which keeps every path key and the host key distinct.
//   next_u64   (next_u32 << 32) | next_u32
//   U01        rng.gen::<f64>()      = (next_u64 >> 11) * 2^-53
//   R(a,b)     rng.gen_range(a..b)   = (bits(0x3FF<<52 | next_u64 >> 12) - 1.0) * (b - a) + a,
//              a result >= b (p ~ 2^-53) becomes the largest value below b
//   B          rng.gen::<bool>()     = top bit of next_u32
//   I(n)       gen_range(0..n)/choose = (next_u32 * n) >> 32
//   f32 mode   U01 = (next_u32 >> 8) * 2^-24;  R(a,b) = (bits(0x7F<<23 | next_u32 >> 9) - 1) * (b-a) + a
//              (rand 0.8.5's f32 forms), one u32 per draw.
// Draw ORDER along a path is the reference's.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt {

struct Rng {
    uint32_t s0, s1, s2, s3;
};

RT_HD uint32_t rng_rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

RT_HD uint64_t rng_mix64(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

RT_HD Rng rng_for_path(uint64_t seed, uint32_t pixel, uint32_t sample) {
    const uint64_t G = 0x9E3779B97F4A7C15ULL;
    uint64_t id = ((uint64_t)pixel << 32) | (uint64_t)sample;
    uint64_t z = seed + 2ULL * id * G;
    uint64_t a = rng_mix64(z + G);
    uint64_t b = rng_mix64(z + 2ULL * G);
    Rng r;
    r.s0 = (uint32_t)a; r.s1 = (uint32_t)(a >> 32);
    r.s2 = (uint32_t)b; r.s3 = (uint32_t)(b >> 32);
    if ((r.s0 | r.s1 | r.s2 | r.s3) == 0) r.s0 = 1;
    return r;
}

RT_HD Rng rng_for_stream(uint64_t seed, uint32_t stream) { return rng_for_path(seed, 0xFFFFFFFFu, stream); }

RT_HD uint32_t rng_u32(Rng& r) {
    uint32_t result = rng_rotl(r.s0 + r.s3, 7) + r.s0;
    uint32_t t = r.s1 << 9;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rng_rotl(r.s3, 11);
    return result;
}

RT_HD uint64_t rng_u64(Rng& r) {
    uint64_t hi = rng_u32(r);
    uint64_t lo = rng_u32(r);
    return (hi << 32) | lo;
}

RT_HD bool rng_bool(Rng& r) { return (rng_u32(r) >> 31) != 0; }
RT_HD uint32_t rng_index(Rng& r, uint32_t n) { return (uint32_t)(((uint64_t)rng_u32(r) * (uint64_t)n) >> 32); }

// ---- f64 draws
RT_HD double rng_u01(Rng& r, double) {
    uint64_t v = rng_u64(r) >> 11;                 // 53 bits; both halves convert exactly
    double hi = (double)(uint32_t)(v >> 32);
    double lo = (double)(uint32_t)v;
    return (hi * 4294967296.0 + lo) * 0x1.0p-53;
}
RT_HD double rng_range(Rng& r, double a, double b) {
    uint64_t bits = 0x3FF0000000000000ULL | (rng_u64(r) >> 12);
    double v12;
#if defined(__HIP_DEVICE_COMPILE__)
    v12 = __longlong_as_double((long long)bits);
#else
    __builtin_memcpy(&v12, &bits, 8);
#endif
    double res = (v12 - 1.0) * (b - a) + a;
    if (!(res < b)) {
        // largest double below b (b finite, b > a): step one ulp towards a
        uint64_t bb;
#if defined(__HIP_DEVICE_COMPILE__)
        bb = (uint64_t)__double_as_longlong(b);
#else
        __builtin_memcpy(&bb, &b, 8);
#endif
        if (b > 0.0) bb -= 1; else if (b < 0.0) bb += 1; else bb = 0x8000000000000001ULL;
#if defined(__HIP_DEVICE_COMPILE__)
        res = __longlong_as_double((long long)bb);
#else
        __builtin_memcpy(&res, &bb, 8);
#endif
    }
    return res;
}

// ---- f32 draws (RT_F32 kernels only)
RT_HD float rng_u01(Rng& r, float) { return (float)(rng_u32(r) >> 8) * 0x1.0p-24f; }
RT_HD float rng_range(Rng& r, float a, float b) {
    uint32_t bits = 0x3F800000u | (rng_u32(r) >> 9);
    float v12;
#if defined(__HIP_DEVICE_COMPILE__)
    v12 = __uint_as_float(bits);
#else
    __builtin_memcpy(&v12, &bits, 4);
#endif
    float res = (v12 - 1.0f) * (b - a) + a;
    if (!(res < b)) {
        uint32_t bb;
#if defined(__HIP_DEVICE_COMPILE__)
        bb = __float_as_uint(b);
#else
        __builtin_memcpy(&bb, &b, 4);
#endif
        if (b > 0.0f) bb -= 1; else if (b < 0.0f) bb += 1; else bb = 0x80000001u;
#if defined(__HIP_DEVICE_COMPILE__)
        res = __uint_as_float(bb);
#else
        __builtin_memcpy(&res, &bb, 4);
#endif
    }
    return res;
}

} // namespace rt

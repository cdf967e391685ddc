// oracle/orc_rng.h — TEST INFRASTRUCTURE ONLY (see oracle/oracle.cpp header).
//
// The seeded random stream shared (by specification, not by code) between the
// CPU oracle and the HIP kernel.  The reference draws every random number from
// rand 0.8.5's `thread_rng()` (OS-seeded ChaCha12; /root/reference/src/main.rs:813,
// src/vec.rs:71,97, src/camera.rs:56, src/pdf.rs:9,168, src/mat.rs:356, src/hit.rs:95,
// src/rect.rs:104, src/medium.rs:28, src/perlin.rs:5,22) so its stream cannot be
// reproduced by anybody; what CAN be kept is the *distribution* of every draw kind
// and the *order* of draws along a path.  This header is the
// oracle's own implementation of that spec; the product's is csrc/rt_rng.h.
//
// Generator: xoshiro128++ (Blackman & Vigna), 32-bit output, 128-bit state.
// Stream key: SplitMix64 outputs number 2*id+1 and 2*id+2 of the sequence started
// at `seed`, id = (pixel << 32) | sample, so distinct (pixel, sample) pairs get
// disjoint SplitMix outputs.
// Draw kinds (rand 0.8.5 semantics, crate not vendored under /root/reference):
//   U01      gen::<f64>()          (next_u64 >> 11) * 2^-53                [Standard, 53 bits]
//   R(a,b)   gen_range(a..b) f64   ((bits(0x3FF<<52 | next_u64>>12)) - 1.0) * (b-a) + a,
//                                  result >= b (probability ~2^-53) is replaced by the
//                                  largest double below b                 [UniformFloat::sample_single]
//   B        gen::<bool>()         top bit of next_u32                    [Standard for bool]
//   I(n)     gen_range(0..n) / choose   (next_u32 * n) >> 32  (widening multiply, no rejection)
//   next_u64 = (next_u32 << 32) | next_u32   (first draw is the high word)
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>

namespace orc {

struct Rng {
    uint32_t s[4];

    static inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

    static inline uint64_t mix64(uint64_t x) {
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
        return x ^ (x >> 31);
    }

    // stream for one camera path
    static inline Rng for_path(uint64_t seed, uint32_t pixel, uint32_t sample) {
        const uint64_t G = 0x9E3779B97F4A7C15ULL;
        uint64_t id = ((uint64_t)pixel << 32) | (uint64_t)sample;
        uint64_t z = seed + 2ULL * id * G;
        uint64_t a = mix64(z + G);
        uint64_t b = mix64(z + 2ULL * G);
        Rng r;
        r.s[0] = (uint32_t)a; r.s[1] = (uint32_t)(a >> 32);
        r.s[2] = (uint32_t)b; r.s[3] = (uint32_t)(b >> 32);
        if ((r.s[0] | r.s[1] | r.s[2] | r.s[3]) == 0) r.s[0] = 1;
        return r;
    }
    // host-side streams (scene construction): pixel 0xFFFFFFFF is never a real pixel; its key equals pixel 0x7FFFFFFF's (the factor
    // 2 drops bit 31), which frames of at most 2^31 - 1 pixels never reach (the product refuses larger frames)
    static inline Rng for_stream(uint64_t seed, uint32_t stream) { return for_path(seed, 0xFFFFFFFFu, stream); }

    inline uint32_t next_u32() {
        uint32_t result = rotl(s[0] + s[3], 7) + s[0];
        uint32_t t = s[1] << 9;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 11);
        return result;
    }
    inline uint64_t next_u64() {
        uint64_t hi = next_u32();
        uint64_t lo = next_u32();
        return (hi << 32) | lo;
    }
    inline double u01() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
    inline double range(double a, double b) {
        uint64_t bits = 0x3FF0000000000000ULL | (next_u64() >> 12);
        double v12;
        std::memcpy(&v12, &bits, 8);
        double res = (v12 - 1.0) * (b - a) + a;
        if (!(res < b)) res = std::nextafter(b, a);
        return res;
    }
    inline bool boolean() { return (next_u32() >> 31) != 0; }
    inline uint32_t index(uint32_t n) { return (uint32_t)(((uint64_t)next_u32() * (uint64_t)n) >> 32); }
};

} // namespace orc

// tools/ubench/ubench.hip — VALU instruction throughput on gfx950 (developer tool, not part of the product).
// Each kernel runs iters x 16 independent instances of one operation per lane with EXACTLY 4 waves on every SIMD: one workgroup of 1024
// threads per CU (128 KB of dynamic LDS keeps a second one off the CU), as many workgroups as the chip has CUs.  Nothing depends on an
// assumed frequency: every wave reads two counters around its loop,
//   * s_memtime  — the shader-clock counter: its rate against the reference counter is the clock the chip really ran at under this load;
//   * s_memrealtime — the constant 100 MHz reference counter: the time base.  The cost of one operation is the kernel's event time (launches are
//     calibrated to ~20 ms) divided by the operations one SIMD issued, in ns and — times the measured clock — in SIMD issue cycles.
// Output is one CSV row per operation; `make -C tools/ubench run > profiles/rNN_ubench.csv` on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(name, TYPE, INIT, OP) \
__global__ void __launch_bounds__(1024) k_##name(TYPE* out, unsigned long long* ticks, TYPE seed, int iters) { \
    TYPE a[16]; for (int i = 0; i < 16; i++) a[i] = INIT; \
    __builtin_amdgcn_sched_barrier(0); const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(); const unsigned long long t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); \
    for (int it = 0; it < iters; it++) { _Pragma("unroll") for (int i = 0; i < 16; i++) { OP; } } \
    TYPE s = a[0]; for (int i = 1; i < 16; i++) s = s + a[i]; \
    asm volatile("" :: "v"(s)); \
    __builtin_amdgcn_sched_barrier(0); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); const unsigned long long r1 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
    if ((threadIdx.x & 63u) == 0u) { ticks[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = t1 - t0; ticks[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) + 1] = r1 - r0; } }

KERNEL(fma_f64, double, seed + i + threadIdx.x, a[i] = __builtin_fma(a[i], 1.0000001, 0.5))
KERNEL(mul_f64, double, seed + i + threadIdx.x, a[i] = a[i] * 1.0000001)
KERNEL(add_f64, double, seed + i + threadIdx.x, a[i] = a[i] + 1.5)
KERNEL(minmax_f64, double, seed + i + threadIdx.x, a[i] = __builtin_fmin(__builtin_fmax(a[i], 0.25), 1.0e300) + 1.0)
KERNEL(rcp_f64, double, seed + i + threadIdx.x, a[i] = __builtin_amdgcn_rcp(a[i]))
KERNEL(rsq_f64, double, seed + i + threadIdx.x, a[i] = __builtin_amdgcn_rsq(a[i]))
KERNEL(div_f64, double, seed + i + threadIdx.x + 1.0, a[i] = 3.0 / a[i])
KERNEL(sqrt_f64, double, seed + i + threadIdx.x + 1.0, a[i] = __builtin_sqrt(a[i]) + 2.0)
KERNEL(cmp_f64, double, seed + i + threadIdx.x, a[i] = (a[i] < 1.0e300) ? a[i] + 1.0 : 0.0)
// libm-class device functions the path calls (rt_kernel.hip m_sin ... m_log): one call per instance, arguments kept in a sane range
KERNEL(sin_f64, double, 0.001 * (seed + i + threadIdx.x), a[i] = ::sin(a[i]) + 1.25)
KERNEL(cos_f64, double, 0.001 * (seed + i + threadIdx.x), a[i] = ::cos(a[i]) + 1.25)
KERNEL(tan_f64, double, 0.001 * (seed + i + threadIdx.x), a[i] = ::tan(a[i]) * 0.5 + 0.3)
KERNEL(atan_f64, double, 0.001 * (seed + i + threadIdx.x), a[i] = ::atan(a[i]) + 0.75)
KERNEL(atan2_f64, double, 0.001 * (seed + i + threadIdx.x), a[i] = ::atan2(a[i], 0.7) + 1.5)
KERNEL(acos_f64, double, 0.0001 * (seed + i + threadIdx.x), a[i] = ::acos(a[i]) * 0.3)
KERNEL(log_f64, double, 1.0 + 0.001 * (seed + i + threadIdx.x), a[i] = ::log(a[i]) + 2.5)
KERNEL(log2_f64, double, 1.0 + 0.001 * (seed + i + threadIdx.x), a[i] = ::log2(a[i]) + 2.5)
KERNEL(pow_f64, double, 1.0 + 0.001 * (seed + i + threadIdx.x), a[i] = ::pow(a[i], 2.2) * 0.25 + 1.0)
KERNEL(floor_f64, double, seed + i + threadIdx.x, a[i] = ::floor(a[i] * 1.5) + 0.25)
KERNEL(fma_f32, float, seed + i + threadIdx.x, a[i] = __builtin_fmaf(a[i], 1.0000001f, 0.5f))
KERNEL(add_u32, unsigned, seed + i + threadIdx.x, a[i] = a[i] + 0x9E3779B9u)
KERNEL(xor_shift_u32, unsigned, seed + i + threadIdx.x, a[i] = a[i] ^ (a[i] << 9))
KERNEL(rot_u32, unsigned, seed + i + threadIdx.x, a[i] = __builtin_rotateleft32(a[i], 7) + 1u)
KERNEL(mul_lo_u32, unsigned, seed + i + threadIdx.x, a[i] = a[i] * 0x9E3779B9u)
KERNEL(mul_hi_u32, unsigned, seed + i + threadIdx.x, a[i] = __umulhi(a[i], 0x9E3779B9u) + 3u)
KERNEL(cndmask, unsigned, seed + i + threadIdx.x, a[i] = (a[i] & 1u) ? a[i] + 3u : a[i] ^ 5u)
KERNEL(mul_u64, unsigned long long, seed + i + threadIdx.x, a[i] = a[i] * 0xBF58476D1CE4E5B9ULL)

static int BLOCKS = 256;                // one 1024-thread workgroup per CU = 4 waves on every SIMD
static const size_t LDS_BYTES = 128 * 1024;
template <typename T, typename K> int run(K kern, const char* name, const char* what) {
    T* d; unsigned long long* dt;
    const size_t n_waves = (size_t)BLOCKS * 16;
    CHK(hipMalloc(&d, sizeof(T) * BLOCKS * 1024)); CHK(hipMalloc(&dt, sizeof(unsigned long long) * n_waves * 2));
    CHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    // calibrate the loop count so that the timed launch lasts about 20 ms: launch cost, the waves' staggered start and the clock's
    // ramp are then below a per cent of it, and the kernel's event time can be divided by the operations a SIMD issued
    int iters = 256; float ms = 0.f;
    for (int pass = 0; pass < 3; pass++) {
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(BLOCKS), dim3(1024), LDS_BYTES, 0, d, dt, (T)1, iters); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
        if (pass < 2) { double want = 20.0 / (ms > 1e-3 ? ms : 1e-3) * iters; if (want > 4.0e6) want = 4.0e6; if (want < 256) want = 256; iters = (int)want; }
    }
    std::vector<unsigned long long> t(n_waves * 2);
    CHK(hipMemcpy(t.data(), dt, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum_t = 0, sum_r = 0; unsigned long long mx = 0;
    for (size_t w = 0; w < n_waves; w++) { sum_t += (double)t[2 * w]; sum_r += (double)t[2 * w + 1]; if (t[2 * w] > mx) mx = t[2 * w]; }
    const double ops = 4.0 * (double)iters * 16.0;              // operations one SIMD issues for its 4 waves
    const double ghz = sum_t / (sum_r * 10.0);                  // s_memtime ticks per nanosecond while the loops ran = the shader clock
    const double ns = (double)ms * 1e6 / ops;                   // kernel time (HIP events) per operation per SIMD
    const double cyc = ns * ghz;                                // ... in shader-clock cycles
    const double cyc_wave = (double)mx / ops;                   // cross-check: the slowest wave's own s_memtime span
    printf("%s,\"%s\",%d,%.3f,%.3f,%.3f,%.3f,%.3f\n", name, what, iters, ms, ns, ghz, cyc, cyc_wave);
    (void)hipFree(d); (void)hipFree(dt); return 0;
}
int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("# %s, %d CUs, clockRate %.0f MHz; iters x 16 independent operations per lane, 4 waves/SIMD on every SIMD, ~20 ms per launch\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000.0);
    BLOCKS = p.multiProcessorCount;
    printf("# ns_per_wave_op: kernel time (HIP events) / operations one SIMD issued (4 waves x iters x 16); shader_GHz: s_memtime ticks per ns of s_memrealtime\n");
    printf("# while the loops ran; simd_cycles_per_wave_op = ns x GHz (what one wave-instruction costs in SIMD issue cycles); slowest_wave_cycles: cross-check from\n");
    printf("# the slowest wave's own s_memtime span.  Weights of workloads.VALU_OP_WEIGHTS = cycles(op) / cycles(fma_f64), composite rows less their add / mul.\n");
    printf("op,what,iters,kernel_ms,ns_per_wave_op,shader_GHz,simd_cycles_per_wave_op,slowest_wave_cycles\n");
    int rc = 0;
    rc |= run<double>(k_fma_f64, "fma_f64", "v_fma_f64"); rc |= run<double>(k_mul_f64, "mul_f64", "v_mul_f64"); rc |= run<double>(k_add_f64, "add_f64", "v_add_f64");
    rc |= run<double>(k_minmax_f64, "minmax_f64", "fmax+fmin+add (3 ops + quieting)"); rc |= run<double>(k_cmp_f64, "cmp_f64", "v_cmp+v_cndmask x2+v_add");
    rc |= run<double>(k_rcp_f64, "rcp_f64", "v_rcp_f64"); rc |= run<double>(k_rsq_f64, "rsq_f64", "v_rsq_f64");
    rc |= run<double>(k_div_f64, "div_f64", "IEEE f64 divide (div_scale x2, rcp, 2 Newton, fmas, fixup)"); rc |= run<double>(k_sqrt_f64, "sqrt_f64", "IEEE f64 sqrt + add");
    rc |= run<double>(k_sin_f64, "sin_f64", "::sin + add"); rc |= run<double>(k_cos_f64, "cos_f64", "::cos + add"); rc |= run<double>(k_tan_f64, "tan_f64", "::tan, mul, add");
    rc |= run<double>(k_atan_f64, "atan_f64", "::atan + add"); rc |= run<double>(k_atan2_f64, "atan2_f64", "::atan2 + add"); rc |= run<double>(k_acos_f64, "acos_f64", "::acos, mul");
    rc |= run<double>(k_log_f64, "log_f64", "::log + add"); rc |= run<double>(k_log2_f64, "log2_f64", "::log2 + add"); rc |= run<double>(k_pow_f64, "pow_f64", "::pow, mul, add");
    rc |= run<double>(k_floor_f64, "floor_f64", "mul, ::floor, add");
    rc |= run<float>(k_fma_f32, "fma_f32", "v_fma_f32"); rc |= run<unsigned>(k_add_u32, "add_u32", "v_add_u32"); rc |= run<unsigned>(k_xor_shift_u32, "xor_shift_u32", "xor+shift (2)");
    rc |= run<unsigned>(k_rot_u32, "rot_u32", "rot+add (2)"); rc |= run<unsigned>(k_mul_lo_u32, "mul_lo_u32", "v_mul_lo_u32"); rc |= run<unsigned>(k_mul_hi_u32, "mul_hi_u32", "mul_hi+add");
    rc |= run<unsigned>(k_cndmask, "cndmask", "and,cmp,add,xor,cnd"); rc |= run<unsigned long long>(k_mul_u64, "mul_u64", "u64 multiply");
    return rc;
}

// tools/ubench/ubench.hip — VALU instruction throughput on gfx950 (developer tool, not part of the product).
// Each kernel runs ITER x 16 independent instances of one operation per lane with 4 waves/SIMD resident everywhere (1024 workgroups of
// 256 threads on 256 CUs).  Two clocks are reported so that nothing depends on an assumed frequency:
//   * cycles per wave-instruction from the SHADER clock (s_memtime deltas inside the kernel, averaged over the waves, divided by the
//     4 waves that share a SIMD) — what one operation costs in SIMD issue cycles;
//   * the shader clock's rate against the host-visible event time (s_memtime ticks per second) — the clock the chip actually ran at.
// Output is one CSV row per operation; `make -C tools/ubench run > profiles/rNN_ubench.csv` on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define ITER 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(name, TYPE, INIT, OP) \
__global__ void __launch_bounds__(256) k_##name(TYPE* out, unsigned long long* ticks, TYPE seed) { \
    TYPE a[16]; for (int i = 0; i < 16; i++) a[i] = INIT; \
    __builtin_amdgcn_sched_barrier(0); const unsigned long long t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 16; i++) { OP; } } \
    TYPE s = a[0]; for (int i = 1; i < 16; i++) s = s + a[i]; \
    asm volatile("" :: "v"(s)); \
    __builtin_amdgcn_sched_barrier(0); const unsigned long long t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
    if ((threadIdx.x & 63u) == 0u) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0; }

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

static const int BLOCKS = 256 * 4;      // 4 blocks of 256 per CU = 4 waves/SIMD
template <typename T, typename K> int run(K kern, const char* name, const char* what) {
    T* d; unsigned long long* dt;
    CHK(hipMalloc(&d, sizeof(T) * BLOCKS * 256)); CHK(hipMalloc(&dt, sizeof(unsigned long long) * BLOCKS * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(BLOCKS), dim3(256), 0, 0, d, dt, (T)1);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(BLOCKS), dim3(256), 0, 0, d, dt, (T)1); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> t(BLOCKS * 4);
    CHK(hipMemcpy(t.data(), dt, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0; unsigned long long mx = 0; for (unsigned long long v : t) { sum += (double)v; if (v > mx) mx = v; }
    const double mean_ticks = sum / (double)t.size();
    // one SIMD runs 4 of these waves side by side: per wave ITER * 16 operations in mean_ticks shader-clock ticks
    const double cyc = mean_ticks / (4.0 * ITER * 16.0);
    // the longest wave's ticks against the kernel's event time: the shader clock's rate (launch overhead makes this a lower bound)
    const double ghz = (double)mx / ((double)ms * 1e6);
    printf("%s,%s,%.4f,%.3f,%.3f\n", name, what, ms, cyc, ghz);
    (void)hipFree(d); (void)hipFree(dt); return 0;
}
int main() {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("# %s, %d CUs, clockRate %.0f MHz; ITER %d x 16 independent operations per lane, 4 waves/SIMD on every SIMD\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000.0, ITER);
    printf("# simd_cycles_per_wave_op: shader-clock (s_memtime) ticks per operation of one wave, divided by the 4 waves sharing the SIMD\n");
    printf("op,what,kernel_ms,simd_cycles_per_wave_op,shader_clock_GHz_lower_bound\n");
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

// tools/ubench/ubench.hip — VALU instruction throughput on gfx950 (developer tool, not part of the product).
// Each kernel runs ITER x 16 independent instances of one instruction per lane with 4 waves/SIMD resident everywhere.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define KERNEL(name, TYPE, INIT, OP) \
__global__ void __launch_bounds__(256) k_##name(TYPE* out, TYPE seed) { \
    TYPE a[16]; for (int i = 0; i < 16; i++) a[i] = INIT; \
    for (int it = 0; it < ITER; it++) { _Pragma("unroll") for (int i = 0; i < 16; i++) { OP; } } \
    TYPE s = a[0]; for (int i = 1; i < 16; i++) s = s + a[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = s; }

KERNEL(fma_f64, double, seed + i + threadIdx.x, a[i] = __builtin_fma(a[i], 1.0000001, 0.5))
KERNEL(mul_f64, double, seed + i + threadIdx.x, a[i] = a[i] * 1.0000001)
KERNEL(add_f64, double, seed + i + threadIdx.x, a[i] = a[i] + 1.5)
KERNEL(rcp_f64, double, seed + i + threadIdx.x, a[i] = __builtin_amdgcn_rcp(a[i]))
KERNEL(rsq_f64, double, seed + i + threadIdx.x, a[i] = __builtin_amdgcn_rsq(a[i]))
KERNEL(div_f64, double, seed + i + threadIdx.x + 1.0, a[i] = 3.0 / a[i])
KERNEL(sqrt_f64, double, seed + i + threadIdx.x + 1.0, a[i] = __builtin_sqrt(a[i]) + 2.0)
KERNEL(fma_f32, float, seed + i + threadIdx.x, a[i] = __builtin_fmaf(a[i], 1.0000001f, 0.5f))
KERNEL(add_u32, unsigned, seed + i + threadIdx.x, a[i] = a[i] + 0x9E3779B9u)
KERNEL(xor_shift_u32, unsigned, seed + i + threadIdx.x, a[i] = a[i] ^ (a[i] << 9))
KERNEL(rot_u32, unsigned, seed + i + threadIdx.x, a[i] = __builtin_rotateleft32(a[i], 7) + 1u)
KERNEL(mul_lo_u32, unsigned, seed + i + threadIdx.x, a[i] = a[i] * 0x9E3779B9u)
KERNEL(mul_hi_u32, unsigned, seed + i + threadIdx.x, a[i] = __umulhi(a[i], 0x9E3779B9u) + 3u)
KERNEL(cndmask, unsigned, seed + i + threadIdx.x, a[i] = (a[i] & 1u) ? a[i] + 3u : a[i] ^ 5u)
KERNEL(mul_u64, unsigned long long, seed + i + threadIdx.x, a[i] = a[i] * 0xBF58476D1CE4E5B9ULL)
KERNEL(cmp_f64, double, seed + i + threadIdx.x, a[i] = (a[i] < 1.0e300) ? a[i] + 1.0 : 0.0)

template <typename T, typename K> double run(K kern, const char* name, int ops_per_iter, double clock_ghz) {
    int blocks = 256 * 4;      // 4 blocks of 256 per CU = 4 waves/SIMD
    T* d; hipMalloc(&d, sizeof(T) * blocks * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, (T)1);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, (T)1); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD = (4 waves) * ITER * 16 * ops_per_iter
    double winst = 4.0 * ITER * 16.0 * ops_per_iter;
    double cyc = ms * 1e-3 * clock_ghz * 1e9 / winst;
    printf("%-14s %8.3f ms  -> %6.2f SIMD-cycles per wave-instruction (at %.2f GHz, x%d ops)\n", name, ms, cyc, clock_ghz, ops_per_iter);
    hipFree(d); return cyc;
}
int main() {
    double ghz = 2.38;
    run<double>(k_fma_f64, "v_fma_f64", 1, ghz); run<double>(k_mul_f64, "v_mul_f64", 1, ghz); run<double>(k_add_f64, "v_add_f64", 1, ghz);
    run<double>(k_rcp_f64, "v_rcp_f64", 1, ghz); run<double>(k_rsq_f64, "v_rsq_f64", 1, ghz);
    run<double>(k_div_f64, "f64 divide", 1, ghz); run<double>(k_sqrt_f64, "f64 sqrt(+add)", 1, ghz);
    run<double>(k_cmp_f64, "cmp+sel+add f64", 1, ghz);
    run<float>(k_fma_f32, "v_fma_f32", 1, ghz); run<unsigned>(k_add_u32, "v_add_u32", 1, ghz); run<unsigned>(k_xor_shift_u32, "xor+shift (2)", 1, ghz);
    run<unsigned>(k_rot_u32, "rot+add (2)", 1, ghz); run<unsigned>(k_mul_lo_u32, "v_mul_lo_u32", 1, ghz); run<unsigned>(k_mul_hi_u32, "mul_hi+add", 1, ghz);
    run<unsigned>(k_cndmask, "and,cmp,add,xor,cnd", 1, ghz); run<unsigned long long>(k_mul_u64, "u64 multiply", 1, ghz);
    return 0;
}

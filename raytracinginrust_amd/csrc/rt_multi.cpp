// csrc/rt_multi.cpp — rt_render_multi: the whole frame on several GPUs of one node from ONE call (SURVEY.md §8(b) `device_mask`,
// §8(e)).  Replaces the reference's render loop src/main.rs:767-835 for a host that owns all the GPUs of the node itself (the Rust
// `main`): one process, one thread.
//
//   * the scene (< 2 MB) is replicated: every selected device holds its own copy of the tables (Scene::DeviceCtx);
//   * the frame's tiles are dealt round-robin (tile t -> device t mod N, the same decomposition as rt_render_device / dist.py), each
//     device renders its share with one persistent kernel launch on its own stream — the launches are asynchronous, so the N
//     kernels run concurrently;
//   * ONE collective at frame end: ncclGather (RCCL over xGMI, communicators from ncclCommInitAll, cached with the scene) of the
//     N packed tile buffers to the root device, each rank's gather ordered after its own kernel on its own stream;
//   * the root un-permutes [rank][local tile][pixel] -> output-order pixels with a copy kernel on the device and the frame goes to
//     the caller's host buffer in one transfer.
// RCCL is loaded with dlopen on first use (librccl.so.1 — the copy already in the process when the host is PyTorch), so that
// librt_amd.so carries no link-time dependency on it and single-GPU users never load it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rt_amd.h"
#include "rt_scene.h"

using namespace rt;

namespace rt {
int set_error(const std::string& m);          // rt_host.cpp: rt_last_error()'s message; returns -1
int device_kernel_ms(Scene& s, int device, float* ms);
}

namespace {

// the few RCCL entry points used, resolved at run time (signatures of rccl.h 2.27: ncclCommInitAll :236, ncclGather :745)
typedef void* comm_t;
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
const int NCCL_DOUBLE = 8;                      // ncclFloat64, rccl.h ncclDataType_t

Rccl* rccl() {
    static Rccl r;
    if (r.lib || !r.err.empty()) return &r;
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.err = std::string("cannot load RCCL (librccl.so.1): ") + dlerror(); return &r; }
    r.CommInitAll = (int (*)(comm_t*, int, const int*))dlsym(r.lib, "ncclCommInitAll");
    r.CommDestroy = (int (*)(comm_t))dlsym(r.lib, "ncclCommDestroy");
    r.GroupStart = (int (*)())dlsym(r.lib, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.lib, "ncclGroupEnd");
    r.Gather = (int (*)(const void*, void*, size_t, int, int, comm_t, hipStream_t))dlsym(r.lib, "ncclGather");
    r.GetErrorString = (const char* (*)(int))dlsym(r.lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Gather || !r.GetErrorString) {
        r.err = "librccl.so.1 lacks ncclCommInitAll / ncclGather / ncclGroupStart / ncclGroupEnd";
        r.lib = nullptr;
    }
    return &r;
}

#define HIPX(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_error(std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)
#define NCCLX(call) do { int r_ = (call); if (r_ != 0) return set_error(std::string(#call) + ": " + R->GetErrorString(r_)); } while (0)

// gathered[rank][q][k][3] (tile t = rank + q * world, pixel k of the tile) -> frame[p][3], p = t * tile_px + k: one thread per
// output double, consecutive threads write consecutive addresses and read runs of tile_px * 3 consecutive ones (HBM-bound copy).
__global__ void unpermute_tiles(const double* __restrict__ gathered, double* __restrict__ frame, unsigned long long n_px,
                                unsigned tile_px, unsigned world, unsigned long long n_local_tiles) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_px * 3ull) return;
    const unsigned long long p = i / 3ull; const unsigned c = (unsigned)(i - p * 3ull);
    const unsigned long long t = p / tile_px; const unsigned k = (unsigned)(p - t * tile_px);
    const unsigned long long rank = t % world, q = t / world;
    frame[i] = gathered[((rank * n_local_tiles + q) * tile_px + k) * 3ull + c];
}

int ensure_buffer(void*& p, size_t& have, size_t need) {
    if (have >= need) return 0;
    if (p) { (void)hipFree(p); p = nullptr; have = 0; }
    HIPX(hipMalloc(&p, need));
    have = need;
    return 0;
}

} // namespace

namespace rt {
void multi_release(Scene& s) {
    Rccl* R = rccl();
    if (R->lib) for (void* c : s.comms) if (c) (void)R->CommDestroy(c);
    s.comms.clear(); s.comm_devices.clear();
}
}

extern "C" {

int rt_render_multi(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                    uint64_t seed, uint32_t flags, uint32_t device_mask, uint32_t tile_px, double* rgb_sum_out) {
    if (!sc || !cam || !bg || !rgb_sum_out) return set_error("null argument");
    const auto t_call = std::chrono::steady_clock::now();
    const int n_visible = rt_device_count();
    if (n_visible <= 0) return set_error("no HIP device: librt_amd has no CPU rendering path");
    std::vector<int> devs;
    for (int d = 0; d < n_visible && d < 32; d++) if (device_mask == 0u || (device_mask >> d) & 1u) devs.push_back(d);
    if (devs.empty() || (device_mask != 0u && n_visible < 32 && (device_mask >> n_visible) != 0u))
        return set_error("device_mask selects a device that is not visible (rt_device_count() = " + std::to_string(n_visible) + ")");
    if (tile_px == 0) tile_px = 67;                 // the default of dist.py: a prime, so that tile columns drift across rows
    // Test hook (tests/test_multi_gpu.py): RT_MULTI_VIRTUAL_RANKS=N deals the tiles to N ranks that all live on the FIRST selected
    // device, one after the other, and "gathers" with device-to-device copies — the tile arithmetic and the un-permute kernel of an
    // N-GPU frame on a one-GPU box.  Never set in production.
    uint32_t virtual_ranks = 0;
    if (const char* v = std::getenv("RT_MULTI_VIRTUAL_RANKS")) { long n = std::strtol(v, nullptr, 10); if (n >= 1 && n <= 64) virtual_ranks = (uint32_t)n; }
    if (virtual_ranks) devs.assign(virtual_ranks, devs[0]);
    const uint32_t N = (uint32_t)devs.size();
    Scene& s = sc->s;
    int cur = 0; HIPX(hipGetDevice(&cur));
    const uint32_t n_local = rt_local_tiles(W, H, tile_px, 0, N);
    const size_t tiles_bytes = (size_t)n_local * tile_px * 3 * sizeof(double);
    const unsigned long long n_px = (unsigned long long)W * H;
    // RT_MULTI_COLLECTIVE: force the collective path on one device too (lets a 1-GPU box exercise the RCCL calls)
    const bool collective = !virtual_ranks && (N > 1 || (flags & RT_MULTI_COLLECTIVE) != 0u);
    std::vector<void*> rank_tiles(N, nullptr);      // each rank's packed tile buffer (virtual ranks: temporaries on the one device)
    Rccl* R = nullptr;
    if (collective) {
        R = rccl();
        if (!R->lib) return set_error(R->err);
        if (s.comm_devices != devs) {
            rt::multi_release(s);
            s.comms.assign(N, nullptr);
            // RCCL prints a version banner to stdout when it initialises; the reference's contract is "the image IS stdout"
            // (`cargo run --release > image.ppm`, README.md:4), so stdout points at stderr while the communicators are created
            std::fflush(stdout);
            const int saved_out = dup(1);
            if (saved_out >= 0) (void)dup2(2, 1);
            const int init_rc = R->CommInitAll(s.comms.data(), (int)N, devs.data());
            std::fflush(stdout);
            if (saved_out >= 0) { (void)dup2(saved_out, 1); (void)close(saved_out); }
            NCCLX(init_rc);
            s.comm_devices = devs;
        }
    }
    int rc = 0;
    // ---- every device: its share of the tiles, asynchronously on its own stream
    for (uint32_t r = 0; r < N && rc == 0; r++) {
        hipError_t e = hipSetDevice(devs[r]);
        if (e != hipSuccess) { rc = set_error(std::string("hipSetDevice: ") + hipGetErrorString(e)); break; }
        Scene::DeviceCtx& c = s.ctx_for(devs[r]);
        if (!c.stream) { hipStream_t st; e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking); if (e != hipSuccess) { rc = set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e)); break; } c.stream = st; }
        if (virtual_ranks) { e = hipMalloc(&rank_tiles[r], tiles_bytes); if (e != hipSuccess) { rc = set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); break; } }
        else { if (ensure_buffer(c.d_tiles, c.tiles_bytes, tiles_bytes)) { rc = -1; break; } rank_tiles[r] = c.d_tiles; }
        if (r == 0) {
            if ((collective || virtual_ranks) && ensure_buffer(c.d_gather, c.gather_bytes, tiles_bytes * N)) { rc = -1; break; }
            if (ensure_buffer(c.d_frame, c.frame_bytes, (size_t)n_px * 3 * sizeof(double))) { rc = -1; break; }
        }
        rc = rt_render_device(sc, cam, bg, W, H, spp, max_depth, seed, flags & ~(uint32_t)RT_MULTI_COLLECTIVE, tile_px, r, N, rank_tiles[r], tiles_bytes, c.stream);
    }
    // ---- one gather to the root device (rank 0), each rank's part ordered behind its own kernel
    Scene::DeviceCtx& root = s.ctx_for(devs[0]);
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (rc == 0) {
        (void)hipSetDevice(devs[0]);
        for (hipEvent_t& e : ev) if (hipEventCreate(&e) != hipSuccess) rc = set_error("hipEventCreate failed");
    }
    if (rc == 0) {
        HIPX(hipEventRecord(ev[0], (hipStream_t)root.stream));
        if (collective) {
            NCCLX(R->GroupStart());
            for (uint32_t r = 0; r < N; r++) {
                Scene::DeviceCtx& c = s.ctx_for(devs[r]);
                int g = R->Gather(rank_tiles[r], r == 0 ? root.d_gather : nullptr, (size_t)n_local * tile_px * 3, NCCL_DOUBLE, 0, s.comms[r], (hipStream_t)c.stream);
                if (g != 0) { (void)R->GroupEnd(); return set_error(std::string("ncclGather: ") + R->GetErrorString(g)); }
            }
            NCCLX(R->GroupEnd());
        } else if (virtual_ranks) {
            for (uint32_t r = 0; r < N; r++)
                HIPX(hipMemcpyAsync((char*)root.d_gather + (size_t)r * tiles_bytes, rank_tiles[r], tiles_bytes, hipMemcpyDeviceToDevice, (hipStream_t)root.stream));
        }
        HIPX(hipSetDevice(devs[0]));
        HIPX(hipEventRecord(ev[1], (hipStream_t)root.stream));
        // ---- un-permute on the root device into output order, then one transfer to the caller's buffer
        const unsigned long long n_out = n_px * 3ull;
        const unsigned block = 256; const unsigned long long grid = (n_out + block - 1) / block;
        hipLaunchKernelGGL(unpermute_tiles, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)root.stream,
                           (const double*)((collective || virtual_ranks) ? root.d_gather : rank_tiles[0]), (double*)root.d_frame, n_px, tile_px, N, (unsigned long long)n_local);
        HIPX(hipGetLastError());
        HIPX(hipEventRecord(ev[2], (hipStream_t)root.stream));
        HIPX(hipMemcpyAsync(rgb_sum_out, root.d_frame, (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)root.stream));
        HIPX(hipEventRecord(ev[3], (hipStream_t)root.stream));
        for (uint32_t r = 0; r < N; r++) { HIPX(hipSetDevice(devs[r])); HIPX(hipStreamSynchronize((hipStream_t)s.ctx_for(devs[r]).stream)); }
        HIPX(hipSetDevice(devs[0]));
        float g_ms = 0.f, u_ms = 0.f, k_max = 0.f;
        (void)hipEventElapsedTime(&g_ms, ev[0], ev[1]);             // on the root's stream: its own kernel + the gather behind it
        (void)hipEventElapsedTime(&u_ms, ev[1], ev[2]);
        for (uint32_t r = 0; r < N; r++) { float k = 0.f; if (device_kernel_ms(s, devs[r], &k) == 0 && k > k_max) k_max = k; }
        s.multi_ms[0] = k_max; s.multi_ms[1] = g_ms; s.multi_ms[2] = u_ms;
        s.multi_ms[3] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    }
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    if (virtual_ranks) for (void* p : rank_tiles) if (p) (void)hipFree(p);
    (void)hipSetDevice(cur);
    return rc;
}

// Timings of the last rt_render_multi (ms): [0] slowest device's path-tracing kernel, [1] root stream from launch to the end of the
// gather (its own kernel + the collective), [2] un-permute kernel, [3] the whole call on the host clock.
int rt_last_multi_ms(rt_scene* sc, double out4[4]) {
    if (!sc || !out4) return set_error("null argument");
    for (int k = 0; k < 4; k++) out4[k] = sc->s.multi_ms[k];
    return 0;
}

} // extern "C"

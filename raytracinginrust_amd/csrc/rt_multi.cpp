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
//
// Entry points: rt_render_multi_device enqueues the frame on every device and returns (the frame stays on the root device),
// rt_multi_sync waits for it and settles the timings, rt_multi_copy_frame fetches it; rt_render_multi = all three.  Every failure
// leaves through one tail that waits for what was already launched, keeps no temporary alive and restores the caller's HIP device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/rt_amd.h"
#include "rt_scene.h"

using namespace rt;

namespace rt {
int set_error(const std::string& m);          // rt_host.cpp: rt_last_error()'s message; returns -1
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
    int (*CommCount)(comm_t, int*) = nullptr;        // optional: how many ranks the communicator really has (rt_last_multi_ranks)
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
    r.CommCount = (int (*)(comm_t, int*))dlsym(r.lib, "ncclCommCount");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Gather || !r.GetErrorString) {
        r.err = "librccl.so.1 lacks ncclCommInitAll / ncclGather / ncclGroupStart / ncclGroupEnd";
        r.lib = nullptr;
    }
    return &r;
}

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

std::string hip_msg(const char* what, hipError_t e) { return std::string(what) + ": " + hipGetErrorString(e); }

bool ensure_buffer(void*& p, size_t& have, size_t need, std::string& err) {
    if (have >= need) return true;
    if (p) { (void)hipFree(p); p = nullptr; have = 0; }
    const hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) { p = nullptr; err = hip_msg("hipMalloc", e); return false; }
    have = need;
    return true;
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Wait for everything the frame in flight put on the devices' streams (also the way out of a failed call: nothing of it may still
// be running when its buffers are reused or freed).  Leaves the current device at the last one visited.
void drain(Scene& s) {
    for (int d : s.multi_devs) {
        for (Scene::DeviceCtx* c : s.ctxs) if (c->device == d && c->stream) { (void)hipSetDevice(d); (void)hipStreamSynchronize((hipStream_t)c->stream); }
    }
}

} // namespace

namespace rt {
void multi_release(Scene& s) {
    if (s.multi_pending) { int cur = 0; (void)hipGetDevice(&cur); drain(s); (void)hipSetDevice(cur); s.multi_pending = false; }
    if (s.multi_ev_device >= 0) {
        int cur = 0; (void)hipGetDevice(&cur);
        (void)hipSetDevice(s.multi_ev_device);
        for (void*& e : s.multi_ev) if (e) { (void)hipEventDestroy((hipEvent_t)e); e = nullptr; }
        s.multi_ev_device = -1;
        (void)hipSetDevice(cur);
    }
    if (!s.virtual_tiles.empty()) {
        int cur = 0; (void)hipGetDevice(&cur);
        (void)hipSetDevice(s.virtual_tiles_device);
        for (void* p : s.virtual_tiles) if (p) (void)hipFree(p);
        s.virtual_tiles.clear(); s.virtual_tiles_bytes = 0; s.virtual_tiles_device = -1;
        (void)hipSetDevice(cur);
    }
    if (s.comms.empty()) return;                 // single-GPU users never touch (or load) RCCL, not even at teardown
    Rccl* R = rccl();
    if (R->lib) for (void* c : s.comms) if (c) (void)R->CommDestroy(c);
    s.comms.clear(); s.comm_devices.clear();
}
}

namespace {

// The frame on N devices, enqueued: everything up to and including the un-permute on the root device's stream.  On failure the
// message is in `err` and the caller runs the cleanup tail.
bool enqueue_frame(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                   uint64_t seed, uint32_t flags, uint32_t device_mask, uint32_t tile_px, std::string& err) {
    Scene& s = sc->s;
    const int n_visible = rt_device_count();
    if (n_visible <= 0) { err = "no HIP device: librt_amd has no CPU rendering path"; return false; }
    std::vector<int> devs;
    for (int d = 0; d < n_visible && d < 32; d++) if (device_mask == 0u || (device_mask >> d) & 1u) devs.push_back(d);
    if (devs.empty() || (device_mask != 0u && n_visible < 32 && (device_mask >> n_visible) != 0u)) {
        err = "device_mask selects a device that is not visible (rt_device_count() = " + std::to_string(n_visible) + ")"; return false;
    }
    // Test hooks (tests/test_multi_gpu.py), never set in production.  RT_MULTI_VIRTUAL_RANKS=N deals the tiles to N ranks that all
    // live on the FIRST selected device, one after the other, and "gathers" with device-to-device copies — the tile arithmetic and the
    // un-permute kernel of an N-GPU frame on a one-GPU box.  RT_MULTI_FAIL_RANK=r makes rank r's launch fail after the ranks before
    // it were launched — the error path of an N-GPU frame.
    uint32_t virtual_ranks = 0;
    if (const char* v = std::getenv("RT_MULTI_VIRTUAL_RANKS")) { long n = std::strtol(v, nullptr, 10); if (n >= 1 && n <= 64) virtual_ranks = (uint32_t)n; }
    long fail_rank = -1;
    if (const char* v = std::getenv("RT_MULTI_FAIL_RANK")) fail_rank = std::strtol(v, nullptr, 10);
    if (virtual_ranks) devs.assign(virtual_ranks, devs[0]);
    const uint32_t N = (uint32_t)devs.size();
    if (!rt::flatten_for_render(s)) { err = rt_last_error(); return false; }
    if (W < 2 || H < 2 || spp == 0 || (uint64_t)W * H > 0x7FFFFFFFull) {      // rt_render_device repeats these with its own messages
        err = "bad frame: W and H must be >= 2, samples_per_pixel >= 1, W*H <= 2^31 - 1"; return false;
    }
    const uint32_t n_local = rt_local_tiles(W, H, tile_px, 0, N);
    const size_t tiles_bytes = (size_t)n_local * tile_px * 3 * sizeof(double);
    const unsigned long long n_px = (unsigned long long)W * H;
    // RT_MULTI_COLLECTIVE: force the collective path on one device too (lets a 1-GPU box exercise the RCCL calls)
    const bool collective = !virtual_ranks && (N > 1 || (flags & RT_MULTI_COLLECTIVE) != 0u);
    s.multi_devs.assign(devs.begin(), devs.end());
    hipError_t e;

    // ---- per-device one-time work (first use of a device: context, code object, scene upload, stream) from one thread per
    //      device, so that an 8-GPU node does not do it eight times in a row; every thread touches its own DeviceCtx only
    std::vector<Scene::DeviceCtx*> ctx(N, nullptr);
    for (uint32_t r = 0; r < N; r++) ctx[r] = &s.ctx_for(devs[r]);
    {
        const uint32_t n_distinct = virtual_ranks ? 1u : N;
        std::vector<std::string> perr(n_distinct);
        auto prep = [&](uint32_t r) {
            Scene::DeviceCtx& c = *ctx[r];
            hipError_t pe = hipSetDevice(devs[r]);
            if (pe != hipSuccess) { perr[r] = hip_msg("hipSetDevice", pe); return; }
            const bool f32 = (flags & RT_F32) != 0u;
            if (!(f32 ? c.dev32.valid : c.dev64.valid) || !c.slots[0].ev_start) {
                if (rt::prepare_device(s, c, flags)) { perr[r] = rt_last_error(); return; }
            }
            if (!c.stream) { hipStream_t st; pe = hipStreamCreateWithFlags(&st, hipStreamNonBlocking); if (pe != hipSuccess) { perr[r] = hip_msg("hipStreamCreate", pe); return; } c.stream = st; }
            if (!virtual_ranks && !ensure_buffer(c.d_tiles, c.tiles_bytes, tiles_bytes, perr[r])) return;
            if (r == 0) {
                if ((collective || virtual_ranks) && !ensure_buffer(c.d_gather, c.gather_bytes, tiles_bytes * N, perr[r])) return;
                const int fb = (int)(s.multi_frames & 1ull);      // consecutive frames alternate between two frame buffers
                if (!ensure_buffer(c.d_frame[fb], c.frame_bytes[fb], (size_t)n_px * 3 * sizeof(double), perr[r])) return;
            }
        };
        if (n_distinct == 1) prep(0);
        else {
            std::vector<std::thread> th;
            for (uint32_t r = 0; r < n_distinct; r++) th.emplace_back(prep, r);
            for (std::thread& t : th) t.join();
        }
        for (uint32_t r = 0; r < n_distinct; r++) if (!perr[r].empty()) { err = "device " + std::to_string(devs[r]) + ": " + perr[r]; return false; }
    }
    Scene::DeviceCtx& root = *ctx[0];
    if ((e = hipSetDevice(devs[0])) != hipSuccess) { err = hip_msg("hipSetDevice", e); return false; }
    if (s.multi_ev_device != devs[0]) {
        if (s.multi_ev_device >= 0) { (void)hipSetDevice(s.multi_ev_device); for (void*& ev : s.multi_ev) if (ev) { (void)hipEventDestroy((hipEvent_t)ev); ev = nullptr; } (void)hipSetDevice(devs[0]); }
        s.multi_ev_device = devs[0];
        for (void*& ev : s.multi_ev) { hipEvent_t x; if ((e = hipEventCreate(&x)) != hipSuccess) { err = hip_msg("hipEventCreate", e); return false; } ev = x; }
    }
    std::vector<void*> rank_tiles(N, nullptr);      // each rank's packed tile buffer (virtual ranks: buffers on the one device, kept with the scene)
    if (virtual_ranks) {
        if (s.virtual_tiles.size() != N || s.virtual_tiles_bytes < tiles_bytes || s.virtual_tiles_device != devs[0]) {
            if (!s.virtual_tiles.empty()) { (void)hipSetDevice(s.virtual_tiles_device); for (void* p : s.virtual_tiles) if (p) (void)hipFree(p); (void)hipSetDevice(devs[0]); }
            s.virtual_tiles.assign(N, nullptr); s.virtual_tiles_bytes = tiles_bytes; s.virtual_tiles_device = devs[0];
            for (uint32_t r = 0; r < N; r++) if ((e = hipMalloc(&s.virtual_tiles[r], tiles_bytes)) != hipSuccess) { s.virtual_tiles[r] = nullptr; err = hip_msg("hipMalloc", e); return false; }
        }
        rank_tiles = s.virtual_tiles;
    } else {
        for (uint32_t r = 0; r < N; r++) rank_tiles[r] = ctx[r]->d_tiles;
    }
    Rccl* R = nullptr;
    if (collective) {
        R = rccl();
        if (!R->lib) { err = R->err; return false; }
        if (s.comm_devices != devs) {
            if (!s.comms.empty()) { for (void* c : s.comms) if (c) (void)R->CommDestroy(c); s.comms.clear(); s.comm_devices.clear(); }
            s.comms.assign(N, nullptr);
            // RCCL prints a version banner to stdout when it initialises; the reference's contract is "the image IS stdout"
            // (`cargo run --release > image.ppm`, README.md:4), so stdout points at stderr while the communicators are created
            std::fflush(stdout);
            const int saved_out = dup(1);
            if (saved_out >= 0) (void)dup2(2, 1);
            const int init_rc = R->CommInitAll(s.comms.data(), (int)N, devs.data());
            std::fflush(stdout);
            if (saved_out >= 0) { (void)dup2(saved_out, 1); (void)close(saved_out); }
            if (init_rc != 0) { s.comms.clear(); err = std::string("ncclCommInitAll: ") + R->GetErrorString(init_rc); return false; }
            s.comm_devices = devs;
        }
    }
    // ---- every device: its share of the tiles, asynchronously on its own stream (one thread: a launch is a few microseconds)
    s.frame_group++; s.group_open = true;          // the N launches are ONE frame: rt_last_stats sums their counters
    struct CloseGroup { Scene& s; ~CloseGroup() { s.group_open = false; } } close_group{s};
    s.multi_rank_seq.assign(N, 0ull); s.multi_rank_dev.assign(N, -1); s.multi_comm_count = 0;
    for (uint32_t r = 0; r < N; r++) {
        if ((e = hipSetDevice(devs[r])) != hipSuccess) { err = hip_msg("hipSetDevice", e); return false; }
        if ((long)r == fail_rank) { err = "rank " + std::to_string(r) + ": injected failure (RT_MULTI_FAIL_RANK)"; return false; }
        if (rt_render_device(sc, cam, bg, W, H, spp, max_depth, seed, flags & ~(uint32_t)RT_MULTI_COLLECTIVE, tile_px, r, N, rank_tiles[r], tiles_bytes, ctx[r]->stream)) {
            err = "rank " + std::to_string(r) + " (device " + std::to_string(devs[r]) + "): " + rt_last_error(); return false;
        }
        s.multi_rank_seq[r] = s.launch_seq; s.multi_rank_dev[r] = devs[r];
    }
    // ---- one gather to the root device (rank 0), each rank's part ordered behind its own kernel on its own stream
    if ((e = hipSetDevice(devs[0])) != hipSuccess) { err = hip_msg("hipSetDevice", e); return false; }
    const hipStream_t rs = (hipStream_t)root.stream;
    if ((e = hipEventRecord((hipEvent_t)s.multi_ev[0], rs)) != hipSuccess) { err = hip_msg("hipEventRecord", e); return false; }   // fires when the root's own kernel is done
    if (collective) {
        int g = R->GroupStart();
        if (g != 0) { err = std::string("ncclGroupStart: ") + R->GetErrorString(g); return false; }
        for (uint32_t r = 0; r < N && g == 0; r++)
            g = R->Gather(rank_tiles[r], r == 0 ? root.d_gather : nullptr, (size_t)n_local * tile_px * 3, NCCL_DOUBLE, 0, s.comms[r], (hipStream_t)ctx[r]->stream);
        const int ge = R->GroupEnd();                                 // always closed, also after a failed ncclGather
        if (g != 0) { err = std::string("ncclGather: ") + R->GetErrorString(g); return false; }
        if (ge != 0) { err = std::string("ncclGroupEnd: ") + R->GetErrorString(ge); return false; }
        int cnt = 0;
        s.multi_comm_count = (R->CommCount && R->CommCount(s.comms[0], &cnt) == 0) ? cnt : (int)N;
    } else if (virtual_ranks) {
        for (uint32_t r = 0; r < N; r++)
            if ((e = hipMemcpyAsync((char*)root.d_gather + (size_t)r * tiles_bytes, rank_tiles[r], tiles_bytes, hipMemcpyDeviceToDevice, rs)) != hipSuccess) { err = hip_msg("hipMemcpyAsync", e); return false; }
    }
    if ((e = hipSetDevice(devs[0])) != hipSuccess) { err = hip_msg("hipSetDevice", e); return false; }
    if ((e = hipEventRecord((hipEvent_t)s.multi_ev[1], rs)) != hipSuccess) { err = hip_msg("hipEventRecord", e); return false; }
    // ---- un-permute on the root device into output order
    const unsigned long long n_out = n_px * 3ull;
    const unsigned block = 256; const unsigned long long grid = (n_out + block - 1) / block;
    hipLaunchKernelGGL(unpermute_tiles, dim3((unsigned)grid), dim3(block), 0, rs,
                       (const double*)((collective || virtual_ranks) ? root.d_gather : rank_tiles[0]), (double*)root.d_frame[s.multi_frames & 1ull], n_px, tile_px, N, (unsigned long long)n_local);
    if ((e = hipGetLastError()) != hipSuccess) { err = hip_msg("un-permute launch", e); return false; }
    if ((e = hipEventRecord((hipEvent_t)s.multi_ev[2], rs)) != hipSuccess) { err = hip_msg("hipEventRecord", e); return false; }
    return true;
}

// Wait for the frame in flight and settle rt_last_multi_ms.
int settle(Scene& s) {
    if (!s.multi_pending) return 0;
    int cur = 0; (void)hipGetDevice(&cur);
    drain(s);
    s.multi_pending = false;
    (void)hipSetDevice(s.multi_devs[0]);
    float g_ms = 0.f, u_ms = 0.f, k_max = 0.f;
    // both on the root's stream: [0] fires when the root's own kernel is done, [1] when the gather behind it is — the collective
    // including the wait for the slowest other device
    const hipError_t e1 = hipEventElapsedTime(&g_ms, (hipEvent_t)s.multi_ev[0], (hipEvent_t)s.multi_ev[1]);
    const hipError_t e2 = hipEventElapsedTime(&u_ms, (hipEvent_t)s.multi_ev[1], (hipEvent_t)s.multi_ev[2]);
    int rc = 0;
    if (e1 != hipSuccess || e2 != hipSuccess) rc = set_error(hip_msg("rt_render_multi: a device reported an error while the frame ran", e1 != hipSuccess ? e1 : e2));
    std::vector<int> seen;
    for (int d : s.multi_devs) {
        bool dup = false; for (int x : seen) dup = dup || x == d;
        if (dup) continue;
        seen.push_back(d);
        float k = 0.f; if (device_kernel_ms(s, d, &k) == 0 && k > k_max) k_max = k;
    }
    s.multi_ms[0] = k_max; s.multi_ms[1] = g_ms; s.multi_ms[2] = u_ms; s.multi_ms[3] = now_ms() - s.multi_t0;
    (void)hipSetDevice(cur);
    return rc;
}

} // namespace

extern "C" {

int rt_render_multi_device(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                           uint64_t seed, uint32_t flags, uint32_t device_mask, uint32_t tile_px, void** d_frame_out) {
    if (!sc || !cam || !bg) return set_error("null argument");
    if (rt_device_count() <= 0) return set_error("no HIP device: librt_amd has no CPU rendering path");
    Scene& s = sc->s;
    // A frame of this scene may still be in flight: it is not waited for.  Every device's work is ordered by its own stream — kernel,
    // gather, (root) un-permute of frame i, then kernel, gather, ... of frame i + 1 — and so are the buffers they share; the host runs
    // at most one frame ahead (a launch waits for the launch before the previous one on its stream, rt_host.cpp acquire_slot), so the
    // devices go from one frame's kernel to the next without a launch gap.  Timings are kept for the most recent frame only.
    if (tile_px == 0) tile_px = 67;                 // the default of dist.py: a prime, so that tile columns drift across rows
    {   // (a frame of another shape or on other devices may reallocate buffers or communicators: that one waits)
        unsigned long long virt = 0; if (const char* v = std::getenv("RT_MULTI_VIRTUAL_RANKS")) virt = (unsigned long long)std::strtol(v, nullptr, 10);
        const unsigned long long sig[4] = {W, H, tile_px, (unsigned long long)device_mask | ((unsigned long long)(flags & (RT_F32 | RT_MULTI_COLLECTIVE)) << 32) | (virt << 48)};
        bool same = true; for (int k = 0; k < 4; k++) same = same && sig[k] == s.multi_sig[k];
        if (!same && settle(s)) return -1;
        for (int k = 0; k < 4; k++) s.multi_sig[k] = sig[k];
    }
    int cur = 0;
    { const hipError_t e = hipGetDevice(&cur); if (e != hipSuccess) return set_error(hip_msg("hipGetDevice", e)); }
    s.multi_t0 = now_ms();
    std::string err;
    const std::vector<int> devs_before = s.multi_devs;
    const bool ok = enqueue_frame(sc, cam, bg, W, H, spp, max_depth, seed, flags, device_mask, tile_px, err);
    if (!ok) {                                      // whatever was launched — of this frame or the one before — finishes before anything is reused
        drain(s);
        const std::vector<int> devs_now = s.multi_devs;
        s.multi_devs = devs_before; drain(s); s.multi_devs = devs_now;
    }
    s.multi_pending = ok;
    if (ok) {
        if (d_frame_out) *d_frame_out = s.ctx_for(s.multi_devs[0]).d_frame[s.multi_frames & 1ull];
        s.multi_frame_doubles = (size_t)W * H * 3; s.multi_frames++;
    }
    (void)hipSetDevice(cur);                        // a PyTorch host keeps its current device
    return ok ? 0 : set_error(err);
}

int rt_multi_sync(rt_scene* sc) {
    if (!sc) return set_error("null argument");
    return settle(sc->s);
}

int rt_multi_copy_frame(rt_scene* sc, double* rgb_sum_out, size_t n_doubles) {
    if (!sc || !rgb_sum_out) return set_error("null argument");
    Scene& s = sc->s;
    if (settle(s)) return -1;
    if (s.multi_devs.empty()) return set_error("no rt_render_multi frame has been rendered for this scene");
    Scene::DeviceCtx& root = s.ctx_for(s.multi_devs[0]);
    if (s.multi_frames == 0) return set_error("no rt_render_multi frame has been rendered for this scene");
    const void* frame = root.d_frame[(s.multi_frames - 1ull) & 1ull];          // the most recent frame's buffer
    if (!frame || n_doubles > s.multi_frame_doubles) return set_error("rt_multi_copy_frame: more doubles asked for than the last frame holds (W*H*3)");
    int cur = 0; (void)hipGetDevice(&cur);
    (void)hipSetDevice(s.multi_devs[0]);
    const hipError_t e = hipMemcpy(rgb_sum_out, frame, n_doubles * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipSetDevice(cur);
    return e == hipSuccess ? 0 : set_error(hip_msg("hipMemcpy(frame)", e));
}

int rt_render_multi(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                    uint64_t seed, uint32_t flags, uint32_t device_mask, uint32_t tile_px, double* rgb_sum_out) {
    if (!rgb_sum_out || !sc || !cam || !bg) return set_error("null argument");
    {   // this entry point is synchronous: a first large frame of a mesh scene measures its loop shape here (rt_host.cpp:
        // calibrate_loop_shape), on the frame's first device; the asynchronous rt_render_multi_device never does
        const int n_visible = rt_device_count();
        int first = -1;
        for (int d = 0; d < n_visible && d < 32 && first < 0; d++) if (device_mask == 0u || (device_mask >> d) & 1u) first = d;
        if (first >= 0) {
            int cur = 0; (void)hipGetDevice(&cur);
            (void)hipSetDevice(first);
            const int rc = rt::calibrate_if_worth_it(sc, cam, bg, W, H, spp, max_depth, seed, flags & ~(uint32_t)RT_MULTI_COLLECTIVE);
            (void)hipSetDevice(cur);
            if (rc) return -1;
        }
    }
    if (rt_render_multi_device(sc, cam, bg, W, H, spp, max_depth, seed, flags, device_mask, tile_px, nullptr)) return -1;
    const double t0 = sc->s.multi_t0;
    if (rt_multi_copy_frame(sc, rgb_sum_out, (size_t)W * H * 3)) return -1;
    sc->s.multi_ms[3] = now_ms() - t0;              // the whole call, transfer to the host included
    return 0;
}

// The ranks of the last rt_render_multi* frame (waits for it): HIP device and kernel time of each, and how many ranks the communicator
// of its gather has according to RCCL (0: no collective ran).  For a caller that must see a rank that rendered nothing, a device used
// twice, or a communicator smaller than the node.
int rt_last_multi_ranks(rt_scene* sc, uint32_t max_ranks, uint32_t* n_ranks_out, int* device_out, double* kernel_ms_out, uint32_t* collective_ranks_out) {
    if (!sc || !n_ranks_out) return set_error("null argument");
    Scene& s = sc->s;
    if (settle(s)) return -1;
    if (settle_all_launches(s)) return -1;
    *n_ranks_out = (uint32_t)s.multi_rank_seq.size();
    if (collective_ranks_out) *collective_ranks_out = (uint32_t)s.multi_comm_count;
    for (uint32_t r = 0; r < s.multi_rank_seq.size() && r < max_ranks; r++) {
        const unsigned long long q = s.multi_rank_seq[r];
        if (device_out) device_out[r] = s.multi_rank_dev[r];
        if (kernel_ms_out) kernel_ms_out[r] = (q != 0ull && s.seq_tag[q % Scene::SEQ_RING] == q) ? s.seq_ms[q % Scene::SEQ_RING] : -1.0;
    }
    return 0;
}

// Timings of the last rt_render_multi* frame (ms; waits for it): [0] slowest device's path-tracing kernel, [1] on the root's stream
// from the end of its own kernel to the end of the gather (the collective, including the wait for the slowest other device),
// [2] un-permute kernel, [3] host clock from the call's entry to the end of the wait (rt_render_multi: of the whole call).
int rt_last_multi_ms(rt_scene* sc, double out4[4]) {
    if (!sc || !out4) return set_error("null argument");
    if (settle(sc->s)) return -1;
    for (int k = 0; k < 4; k++) out4[k] = sc->s.multi_ms[k];
    return 0;
}

} // extern "C"

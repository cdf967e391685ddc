// csrc/rt_scene.h — host-side scene: the builder tree behind the C-ABI handles, and its flattened form.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "rt_ir.h"
#include "rt_rng.h"

namespace rt {

// One node per reference constructor call (Sphere::new, Translate::new, ...).
struct HNode {
    enum Kind { SPHERE, MSPHERE, RECT, CUBE, TRI, LIST, FLIP, TRANSLATE, ROTATE, MEDIUM, BVH } kind;
    double v[12] = {0};        // SPHERE: c, r | MSPHERE: c0, c1, t0, t1, r | RECT: a0,a1,b0,b1,k | CUBE: min,max | TRI: 9 | TRANSLATE: offset | ROTATE: angle | MEDIUM: density
    int plane_or_axis = 0;     // RECT: plane; ROTATE: axis
    int mat = -1;              // prims; MEDIUM: texture id
    int child = -1;            // wrappers / MEDIUM boundary
    std::vector<int> items;    // LIST items, BVH children
};

struct HPerlin { double rd_vec[256 * 3]; uint8_t perm[3][256]; };

struct HostFlat {             // canonical f64 flattening
    std::vector<DObject> objects;  // the world's top-level objects [0, n_top), then the sub-objects BVH leaves of kind G_OBJ refer to
    uint32_t n_top = 0;
    uint32_t n_alt = 0;            // list scenes with a room (form_room): objects[n_top, n_top + n_alt) = the world list as the reference has it
    std::vector<DOp<double>> ops;
    std::vector<DRect<double>> rects;
    std::vector<DSphere<double>> spheres;
    std::vector<DMSphere<double>> mspheres;
    std::vector<DTri<double>> tris;
    std::vector<DBvhNode<double>> bvh;
    std::vector<DFNode> bvh_f;     // f32 companions of bvh[] for the filtered walk (boxes rounded outward; rt_flatten.cpp: make_filter_nodes)
    float rect_m = 0.0f;           // >= every |coordinate| of every rect; 0: the Cube fast path is off (an inverted Cube or a non-finite rect)
    float filter_m = 0.0f;         // >= every |coordinate| in bvh_f, >= 1; 0: no filter (a box is not finite, inverted, or beyond 2^40)
    std::vector<DMaterial<double>> materials;
    std::vector<DTexture<double>> textures;
    std::vector<DMedium<double>> media;
    std::vector<DLight> lights;
    uint32_t feats = 0;
    uint32_t bvh_depth = 0;
    bool bvh_tame = true;          // all BVH boxes finite, |.| < 1e300 (1e30 matters for the f32 variant: checked there too), min <= max
};

template <typename T> struct DeviceScene {   // device copies of HostFlat for one arithmetic type
    bool valid = false;
    void* objects = nullptr; void* ops = nullptr; void* rects = nullptr; void* spheres = nullptr; void* mspheres = nullptr;
    void* tris = nullptr; void* bvh = nullptr; void* materials = nullptr; void* textures = nullptr; void* media = nullptr;
    void* lights = nullptr; void* perlins = nullptr; void* image = nullptr; void* pbr = nullptr; void* bvh_f = nullptr;
};

struct Scene {
    std::vector<HNode> nodes;
    std::vector<DMaterial<double>> materials;
    std::vector<DTexture<double>> textures;
    std::vector<HPerlin> perlins;
    std::vector<DPbr<double>> pbr;
    std::vector<uint8_t> image_bytes;
    int world = -1;
    unsigned trav_hi = 56, trav_lo = 16, trav_leaf = 32;      // persistent-traversal schedule (rt_scene_set_traversal_schedule); measured best on the teapot room (round 5, the filtered walk: profiles/r05_trav_schedule_sweep.log; 40 / 24 / 24 before)
    int bvh_builder = 0;      // 0: the reference's widest-axis object-median split (bvh.rs:18-73); 1: binned SAH (opt-in)
    // Mesh scenes: which loop shape (1 persistent traversal, 0 lock-step; -1: the size rule decides) — same samples either way.  Set by a
    // calibration (rt_scene_calibrate, or the first large frame of a SYNCHRONOUS entry point: rt_host.cpp calibrate_loop_shape), which is
    // valid for the view it measured (loop_key: camera, frame size, precision — another view falls back to the size rule until it is
    // calibrated itself), or by rt_scene_set_loop_shape (any view, until the scene changes: how a launcher hands rank 0's choice to the others).
    int loop_choice = -1;
    int loop_how = 0;         // 0 nothing stored, 2 calibration (keyed), 4 rt_scene_set_loop_shape (not keyed)
    unsigned long long loop_key = 0;
    float loop_ms[2] = {0.f, 0.f};     // the calibration's kernel times: [0] lock-step, [1] persistent traversal (ms; 0: not measured)
    void* d_calib = nullptr; size_t calib_bytes = 0; int calib_device = -1;     // the calibration renders' frame buffer, kept with the scene
    // what the most recent launch ran (rt_last_loop_info): [0] loop shape (0 list kernel, 1 lock-step BVH, 2 persistent traversal), [1] the
    // FEATS template argument of the instantiation, [2] how the shape was chosen (0 the scene leaves no choice, 1 size rule, 2 calibration,
    // 3 the caller's RT_PERSISTENT_BVH / RT_LOCKSTEP_BVH flag, 4 rt_scene_set_loop_shape), [3] precision (0 f64, 1 f32)
    int last_loop[4] = {0, 0, 0, 0};
    // Worlds that are ONE bare BVH: the view whose estimated pass rates the filter tree's contraction was last tuned for (0: the area
    // rule's tree as flattened) — rt_flatten.cpp tune_filter_tree, rt_host.cpp tune_for_view; any tree gives the same samples
    unsigned long long filter_key = 0;
    std::vector<int> lights;
    std::string error;

    bool flat_valid = false;
    HostFlat flat;
    // Launch scratch (device queue counter + counter block, a pair of events), one set per stream in use: launches of one
    // scene on different streams may overlap (a frame's drain with the next frame's start), launches on one stream are
    // ordered by the stream.  A slot is bound to the stream that used it last; when all are bound to other streams the
    // least recently used one is waited for and rebound.
    static const int N_SLOTS = 4;
    struct LaunchSlot {
        void* d_queue = nullptr; void* d_stats = nullptr;
        void* ev_start = nullptr; void* ev_stop = nullptr;
        void* stream = nullptr; bool recorded = false, timed = true; unsigned long long seq = 0;
        unsigned long long group = 0; bool harvested = true;   // the frame this launch is a share of; its counters have joined Scene::acc_stats
    };
    // Everything that lives on ONE HIP device: the replicated scene tables (the scene is < 2 MB: every device that renders
    // a share of the frame holds its own copy), launch scratch, and rt_render_multi's per-device buffers.
    struct DeviceCtx {
        int device = -1;
        DeviceScene<double> dev64;
        DeviceScene<float> dev32;
        LaunchSlot slots[N_SLOTS];
        int last_slot = -1;                // slot of the most recent launch on this device
        void* stream = nullptr;            // rt_render_multi: this device's stream, tile buffer, and (root only) gather / frame buffers
        void* d_tiles = nullptr; size_t tiles_bytes = 0;
        void* d_gather = nullptr; size_t gather_bytes = 0;
        void* d_frame[2] = {nullptr, nullptr}; size_t frame_bytes[2] = {0, 0};      // root only: consecutive frames alternate, so that frame i can still be read while frame i + 1 is un-permuted
    };
    std::vector<DeviceCtx*> ctxs;          // created on first use of a device
    int last_device = -1;                  // device of the most recent launch (what rt_last_* report)
    unsigned long long launch_seq = 0;
    uint32_t launch_info[6] = {0, 0, 0, 0, 0, 0};      // rt_last_launch_info
    // rt_render_multi: RCCL communicators of the last device set (csrc/rt_multi.cpp)
    std::vector<int> comm_devices; std::vector<void*> comms;
    double multi_ms[4] = {0, 0, 0, 0};     // last rt_render_multi*: slowest device's kernel, gather, un-permute, whole call (wall)
    // state of the rt_render_multi* frame in flight (rt_multi_sync settles it): the devices it runs on, four events on the root
    // device's stream, the host clock at the call's entry; the virtual-rank test hook's tile buffers (all on the root device)
    std::vector<int> multi_devs; bool multi_pending = false; double multi_t0 = 0.0;
    unsigned long long multi_sig[4] = {0, 0, 0, 0};      // shape of the frame in flight (W, H, tile_px, device mask | flags): a differently shaped next frame waits for it
    void* multi_ev[4] = {nullptr, nullptr, nullptr, nullptr}; int multi_ev_device = -1;
    std::vector<void*> virtual_tiles; size_t virtual_tiles_bytes = 0; int virtual_tiles_device = -1;
    double kernel_ms_total = 0.0; unsigned long long kernel_launches_timed = 0;      // rt_kernel_time_total
    // per-launch kernel times by launch number (a small ring: a slot's events are reused by later launches) and, for the last
    // rt_render_multi* frame, the launch number / HIP device of every rank and the size RCCL reports for the communicator the gather
    // ran on (0: no collective ran — one device, or the virtual-rank test hook) — rt_last_multi_ranks
    static const int SEQ_RING = 64;
    double seq_ms[SEQ_RING] = {0}; unsigned long long seq_tag[SEQ_RING] = {0};
    std::vector<unsigned long long> multi_rank_seq; std::vector<int> multi_rank_dev; int multi_comm_count = 0;
    // One frame may be several launches (rt_render_multi: one per device or virtual rank).  Every launch carries its frame's number;
    // the counters of a frame's launches are summed in acc_stats (rt_last_stats reports the whole frame, not one rank's share).
    unsigned long long frame_group = 0; bool group_open = false;      // group_open: rt_render_multi is enqueueing the launches of ONE frame
    unsigned long long acc_group = 0; unsigned long long acc_stats[16] = {0};
    unsigned long long multi_frames = 0; size_t multi_frame_doubles = 0;      // rt_render_multi* frames enqueued so far; W*H*3 of the most recent one

    // debugging aid (rt_debug_trace_path; -DRT_TRACE_PATH builds of the kernels): the path whose hits are recorded, and the device buffer
    long long trace_px = -1, trace_s = -1; void* d_trace = nullptr; int trace_device = -1; uint32_t trace_levels = 0;     // trace_levels: 16-double records d_trace holds

    void invalidate() { flat_valid = false; loop_choice = -1; loop_how = 0; loop_ms[0] = loop_ms[1] = 0.f; filter_key = 0; }
    DeviceCtx& ctx_for(int device) {
        for (DeviceCtx* c : ctxs) if (c->device == device) return *c;
        DeviceCtx* c = new DeviceCtx(); c->device = device; ctxs.push_back(c); return *c;
    }
    DeviceCtx* last_ctx() { for (DeviceCtx* c : ctxs) if (c->device == last_device) return c; return nullptr; }
};

// rt_flatten.cpp
bool flatten_scene(Scene& s);
template <typename T> struct DCamera;
bool tune_filter_tree(Scene& s, const DCamera<double>& cam);      // one-BVH worlds: contraction by a view's estimated pass rates; true if the filter tree was rebuilt
// rt_host.cpp (shared with rt_multi.cpp)
int set_error(const std::string& m);                              // leaves the message for rt_last_error(); returns -1
int device_kernel_ms(Scene& s, int device, float* ms);            // duration of the last path-tracing kernel launched on `device`
int settle_all_launches(Scene& s);                                // wait for every launch in flight; its kernel time joins the totals and the per-launch ring
int prepare_device(Scene& s, Scene::DeviceCtx& c, uint32_t flags); // rt_scene_prepare's work for one device (current device = c.device)
bool flatten_for_render(Scene& s);                                // flatten_scene + rt_last_error on failure
void multi_release(Scene& s);                                     // rt_multi.cpp: destroys the cached RCCL communicators
// camera (src/camera.rs:19-49)
struct rt_camera_args { double lookfrom[3], lookat[3], vup[3], vfov, aspect, aperture, focus_dist, time0, time1; };
void camera_new(const rt_camera_args& a, DCamera<double>& out);

} // namespace rt

struct rt_scene { rt::Scene s; };       // the opaque handle of include/rt_amd.h
struct rt_camera;
namespace rt {
int calibrate_if_worth_it(::rt_scene* sc, const ::rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth, uint64_t seed, uint32_t flags);   // rt_render_multi: the implicit loop-shape calibration of a first large frame (current device = the frame's first)
}

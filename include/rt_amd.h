/* include/rt_amd.h — C-ABI of the MI355X-native path tracer (librt_amd.so).
 *
 * Drop-in boundary for the per-pixel sample loop of 4meame/RayTracingInRust.  The reference has
 * no FFI/plugin interface of its own (one binary, `fn main`, src/main.rs:577); the seam is the
 * per-pixel closure src/main.rs:811-830 + `format_color` src/main.rs:832.  A Rust `main.rs`
 * keeps its scene functions and camera set-up, mirrors them through the builder entry points
 * below (one entry point per reference constructor, cited next to each), and replaces the
 * `for j / for i / into_par_iter().map().sum()` loop nest with ONE call to rt_render(); see
 * INTEGRATION.md for the `extern "C"` block.
 *
 * Conventions: plain C, no torch / HIP types.  Handles are small non-negative ints scoped to one
 * rt_scene; every function that can fail returns a negative value (builders) or non-zero status
 * (render) and leaves a message for rt_last_error().  The reference's failure mode is panic
 * (src/bvh.rs:28,55,61; src/hit.rs:95; src/main.rs:431) — an error code here.  All reals are f64,
 * as in the reference (src/vec.rs:10-12).
 */
#ifndef RT_AMD_H
#define RT_AMD_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_scene rt_scene;
typedef struct rt_rng rt_rng;

/* enum Plane, src/rect.rs:9-13;  enum Axis, src/rotate.rs:8-12 */
enum { RT_PLANE_XY = 0, RT_PLANE_XZ = 1, RT_PLANE_YZ = 2 };
enum { RT_AXIS_X = 0, RT_AXIS_Y = 1, RT_AXIS_Z = 2 };

/* Camera::new arguments, src/camera.rs:19 */
typedef struct rt_camera {
    double lookfrom[3], lookat[3], vup[3];
    double vfov, aspect, aperture, focus_dist, time0, time1;
} rt_camera;

/* render flags */
enum {
    RT_F64 = 0,            /* reference precision (default): every operation in f64                         */
    RT_F32 = 1,            /* throughput variant: f32 arithmetic (statistical parity only)                   */
    RT_STOP_ON_ZERO = 2,   /* opt-in: end a path whose throughput is exactly (0,0,0); differs from the
                              reference only where a later bounce would have produced NaN (0*NaN)            */
    RT_NEAR_FIRST_BVH = 8, /* opt-in order (fewer node visits; faster only on some scenes): BVH children are visited nearer-first (by the ray's sign on the split axis) instead
                              of the reference's always-left-first (src/bvh.rs:81-84).  Same closest hit; exact-t ties are
                              still resolved by the reference's DFS order.  Can differ only where a last-ulp box cull depends
                              on the order hits are found.                                                                   */
    RT_PERSISTENT_BVH = 16, /* scheduling only, same samples: lanes keep their place inside a BVH while the rest of the wavefront
                              shades / regenerates (mesh kernels).  Chosen automatically for triangle-mesh BVHs that stand
                              beside other top-level objects: by tree size, or by a calibration of the view (rt_scene_calibrate;
                              rt_render / rt_render_multi run one at a scene's first frame of >= 1e8 samples); this flag forces it on ... */
    RT_LOCKSTEP_BVH = 32,  /* ... and this one forces the lock-step loop                                                      */
    RT_MULTI_COLLECTIVE = 64, /* rt_render_multi only: run the RCCL gather even when one device is selected (a one-GPU box then
                              exercises the same collective calls as an 8-GPU node)                                          */
    RT_SPECULATE_BVH = 1024, /* scheduling only, same samples (lock-step BVH kernel): a lane that has reached a leaf walks on while it waits for the
                              leaf step (rt_kernel.hip: bvh_hit_filt, SPEC).  Chosen automatically for scenes whose world is one BVH (every ray
                              enters it); this flag forces it on ...                                                          */
    RT_NO_SPECULATE_BVH = 2048, /* ... and this one off                                                                        */
    RT_ISOTROPIC_SCATTER = 4 /* opt-in, NOT the committed reference behaviour: Isotropic (constant media) scatters with its
                              old `scatter` (src/mat.rs:417-421) instead of absorbing — the look of img/volume.png       */
};

const char* rt_last_error(void);
/* number of visible HIP devices (0 without a GPU; never fails) */
int rt_device_count(void);

/* ---- seeded stream replacing rand::thread_rng() for HOST-side scene construction
 *      (src/main.rs:154,457; src/perlin.rs:5,22).  Spec: raytracinginrust_amd/csrc/rt_rng.h          */
rt_rng* rt_rng_create(uint64_t seed, uint32_t stream);
void rt_rng_destroy(rt_rng*);
double rt_rng_f64(rt_rng*);                       /* rng.gen::<f64>()                 */
double rt_rng_range(rt_rng*, double a, double b); /* rng.gen_range(a..b)              */
int rt_rng_bool(rt_rng*);                         /* rng.gen::<bool>()                */
uint32_t rt_rng_index(rt_rng*, uint32_t n);       /* rng.gen_range(0..n)              */
uint32_t rt_rng_u32(rt_rng*);
void rt_rng_path(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t state_out[4]);

/* ---- scene ------------------------------------------------------------------------------- */
rt_scene* rt_scene_create(void);
void rt_scene_destroy(rt_scene*);
const char* rt_scene_error(rt_scene*);

/* textures, src/texture.rs:16,37,63,90 */
int rt_texture_constant(rt_scene*, const double rgb[3]);              /* ConstantTexture::new  */
int rt_texture_check(rt_scene*, int odd, int even);                   /* CheckTexture::new     */
int rt_texture_noise(rt_scene*, double scale, rt_rng* rng);           /* NoiseTexture::new (Perlin::new draws from rng, src/perlin.rs:67-75) */
int rt_texture_image(rt_scene*, const uint8_t* rgb8, uint32_t width, uint32_t height);   /* ImageTexture::new */

/* materials, src/mat.rs:101,205,260,303,383,410 */
int rt_material_lambertian(rt_scene*, int texture);
int rt_material_metal(rt_scene*, const double albedo[3], double fuzz);
int rt_material_dielectric(rt_scene*, double index_of_refraction);
int rt_material_diffuse_light(rt_scene*, int texture);
int rt_material_isotropic(rt_scene*, int texture);
/* PBR::new (the principled "Disney" material), src/mat.rs:101: params = metallic, subsurface, specular, roughness,
 * specular_tint, anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss (the constructor's argument order) */
int rt_material_pbr(rt_scene*, int base_color_texture, const double params[10]);

/* hittables */
int rt_sphere(rt_scene*, const double center[3], double radius, int material);                         /* Sphere::new, src/sphere.rs:46        */
int rt_moving_sphere(rt_scene*, const double c0[3], const double c1[3], double t0, double t1,
                     double radius, int material);                                                     /* MovingSphere::new, src/sphere.rs:133 */
int rt_aarect(rt_scene*, int plane, double a0, double a1, double b0, double b1, double k, int material); /* AARect::new, src/rect.rs:35         */
int rt_cube(rt_scene*, const double min[3], const double max[3], int material);                        /* Cube::new, src/cube.rs:14            */
int rt_triangle(rt_scene*, const double v[9], int material);                                           /* Triangle::new, src/tri.rs:15         */
int rt_list_create(rt_scene*);                                                                         /* HittableList::default, src/hit.rs:46 */
int rt_list_push(rt_scene*, int list, int hittable);                                                   /* HittableList::push, src/hit.rs:52    */
int rt_mesh(rt_scene*, const double* positions, uint32_t n_positions, const uint32_t* indices,
            uint32_t n_indices, int material);                       /* Mesh::new, src/mesh.rs:16 — returns the `tris` HittableList */
/* Mesh::load_obj(path, offset, scale, material), src/mesh.rs:33-61: tobj semantics for what the path uses (f32 positions widened,
 * fan triangulation, models[0] only, then `* scale + offset`); returns the `tris` HittableList; a malformed file is an error
 * (the reference unwraps tobj's Err, src/main.rs:431).  rt_parse_obj is its parsing step on a buffer: malloc'ed
 * positions (3 doubles each) and triangle indices, release with rt_free. */
int rt_mesh_load_obj(rt_scene*, const char* path, const double offset[3], double scale, int material);
int rt_parse_obj(const char* data, size_t size, const double offset[3], double scale, double** positions_out,
                 uint32_t* n_positions_out, uint32_t** indices_out, uint32_t* n_indices_out);
int rt_flip_normal(rt_scene*, int hittable);                                                           /* FlipNormal::new, src/hit.rs:105      */
int rt_translate(rt_scene*, int hittable, const double offset[3]);                                     /* Translate::new, src/translate.rs:13  */
int rt_rotate(rt_scene*, int axis, int hittable, double angle_deg);                                    /* Rotate::new, src/rotate.rs:32        */
int rt_constant_medium(rt_scene*, int boundary, double density, int texture);                          /* ConstantMedium::new, src/medium.rs:17 */
/* BVH::new, src/bvh.rs:18-73: `hittables` may be handles of ANY kind, as the reference's Vec<Box<dyn Hittable>> — bare primitives, lists,
 * FlipNormal / Translate / Rotate of anything (a Rotate child has the whole-space box of src/rotate.rs:40-57: such a tree is walked with
 * the exact box test everywhere), ConstantMedium, another BVH (one level of BVHs inside BVH leaves; deeper is an error at flatten time).
 * Errors where the reference panics: n = 0 ("no object in the scene", src/bvh.rs:55); a child without a bounding box — an empty list or a
 * wrapper of one ("no bounding box in bvh node", src/bvh.rs:28,61) — at flatten time. */
int rt_bvh(rt_scene*, const int* hittables, uint32_t n, double time0, double time1);
int rt_bvh_of_list(rt_scene*, int list, double time0, double time1);                                   /* BVH::new(list.list, ..), src/main.rs:442 */

/* the (world, lights) pair every scene fn returns, src/main.rs:153,278,348,453 */
int rt_scene_set_world(rt_scene*, int hittable);
int rt_lights_push(rt_scene*, int hittable);

/* ---- host-side pieces of the boundary ------------------------------------------------------ */
/* Camera::new, src/camera.rs:19-49: origin, lower_left_corner, horizontal, vertical, cu, cv (3 each),
 * lens_radius, time0, time1 */
void rt_camera_fields(const rt_camera*, double out21[21]);
/* Vec3::format_color, src/vec.rs:125-131 (NaN -> 0, +inf -> 255) */
void rt_format_color(const double rgb_sum[3], uint64_t samples_per_pixel, uint64_t out3[3]);
/* PPM emitter, src/main.rs:767-769,832: "P3\nW H\n255\n" then one "r g b" line per pixel, rows top to
 * bottom.  rgb_sum is W*H*3 in output order (row 0 = top).  path NULL or "-" writes to stdout. */
int rt_write_ppm(const char* path, const double* rgb_sum, uint32_t W, uint32_t H, uint64_t samples_per_pixel);

/* Host asset ingest, `image::open(path).to_rgb8()` (src/main.rs:248,491): decodes a baseline JPEG (8-bit, grey or YCbCr,
 * 1x1 sampling — the class of the reference's earthmap.jpg) held in memory into a malloc'ed interleaved RGB8 buffer
 * (release with rt_free); NULL + rt_last_error() on anything else.  The result feeds rt_texture_image(). */
uint8_t* rt_decode_jpeg_rgb8(const uint8_t* data, size_t size, uint32_t* width, uint32_t* height);
void rt_free(void*);

/* Flatten the Hittable tree into the device scene (no GPU needed); fills counts for inspection:
 * objects, ops, rects, spheres, moving spheres, triangles, bvh nodes, materials, textures, lights, media, perlins */
int rt_scene_flatten(rt_scene*, uint32_t counts_out[12]);

/* ---- the hot path: replaces src/main.rs:772-833 -------------------------------------------- */
/* Renders the whole frame on the current HIP device and returns, per pixel, the SUM over samples of
 * ray_color (what `.sum()` yields at src/main.rs:830) as W*H*3 doubles in output order (row 0 = top,
 * i.e. j = H-1).  Fails (non-zero) if no GPU/HIP device is present: there is no CPU fallback. */
int rt_render(rt_scene*, const rt_camera*, const double background[3], uint32_t W, uint32_t H,
              uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, uint32_t flags,
              double* rgb_sum_out);

/* Tile-sharded form for one-process-per-GPU use.  The image's W*H pixels (output order) are cut into
 * tiles of `tile_px` consecutive pixels; this call renders tiles t with t % world_size == rank into
 * d_out (DEVICE pointer, n_local_tiles * tile_px * 3 doubles, tile-major; pixels past W*H are zero),
 * asynchronously on `hip_stream` (a hipStream_t, may be NULL).  n_local_tiles =
 * rt_local_tiles(W,H,tile_px,rank,world_size) is the same on every rank (padded), so a plain gather
 * reassembles the frame. */
uint32_t rt_local_tiles(uint32_t W, uint32_t H, uint32_t tile_px, uint32_t rank, uint32_t world_size);
int rt_render_device(rt_scene*, const rt_camera*, const double background[3], uint32_t W, uint32_t H,
                     uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, uint32_t flags,
                     uint32_t tile_px, uint32_t rank, uint32_t world_size,
                     void* d_out, size_t d_out_bytes, void* hip_stream);
/* (rt_render_device and rt_render_multi_device never wait: everything that has to — the first use of a scene on a device, see
 * rt_scene_prepare, and the loop-shape calibration of a mesh scene, see rt_scene_calibrate — is done inside them only as far as it can
 * be done without a wait; call those two first where it matters.) */
/* The whole frame on several GPUs of this node from ONE call: what a host that owns the node's GPUs itself (the reference's `main`,
 * src/main.rs:767-835) calls instead of rt_render.  device_mask: bit d selects HIP device d (0 = every visible device).  The scene
 * is replicated on each selected device; tiles of tile_px output-order pixels (0 = the default, 67) are dealt round-robin, each
 * device renders its share with one persistent launch, ONE ncclGather (RCCL over xGMI; communicators from ncclCommInitAll, cached
 * with the scene; librccl is loaded on first use) brings the packed tiles to the first selected device, which un-permutes them on the
 * device; rgb_sum_out receives W*H*3 doubles in output order, as from rt_render.  Synchronous.  A failure on any device leaves
 * nothing in flight, keeps no temporary and restores the caller's current HIP device.
 * rt_last_multi_ms (waits for the frame): [0] slowest device's kernel, [1] the gather as the first device's stream sees it
 * (from the end of its own kernel: the collective including the wait for the slowest other device), [2] un-permute, [3] host clock
 * from the call's entry to the end of the wait (rt_render_multi: of the whole call, transfer included), in ms. */
int rt_render_multi(rt_scene*, const rt_camera*, const double background[3], uint32_t W, uint32_t H,
                    uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, uint32_t flags,
                    uint32_t device_mask, uint32_t tile_px, double* rgb_sum_out);
/* The same frame left in device memory: returns once every device's work is enqueued (kernels, gather, un-permute); *d_frame_out (may
 * be NULL) receives a DEVICE pointer on the first selected device to W*H*3 doubles in output order, owned by the scene.  Consecutive
 * frames alternate between TWO such buffers, so a frame's pointer stays valid (and its pixels untouched) until the next-but-one
 * rt_render_multi* call: a caller can read frame i while frame i + 1 is rendered.  Calls may follow each other without waiting: each device's work is ordered by its own
 * stream and the host runs at most one frame ahead, so the devices go from frame to frame without a launch gap (a frame of another
 * shape or device set first waits for the one in flight); timings are kept for the most recent frame.
 * rt_multi_sync waits for the frames in flight; rt_multi_copy_frame waits and copies the first n_doubles (<= W*H*3) doubles of the most
 * recent frame to host memory.
 * rt_render_multi = rt_render_multi_device + rt_multi_copy_frame. */
int rt_render_multi_device(rt_scene*, const rt_camera*, const double background[3], uint32_t W, uint32_t H,
                           uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, uint32_t flags,
                           uint32_t device_mask, uint32_t tile_px, void** d_frame_out);
int rt_multi_sync(rt_scene*);
int rt_multi_copy_frame(rt_scene*, double* rgb_sum_out, size_t n_doubles);
int rt_last_multi_ms(rt_scene*, double out4[4]);
/* The ranks of the most recent rt_render_multi* frame (waits for it): *n_ranks_out = how many launches made the frame; for the first
 * max_ranks of them device_out[r] = the HIP device rank r ran on, kernel_ms_out[r] = its path-tracing kernel's duration (HIP events
 * on its stream; -1 if no longer known); *collective_ranks_out = the size RCCL reports for the communicator the frame's gather ran
 * on (ncclCommCount), 0 when no collective ran (one device without RT_MULTI_COLLECTIVE, or the virtual-rank test hook).  What a
 * multi-GPU caller checks before it trusts a frame: N distinct devices, N ranks in the collective, no rank with a ~0 ms kernel. */
int rt_last_multi_ranks(rt_scene*, uint32_t max_ranks, uint32_t* n_ranks_out, int* device_out, double* kernel_ms_out,
                        uint32_t* collective_ranks_out);
/* How BVH objects are built when the scene is flattened (at the first render / rt_scene_prepare after a change).
 * RT_BVH_MEDIAN (default) is BVH::new, src/bvh.rs:18-73: widest axis, object median.  RT_BVH_SAH is an opt-in fast mode
 * (binned surface-area heuristic, one object per leaf as in the reference): same closest hits; the order in which
 * exactly-equal-t hits are met, and last-ulp box culls, may differ — like RT_NEAR_FIRST_BVH. */
enum rt_bvh_builder { RT_BVH_MEDIAN = 0, RT_BVH_SAH = 1 };
/* Tuning knob of the persistent-traversal loop (scheduling only, never results): a traversal pass starts once `start_at` lanes of a
 * wavefront are inside a BVH and runs until fewer than `stop_below` are still walking; primitives are tested once `leaf_share64`/64
 * of the walking lanes hold a pending leaf.  Defaults 56, 16, 32 (measured best on the teapot room; the optimum is flat). */
int rt_scene_set_traversal_schedule(rt_scene*, uint32_t start_at, uint32_t stop_below, uint32_t leaf_share64);
int rt_scene_set_bvh_builder(rt_scene*, int mode);
/* Optional: do now what the first render of this scene would do once inside its call (flatten, upload for the precision in
 * `flags`, load the kernel's code object).  Launches nothing. */
int rt_scene_prepare(rt_scene*, uint32_t flags);
/* Mesh scenes (triangle-mesh BVHs beside other top-level objects) run a persistent-traversal or a lock-step loop — bit-identical samples;
 * which is faster depends on what the rays of a VIEW do inside the trees, not on the trees' size.  rt_scene_calibrate measures it:
 * the same view at <= 1024 x 1024 x 16 samples, twice in each shape (four launches, ~30 ms), SYNCHRONOUSLY on the calling thread's
 * current device, and keeps the faster shape for that view (camera, W, H, precision; another view falls back to the size rule until it
 * is calibrated itself; a change to the scene forgets everything).  A no-op for every other kind of scene and for a view already
 * measured.  rt_render, rt_render_samples and rt_render_multi do this themselves at a scene's first frame of >= 1e8 samples;
 * rt_render_device and rt_render_multi_device — asynchronous — never do.
 * The same call (and the same three synchronous entry points, for every new view) tunes the filter tree of a world that is ONE bare BVH —
 * the random-spheres scene — for the view: which inner nodes the kernels' box steps skip over is decided by the nodes' estimated pass
 * rates (a few thousand rays through the tree on the host, < 1 ms) instead of by box areas; random spheres +6 %, samples unchanged
 * (rt_flatten.cpp tune_filter_tree).
 * rt_scene_set_loop_shape: 1 persistent traversal, 0 lock-step, for every view until the scene changes, -1 forgets (how the ranks of a
 * one-process-per-GPU job all run the shape rank 0 measured).
 * rt_last_loop_info: what the most recent launch ran — out4[0] loop shape (0: a list scene's kernel, 1: lock-step BVH, 2: persistent
 * traversal), [1] the FEATS template argument of the instantiation (its name in a profile: rt::pathtrace_kernel<double, FEATSu>),
 * [2] how the shape was chosen (0 the scene leaves no choice, 1 size rule, 2 calibration, 3 the caller's flag, 4 rt_scene_set_loop_shape),
 * [3] precision (0 f64, 1 f32); calibration_ms2 (may be NULL): the stored calibration's kernel times {lock-step, persistent}, 0 if none. */
int rt_scene_calibrate(rt_scene*, const rt_camera*, const double background[3], uint32_t W, uint32_t H,
                       uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, uint32_t flags);
int rt_scene_set_loop_shape(rt_scene*, int shape);
int rt_scene_loop_shape(rt_scene*);      /* the stored shape: 1 / 0, -1 none (a calibration's result applies to the view it measured) */
int rt_last_loop_info(rt_scene*, int32_t out4[4], float calibration_ms2[2]);
/* Milliseconds of the most recent path-tracing kernel launched by this library on this thread's scene,
 * from HIP events recorded on the launch stream (blocks until that kernel finishes). */
int rt_last_kernel_ms(rt_scene*, float* ms_out);
/* Total duration (ms, HIP events on the launch streams) and number of this scene's path-tracing kernels since the last
 * reset; waits for launches still in flight.  Launches of one scene on different streams may overlap (up to four streams; a
 * frame's drain then hides behind the next frame's start); launches on one stream are ordered by the stream. */
int rt_kernel_time_total(rt_scene*, double* ms_total, unsigned long long* n_launches, int reset);
/* Counters of the most recent FRAME — one launch for rt_render / rt_render_device, the N launches of an rt_render_multi* frame summed
 * (N devices or virtual ranks) —: [0] samples whose radiance was not finite (the reference's
 * 0*inf / x/0 cases), [1] bounce-loop iterations summed over wavefronts, [2] lane-iterations
 * that carried a live path ([2] / (64*[1]) = lane utilisation). */
int rt_last_stats(rt_scene*, unsigned long long out3[3]);
/* Geometry of the most recent launch: [0] workgroups, [1] threads per workgroup, [2] dynamic LDS bytes per workgroup, [3] BVH nodes
 * staged in LDS by each workgroup (the top levels of the trees), [4] BVH nodes in the scene, [5] resident workgroups per CU. */
int rt_last_launch_info(rt_scene*, uint32_t out6[6]);
/* Per-pixel accumulator flushes of the most recent finished launch; each is three f64 atomic adds to the frame — the
 * kernel's only global write traffic (used to calibrate the WRITE_SIZE counter). */
int rt_last_flush_count(rt_scene*, unsigned long long* out);
/* Scenes with a BVH (persistent-traversal kernel): [0] advance passes, [1] lanes taking part in them, [2] traversal
 * steps, [3] lanes stepping, all summed over wavefronts ([3] / (64*[2]) = lane utilisation of the traversal). */
int rt_last_traversal_stats(rt_scene*, unsigned long long out4[4]);
/* Persistent-traversal kernels: of the traversal steps above, [0] the leaf steps (primitive tests) and [1] the lanes in them. */
int rt_last_leaf_steps(rt_scene*, unsigned long long out2[2]);
/* Diagnostic builds (-DRT_DIAG) only (zeros in a normal build): [0..5] wave-cycle sums of the six kernel sections, [6] rect tests
 * counted per wavefront, [7] those among them in which no lane's plane distance lay in [t_min, closest] (-DRT_DIAG_RECTS builds). */
int rt_debug_section_cycles(rt_scene*, unsigned long long out8[8]);
/* Test aid (host only, no GPU): what rt_scene_calibrate does to the filter tree of a world that is ONE bare BVH — inner nodes leave it by
 * their estimated pass rate for the view instead of by box areas (rt_flatten.cpp tune_filter_tree; scheduling only: any conservative
 * hierarchy over the leaves gives the same samples) — without touching a device copy.  1: rebuilt, 0: not that kind of scene, -1: error. */
int rt_debug_tune_filter(rt_scene*, const rt_camera*);
/* Test aid (host only, no GPU): the flattened object table, out[8 i ..] = {geometry kind (0 rect, 1 sphere, 2 moving sphere, 3 triangle,
 * 4 BVH root), first primitive / root node, count, first wrapper op, number of wrapper ops, medium index (0xFFFFFFFF: none), is_cube
 * (1: the six rects are one Cube's faces; 2 | map << 8: a ROOM — bare AARects of a list scene that are exact faces of one axis-aligned box,
 * searched as one object where the last of them stood, through the Cube fast path with a face map: three bits per face in cube.rs:17-24
 * order = the wall's place in the room's run of rect records, 7 = no such wall; first_op then holds five bits per wall, the index of the
 * first object that stood after that wall and is searched before the room — an exact tie with an object at or beyond it goes to that
 * object, as hit.rs:62-68 gives it to the later item; rt_flatten.cpp form_room), nest (sub-objects: wrapper ops outside the enclosing BVH | ops outside the medium << 8; bit 16:
 * every wrapper is a FlipNormal)}: the
 * world's top-level objects (HittableList push order,
 * runs of bare primitives merged) first — *n_top_out of them — then the sub-objects BVH leaves of other Hittable kinds refer to, or — a list
 * scene in which a room was formed — the world list as the reference has it (what a wave searches when one of its rays could produce a NaN
 * plane distance, rt_kernel.hip world_hit_list).
 * Returns the number of objects or -1. */
int rt_debug_objects(rt_scene*, uint32_t* out, uint32_t max_objects, uint32_t* n_top_out);
/* Test aid (host only, no GPU): the flattened BVH's link words, out[4*i..] = node i's {a, b, c, skip}: `a` bit 31 marks a leaf (then kind
 * and first primitive; b = count, c = the leaf's rank in the reference's depth-first order), otherwise a = split axis, b = right child,
 * c = left child; skip = the node BVH::hit's recursion (src/bvh.rs:77-91) reaches next once this node's subtree is finished or culled
 * (0xFFFFFFFF: the search is over).  roots_out: root node of every BVH object of the world list.  Returns the node count or -1. */
int rt_debug_bvh_links(rt_scene*, uint32_t* out, uint32_t max_nodes, uint32_t* roots_out, uint32_t max_roots, uint32_t* n_roots_out);
/* Test aid (host only, no GPU): the filter tree the kernels' box steps walk — the f32 companion of every node of the tree above
 * (same ids): boxes6_out[6 i ..] = {min.x, max.x, min.y, max.y, min.z, max.z} rounded OUTWARD to f32, links2_out[2 i ..] = {skip, info}
 * (info: the first child, or for a leaf its own id | 0x40000000; near-duplicate inner nodes have been taken out of these links),
 * f64_boxes_out[6 i ..] = the exact box {min[3], max[3]}; *filter_m_out >= every |coordinate| of the f32 boxes (0: filter off).
 * Any of the output pointers may be NULL.  Returns the node count or -1. */
int rt_debug_filter_nodes(rt_scene*, float* boxes6_out, uint32_t* links2_out, double* f64_boxes_out, uint32_t max_nodes, float* filter_m_out);
/* Test aid: AABB::hit (src/aabb.rs:19-36) evaluated on the device for n (box, ray, [t_min, t_max]) triples given as host arrays
 * (boxes: min[3] max[3]; rays: origin[3] direction[3]).  out[i] bit 0: hit by the reference's form; bit 1: by the NaN-free form the
 * leaf steps use for tame rays; bit 2: the ray qualifies for that form (finite 1/d, |origin| < 1e300); bit 3: ray and box are inside the
 * ranges of the box steps' conservative f32 filter (rt_kernel.hip: make_filter); bit 4: that filter lets the box through (it must
 * wherever bit 0 is set; it may elsewhere); bits 5, 6, 7: for the f32 kernels (RT_F32) — AABB::hit's form in f32 on the inputs rounded to
 * nearest (what their leaf steps run), tame ray and filter ranges, their filter's verdict (it must pass wherever bit 5 is set).
 * Non-zero on a HIP error. */
int rt_debug_aabb_hit(uint32_t n, const double* boxes, const double* rays, const double* tlim, int* out);
/* Test aid: Cube::hit (src/cube.rs:14-36: HittableList::hit over six AARects) on the device for n (cube, ray, [t_min, t_max]) triples
 * given as host arrays (boxes: min[3] max[3]; rays: origin[3] direction[3]); rect_m as KParams::rect_m (>= every |coordinate|).
 * out[4 i ..] = t of the six exact rect tests in cube.rs order (NaN: no face accepted), the accepted face (0..5 in cube.rs:17-24 order,
 * -1), t of the fast path (rt_kernel.hip: cube_fast; NaN: no hit), 8 * clear + (face + 1) of the fast path (clear:
 * the lane's outcome is outside the approximation's margin — otherwise the kernels run the six tests).  Non-zero on a HIP error. */
int rt_debug_cube_hit(uint32_t n, double rect_m, const double* boxes, const double* rays, const double* tlim, double* out);
/* The same for the fast path's ROOM form (round 6: walls of a list scene that are faces of one box, rt_flatten.cpp form_room): masks[i]
 * says which of the six faces exist (bit f = face f in cube.rs:17-24 order); the exact side is HittableList::hit (hit.rs:59-71) over
 * the AARects of those faces.  out as above. */
int rt_debug_room_hit(uint32_t n, double rect_m, const double* boxes, const double* rays, const double* tlim, const uint32_t* masks, double* out);
/* Known-answer access to the closest-hit search of a LIST scene (no feature bit: what the lean kernels serve — rects, Cubes, wrapped ones):
 * world.hit (main.rs:48, HittableList::hit hit.rs:59-71) and the hit record for n given rays through the kernels' own search.  rays: n x
 * (origin[3], direction[3]); t_min: n; out: n x 12 = hit (0 / 1), t, position[3], normal[3], front_face, object index, primitive index,
 * material.  Host pointers.  What the room form's order argument rests on is tested through this: rays that start ON planes, with zero
 * direction components (0 / 0 plane distances), non-finite rays. */
int rt_debug_list_hit(rt_scene*, uint32_t n, const double* rays, const double* t_min, double* out);
/* Debugging aid for parity work: the hits of ONE camera path, level by level.  rt_debug_trace_path chooses the path (local pixel index =
 * output-order pixel for an unsharded render, sample index; -1 switches it off); the following renders record, per level of ray_color
 * that found a hit, 16 doubles at out[16 * level]: t, position[3], normal[3], front_face, object, primitive kind, primitive index,
 * material, incoming direction[3], 1.0 (a level without a hit stays all zero); rt_debug_get_trace fetches n_levels of them.  Only the
 * lock-step kernels of a -DRT_TRACE_PATH build of the library write the record (tools/mkab.sh trace "-DRT_TRACE_PATH" "-DRT_TRACE_PATH";
 * the shipped build leaves it zero) — tests/sweeps/fuzz_probe.py compares it with the oracle's orc_trace_path. */
int rt_debug_trace_path(rt_scene*, long long local_pixel, long long sample);
int rt_debug_get_trace(rt_scene*, double* out, uint32_t n_levels);
/* Debug/parity aid: like rt_render but also returns every sample's radiance (W*H*spp*3 doubles). */
int rt_render_samples(rt_scene*, const rt_camera*, const double background[3], uint32_t W, uint32_t H,
                      uint32_t samples_per_pixel, uint32_t max_depth, uint64_t seed, uint32_t flags,
                      double* rgb_sum_out, double* samples_out);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */

// csrc/rt_ir.h — the flat device scene ("IR") the host flattener produces and the kernels read.
//
// The reference scene is a tree of `Box<dyn Hittable>` with `&dyn Material` / `dyn Texture` leaves
// (src/hit.rs:26-31, src/mat.rs:54-77, src/texture.rs:5-7).  On the GPU every `dyn` becomes a small
// integer tag and every Box a table index:
//
//   world  = objects[0..n_objects)   in HittableList push order (src/hit.rs:59-71)
//   object = wrapper chain ops[first_op .. first_op+n_ops) outermost first (Translate / Rotate / FlipNormal),
//            optional ConstantMedium (outermost only), and one geometry:
//              a typed range of primitives (1 rect, 6 rects = Cube, n triangles = Mesh list, ...) or a BVH root
//   BVH    = nodes of all the scene's trees ordered by depth (roots first), both children stored: the first `n_cached` ids — the top
//            levels, where most visits go — are staged in LDS by each workgroup, the rest are fetched per lane; leaves hold a typed
//            primitive range (a sphere, a moving sphere, a triangle, or a Cube's 6 rects) — or, for any other Hittable BVH::new accepts
//            (bvh.rs:18-31: a list, a wrapped object, a medium, another BVH), a run of sub-objects (G_OBJ) — and their rank in DFS preorder
//   lights = light records (rect / sphere / "other" = trait default pdf 0, random (1,0,0))
//
// Records are templated on the arithmetic type: the f64 build is the reference-precision product path,
// the f32 build the throughput variant.  Top-level tables are read with wave-uniform indices (scalar
// loads); BVH nodes and leaf primitives with per-lane indices (vector gathers, node = one 64-byte line
// in f64).
#pragma once
#include <stdint.h>

namespace rt {

enum GeomKind : uint32_t { G_RECT = 0, G_SPHERE = 1, G_MSPHERE = 2, G_TRI = 3, G_BVH = 4, G_OBJ = 5 };      // G_OBJ: BVH leaves only — a run of sub-objects (below)
enum OpKind : uint32_t { OP_TRANSLATE = 0, OP_ROTATE = 1, OP_FLIP = 2 };
enum MatKind : uint32_t { M_LAMBERTIAN = 0, M_METAL = 1, M_DIELECTRIC = 2, M_DIFFUSE_LIGHT = 3, M_ISOTROPIC = 4, M_PBR = 5 };
// DMaterial::kind also carries, above the MatKind byte, whether the material's texture graph reads (u, v) at all (only ImageTexture
// does, texture.rs:99-120): the hit record's u, v — atan2 + acos per sphere hit — are computed only when something will read them.
static const uint32_t MAT_KIND_MASK = 0xFFu, MAT_NEEDS_UV = 0x100u;
enum TexKind : uint32_t { T_CONSTANT = 0, T_CHECK = 1, T_NOISE = 2, T_IMAGE = 3 };
enum LightKind : uint32_t { L_RECT = 0, L_SPHERE = 1, L_OTHER = 2 };

// scene feature bits: the host picks the leanest kernel instantiation that covers the scene
enum Feat : uint32_t {
    F_BVH = 1u << 0,        // any G_BVH geometry
    F_SPHERES = 1u << 1,    // spheres / moving spheres
    F_TRIS = 1u << 2,       // triangles
    F_MEDIUM = 1u << 3,     // ConstantMedium
    F_TEXTURES = 1u << 4,   // Check / Noise / Image textures
    F_DIELECTRIC = 1u << 5, // Dielectric material
    F_PBR = 1u << 6,        // principled material (PBR) + PDF::BRDF
    F_ALL = 0x7F,
    F_NEAR_FIRST = 1u << 7, // not a scene feature: selects the near-first BVH traversal instantiations (RT_NEAR_FIRST_BVH)
    F_PERSIST = 1u << 8,    // not a scene feature: selects the persistent-traversal loop (mesh scenes whose BVH few rays enter)
    F_NESTED = 1u << 9,     // a BVH leaf that is not a bare primitive / Cube (any other Hittable: a list, a wrapped object, a medium, another BVH), or a
                            // ConstantMedium under a wrapper: served by the all-features instantiation only (no BASELINE scene has either)
    F_SPEC = 1u << 11           // not a scene feature: lock-step BVH walk with speculative box steps (scenes whose world IS one BVH; RT_SPECULATE_BVH)
};

static const uint32_t BVH_LEAF = 0x80000000u;     // leaf: node.a = bit 31 | GeomKind << 28 | first index, node.b = count, node.c = rank in DFS
                                                   //       preorder (= the reference's visiting order: resolves exact-t ties in near-first mode);
                                                   // inner: node.a = split axis (0..2), node.b = right child, node.c = left child
static const int RT_MAX_OPS = 8;                   // wrapper chain length limit (a sub-object's chain counts the wrappers around its BVHs too)
static const int RT_MAX_BVH_DEPTH = 48;
static const int RT_MAX_NEST = 1;                  // BVHs inside BVH leaves: a BVH (level 0) may hold BVHs (level 1), whose own leaves may be any Hittable but another BVH
                                                   // (the kernel holds one copy of the walk per level: rt_kernel.hip bvh_hit_filt<.., NEST>)

// Records are 16-byte aligned (the BVH node a whole 64 / 32 bytes) so that a per-lane fetch is a few 16-byte loads inside one
// cache line instead of a string of 8-byte loads straddling two.
template <typename T> struct alignas(16) DRect { T a0, a1, b0, b1, k; uint32_t plane, mat; };            // src/rect.rs:16-24
template <typename T> struct alignas(16) DSphere { T c[3], r; uint32_t mat, pad; };                       // src/sphere.rs:38-43
template <typename T> struct alignas(16) DMSphere { T c0[3], c1[3], t0, t1, r; uint32_t mat, pad; };      // src/sphere.rs:122-129
template <typename T> struct alignas(16) DTri { T v0[3], e1[3], e2[3]; uint32_t mat, pad; };              // src/tri.rs:9-12 (e1 = v1-v0, e2 = v2-v0, as tri.rs:27-28 computes per hit)
template <typename T> struct alignas(16) DOp { uint32_t kind, axis; T x, y, z; };             // translate: offset; rotate: x = sin, y = cos (src/rotate.rs:23-30)
struct alignas(16) DObject { uint32_t geom_kind, geom_first, geom_count, first_op, n_ops; int32_t medium; uint32_t is_cube, nest; };   // is_cube: the 6 rects are ONE Cube's faces in cube.rs:17-24 order (a run of six bare AARects is not)
// nest (F_NESTED kernels; 0 everywhere else) = n_outer | med_at << 8.  A SUB-OBJECT — what a BVH leaf of kind G_OBJ holds: objects[first ..
// first + count) behind the world's n_objects — carries its WHOLE wrapper chain from the world down (the hit record is rebuilt from the
// world ray), of which the first n_outer ops lie outside its BVH: the walk's ray has them applied already, a leaf step applies the rest.
// med_at = the ops that lie outside the object's ConstantMedium (its free-flight length is measured with the ray as the medium receives
// it, medium.rs:40; 0 when the medium is the outermost wrapper, the only form the other kernels serve).
template <typename T> struct alignas(16) DBvhNode { T mn[3], mx[3]; uint32_t a, b, c, skip; };      // f64: 64 B = four 16-byte pieces of one line; f32: 48 B
// Conservative f32 companion of a BVH node (same id, same skip link), what the f64 kernels' box steps read (rt_kernel.hip: "filtered walk"):
// b = {min.x, max.x, min.y, max.y, min.z, max.z} rounded OUTWARD to f32; info = left child id (inner) or own id | FNODE_LEAF (leaf).
struct alignas(16) DFNode { float b[6]; uint32_t skip, info; };
static const uint32_t FNODE_LEAF = 0x40000000u;
template <typename T> struct alignas(16) DMaterial { uint32_t kind, tex; T albedo[3]; T param; };   // metal: albedo, fuzz; dielectric: param = ir; PBR: tex = base colour, albedo[0] = index into pbr[]
template <typename T> struct DPbr { T metallic, subsurface, specular, roughness, specular_tint, anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss; };   // src/mat.rs:85-97
template <typename T> struct alignas(16) DTexture { uint32_t kind, a, b, c; T color[3]; T scale; }; // check: a = odd, b = even; noise: a = perlin; image: a = byte offset, b = width, c = height
template <typename T> struct DMedium { T neg_inv_density; uint32_t mat, pad; };               // -(1.0/density), src/medium.rs:42
struct DLight { uint32_t kind, index; };
template <typename T> struct DPerlin { T rd_vec[256 * 3]; uint8_t perm_x[256], perm_y[256], perm_z[256]; };   // src/perlin.rs:59-65
template <typename T> struct DCamera {                                                        // src/camera.rs:6-16
    T origin[3], lower_left_corner[3], horizontal[3], vertical[3], cu[3], cv[3];
    T lens_radius, time0, time1;
};

template <typename T> struct KParams {
    // scene
    const DObject* objects; uint32_t n_objects;
    const DOp<T>* ops;
    const DRect<T>* rects;
    const DSphere<T>* spheres;
    const DMSphere<T>* mspheres;
    const DTri<T>* tris;
    const DBvhNode<T>* bvh;
    uint32_t n_bvh;
    const DMaterial<T>* materials;
    const DTexture<T>* textures;
    const DMedium<T>* media;
    const DLight* lights; uint32_t n_lights;
    const DPerlin<T>* perlins;
    const DPbr<T>* pbr;
    const uint8_t* image_bytes;
    uint32_t stack_depth;          // per-lane BVH stack entries staged in LDS: 0 in the reference's traversal order (skip links), the tree depth for near-first
    uint32_t queue_entries;        // camera paths each wave's LDS queue holds (16, 32 or 64; 80 B each)
    uint32_t n_cached;             // BVH nodes [0, n_cached) are copied into LDS by every workgroup at launch (depth order: the top levels)
    uint32_t bvh_tame;             // every BVH box is finite, below 1e300 in magnitude and has min <= max: rays that are tame too may
                                   // take the NaN-free form of AABB::hit (rt_kernel.hip: box_inside_tame) — same answers
    // frame
    DCamera<T> cam;
    T background[3];
    uint32_t W, H, spp, max_depth;
    uint64_t seed;
    uint32_t flags;
    // work decomposition: this launch owns tiles t = rank + q*world, q in [0, n_local_tiles)
    uint32_t tile_px, rank, world, n_local_tiles;
    // dequeue unit: chunk c covers samples [sub*chunk_spp, ..) of chunk_px consecutive local pixels starting at
    // (c / chunks_per_px) * chunk_px, sub = c % chunks_per_px.  Either chunks_per_px == 1 (whole pixels) or chunk_px == 1.
    // With chunk_px == 1 the first n_coarse_px pixels are whole-pixel chunks and only the rest are split (guided
    // self-scheduling: fine chunks where they shorten the end-of-frame tail, coarse ones elsewhere).
    uint32_t chunk_px, chunks_per_px, chunk_spp, n_coarse_px, n_chunks;
    uint32_t* queue;               // zeroed before launch
    double* out;                   // n_local_tiles * tile_px * 3 (always f64: per-pixel sums)
    double* samples_out;           // optional: local_px * spp * 3
    unsigned long long* stats;     // [0] non-finite samples, [1] bounce iterations, [2] lane-iterations active, [9] traversal steps, [10] lanes stepping, [11] accumulator flushes (3 f64 atomics each)
    // BVH scenes (persistent traversal): a traversal pass starts once trav_hi lanes are inside a BVH and runs until fewer
    // than trav_lo are still walking
    uint32_t trav_hi, trav_lo, trav_leaf;      // trav_leaf: a leaf step runs once trav_leaf/64 of the walking lanes hold a pending leaf
    // debugging aid (-DRT_TRACE_PATH builds, rt_debug_trace_path): the path (trace_px, trace_s) writes 16 doubles per level to trace_out
    double* trace_out; uint32_t trace_px, trace_s;
    // (new fields go here, at the end: the list-scene kernels are sensitive to the kernel-argument layout of the fields above)
    // filtered walk (f64 kernels, reference traversal order): the f32 companions of bvh[] — the first n_cached of THEM are what the
    // workgroups stage in LDS then — and filter_m >= every |coordinate| of theirs (>= 1; 0 = no filter: some box is non-finite or huge)
    const DFNode* bvh_f; float filter_m;
    // Cube fast path (rt_kernel.hip: cube_hit): >= every |coordinate| of every rect; 0 = off (a Cube with min > max on some axis, or a
    // non-finite coordinate: the six exact tests then)
    float rect_m;
    // Room form (list scenes; rt_flatten.cpp form_room): objects[n_objects, n_objects + n_objects_alt) is the world list AS THE REFERENCE HAS
    // IT, behind the list with the room.  A wave in which some lane's ray has a zero or non-finite component — the only rays for which a
    // plane distance can be NaN, and a NaN hit makes HittableList::hit depend on the order of the items — searches that list instead
    // (rt_kernel.hip: world_hit_list).  0: none.
    uint32_t n_objects_alt;
};

} // namespace rt

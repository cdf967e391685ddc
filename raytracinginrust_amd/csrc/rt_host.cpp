// csrc/rt_host.cpp — C-ABI of librt_amd.so (include/rt_amd.h): scene builder handles, device upload,
// kernel launch, format_color / PPM emitter.  There is NO CPU rendering path in this library: without a
// HIP device rt_render* fail with an error.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <exception>
#include <string>
#include <vector>
#include "../../include/rt_amd.h"
#include "rt_scene.h"
#include "rt_launch.h"

using namespace rt;

namespace rt {
bool decode_jpeg_rgb8(const uint8_t* data, size_t size, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height, std::string& err);
bool parse_obj(const char* data, size_t size, const double offset[3], double scale, std::vector<double>& positions, std::vector<uint32_t>& indices, std::string& err);
}

struct rt_rng { Rng r; };

static thread_local std::string g_err;
static int set_err(const std::string& m) { g_err = m; return -1; }
namespace rt { int set_error(const std::string& m) { return set_err(m); } }
static int scene_err(rt_scene* sc, const std::string& m) { sc->s.error = m; g_err = m; return -1; }

#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { g_err = std::string(#call) + ": " + hipGetErrorString(e_); return -1; } } while (0)

static void free_dev(void*& p) { if (p) { (void)hipFree(p); p = nullptr; } }
template <typename T> static void free_device_scene(DeviceScene<T>& d) {
    free_dev(d.objects); free_dev(d.ops); free_dev(d.rects); free_dev(d.spheres); free_dev(d.mspheres); free_dev(d.tris);
    free_dev(d.bvh); free_dev(d.materials); free_dev(d.textures); free_dev(d.media); free_dev(d.lights); free_dev(d.perlins); free_dev(d.image); free_dev(d.pbr); free_dev(d.bvh_f);
    d.valid = false;
}

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
int rt_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

// ---------------------------------------------------------------- rng
rt_rng* rt_rng_create(uint64_t seed, uint32_t stream) { rt_rng* r = new rt_rng(); r->r = rng_for_stream(seed, stream); return r; }
void rt_rng_destroy(rt_rng* r) { delete r; }
double rt_rng_f64(rt_rng* r) { return rng_u01(r->r, 0.0); }
double rt_rng_range(rt_rng* r, double a, double b) { return rng_range(r->r, a, b); }
int rt_rng_bool(rt_rng* r) { return rng_bool(r->r) ? 1 : 0; }
uint32_t rt_rng_index(rt_rng* r, uint32_t n) { return rng_index(r->r, n); }
uint32_t rt_rng_u32(rt_rng* r) { return rng_u32(r->r); }
void rt_rng_path(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t out[4]) {
    Rng g = rng_for_path(seed, pixel, sample);
    out[0] = g.s0; out[1] = g.s1; out[2] = g.s2; out[3] = g.s3;
}

// ---------------------------------------------------------------- scene
rt_scene* rt_scene_create(void) { return new rt_scene(); }

void rt_scene_destroy(rt_scene* sc) {
    if (!sc) return;
    int cur = 0; (void)hipGetDevice(&cur);
    rt::multi_release(sc->s);
    if (sc->s.d_trace) { (void)hipSetDevice(sc->s.trace_device); (void)hipFree(sc->s.d_trace); sc->s.d_trace = nullptr; (void)hipSetDevice(cur); }
    if (sc->s.d_calib) { (void)hipSetDevice(sc->s.calib_device); (void)hipDeviceSynchronize(); (void)hipFree(sc->s.d_calib); sc->s.d_calib = nullptr; (void)hipSetDevice(cur); }
    for (Scene::DeviceCtx* c : sc->s.ctxs) {
        (void)hipSetDevice(c->device);
        free_device_scene(c->dev64);
        free_device_scene(c->dev32);
        for (Scene::LaunchSlot& l : c->slots) {
            if (l.recorded) (void)hipEventSynchronize((hipEvent_t)l.ev_stop);
            free_dev(l.d_queue); free_dev(l.d_stats);
            if (l.ev_start) (void)hipEventDestroy((hipEvent_t)l.ev_start);
            if (l.ev_stop) (void)hipEventDestroy((hipEvent_t)l.ev_stop);
        }
        free_dev(c->d_tiles); free_dev(c->d_gather); free_dev(c->d_frame[0]); free_dev(c->d_frame[1]);
        if (c->stream) (void)hipStreamDestroy((hipStream_t)c->stream);
        delete c;
    }
    if (!sc->s.ctxs.empty()) (void)hipSetDevice(cur);
    delete sc;
}
const char* rt_scene_error(rt_scene* sc) { return sc->s.error.c_str(); }

static void touch(rt_scene* sc) {
    sc->s.invalidate();
    if (sc->s.ctxs.empty()) return;
    int cur = 0; (void)hipGetDevice(&cur);
    for (Scene::DeviceCtx* c : sc->s.ctxs) {
        (void)hipSetDevice(c->device); free_device_scene(c->dev64); free_device_scene(c->dev32);
    }
    (void)hipSetDevice(cur);
}
int rt_scene_set_traversal_schedule(rt_scene* sc, uint32_t start_at, uint32_t stop_below, uint32_t leaf_share64) {
    if (!sc) return set_err("null argument");
    if (start_at < 1u || start_at > 64u || stop_below < 1u || stop_below > start_at || leaf_share64 < 1u || leaf_share64 > 64u)
        return set_err("traversal schedule: need 1 <= stop_below <= start_at <= 64 and 1 <= leaf_share64 <= 64");
    sc->s.trav_hi = start_at; sc->s.trav_lo = stop_below; sc->s.trav_leaf = leaf_share64;
    return 0;
}
int rt_scene_set_bvh_builder(rt_scene* sc, int mode) {
    if (!sc) return set_err("null argument");
    if (mode != RT_BVH_MEDIAN && mode != RT_BVH_SAH) return set_err("unknown BVH builder");
    touch(sc);
    sc->s.bvh_builder = mode;
    return 0;
}
static bool tex_ok(rt_scene* sc, int t) { return t >= 0 && t < (int)sc->s.textures.size(); }
static bool mat_ok(rt_scene* sc, int m) { return m >= 0 && m < (int)sc->s.materials.size(); }
static bool node_ok(rt_scene* sc, int h) { return h >= 0 && h < (int)sc->s.nodes.size(); }
static int add_node(rt_scene* sc, const HNode& n) { touch(sc); sc->s.nodes.push_back(n); return (int)sc->s.nodes.size() - 1; }
static int add_tex(rt_scene* sc, const DTexture<double>& t) { touch(sc); sc->s.textures.push_back(t); return (int)sc->s.textures.size() - 1; }
static int add_mat(rt_scene* sc, const DMaterial<double>& m) { touch(sc); sc->s.materials.push_back(m); return (int)sc->s.materials.size() - 1; }

int rt_texture_constant(rt_scene* sc, const double rgb[3]) {
    DTexture<double> t{}; t.kind = T_CONSTANT; t.color[0] = rgb[0]; t.color[1] = rgb[1]; t.color[2] = rgb[2];
    return add_tex(sc, t);
}
int rt_texture_check(rt_scene* sc, int odd, int even) {
    if (!tex_ok(sc, odd) || !tex_ok(sc, even)) return scene_err(sc, "CheckTexture: bad texture handle");
    DTexture<double> t{}; t.kind = T_CHECK; t.a = (uint32_t)odd; t.b = (uint32_t)even;
    return add_tex(sc, t);
}
// NoiseTexture::new -> Perlin::new (src/perlin.rs:67-75): generate_vector (:13-19), then generate_perm x3 (:21-37)
int rt_texture_noise(rt_scene* sc, double scale, rt_rng* rng) {
    if (!rng) return scene_err(sc, "NoiseTexture: rng is null");
    HPerlin p;
    for (int i = 0; i < 256; i++) {
        for (;;) {                                               // Vec3::random_in_unit_sphere, src/vec.rs:78-85
            double a = rng_range(rng->r, -1.0, 1.0), b = rng_range(rng->r, -1.0, 1.0), c = rng_range(rng->r, -1.0, 1.0);
            if (std::sqrt(a * a + b * b + c * c) < 1.0) { p.rd_vec[i * 3] = a; p.rd_vec[i * 3 + 1] = b; p.rd_vec[i * 3 + 2] = c; break; }
        }
    }
    for (int k = 0; k < 3; k++) {
        for (int i = 0; i < 256; i++) p.perm[k][i] = (uint8_t)i;
        for (int i = 255; i >= 0; i--) {
            uint32_t target = rng_index(rng->r, (uint32_t)i + 1u);   // gen_range(0..=i)
            uint8_t tmp = p.perm[k][i]; p.perm[k][i] = p.perm[k][target]; p.perm[k][target] = tmp;
        }
    }
    sc->s.perlins.push_back(p);
    DTexture<double> t{}; t.kind = T_NOISE; t.a = (uint32_t)sc->s.perlins.size() - 1; t.scale = scale;
    return add_tex(sc, t);
}
int rt_texture_image(rt_scene* sc, const uint8_t* rgb8, uint32_t width, uint32_t height) {
    if (!rgb8 || width == 0 || height == 0) return scene_err(sc, "ImageTexture: empty image");
    DTexture<double> t{}; t.kind = T_IMAGE; t.a = (uint32_t)sc->s.image_bytes.size(); t.b = width; t.c = height;
    sc->s.image_bytes.insert(sc->s.image_bytes.end(), rgb8, rgb8 + (size_t)3 * width * height);
    return add_tex(sc, t);
}

int rt_material_lambertian(rt_scene* sc, int tex) {
    if (!tex_ok(sc, tex)) return scene_err(sc, "Lambertian: bad texture handle");
    DMaterial<double> m{}; m.kind = M_LAMBERTIAN; m.tex = (uint32_t)tex; return add_mat(sc, m);
}
int rt_material_metal(rt_scene* sc, const double albedo[3], double fuzz) {
    DMaterial<double> m{}; m.kind = M_METAL; m.albedo[0] = albedo[0]; m.albedo[1] = albedo[1]; m.albedo[2] = albedo[2]; m.param = fuzz; return add_mat(sc, m);
}
int rt_material_dielectric(rt_scene* sc, double ir) { DMaterial<double> m{}; m.kind = M_DIELECTRIC; m.param = ir; return add_mat(sc, m); }
int rt_material_diffuse_light(rt_scene* sc, int tex) {
    if (!tex_ok(sc, tex)) return scene_err(sc, "DiffuseLight: bad texture handle");
    DMaterial<double> m{}; m.kind = M_DIFFUSE_LIGHT; m.tex = (uint32_t)tex; return add_mat(sc, m);
}
int rt_material_pbr(rt_scene* sc, int tex, const double p[10]) {
    if (!tex_ok(sc, tex)) return scene_err(sc, "PBR: bad texture handle");
    if (!p) return scene_err(sc, "PBR: null parameters");
    DPbr<double> d{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9]};
    sc->s.pbr.push_back(d);
    DMaterial<double> m{}; m.kind = M_PBR; m.tex = (uint32_t)tex; m.albedo[0] = (double)(sc->s.pbr.size() - 1); return add_mat(sc, m);
}
int rt_material_isotropic(rt_scene* sc, int tex) {
    if (!tex_ok(sc, tex)) return scene_err(sc, "Isotropic: bad texture handle");
    DMaterial<double> m{}; m.kind = M_ISOTROPIC; m.tex = (uint32_t)tex; return add_mat(sc, m);
}

int rt_sphere(rt_scene* sc, const double c[3], double r, int mat) {
    if (!mat_ok(sc, mat)) return scene_err(sc, "Sphere: bad material handle");
    HNode n; n.kind = HNode::SPHERE; n.v[0] = c[0]; n.v[1] = c[1]; n.v[2] = c[2]; n.v[3] = r; n.mat = mat; return add_node(sc, n);
}
int rt_moving_sphere(rt_scene* sc, const double c0[3], const double c1[3], double t0, double t1, double r, int mat) {
    if (!mat_ok(sc, mat)) return scene_err(sc, "MovingSphere: bad material handle");
    HNode n; n.kind = HNode::MSPHERE;
    for (int k = 0; k < 3; k++) { n.v[k] = c0[k]; n.v[3 + k] = c1[k]; }
    n.v[6] = t0; n.v[7] = t1; n.v[8] = r; n.mat = mat; return add_node(sc, n);
}
int rt_aarect(rt_scene* sc, int plane, double a0, double a1, double b0, double b1, double k, int mat) {
    if (!mat_ok(sc, mat)) return scene_err(sc, "AARect: bad material handle");
    if (plane < 0 || plane > 2) return scene_err(sc, "AARect: bad plane");
    HNode n; n.kind = HNode::RECT; n.v[0] = a0; n.v[1] = a1; n.v[2] = b0; n.v[3] = b1; n.v[4] = k; n.plane_or_axis = plane; n.mat = mat; return add_node(sc, n);
}
int rt_cube(rt_scene* sc, const double mn[3], const double mx[3], int mat) {
    if (!mat_ok(sc, mat)) return scene_err(sc, "Cube: bad material handle");
    HNode n; n.kind = HNode::CUBE; for (int k = 0; k < 3; k++) { n.v[k] = mn[k]; n.v[3 + k] = mx[k]; } n.mat = mat; return add_node(sc, n);
}
int rt_triangle(rt_scene* sc, const double v[9], int mat) {
    if (!mat_ok(sc, mat)) return scene_err(sc, "Triangle: bad material handle");
    HNode n; n.kind = HNode::TRI; for (int k = 0; k < 9; k++) n.v[k] = v[k]; n.mat = mat; return add_node(sc, n);
}
int rt_list_create(rt_scene* sc) { HNode n; n.kind = HNode::LIST; return add_node(sc, n); }
int rt_list_push(rt_scene* sc, int list, int h) {
    if (!node_ok(sc, list) || sc->s.nodes[list].kind != HNode::LIST) return scene_err(sc, "HittableList::push: not a list");
    if (!node_ok(sc, h)) return scene_err(sc, "HittableList::push: bad hittable handle");
    touch(sc); sc->s.nodes[list].items.push_back(h); return 0;
}
int rt_mesh(rt_scene* sc, const double* positions, uint32_t n_positions, const uint32_t* indices, uint32_t n_indices, int mat) {
    if (!mat_ok(sc, mat)) return scene_err(sc, "Mesh: bad material handle");
    for (uint32_t i = 0; i < n_indices; i++) if (indices[i] >= n_positions) return scene_err(sc, "Mesh: index out of range");
    int list = rt_list_create(sc);
    for (uint32_t i = 0; i < n_indices / 3; i++) {                    // src/mesh.rs:19-26
        double v[9];
        for (int c = 0; c < 3; c++) for (int k = 0; k < 3; k++) v[c * 3 + k] = positions[(size_t)indices[i * 3 + c] * 3 + k];
        int t = rt_triangle(sc, v, mat);
        sc->s.nodes[list].items.push_back(t);
    }
    return list;
}
// tobj::load_obj's part of Mesh::load_obj (src/mesh.rs:40-54): csrc/rt_obj.cpp
int rt_parse_obj(const char* data, size_t size, const double offset[3], double scale, double** positions_out, uint32_t* n_positions_out,
                 uint32_t** indices_out, uint32_t* n_indices_out) {
    if (!data || !offset || !positions_out || !n_positions_out || !indices_out || !n_indices_out) return set_err("null argument");
    try {
        std::vector<double> pos; std::vector<uint32_t> idx; std::string err;
        if (!parse_obj(data, size, offset, scale, pos, idx, err)) return set_err("Failed to load obj file: " + err);
        double* p = (double*)std::malloc((pos.size() ? pos.size() : 1) * sizeof(double));
        uint32_t* ix = (uint32_t*)std::malloc((idx.size() ? idx.size() : 1) * sizeof(uint32_t));
        if (!p || !ix) { std::free(p); std::free(ix); return set_err("out of memory"); }
        if (!pos.empty()) std::memcpy(p, pos.data(), pos.size() * sizeof(double));
        if (!idx.empty()) std::memcpy(ix, idx.data(), idx.size() * sizeof(uint32_t));
        *positions_out = p; *n_positions_out = (uint32_t)(pos.size() / 3); *indices_out = ix; *n_indices_out = (uint32_t)idx.size();
        return 0;
    } catch (const std::exception& e) { return set_err(std::string("Failed to load obj file: ") + e.what()); }
}
// Mesh::load_obj(path, offset, scale, material), src/mesh.rs:33-61 -> the `tris` HittableList (like rt_mesh)
int rt_mesh_load_obj(rt_scene* sc, const char* path, const double offset[3], double scale, int mat) {
    if (!sc || !path || !offset) return set_err("null argument");
    if (!mat_ok(sc, mat)) return scene_err(sc, "Mesh: bad material handle");
    try {
        FILE* f = std::fopen(path, "rb");
        if (!f) return scene_err(sc, std::string("Failed to load obj file: cannot open ") + path);
        std::string data; char buf[1 << 16]; size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) data.append(buf, n);
        std::fclose(f);
        std::vector<double> pos; std::vector<uint32_t> idx; std::string err;
        if (!parse_obj(data.data(), data.size(), offset, scale, pos, idx, err)) return scene_err(sc, "Failed to load obj file: " + err);
        return rt_mesh(sc, pos.data(), (uint32_t)(pos.size() / 3), idx.data(), (uint32_t)idx.size(), mat);
    } catch (const std::exception& e) { return scene_err(sc, std::string("Failed to load obj file: ") + e.what()); }
}
int rt_flip_normal(rt_scene* sc, int h) {
    if (!node_ok(sc, h)) return scene_err(sc, "FlipNormal: bad hittable handle");
    HNode n; n.kind = HNode::FLIP; n.child = h; return add_node(sc, n);
}
int rt_translate(rt_scene* sc, int h, const double off[3]) {
    if (!node_ok(sc, h)) return scene_err(sc, "Translate: bad hittable handle");
    HNode n; n.kind = HNode::TRANSLATE; n.child = h; n.v[0] = off[0]; n.v[1] = off[1]; n.v[2] = off[2]; return add_node(sc, n);
}
int rt_rotate(rt_scene* sc, int axis, int h, double angle) {
    if (!node_ok(sc, h)) return scene_err(sc, "Rotate: bad hittable handle");
    if (axis < 0 || axis > 2) return scene_err(sc, "Rotate: bad axis");
    HNode n; n.kind = HNode::ROTATE; n.child = h; n.plane_or_axis = axis; n.v[0] = angle; return add_node(sc, n);
}
int rt_constant_medium(rt_scene* sc, int boundary, double density, int tex) {
    if (!node_ok(sc, boundary)) return scene_err(sc, "ConstantMedium: bad boundary handle");
    if (!tex_ok(sc, tex)) return scene_err(sc, "ConstantMedium: bad texture handle");
    HNode n; n.kind = HNode::MEDIUM; n.child = boundary; n.v[0] = density; n.mat = tex; return add_node(sc, n);
}
int rt_bvh(rt_scene* sc, const int* hs, uint32_t nh, double, double) {
    if (nh == 0) return scene_err(sc, "no object in the scene");                 // src/bvh.rs:55 panics
    HNode n; n.kind = HNode::BVH;
    for (uint32_t i = 0; i < nh; i++) { if (!node_ok(sc, hs[i])) return scene_err(sc, "BVH: bad hittable handle"); n.items.push_back(hs[i]); }
    return add_node(sc, n);
}
int rt_bvh_of_list(rt_scene* sc, int list, double t0, double t1) {
    if (!node_ok(sc, list) || sc->s.nodes[list].kind != HNode::LIST) return scene_err(sc, "BVH: not a list");
    std::vector<int> items = sc->s.nodes[list].items;
    return rt_bvh(sc, items.data(), (uint32_t)items.size(), t0, t1);
}
int rt_scene_set_world(rt_scene* sc, int h) {
    if (!node_ok(sc, h)) return scene_err(sc, "set_world: bad hittable handle");
    touch(sc); sc->s.world = h; return 0;
}
int rt_lights_push(rt_scene* sc, int h) {
    if (!node_ok(sc, h)) return scene_err(sc, "lights.push: bad hittable handle");
    touch(sc); sc->s.lights.push_back(h); return 0;
}

// ---------------------------------------------------------------- host-side boundary pieces
void rt_camera_fields(const rt_camera* c, double out21[21]) {
    rt_camera_args a; std::memcpy(&a, c, sizeof(a));
    DCamera<double> d; camera_new(a, d);
    const double* vs[6] = {d.origin, d.lower_left_corner, d.horizontal, d.vertical, d.cu, d.cv};
    for (int i = 0; i < 6; i++) for (int k = 0; k < 3; k++) out21[i * 3 + k] = vs[i][k];
    out21[18] = d.lens_radius; out21[19] = d.time0; out21[20] = d.time1;
}
// Vec3::format_color, src/vec.rs:125-131: (256.0 * (c / spp).sqrt().clamp(0.0, 0.999)) as u64
void rt_format_color(const double rgb_sum[3], uint64_t spp, uint64_t out3[3]) {
    for (int k = 0; k < 3; k++) {
        double x = std::sqrt(rgb_sum[k] / (double)spp);
        if (x < 0.0) x = 0.0; else if (x > 0.999) x = 0.999;       // f64::clamp: NaN stays NaN
        double y = 256.0 * x;
        out3[k] = (y > 0.0) ? (y >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)y) : 0;   // `as u64` saturates, NaN -> 0
    }
}
int rt_write_ppm(const char* path, const double* rgb_sum, uint32_t W, uint32_t H, uint64_t spp) {
    FILE* f = (!path || std::strcmp(path, "-") == 0) ? stdout : std::fopen(path, "w");
    if (!f) return set_err(std::string("cannot open ") + path);
    std::fprintf(f, "P3\n%u %u\n255\n", W, H);                      // src/main.rs:767-769
    for (size_t p = 0; p < (size_t)W * H; p++) {
        uint64_t c[3]; rt_format_color(rgb_sum + p * 3, spp, c);
        std::fprintf(f, "%llu %llu %llu\n", (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[2]);   // src/main.rs:832
    }
    if (f != stdout) std::fclose(f); else std::fflush(f);
    return 0;
}

// image::open(path).to_rgb8() for the reference's asset class (src/main.rs:248,491): baseline JPEG -> RGB8 (csrc/rt_jpeg.cpp)
uint8_t* rt_decode_jpeg_rgb8(const uint8_t* data, size_t size, uint32_t* width, uint32_t* height) {
    if (!data || !width || !height) { set_err("null argument"); return nullptr; }
    try {                                                               // no C++ exception may cross the C boundary
        std::vector<uint8_t> rgb; std::string err; uint32_t w = 0, h = 0;
        if (!decode_jpeg_rgb8(data, size, rgb, w, h, err)) { set_err("image not found / not decodable: " + err); return nullptr; }
        uint8_t* out = (uint8_t*)std::malloc(rgb.size());
        if (!out) { set_err("out of memory"); return nullptr; }
        std::memcpy(out, rgb.data(), rgb.size());
        *width = w; *height = h;
        return out;
    } catch (const std::exception& e) { set_err(std::string("image not decodable: ") + e.what()); return nullptr; }
}
void rt_free(void* p) { std::free(p); }

int rt_scene_flatten(rt_scene* sc, uint32_t counts[12]) {
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    const HostFlat& f = sc->s.flat;
    if (counts) {
        uint32_t c[12] = {(uint32_t)f.objects.size(), (uint32_t)f.ops.size(), (uint32_t)f.rects.size(), (uint32_t)f.spheres.size(),
                          (uint32_t)f.mspheres.size(), (uint32_t)f.tris.size(), (uint32_t)f.bvh.size(), (uint32_t)f.materials.size(),
                          (uint32_t)f.textures.size(), (uint32_t)f.lights.size(), (uint32_t)f.media.size(), (uint32_t)sc->s.perlins.size()};
        std::memcpy(counts, c, sizeof(c));
    }
    return 0;
}

// Test aid (host only, no GPU): tune the filter tree of a one-BVH world for a view (rt_flatten.cpp tune_filter_tree) without touching any
// device copy; 1 if the tree was rebuilt, 0 if the scene is not of that kind, -1 on error.  rt_debug_filter_nodes then shows the result.
int rt_debug_tune_filter(rt_scene* sc, const rt_camera* cam) {
    if (!sc || !cam) return set_err("null argument");
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    rt_camera_args ca; std::memcpy(&ca, cam, sizeof(ca));
    DCamera<double> dcam; camera_new(ca, dcam);
    return tune_filter_tree(sc->s, dcam) ? 1 : 0;
}

// Test aid (host only): the flattened object table, out[8*i ..] = object i's {geom_kind, geom_first, geom_count, first_op, n_ops, medium,
// is_cube, nest} (rt_ir.h DObject): the world's top-level objects first, then the sub-objects of G_OBJ leaves.  Returns the number of
// objects (all of them), *n_top_out the number of top-level ones; -1 on error.
int rt_debug_objects(rt_scene* sc, uint32_t* out, uint32_t max_objects, uint32_t* n_top_out) {
    if (!sc) return set_err("null argument");
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    const HostFlat& f = sc->s.flat;
    if (n_top_out) *n_top_out = f.n_top;
    for (size_t i = 0; i < f.objects.size() && i < max_objects && out; i++) {
        const DObject& o = f.objects[i];
        const uint32_t w[8] = {o.geom_kind, o.geom_first, o.geom_count, o.first_op, o.n_ops, (uint32_t)o.medium, o.is_cube, o.nest};
        std::memcpy(out + 8 * i, w, sizeof(w));
    }
    return (int)f.objects.size();
}

// Test aid (host only): the flattened BVH's link words.  out[4*i ..] = node i's {a, b, c, skip} (rt_ir.h DBvhNode); roots_out receives
// the root node of every BVH object in list order (at most max_roots).  Returns the number of nodes, or -1.
int rt_debug_bvh_links(rt_scene* sc, uint32_t* out, uint32_t max_nodes, uint32_t* roots_out, uint32_t max_roots, uint32_t* n_roots_out) {
    if (!sc) return -1;
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    const HostFlat& f = sc->s.flat;
    if (out) for (size_t i = 0; i < f.bvh.size() && i < max_nodes; i++) { out[4 * i] = f.bvh[i].a; out[4 * i + 1] = f.bvh[i].b; out[4 * i + 2] = f.bvh[i].c; out[4 * i + 3] = f.bvh[i].skip; }
    uint32_t n_roots = 0;
    for (const DObject& ob : f.objects) if (ob.geom_kind == G_BVH) { if (roots_out && n_roots < max_roots) roots_out[n_roots] = ob.geom_first; n_roots++; }
    if (n_roots_out) *n_roots_out = n_roots;
    return (int)f.bvh.size();
}

// Test aid (host only): the filter tree the f64 kernels' box steps walk (rt_ir.h DFNode, rt_flatten.cpp make_filter_nodes), node for
// node beside rt_debug_bvh_links' f64 tree: boxes6_out[6*i ..] = {min.x, max.x, min.y, max.y, min.z, max.z} rounded outward to f32,
// links2_out[2*i ..] = {skip, info} (info: first child, or own id | 0x40000000 for a leaf), f64_boxes_out[6*i ..] = the exact box
// {min[3], max[3]}.  *filter_m_out = KParams::filter_m.  Returns the number of nodes, or -1.
int rt_debug_filter_nodes(rt_scene* sc, float* boxes6_out, uint32_t* links2_out, double* f64_boxes_out, uint32_t max_nodes, float* filter_m_out) {
    if (!sc) return -1;
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    const HostFlat& f = sc->s.flat;
    for (size_t i = 0; i < f.bvh_f.size() && i < max_nodes; i++) {
        if (boxes6_out) for (int k = 0; k < 6; k++) boxes6_out[6 * i + k] = f.bvh_f[i].b[k];
        if (links2_out) { links2_out[2 * i] = f.bvh_f[i].skip; links2_out[2 * i + 1] = f.bvh_f[i].info; }
        if (f64_boxes_out) for (int k = 0; k < 3; k++) { f64_boxes_out[6 * i + k] = f.bvh[i].mn[k]; f64_boxes_out[6 * i + 3 + k] = f.bvh[i].mx[k]; }
    }
    if (filter_m_out) *filter_m_out = f.filter_m;
    return (int)f.bvh_f.size();
}

uint32_t rt_local_tiles(uint32_t W, uint32_t H, uint32_t tile_px, uint32_t rank, uint32_t world) {
    (void)rank;
    if (tile_px == 0 || world == 0) return 0;
    uint64_t n_px = (uint64_t)W * H;
    uint64_t n_tiles = (n_px + tile_px - 1) / tile_px;
    return (uint32_t)((n_tiles + world - 1) / world);       // padded: identical on every rank
}

} // extern "C"

// ---------------------------------------------------------------- device upload
namespace {

template <typename T> void cv(const DRect<double>& a, DRect<T>& b) { b.a0 = (T)a.a0; b.a1 = (T)a.a1; b.b0 = (T)a.b0; b.b1 = (T)a.b1; b.k = (T)a.k; b.plane = a.plane; b.mat = a.mat; }
template <typename T> void cv(const DSphere<double>& a, DSphere<T>& b) { for (int k = 0; k < 3; k++) b.c[k] = (T)a.c[k]; b.r = (T)a.r; b.mat = a.mat; b.pad = 0; }
template <typename T> void cv(const DMSphere<double>& a, DMSphere<T>& b) { for (int k = 0; k < 3; k++) { b.c0[k] = (T)a.c0[k]; b.c1[k] = (T)a.c1[k]; } b.t0 = (T)a.t0; b.t1 = (T)a.t1; b.r = (T)a.r; b.mat = a.mat; b.pad = 0; }
template <typename T> void cv(const DTri<double>& a, DTri<T>& b) { for (int k = 0; k < 3; k++) { b.v0[k] = (T)a.v0[k]; b.e1[k] = (T)a.e1[k]; b.e2[k] = (T)a.e2[k]; } b.mat = a.mat; b.pad = 0; }
template <typename T> void cv(const DOp<double>& a, DOp<T>& b) { b.kind = a.kind; b.axis = a.axis; b.x = (T)a.x; b.y = (T)a.y; b.z = (T)a.z; }
template <typename T> void cv(const DBvhNode<double>& a, DBvhNode<T>& b) { for (int k = 0; k < 3; k++) { b.mn[k] = (T)a.mn[k]; b.mx[k] = (T)a.mx[k]; } b.a = a.a; b.b = a.b; b.c = a.c; b.skip = a.skip; }
template <typename T> void cv(const DMaterial<double>& a, DMaterial<T>& b) { b.kind = a.kind; b.tex = a.tex; for (int k = 0; k < 3; k++) b.albedo[k] = (T)a.albedo[k]; b.param = (T)a.param; }
template <typename T> void cv(const DTexture<double>& a, DTexture<T>& b) { b.kind = a.kind; b.a = a.a; b.b = a.b; b.c = a.c; for (int k = 0; k < 3; k++) b.color[k] = (T)a.color[k]; b.scale = (T)a.scale; }
template <typename T> void cv(const DPbr<double>& a, DPbr<T>& b) {
    b.metallic = (T)a.metallic; b.subsurface = (T)a.subsurface; b.specular = (T)a.specular; b.roughness = (T)a.roughness; b.specular_tint = (T)a.specular_tint;
    b.anisotropic = (T)a.anisotropic; b.sheen = (T)a.sheen; b.sheen_tint = (T)a.sheen_tint; b.clearcoat = (T)a.clearcoat; b.clearcoat_gloss = (T)a.clearcoat_gloss;
}
template <typename T> void cv(const DMedium<double>& a, DMedium<T>& b) { b.neg_inv_density = (T)a.neg_inv_density; b.mat = a.mat; b.pad = 0; }

template <typename DT, typename ST> int upload_vec(const std::vector<ST>& src, void*& dst) {
    dst = nullptr;
    size_t n = src.size();
    std::vector<DT> tmp(n + 1);            // one zeroed padding record: the kernel prefetches record i+1
    std::memset((void*)tmp.data(), 0, tmp.size() * sizeof(DT));
    for (size_t i = 0; i < n; i++) cv(src[i], tmp[i]);
    HIP_OK(hipMalloc(&dst, tmp.size() * sizeof(DT)));
    HIP_OK(hipMemcpy(dst, tmp.data(), tmp.size() * sizeof(DT), hipMemcpyHostToDevice));
    return 0;
}
template <typename PT> int upload_raw(const std::vector<PT>& src, void*& dst) {
    dst = nullptr;
    size_t bytes = (src.size() ? src.size() : 1) * sizeof(PT);
    HIP_OK(hipMalloc(&dst, bytes));
    if (!src.empty()) HIP_OK(hipMemcpy(dst, src.data(), src.size() * sizeof(PT), hipMemcpyHostToDevice));
    return 0;
}

template <typename T> int ensure_uploaded(Scene& s, DeviceScene<T>& d) {
    if (d.valid) return 0;
    const HostFlat& f = s.flat;
    if (upload_raw(f.objects, d.objects)) return -1;
    if (upload_vec<DOp<T>>(f.ops, d.ops)) return -1;
    if (upload_vec<DRect<T>>(f.rects, d.rects)) return -1;
    if (upload_vec<DSphere<T>>(f.spheres, d.spheres)) return -1;
    if (upload_vec<DMSphere<T>>(f.mspheres, d.mspheres)) return -1;
    if (upload_vec<DTri<T>>(f.tris, d.tris)) return -1;
    if (upload_vec<DBvhNode<T>>(f.bvh, d.bvh)) return -1;
    { std::vector<DFNode> fn(f.bvh_f); fn.push_back(DFNode{}); if (upload_raw(fn, d.bvh_f)) return -1; }      // (one padding record, as upload_vec)
    if (upload_vec<DMaterial<T>>(f.materials, d.materials)) return -1;
    if (upload_vec<DTexture<T>>(f.textures, d.textures)) return -1;
    if (upload_vec<DMedium<T>>(f.media, d.media)) return -1;
    if (upload_raw(f.lights, d.lights)) return -1;
    std::vector<DPerlin<T>> pl(s.perlins.size());
    for (size_t i = 0; i < pl.size(); i++) {
        for (int k = 0; k < 768; k++) pl[i].rd_vec[k] = (T)s.perlins[i].rd_vec[k];
        std::memcpy(pl[i].perm_x, s.perlins[i].perm[0], 256);
        std::memcpy(pl[i].perm_y, s.perlins[i].perm[1], 256);
        std::memcpy(pl[i].perm_z, s.perlins[i].perm[2], 256);
    }
    if (upload_raw(pl, d.perlins)) return -1;
    if (upload_vec<DPbr<T>>(s.pbr, d.pbr)) return -1;
    if (upload_raw(s.image_bytes, d.image)) return -1;
    d.valid = true;
    return 0;
}

template <typename T> DeviceScene<T>& dev_of(Scene::DeviceCtx& c);
template <> DeviceScene<double>& dev_of<double>(Scene::DeviceCtx& c) { return c.dev64; }
template <> DeviceScene<float>& dev_of<float>(Scene::DeviceCtx& c) { return c.dev32; }
// the context of the calling thread's current HIP device
static int current_ctx(Scene& s, Scene::DeviceCtx** out) {
    int dev = 0; HIP_OK(hipGetDevice(&dev));
    *out = &s.ctx_for(dev);
    return 0;
}

// Events and counter blocks belong to one device: waits, elapsed times and copies run with that device current.
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int device) { if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess; }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
// A finished launch's kernel time joins the running total exactly once (rt_kernel_time_total).  Current device: the slot's.
static int settle_slot(Scene& s, Scene::LaunchSlot& l) {
    if (!l.recorded || l.timed) return 0;
    HIP_OK(hipEventSynchronize((hipEvent_t)l.ev_stop));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, (hipEvent_t)l.ev_start, (hipEvent_t)l.ev_stop));
    s.kernel_ms_total += ms; s.kernel_launches_timed++;
    s.seq_ms[l.seq % Scene::SEQ_RING] = ms; s.seq_tag[l.seq % Scene::SEQ_RING] = l.seq;
    l.timed = true;
    return 0;
}
// A finished launch's counters join its frame's sums exactly once (Scene::acc_stats; rt_last_stats and friends report the whole frame,
// also when it was N launches on N devices or N virtual ranks).  Counters of frames older than the one being summed are dropped.
// Current device: the slot's.
static int harvest_slot(Scene& s, Scene::LaunchSlot& l) {
    if (!l.recorded || l.harvested) return 0;
    l.harvested = true;
    if (l.group < s.acc_group) return 0;
    HIP_OK(hipEventSynchronize((hipEvent_t)l.ev_stop));
    unsigned long long raw[RT_STATS_ROWS * RT_STATS_SLOTS];
    HIP_OK(hipMemcpy(raw, l.d_stats, sizeof(raw), hipMemcpyDeviceToHost));
    if (l.group > s.acc_group) { s.acc_group = l.group; for (unsigned long long& x : s.acc_stats) x = 0; }
    for (uint32_t k = 0; k < RT_STATS_SLOTS; k++) for (uint32_t r = 0; r < RT_STATS_ROWS; r++) s.acc_stats[k] += raw[r * RT_STATS_SLOTS + k];
    return 0;
}
// The launch slot for `stream`.  A stream alternates between (up to) two slots, so that the host can enqueue a frame's kernel while
// the previous one on that stream still runs — the kernels follow each other on the GPU without a launch gap — and waits only for the
// launch before that: the slot this stream used least recently if it already has two, else an unused one, else the least recently
// used one of all once its launch has finished.
static int acquire_slot(Scene& s, Scene::DeviceCtx& c, hipStream_t stream, Scene::LaunchSlot** out) {
    Scene::LaunchSlot* pick = nullptr;
    Scene::LaunchSlot* mine_lru = nullptr; int n_mine = 0;
    for (Scene::LaunchSlot& l : c.slots) if (l.recorded && l.stream == (void*)stream) { n_mine++; if (!mine_lru || l.seq < mine_lru->seq) mine_lru = &l; }
    if (n_mine >= 2) pick = mine_lru;
    if (!pick) for (Scene::LaunchSlot& l : c.slots) if (!l.recorded) { pick = &l; break; }
    if (!pick && mine_lru) pick = mine_lru;
    if (!pick) { pick = &c.slots[0]; for (Scene::LaunchSlot& l : c.slots) if (l.seq < pick->seq) pick = &l; }
    if (settle_slot(s, *pick)) return -1;             // also waits for a launch that may still be running in this slot
    if (pick->group == s.frame_group && harvest_slot(s, *pick)) return -1;      // an earlier share of the frame being enqueued (virtual ranks): keep its counters
    if (!pick->d_queue) HIP_OK(hipMalloc(&pick->d_queue, 64));
    if (!pick->d_stats) HIP_OK(hipMalloc(&pick->d_stats, RT_STATS_BYTES));
    if (!pick->ev_start) { hipEvent_t e; HIP_OK(hipEventCreate(&e)); pick->ev_start = e; }
    if (!pick->ev_stop) { hipEvent_t e; HIP_OK(hipEventCreate(&e)); pick->ev_stop = e; }
    pick->stream = (void*)stream;
    *out = pick;
    return 0;
}

// Loop shape for mesh scenes (same samples either way): a triangle-mesh BVH that stands beside other top-level objects is
// entered by a minority of the rays, which is where persistent traversal pays (measured +20 % on the teapot room); when
// every ray walks the BVH (the BVH is the world) the lock-step loop is faster.
// LDS traversal stack entries per lane: none in the reference's left-then-right order (skip links, rt_ir.h DBvhNode), the tree's depth
// when children are visited nearer-first
static uint32_t stack_depth_of(const HostFlat& f, uint32_t effective) { return ((effective & RT_NEAR_FIRST_BVH) && (f.feats & F_BVH)) ? f.bvh_depth : 0u; }
// A mesh scene whose loop shape is open: triangle-mesh BVHs beside other top-level objects, neither loop asked for.
static bool loop_shape_is_open(const HostFlat& f, uint32_t flags, size_t* n_bvh_objects_out = nullptr) {
    if ((flags & (RT_PERSISTENT_BVH | RT_LOCKSTEP_BVH)) || !(f.feats & F_BVH) || (f.feats & ~(uint32_t)(F_BVH | F_TRIS)) != 0u) return false;
    size_t n_bvh_objects = 0;
    for (uint32_t i = 0; i < f.n_top; i++) n_bvh_objects += f.objects[i].geom_kind == G_BVH ? 1u : 0u;
    if (n_bvh_objects_out) *n_bvh_objects_out = n_bvh_objects;
    return n_bvh_objects != 0 && n_bvh_objects < f.n_top;
}
static uint32_t effective_flags(const HostFlat& f, uint32_t flags, int loop_choice = -1, int* how_out = nullptr) {
    uint32_t out = flags;
    size_t n_bvh_objects = 0;
    if (how_out) *how_out = (flags & (RT_PERSISTENT_BVH | RT_LOCKSTEP_BVH)) ? 3 : 0;
    if (loop_shape_is_open(f, flags, &n_bvh_objects)) {
        if (how_out) *how_out = loop_choice >= 0 ? 2 : 1;
        // Measured for this scene and view where a frame is big enough to be worth two small calibration launches (calibrate_loop_shape);
        // otherwise by tree size: *measured* (tools/mesh_size_probe.py: a lit room with one / two random triangle meshes, persistent ÷
        // lock-step kernel time; round 4 / round 5 with the filtered walk) one tree of 63 / 199 / 599 / 1999 / 5999 nodes 1.24 / 1.09 / 0.99 /
        // 0.85 / 0.80 then, 1.26 / 1.18 / 1.13 / 1.10 / 0.95 now; two trees of 126 ... 11998 nodes 1.73 ... 0.86 then, 1.82 ... 1.05 now —
        // while the teapot room (one tree, 2047 nodes, a closed surface that fills a third of the view) is 0.85 then and 0.90 now: node
        // counts alone do not decide it, which is why it is measured where it matters.
        const bool by_size = f.bvh.size() >= 640u * n_bvh_objects * n_bvh_objects;
        if (loop_choice >= 0 ? loop_choice == 1 : by_size) out |= RT_PERSISTENT_BVH;
    }
    if (flags & RT_LOCKSTEP_BVH) out &= ~(uint32_t)RT_PERSISTENT_BVH;
    // BVHs with object leaves (any Hittable as a BVH child) are served by one instantiation: the reference's order, the lock-step loop
    // (the nearer-first order is an opt-in with the same closest hits; a walk inside a walk would need a second LDS stack for it)
    if (f.feats & F_NESTED) out &= ~(uint32_t)(RT_NEAR_FIRST_BVH | RT_PERSISTENT_BVH);
    // Speculative box steps (scheduling only): for the lock-step all-features-but-PBR kernel when the world is ONE bare BVH — every ray
    // enters it, which is where walking on past an untested leaf pays (*measured* random spheres +2.6 %; scenes whose trees few lanes
    // enter lose 3 %)
    {
        const bool one_bvh = f.n_top == 1 && f.objects[0].geom_kind == G_BVH && f.objects[0].medium < 0;
        const bool can = (f.feats & F_BVH) && !(f.feats & F_PBR) && (f.feats & ~(uint32_t)(F_BVH | F_TRIS)) != 0u &&
                         !(out & (RT_NEAR_FIRST_BVH | RT_PERSISTENT_BVH));
        if (can && one_bvh && !(flags & RT_NO_SPECULATE_BVH)) out |= RT_SPECULATE_BVH;
        if (!can || (flags & RT_NO_SPECULATE_BVH)) out &= ~(uint32_t)RT_SPECULATE_BVH;
    }
    return out;
}

// The view a calibration is valid for: camera, frame size, precision (FNV-1a over their bytes).
static unsigned long long loop_view_key(const rt_camera* cam, uint32_t W, uint32_t H, uint32_t flags) {
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
    const uint32_t tail[3] = {W, H, flags & (uint32_t)RT_F32};
    mix(cam, sizeof(rt_camera_args)); mix(tail, sizeof(tail));
    return h ? h : 1ull;
}
// The stored loop shape as it applies to this view: a calibration only for the view it measured, a set shape for every view.
static int loop_choice_for(const Scene& s, const rt_camera* cam, uint32_t W, uint32_t H, uint32_t flags) {
    if (s.loop_choice < 0) return -1;
    if (s.loop_how == 4) return s.loop_choice;
    return (s.loop_how == 2 && s.loop_key == loop_view_key(cam, W, H, flags)) ? s.loop_choice : -1;
}

// BVH nodes (depth order: the top levels first) that fit into the LDS a one-workgroup-per-CU kernel leaves free beside its waves'
// queues and stacks; 0 for the list-scene kernels.
template <typename T> size_t lds_node_bytes(uint32_t effective) { return filtered_walk(effective) ? sizeof(DFNode) : sizeof(DBvhNode<T>); }
template <typename T> uint32_t cached_nodes(const LaunchShape& g, const HostFlat& f, const hipDeviceProp_t& prop, uint32_t stack_depth, uint32_t effective) {
    if (!g.one_per_cu || f.bvh.empty()) return 0u;
    size_t lds_total = (size_t)prop.maxSharedMemoryPerMultiProcessor;
    if (lds_total < 65536u) lds_total = 65536u;
    const size_t nb = lds_node_bytes<T>(effective);
    const size_t fixed = pathtrace_lds_bytes(g, stack_depth, 0u, nb);
    if (fixed + nb > lds_total) return 0u;
    size_t room = (lds_total - fixed) / nb;
    // RT_NODE_CACHE_MAX (tests, A/B runs): stage at most that many nodes (0 = every node comes from global memory); scheduling only
    if (const char* v = std::getenv("RT_NODE_CACHE_MAX")) { const long n = std::strtol(v, nullptr, 10); if (n >= 0 && (size_t)n < room) room = (size_t)n; }
    return (uint32_t)std::min(room, f.bvh.size());
}

// RT_ROOM_NO_NAN_PATH (tests only): the kernels never leave the list with the room for the list as the reference has it — what
// tests/test_fuzz_gpu.py uses to show that its hostile rays do find the order-dependent NaN hits that path exists for.
static bool room_nan_path() { return std::getenv("RT_ROOM_NO_NAN_PATH") == nullptr; }

template <typename T>
int render_impl(Scene& s, const rt_camera* camp, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                uint64_t seed, uint32_t flags, uint32_t tile_px, uint32_t rank, uint32_t world, void* d_out, size_t d_out_bytes,
                void* d_samples, hipStream_t stream) {
    Scene::DeviceCtx* cp = nullptr;
    if (current_ctx(s, &cp)) return -1;
    Scene::DeviceCtx& c = *cp;
    DeviceScene<T>& d = dev_of<T>(c);
    if (ensure_uploaded<T>(s, d)) return -1;
    const HostFlat& f = s.flat;
    KParams<T> P;
    std::memset((void*)&P, 0, sizeof(P));
    P.objects = (const DObject*)d.objects; P.n_objects = f.n_top; P.n_objects_alt = room_nan_path() ? f.n_alt : 0u;
    P.ops = (const DOp<T>*)d.ops; P.rects = (const DRect<T>*)d.rects; P.spheres = (const DSphere<T>*)d.spheres;
    P.mspheres = (const DMSphere<T>*)d.mspheres; P.tris = (const DTri<T>*)d.tris; P.bvh = (const DBvhNode<T>*)d.bvh;
    P.n_bvh = (uint32_t)f.bvh.size();
    P.bvh_f = (const DFNode*)d.bvh_f; P.filter_m = f.filter_m; P.rect_m = f.rect_m;
    P.materials = (const DMaterial<T>*)d.materials; P.textures = (const DTexture<T>*)d.textures; P.media = (const DMedium<T>*)d.media;
    P.lights = (const DLight*)d.lights; P.n_lights = (uint32_t)f.lights.size();
    P.perlins = (const DPerlin<T>*)d.perlins; P.pbr = (const DPbr<T>*)d.pbr; P.image_bytes = (const uint8_t*)d.image;
    {   // the f32 tables are rounded copies: the tame bound is checked at the precision that is uploaded
        const double big = sizeof(T) == 8 ? 1e300 : 1e30;
        bool tame = f.bvh_tame;
        for (const DBvhNode<double>& nd : f.bvh) for (int k = 0; k < 3 && tame; k++) tame = std::fabs(nd.mn[k]) < big && std::fabs(nd.mx[k]) < big;
        P.bvh_tame = tame ? 1u : 0u;
    }
    rt_camera_args ca; std::memcpy(&ca, camp, sizeof(ca));
    DCamera<double> cam; camera_new(ca, cam);
    for (int k = 0; k < 3; k++) {
        P.cam.origin[k] = (T)cam.origin[k]; P.cam.lower_left_corner[k] = (T)cam.lower_left_corner[k];
        P.cam.horizontal[k] = (T)cam.horizontal[k]; P.cam.vertical[k] = (T)cam.vertical[k];
        P.cam.cu[k] = (T)cam.cu[k]; P.cam.cv[k] = (T)cam.cv[k]; P.background[k] = (T)bg[k];
    }
    P.cam.lens_radius = (T)cam.lens_radius; P.cam.time0 = (T)cam.time0; P.cam.time1 = (T)cam.time1;
    P.W = W; P.H = H; P.spp = spp; P.max_depth = max_depth; P.seed = seed; int loop_how = 0;
    P.flags = effective_flags(f, flags, loop_choice_for(s, camp, W, H, flags), &loop_how);
    if (loop_how == 2 && s.loop_how == 4) loop_how = 4;
    P.stack_depth = stack_depth_of(f, P.flags);
    P.tile_px = tile_px; P.rank = rank; P.world = world;
    P.n_local_tiles = rt_local_tiles(W, H, tile_px, rank, world);
    P.trav_hi = s.trav_hi; P.trav_lo = s.trav_lo; P.trav_leaf = s.trav_leaf;
    if (P.trav_hi < 1u) P.trav_hi = 1u; if (P.trav_hi > 64u) P.trav_hi = 64u;
    if (P.trav_lo < 1u) P.trav_lo = 1u; if (P.trav_lo > P.trav_hi) P.trav_lo = P.trav_hi;
    if (P.trav_leaf < 1u) P.trav_leaf = 1u; if (P.trav_leaf > 64u) P.trav_leaf = 64u;      // >= 1: a box step must win the vote when no leaf is pending
    uint64_t n_local_px = (uint64_t)P.n_local_tiles * tile_px;
    if (n_local_px >= 0xFFFFFFFFull) return set_err("too many local pixels");
    if ((size_t)n_local_px * 3 * sizeof(double) > d_out_bytes) return set_err("output buffer too small for n_local_tiles * tile_px * 3 doubles");
    if (!s.group_open) s.frame_group++;              // a frame of its own (rt_render_multi numbers its N launches as one)
    Scene::LaunchSlot* slot = nullptr;
    if (acquire_slot(s, c, stream, &slot)) return -1;
    P.queue = (uint32_t*)slot->d_queue; P.stats = (unsigned long long*)slot->d_stats;
    P.out = (double*)d_out; P.samples_out = (double*)d_samples;

    hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, c.device));
    LaunchShape shape = pathtrace_shape(f.feats, P.flags);
    P.n_cached = cached_nodes<T>(shape, f, prop, P.stack_depth, P.flags);
    // the camera-path queue: the kernel family's minimum, widened (refills at full lane occupancy) while every node still fits
    for (uint32_t q = 64u; q > shape.queue_entries; q >>= 1) {
        LaunchShape wide = shape; wide.queue_entries = q;
        if (cached_nodes<T>(wide, f, prop, P.stack_depth, P.flags) == P.n_cached) { shape = wide; break; }
    }
    // a tree that does not fit anyway: 32 entries when that costs the node cache less than a fifth of its nodes — a 16-entry queue runs
    // the camera code four times as often at a quarter of the lanes (5 % of a final-scene frame), and the deepest cached levels are
    // worth less (*measured* final scene, 2240 nodes + 16 entries 44.6 ms, 1920 + 32 42.7 ms, 1280 + 64 46.5 ms per 64 spp)
    if (shape.queue_entries < 32u && P.n_cached < (uint32_t)f.bvh.size()) {
        LaunchShape wide = shape; wide.queue_entries = 32u;
        const uint32_t n32 = cached_nodes<T>(wide, f, prop, P.stack_depth, P.flags);
        if ((uint64_t)n32 * 5u >= (uint64_t)P.n_cached * 4u) { shape = wide; P.n_cached = n32; }
    }
    if (const char* v = std::getenv("RT_QUEUE_ENTRIES")) {       // A/B runs only: the queue first, the node cache gets what is left
        const long n = std::strtol(v, nullptr, 10);
        if (n == 16 || n == 32 || n == 64) { shape.queue_entries = (uint32_t)n; P.n_cached = cached_nodes<T>(shape, f, prop, P.stack_depth, P.flags); }
    }
    P.queue_entries = shape.queue_entries;
    size_t shmem = pathtrace_lds_bytes(shape, P.stack_depth, P.n_cached, lds_node_bytes<T>(P.flags));
    int bpc = pathtrace_blocks_per_cu<T>(f.feats, P.flags, shmem);
    if (bpc <= 0) return set_err("occupancy query failed for the path-tracing kernel (LDS: " + std::to_string(shmem) + " bytes per workgroup)");
    const uint64_t waves_per_block = shape.threads / 64u;
    uint64_t waves_needed = (n_local_px * spp + 63) / 64;
    uint64_t blocks_needed = (waves_needed + waves_per_block - 1) / waves_per_block;
    uint64_t n_blocks = (uint64_t)prop.multiProcessorCount * (uint64_t)bpc;
    if (n_blocks > blocks_needed) n_blocks = blocks_needed;
    if (n_blocks == 0) n_blocks = 1;
    {   // Work chunks of at most 256 samples: several pixels per chunk at low spp, several chunks per pixel at high spp.  A small frame
        // (BASELINE config 1: 1400 samples per resident wave) gets smaller ones, down to 64 samples, so that every wave still sees about
        // 32 of them and the frame does not end with a few waves finishing a last big chunk alone: *measured* on C1, 64-sample chunks
        // 4.54 ms, 128 4.88 ms, 256 5.46 ms per frame.
        uint64_t per_wave = (n_local_px * spp) / (n_blocks * waves_per_block * 32ull);
        uint32_t CH = (uint32_t)std::min<uint64_t>(256u, std::max<uint64_t>(64u, per_wave));
        if (const char* v = std::getenv("RT_CHUNK_SAMPLES")) { const long n = std::strtol(v, nullptr, 10); if (n >= 16 && n <= 4096) CH = (uint32_t)n; }   // A/B runs only
        if (spp >= 2u * CH) { P.chunk_px = 1u; P.chunks_per_px = (spp + CH - 1) / CH; P.chunk_spp = (spp + P.chunks_per_px - 1) / P.chunks_per_px; }
        else { P.chunk_px = (CH + spp - 1) / spp; P.chunks_per_px = 1u; P.chunk_spp = spp; }
        P.n_coarse_px = 0;
    }
    {   // split only the last ~1.5 pixels per resident wave into fine chunks
        uint64_t fine_px = n_blocks * waves_per_block * 3ull / 2ull;
        if (P.chunks_per_px > 1 && n_local_px > fine_px) P.n_coarse_px = (uint32_t)(n_local_px - fine_px);
        uint64_t n_chunks64 = P.chunks_per_px > 1
            ? (uint64_t)P.n_coarse_px + (n_local_px - P.n_coarse_px) * (uint64_t)P.chunks_per_px
            : (n_local_px + P.chunk_px - 1) / P.chunk_px;
        if (n_chunks64 >= 0xFFFFFFFFull) return set_err("too many work chunks");
        P.n_chunks = (uint32_t)n_chunks64;
    }

    P.trace_out = nullptr; P.trace_px = 0u; P.trace_s = 0u;
    if (s.trace_px >= 0) {                          // debugging aid: 16 doubles per level of one path (written by -DRT_TRACE_PATH builds only)
        if (world > 1) return set_err("rt_debug_trace_path: tracing a path is for unsharded renders (world = 1)");
        const size_t bytes = ((size_t)max_depth + 1u) * 16u * sizeof(double);
        if (!s.d_trace || s.trace_device != c.device || s.trace_levels != max_depth + 1u) {      // one buffer per (device, depth): earlier launches keep a valid pointer
            if (s.d_trace) { DeviceGuard guard(s.trace_device); (void)hipDeviceSynchronize(); (void)hipFree(s.d_trace); s.d_trace = nullptr; }
            HIP_OK(hipMalloc(&s.d_trace, bytes));
            s.trace_device = c.device; s.trace_levels = max_depth + 1u;
        }
        HIP_OK(hipMemsetAsync(s.d_trace, 0, bytes, stream));
        P.trace_out = (double*)s.d_trace; P.trace_px = (uint32_t)s.trace_px; P.trace_s = (uint32_t)s.trace_s;
    }
    HIP_OK(hipMemsetAsync(slot->d_queue, 0, 64, stream));
    HIP_OK(hipMemsetAsync(slot->d_stats, 0, RT_STATS_BYTES, stream));
    HIP_OK(hipMemsetAsync(d_out, 0, (size_t)n_local_px * 3 * sizeof(double), stream));
    HIP_OK(hipEventRecord((hipEvent_t)slot->ev_start, stream));
    HIP_OK(launch_pathtrace<T>(P, f.feats, (uint32_t)n_blocks, shmem, stream));
    HIP_OK(hipEventRecord((hipEvent_t)slot->ev_stop, stream));
    slot->recorded = true; slot->timed = false; slot->seq = ++s.launch_seq; slot->group = s.frame_group; slot->harvested = false;
    c.last_slot = (int)(slot - c.slots);
    s.last_device = c.device;
    s.launch_info[0] = (uint32_t)n_blocks; s.launch_info[1] = shape.threads; s.launch_info[2] = (uint32_t)shmem; s.launch_info[3] = P.n_cached;
    s.launch_info[4] = (uint32_t)f.bvh.size(); s.launch_info[5] = (uint32_t)bpc;
    s.last_loop[1] = (int)pathtrace_feats(f.feats, P.flags);
    s.last_loop[0] = !(f.feats & F_BVH) ? 0 : ((s.last_loop[1] & (int)F_PERSIST) ? 2 : 1);
    s.last_loop[2] = (f.feats & F_BVH) ? loop_how : 0; s.last_loop[3] = sizeof(T) == 4 ? 1 : 0;
    return 0;
}

int check_frame_args(rt_scene* sc, const rt_camera* cam, const double* bg, uint32_t W, uint32_t H, uint32_t spp) {
    if (!sc || !cam || !bg) return set_err("null argument");
    if (W < 2 || H < 2) return set_err("W and H must be >= 2 (u,v divide by W-1 and H-1, src/main.rs:817-818)");
    if (spp == 0) return set_err("samples_per_pixel must be >= 1");
    if ((uint64_t)W * H > 0x7FFFFFFFull) return set_err("frame too large: W*H must be <= 2^31 - 1 (per-path RNG keys, csrc/rt_rng.h)");
    return 0;
}

// Worlds that are ONE bare BVH: tune the filter tree's contraction for this view (rt_flatten.cpp tune_filter_tree: host arithmetic, < 1 ms)
// and refresh every device copy of the filter nodes.  SYNCHRONOUS (it waits for the scene's launches before it overwrites a table they
// read), so it runs where calibrate_loop_shape runs: rt_scene_calibrate and the synchronous entry points, once per view; a no-op for
// every other scene.  Samples do not depend on it (any conservative hierarchy over the leaves gives the reference's).
static int tune_for_view(Scene& s, const rt_camera* cam, uint32_t W, uint32_t H) {
    const unsigned long long key = loop_view_key(cam, W, H, 0u);
    if (s.filter_key == key) return 0;
    s.filter_key = key;
    rt_camera_args ca; std::memcpy(&ca, cam, sizeof(ca));
    DCamera<double> dcam; camera_new(ca, dcam);
    if (!tune_filter_tree(s, dcam)) return 0;
    if (settle_all_launches(s)) return -1;
    std::vector<DFNode> fn(s.flat.bvh_f); fn.push_back(DFNode{});
    for (Scene::DeviceCtx* c : s.ctxs) {
        DeviceGuard guard(c->device);
        if (c->dev64.valid && c->dev64.bvh_f) HIP_OK(hipMemcpy(c->dev64.bvh_f, fn.data(), fn.size() * sizeof(DFNode), hipMemcpyHostToDevice));
        if (c->dev32.valid && c->dev32.bvh_f) HIP_OK(hipMemcpy(c->dev32.bvh_f, fn.data(), fn.size() * sizeof(DFNode), hipMemcpyHostToDevice));
    }
    return 0;
}

// Mesh scenes: which loop shape is faster depends on what the rays of THIS view do inside the trees (a closed surface that fills a third
// of the frame: the persistent loop by 10 %; a sparse cloud of triangles in a corner: the lock-step loop by 10 ... 80 %), not on anything
// the flattener can see.  Both give the same samples bit for bit, so a calibration renders the same view at <= 1024 x 1024 x 16 twice in
// each shape and keeps the faster one FOR THAT VIEW (Scene::loop_key: camera, frame size, precision; forgotten when the scene changes).
// Within 3 % of each other the size rule stands.  It is a SYNCHRONOUS piece of work (it waits for every launch of the scene, reads four
// event pairs) and therefore runs only where the caller is synchronous anyway: rt_scene_calibrate, and the first render of a frame of
// >= 1e8 samples through rt_render / rt_render_samples / rt_render_multi.  The asynchronous entry points (rt_render_device,
// rt_render_multi_device) never calibrate: they use what is stored for their view, else the size rule.  The calibration launches do not
// count in rt_kernel_time_total and leave no trace in rt_last_* (the render that follows overwrites them); their frame buffer is kept
// with the scene (no allocation, no hipFree — which would synchronise the device — after the first time).
static int calibrate_loop_shape(Scene& s, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                                uint64_t seed, uint32_t flags, hipStream_t stream, bool only_if_worth_it) {
    if (!loop_shape_is_open(s.flat, flags) || (flags & RT_NEAR_FIRST_BVH)) return 0;
    if (s.loop_how == 4 || loop_choice_for(s, cam, W, H, flags) >= 0) return 0;          // a shape was set, or this view has been measured
    if (only_if_worth_it && ((uint64_t)W * H * spp < 100000000ull || std::getenv("RT_NO_LOOP_CALIBRATION"))) return 0;   // (the variable: A/B runs and tests of the size rule)
    // (the copies must be far above the ~0.5 ms a persistent launch spends filling and draining the chip, and long enough for the loops'
    // steady state: *measured* on the teapot room, persistent / lock-step 1.05 at 4 M samples, 0.94 at 20 M, 0.92 at 41 M, 0.85 at 4 G)
    uint32_t k = 1; while ((uint64_t)(W / k) * (H / k) > 1048576ull) k++;
    const uint32_t Wc = std::max(2u, W / k), Hc = std::max(2u, H / k), sc_spp = std::min(spp, 16u);
    const size_t bytes = (size_t)Wc * Hc * 3 * sizeof(double);
    int dev = 0; HIP_OK(hipGetDevice(&dev));
    if (settle_all_launches(s)) return -1;
    if (!s.d_calib || s.calib_device != dev || s.calib_bytes < bytes) {
        if (s.d_calib) { DeviceGuard guard(s.calib_device); (void)hipDeviceSynchronize(); (void)hipFree(s.d_calib); s.d_calib = nullptr; }
        HIP_OK(hipMalloc(&s.d_calib, bytes));
        s.calib_device = dev; s.calib_bytes = bytes;
    }
    const double keep_ms = s.kernel_ms_total; const unsigned long long keep_n = s.kernel_launches_timed;
    // (inside an rt_render_multi* frame the N launches share one frame number: the calibration launches get their own, and the frame a fresh one)
    const bool in_group = s.group_open; s.group_open = false;
    float best[2] = {1e30f, 1e30f};
    int rc = 0;
    for (int round = 0; round < 2 && rc == 0; round++)
        for (int shape = 0; shape < 2 && rc == 0; shape++) {
            const uint32_t fl = flags | (shape ? RT_PERSISTENT_BVH : RT_LOCKSTEP_BVH);
            rc = (flags & RT_F32) ? render_impl<float>(s, cam, bg, Wc, Hc, sc_spp, max_depth, seed, fl, Wc * Hc, 0, 1, s.d_calib, bytes, nullptr, stream)
                                  : render_impl<double>(s, cam, bg, Wc, Hc, sc_spp, max_depth, seed, fl, Wc * Hc, 0, 1, s.d_calib, bytes, nullptr, stream);
            float ms = 0.f;
            if (rc == 0 && device_kernel_ms(s, dev, &ms) == 0) best[shape] = std::min(best[shape], ms); else rc = -1;
        }
    if (settle_all_launches(s)) rc = -1;
    s.kernel_ms_total = keep_ms; s.kernel_launches_timed = keep_n;
    s.group_open = in_group; if (in_group) s.frame_group++;
    if (rc != 0) return -1;
    if (best[1] < 0.97f * best[0]) s.loop_choice = 1;
    else if (best[0] < 0.97f * best[1]) s.loop_choice = 0;
    else s.loop_choice = (effective_flags(s.flat, flags, -1) & RT_PERSISTENT_BVH) ? 1 : 0;
    s.loop_how = 2; s.loop_key = loop_view_key(cam, W, H, flags); s.loop_ms[0] = best[0]; s.loop_ms[1] = best[1];
    return 0;
}

int render_any(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
               uint64_t seed, uint32_t flags, uint32_t tile_px, uint32_t rank, uint32_t world, void* d_out, size_t d_out_bytes,
               void* d_samples, hipStream_t stream, bool may_calibrate) {
    if (check_frame_args(sc, cam, bg, W, H, spp)) return -1;
    if (tile_px == 0 || world == 0 || rank >= world) return set_err("bad tile decomposition");
    if (rt_device_count() <= 0) return set_err("no HIP device: librt_amd has no CPU rendering path");
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    if (may_calibrate && (tune_for_view(sc->s, cam, W, H) || calibrate_loop_shape(sc->s, cam, bg, W, H, spp, max_depth, seed, flags, stream, true))) return -1;
    if (flags & RT_F32) return render_impl<float>(sc->s, cam, bg, W, H, spp, max_depth, seed, flags, tile_px, rank, world, d_out, d_out_bytes, d_samples, stream);
    return render_impl<double>(sc->s, cam, bg, W, H, spp, max_depth, seed, flags, tile_px, rank, world, d_out, d_out_bytes, d_samples, stream);
}

} // namespace

extern "C" {

int rt_render_device(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                     uint64_t seed, uint32_t flags, uint32_t tile_px, uint32_t rank, uint32_t world, void* d_out, size_t d_out_bytes, void* hip_stream) {
    return render_any(sc, cam, bg, W, H, spp, max_depth, seed, flags, tile_px, rank, world, d_out, d_out_bytes, nullptr, (hipStream_t)hip_stream, false);
}

// Mesh scenes: measure which loop shape is faster for this view (see calibrate_loop_shape) — synchronous, on the calling thread's current
// device; a no-op for every other scene, for a view already measured and after rt_scene_set_loop_shape.
int rt_scene_calibrate(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                       uint64_t seed, uint32_t flags) {
    if (check_frame_args(sc, cam, bg, W, H, spp)) return -1;
    if (rt_device_count() <= 0) return set_err("no HIP device: librt_amd has no CPU rendering path");
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    if (tune_for_view(sc->s, cam, W, H)) return -1;
    return calibrate_loop_shape(sc->s, cam, bg, W, H, spp, max_depth, seed, flags, nullptr, false);
}
int rt_scene_set_loop_shape(rt_scene* sc, int shape) {
    if (!sc) return set_err("null argument");
    if (shape < -1 || shape > 1) return set_err("rt_scene_set_loop_shape: -1 (forget), 0 (lock-step) or 1 (persistent traversal)");
    sc->s.loop_choice = shape; sc->s.loop_how = shape < 0 ? 0 : 4;
    if (shape < 0) sc->s.loop_ms[0] = sc->s.loop_ms[1] = 0.f;
    return 0;
}
int rt_scene_loop_shape(rt_scene* sc) { return sc ? sc->s.loop_choice : -1; }
int rt_last_loop_info(rt_scene* sc, int32_t out4[4], float calibration_ms2[2]) {
    if (!sc || !out4) return set_err("null argument");
    for (int k = 0; k < 4; k++) out4[k] = sc->s.last_loop[k];
    if (calibration_ms2) { calibration_ms2[0] = sc->s.loop_ms[0]; calibration_ms2[1] = sc->s.loop_ms[1]; }
    return 0;
}

// One-time work a first render would otherwise do inside its call: flatten, upload the tables for the chosen precision,
// create events / scratch words, and make the runtime load the kernel's code object (occupancy query).  No kernel is launched.
int rt_scene_prepare(rt_scene* sc, uint32_t flags) {
    if (!sc) return set_err("null argument");
    if (rt_device_count() <= 0) return set_err("no HIP device: librt_amd has no CPU rendering path");
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    Scene::DeviceCtx* cp = nullptr;
    if (current_ctx(sc->s, &cp)) return -1;
    return rt::prepare_device(sc->s, *cp, flags);
}

} // extern "C"
namespace rt {
// what the synchronous multi-GPU entry point (rt_render_multi) does before it enqueues its frame: the implicit calibration of a first large frame
int calibrate_if_worth_it(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                          uint64_t seed, uint32_t flags) {
    if (check_frame_args(sc, cam, bg, W, H, spp)) return -1;
    if (rt_device_count() <= 0) return set_err("no HIP device: librt_amd has no CPU rendering path");
    if (!flatten_scene(sc->s)) { g_err = sc->s.error; return -1; }
    if (tune_for_view(sc->s, cam, W, H)) return -1;
    return calibrate_loop_shape(sc->s, cam, bg, W, H, spp, max_depth, seed, flags, nullptr, true);
}
// rt_scene_prepare's work for one device context; the calling thread's current HIP device must be c.device and the scene must be
// flattened.  Touches only `c` and reads s.flat, so rt_render_multi may run it for several devices from several threads at once.
int prepare_device(Scene& s, Scene::DeviceCtx& c, uint32_t flags) {
    if (flags & RT_F32) { if (ensure_uploaded<float>(s, c.dev32)) return -1; }
    else { if (ensure_uploaded<double>(s, c.dev64)) return -1; }
    for (Scene::LaunchSlot& l : c.slots) {
        if (!l.d_queue) HIP_OK(hipMalloc(&l.d_queue, 64));
        if (!l.d_stats) HIP_OK(hipMalloc(&l.d_stats, RT_STATS_BYTES));
        if (!l.ev_start) { hipEvent_t e; HIP_OK(hipEventCreate(&e)); l.ev_start = e; }
        if (!l.ev_stop) { hipEvent_t e; HIP_OK(hipEventCreate(&e)); l.ev_stop = e; }
    }
    const uint32_t eff = effective_flags(s.flat, flags, s.loop_how == 4 ? s.loop_choice : -1);
    const LaunchShape shape = pathtrace_shape(s.flat.feats, eff);
    hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, c.device));
    int bpc;
    const uint32_t sd = stack_depth_of(s.flat, eff);
    if (flags & RT_F32) bpc = pathtrace_blocks_per_cu<float>(s.flat.feats, eff, pathtrace_lds_bytes(shape, sd, cached_nodes<float>(shape, s.flat, prop, sd, eff), lds_node_bytes<float>(eff)));
    else bpc = pathtrace_blocks_per_cu<double>(s.flat.feats, eff, pathtrace_lds_bytes(shape, sd, cached_nodes<double>(shape, s.flat, prop, sd, eff), lds_node_bytes<double>(eff)));
    if (bpc <= 0) return set_err("occupancy query failed for the path-tracing kernel");
    HIP_OK(hipDeviceSynchronize());
    return 0;
}
int settle_all_launches(Scene& s) {
    for (Scene::DeviceCtx* c : s.ctxs) { DeviceGuard guard(c->device); for (Scene::LaunchSlot& l : c->slots) if (settle_slot(s, l)) return -1; }
    return 0;
}
bool flatten_for_render(Scene& s) { if (flatten_scene(s)) return true; set_err(s.error); return false; }
int device_kernel_ms(Scene& s, int device, float* ms) {
    for (Scene::DeviceCtx* c : s.ctxs) if (c->device == device && c->last_slot >= 0) {
        Scene::LaunchSlot& l = c->slots[c->last_slot];
        DeviceGuard guard(c->device);
        HIP_OK(hipEventSynchronize((hipEvent_t)l.ev_stop));
        HIP_OK(hipEventElapsedTime(ms, (hipEvent_t)l.ev_start, (hipEvent_t)l.ev_stop));
        return 0;
    }
    return set_err("no kernel has been launched on that device");
}
}
extern "C" {

int rt_last_kernel_ms(rt_scene* sc, float* ms_out) {
    Scene::DeviceCtx* c = sc ? sc->s.last_ctx() : nullptr;
    if (!sc || !ms_out || !c || c->last_slot < 0) return set_err("no kernel has been launched for this scene");
    Scene::LaunchSlot& l = c->slots[c->last_slot];
    DeviceGuard guard(c->device);
    HIP_OK(hipEventSynchronize((hipEvent_t)l.ev_stop));
    HIP_OK(hipEventElapsedTime(ms_out, (hipEvent_t)l.ev_start, (hipEvent_t)l.ev_stop));
    return 0;
}

// [0] non-finite samples, [1] wave bounce-loop iterations, [2] lane-iterations with a live path  (last finished launch)
// Sum of the path-tracing kernels' durations (HIP events on their launch streams) and their number since the last reset; waits
// for the launches still in flight.  For callers that keep several frames in flight and must not stop after each one.
int rt_kernel_time_total(rt_scene* sc, double* ms_total, unsigned long long* n_launches, int reset) {
    if (!sc) return set_err("null argument");
    if (settle_all_launches(sc->s)) return -1;
    if (ms_total) *ms_total = sc->s.kernel_ms_total;
    if (n_launches) *n_launches = sc->s.kernel_launches_timed;
    if (reset) { sc->s.kernel_ms_total = 0.0; sc->s.kernel_launches_timed = 0; }
    return 0;
}
// The kernel spreads its end-of-launch counter atomics over RT_STATS_ROWS copies of the counter block (row = block index mod
// rows): 4096 waves adding to one address serialise in the L2 atomic unit.  Readers sum the rows — and the launches of the most
// recent FRAME: one for rt_render / rt_render_device, one per device (or virtual rank) for rt_render_multi*.
static int read_stats(rt_scene* sc, unsigned long long h[RT_STATS_SLOTS]) {
    Scene::DeviceCtx* lc = sc ? sc->s.last_ctx() : nullptr;
    if (!sc || !lc || lc->last_slot < 0) return set_err("no kernel has been launched for this scene");
    Scene& s = sc->s;
    const unsigned long long group = lc->slots[lc->last_slot].group;
    for (Scene::DeviceCtx* c : s.ctxs) {
        DeviceGuard guard(c->device);
        for (Scene::LaunchSlot& l : c->slots) if (l.recorded && l.group == group && harvest_slot(s, l)) return -1;
    }
    for (uint32_t k = 0; k < RT_STATS_SLOTS; k++) h[k] = s.acc_group == group ? s.acc_stats[k] : 0ull;
    return 0;
}
// Geometry of the most recent launch: [0] workgroups, [1] threads per workgroup, [2] dynamic LDS bytes per workgroup, [3] BVH nodes
// staged in LDS, [4] BVH nodes in the scene, [5] resident workgroups per CU
int rt_last_launch_info(rt_scene* sc, uint32_t out6[6]) {
    if (!sc || !out6) return set_err("null argument");
    for (int k = 0; k < 6; k++) out6[k] = sc->s.launch_info[k];
    return 0;
}
int rt_last_stats(rt_scene* sc, unsigned long long out[3]) {
    unsigned long long h[RT_STATS_SLOTS];
    if (read_stats(sc, h)) return -1;
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    return 0;
}
// Accumulator flushes of the last launch: each is three hardware f64 atomic adds to the frame (the kernel's only global writes)
int rt_last_flush_count(rt_scene* sc, unsigned long long* out) {
    if (!out) return set_err("null argument");
    unsigned long long h[RT_STATS_SLOTS];
    if (read_stats(sc, h)) return -1;
    *out = h[11];
    return 0;
}
// BVH scenes: {advance passes, lanes advancing, traversal steps, lanes stepping} summed over waves; zeros for list scenes
int rt_last_traversal_stats(rt_scene* sc, unsigned long long out[4]) {
    unsigned long long h[RT_STATS_SLOTS];
    if (read_stats(sc, h)) return -1;
    out[0] = h[1]; out[1] = h[2]; out[2] = h[9]; out[3] = h[10];
    return 0;
}
// persistent-traversal kernels: {leaf steps, lanes in them} of the traversal steps above (the rest are box steps)
int rt_last_leaf_steps(rt_scene* sc, unsigned long long out[2]) {
    unsigned long long h[RT_STATS_SLOTS];
    if (read_stats(sc, h)) return -1;
    out[0] = h[12]; out[1] = h[13];
    return 0;
}
// diagnostic builds (-DRT_DIAG) only: wave-cycle sums of the six kernel sections; [6] wave-level rect tests, [7] those in which no
// lane's t lay in [t_min, closest]; zeros otherwise
int rt_debug_section_cycles(rt_scene* sc, unsigned long long out[8]) {
    unsigned long long h[RT_STATS_SLOTS];
    if (read_stats(sc, h)) return -1;
    for (int k = 0; k < 6; k++) out[k] = h[3 + k];
    out[6] = h[14]; out[7] = h[15];
    return 0;
}

// Debugging aid (see include/rt_amd.h): choose the path whose hits the next renders record / fetch the record
// Known-answer access to the closest-hit search of a LIST scene (no feature bit set: what the lean kernels serve): world.hit (main.rs:48)
// and the hit record for n given rays, through the kernels' own world_hit / finalize_hit.  rays: n x (origin[3], direction[3]); t_min: n;
// out: n x 12 = hit, t, position[3], normal[3], front_face, object index, primitive index, material.  Host pointers.
int rt_debug_list_hit(rt_scene* sc, uint32_t n, const double* rays, const double* t_min, double* out) {
    if (!sc || !rays || !t_min || !out) return set_err("null argument");
    if (n == 0) return 0;
    Scene& s = sc->s;
    if (!flatten_scene(s)) { g_err = s.error; return -1; }
    if (s.flat.feats != 0u) return set_err("rt_debug_list_hit serves list scenes only (no BVH, sphere, triangle, medium, texture, dielectric or PBR)");
    Scene::DeviceCtx* cp = nullptr;
    if (current_ctx(s, &cp)) return -1;
    DeviceScene<double>& d = dev_of<double>(*cp);
    if (ensure_uploaded<double>(s, d)) return -1;
    const HostFlat& f = s.flat;
    KParams<double> P;
    std::memset((void*)&P, 0, sizeof(P));
    P.objects = (const DObject*)d.objects; P.n_objects = f.n_top; P.n_objects_alt = room_nan_path() ? f.n_alt : 0u;
    P.ops = (const DOp<double>*)d.ops; P.rects = (const DRect<double>*)d.rects; P.rect_m = f.rect_m;
    P.materials = (const DMaterial<double>*)d.materials; P.textures = (const DTexture<double>*)d.textures;
    double *dr = nullptr, *dt = nullptr, *dout = nullptr;
    int rc = -1;
    if (hipMalloc(&dr, n * 48ull) == hipSuccess && hipMalloc(&dt, n * 8ull) == hipSuccess && hipMalloc(&dout, n * 96ull) == hipSuccess &&
        hipMemcpy(dr, rays, n * 48ull, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dt, t_min, n * 8ull, hipMemcpyHostToDevice) == hipSuccess &&
        launch_list_hit_kat(P, n, dr, dt, dout, nullptr) == hipSuccess && hipMemcpy(out, dout, n * 96ull, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
    (void)hipFree(dr); (void)hipFree(dt); (void)hipFree(dout);
    return rc == 0 ? 0 : set_err("rt_debug_list_hit: a HIP call failed");
}
int rt_debug_trace_path(rt_scene* sc, long long local_pixel, long long sample) {
    if (!sc) return set_err("null argument");
    sc->s.trace_px = local_pixel; sc->s.trace_s = sample;
    return 0;
}
int rt_debug_get_trace(rt_scene* sc, double* out, uint32_t n_levels) {
    if (!sc || !out) return set_err("null argument");
    if (!sc->s.d_trace) return set_err("no path has been traced (rt_debug_trace_path, then a render)");
    if (n_levels > sc->s.trace_levels) return set_err("rt_debug_get_trace: the last traced render recorded " + std::to_string(sc->s.trace_levels) + " levels (max_depth + 1)");
    DeviceGuard guard(sc->s.trace_device);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipMemcpy(out, sc->s.d_trace, (size_t)n_levels * 16u * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int rt_render_samples(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
                      uint64_t seed, uint32_t flags, double* rgb_sum_out, double* samples_out) {
    if (!rgb_sum_out) return set_err("null output");
    if (check_frame_args(sc, cam, bg, W, H, spp)) return -1;
    size_t n_px = (size_t)W * H;
    void* d_out = nullptr; void* d_samples = nullptr;
    if (rt_device_count() <= 0) return set_err("no HIP device: librt_amd has no CPU rendering path");
    HIP_OK(hipMalloc(&d_out, n_px * 3 * sizeof(double)));
    if (samples_out) {
        hipError_t e = hipMalloc(&d_samples, n_px * spp * 3 * sizeof(double));
        if (e != hipSuccess) { (void)hipFree(d_out); return set_err(std::string("hipMalloc(samples): ") + hipGetErrorString(e)); }
    }
    int rc = render_any(sc, cam, bg, W, H, spp, max_depth, seed, flags, (uint32_t)n_px, 0, 1, d_out, n_px * 3 * sizeof(double), d_samples, nullptr, true);
    if (rc == 0) {
        hipError_t e = hipStreamSynchronize(nullptr);
        if (e == hipSuccess) e = hipMemcpy(rgb_sum_out, d_out, n_px * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess && samples_out) e = hipMemcpy(samples_out, d_samples, n_px * spp * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { g_err = std::string("copy back: ") + hipGetErrorString(e); rc = -1; }
    }
    (void)hipFree(d_out);
    if (d_samples) (void)hipFree(d_samples);
    return rc;
}

int rt_render(rt_scene* sc, const rt_camera* cam, const double bg[3], uint32_t W, uint32_t H, uint32_t spp, uint32_t max_depth,
              uint64_t seed, uint32_t flags, double* rgb_sum_out) {
    return rt_render_samples(sc, cam, bg, W, H, spp, max_depth, seed, flags, rgb_sum_out, nullptr);
}

} // extern "C"

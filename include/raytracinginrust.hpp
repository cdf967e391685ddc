// include/raytracinginrust.hpp — C++ host API mirroring the reference's Rust scene-builder surface
// (4meame/RayTracingInRust) over the C-ABI of include/rt_amd.h.
//
// The reference is one Rust binary whose `main` builds `(world, lights)` with constructors such as
// `Sphere::new`, `AARect::new(Plane::XZ, ..)`, `Translate::new(Rotate::new(Axis::Y, Cube::new(..), -18.0), ..)`
// (src/main.rs:278-311) and then runs the per-pixel sample loop (src/main.rs:772-833).  No Rust toolchain exists in
// this environment, so the host side is C++ with the same names and argument order; every `T::new_(..)` below
// forwards to the C-ABI entry point a Rust `extern "C"` block would bind.  Failures the
// reference reports by panic (src/bvh.rs:55, src/main.rs:431) surface as rtr::Error.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "rt_amd.h"

namespace rtr {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

struct Vec3 {                                    // src/vec.rs:10-22
    double e[3];
    Vec3() : e{0, 0, 0} {}
    Vec3(double a, double b, double c) : e{a, b, c} {}
    static Vec3 new_(double a, double b, double c) { return Vec3(a, b, c); }
    double x() const { return e[0]; } double y() const { return e[1]; } double z() const { return e[2]; }
    Vec3 operator+(const Vec3& o) const { return Vec3(e[0] + o.e[0], e[1] + o.e[1], e[2] + o.e[2]); }
    Vec3 operator*(const Vec3& o) const { return Vec3(e[0] * o.e[0], e[1] * o.e[1], e[2] * o.e[2]); }
    Vec3 operator*(double s) const { return Vec3(e[0] * s, e[1] * s, e[2] * s); }
};
using Point3 = Vec3;
using Color = Vec3;

enum class Plane { XY = RT_PLANE_XY, XZ = RT_PLANE_XZ, YZ = RT_PLANE_YZ };   // src/rect.rs:9-13
enum class Axis { X = RT_AXIS_X, Y = RT_AXIS_Y, Z = RT_AXIS_Z };             // src/rotate.rs:8-12

// rand::thread_rng() stand-in for scene construction (seeded; csrc/rt_rng.h)
class Rng {
public:
    Rng(uint64_t seed, uint32_t stream) : r_(rt_rng_create(seed, stream)) {}
    ~Rng() { rt_rng_destroy(r_); }
    Rng(const Rng&) = delete; Rng& operator=(const Rng&) = delete;
    double gen_f64() { return rt_rng_f64(r_); }                                  // rng.gen::<f64>()
    double gen_range(double a, double b) { return rt_rng_range(r_, a, b); }      // rng.gen_range(a..b)
    Color color_random(double a, double b) { double x = gen_range(a, b), y = gen_range(a, b), z = gen_range(a, b); return Color(x, y, z); }   // Color::random, vec.rs:70-76
    rt_rng* raw() { return r_; }
private:
    rt_rng* r_;
};

// One scene: owns the C handle.  Texture / Material / Hittable are typed ids into it.
class Scene;
struct Texture { int id; };
struct Material { int id; };
struct Hittable { int id; };

class Scene {
public:
    Scene() : s_(rt_scene_create()) { if (!s_) throw Error("rt_scene_create failed"); }
    ~Scene() { rt_scene_destroy(s_); }
    Scene(const Scene&) = delete; Scene& operator=(const Scene&) = delete;
    rt_scene* raw() const { return s_; }
    int chk(int rc) const { if (rc < 0) throw Error(rt_scene_error(s_)); return rc; }
    void set(Hittable world, const std::vector<Hittable>& lights) {              // the (world, lights) pair, main.rs:153
        chk(rt_scene_set_world(s_, world.id));
        for (auto l : lights) chk(rt_lights_push(s_, l.id));
    }
    // not in the reference: opt-in SAH tree for `BVH::new_` (default: the reference's widest-axis object-median split, bvh.rs:18-73)
    void set_bvh_builder(rt_bvh_builder mode) { if (rt_scene_set_bvh_builder(s_, (int)mode) != 0) throw Error(rt_last_error()); }
private:
    rt_scene* s_;
};

// ---- textures (src/texture.rs)
struct ConstantTexture { static Texture new_(Scene& s, Color c) { return {s.chk(rt_texture_constant(s.raw(), c.e))}; } };
struct CheckTexture { static Texture new_(Scene& s, Texture odd, Texture even) { return {s.chk(rt_texture_check(s.raw(), odd.id, even.id))}; } };
struct NoiseTexture { static Texture new_(Scene& s, double scale, Rng& rng) { return {s.chk(rt_texture_noise(s.raw(), scale, rng.raw()))}; } };
struct ImageTexture { static Texture new_(Scene& s, const std::vector<uint8_t>& data, uint32_t w, uint32_t h) { return {s.chk(rt_texture_image(s.raw(), data.data(), w, h))}; } };

// ---- materials (src/mat.rs)
struct Lambertian { static Material new_(Scene& s, Texture albedo) { return {s.chk(rt_material_lambertian(s.raw(), albedo.id))}; } };
struct Metal { static Material new_(Scene& s, Color albedo, double fuzz) { return {s.chk(rt_material_metal(s.raw(), albedo.e, fuzz))}; } };
struct Dielectric { static Material new_(Scene& s, double ir) { return {s.chk(rt_material_dielectric(s.raw(), ir))}; } };
struct DiffuseLight { static Material new_(Scene& s, Texture emit) { return {s.chk(rt_material_diffuse_light(s.raw(), emit.id))}; } };
struct Isotropic { static Material new_(Scene& s, Texture albedo) { return {s.chk(rt_material_isotropic(s.raw(), albedo.id))}; } };
struct PBR {                                            // src/mat.rs:101
    static Material new_(Scene& s, Texture base_color, double metallic, double subsurface, double specular, double roughness, double specular_tint,
                         double anisotropic, double sheen, double sheen_tint, double clearcoat, double clearcoat_gloss) {
        const double p[10] = {metallic, subsurface, specular, roughness, specular_tint, anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss};
        return {s.chk(rt_material_pbr(s.raw(), base_color.id, p))};
    }
};

// ---- hittables
struct Sphere { static Hittable new_(Scene& s, Point3 c, double r, Material m) { return {s.chk(rt_sphere(s.raw(), c.e, r, m.id))}; } };
struct MovingSphere { static Hittable new_(Scene& s, Point3 c0, Point3 c1, double t0, double t1, double r, Material m) { return {s.chk(rt_moving_sphere(s.raw(), c0.e, c1.e, t0, t1, r, m.id))}; } };
struct AARect { static Hittable new_(Scene& s, Plane p, double a0, double a1, double b0, double b1, double k, Material m) { return {s.chk(rt_aarect(s.raw(), (int)p, a0, a1, b0, b1, k, m.id))}; } };
struct Cube { static Hittable new_(Scene& s, Point3 mn, Point3 mx, Material m) { return {s.chk(rt_cube(s.raw(), mn.e, mx.e, m.id))}; } };
struct Triangle { static Hittable new_(Scene& s, const std::array<Point3, 3>& v, Material m) { double f[9]; for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) f[i * 3 + k] = v[i].e[k]; return {s.chk(rt_triangle(s.raw(), f, m.id))}; } };
struct FlipNormal { static Hittable new_(Scene& s, Hittable h) { return {s.chk(rt_flip_normal(s.raw(), h.id))}; } };
struct Translate { static Hittable new_(Scene& s, Hittable h, Vec3 off) { return {s.chk(rt_translate(s.raw(), h.id, off.e))}; } };
struct Rotate { static Hittable new_(Scene& s, Axis a, Hittable h, double angle) { return {s.chk(rt_rotate(s.raw(), (int)a, h.id, angle))}; } };
struct ConstantMedium { static Hittable new_(Scene& s, Hittable boundary, double density, Texture t) { return {s.chk(rt_constant_medium(s.raw(), boundary.id, density, t.id))}; } };

class HittableList {                                    // src/hit.rs:46-56
public:
    explicit HittableList(Scene& s) : s_(s), h_{s.chk(rt_list_create(s.raw()))} {}
    void push(Hittable h) { s_.chk(rt_list_push(s_.raw(), h_.id, h.id)); }
    operator Hittable() const { return h_; }
    Hittable handle() const { return h_; }
private:
    Scene& s_; Hittable h_;
};

struct BVH {                                            // src/bvh.rs:18
    static Hittable new_(Scene& s, const std::vector<Hittable>& hit, double t0, double t1) {
        std::vector<int> ids; for (auto h : hit) ids.push_back(h.id);
        return {s.chk(rt_bvh(s.raw(), ids.data(), (uint32_t)ids.size(), t0, t1))};
    }
    static Hittable of_list(Scene& s, Hittable list, double t0, double t1) { return {s.chk(rt_bvh_of_list(s.raw(), list.id, t0, t1))}; }   // BVH::new(obj.tris.list, ..), main.rs:442
};

struct Mesh {                                           // src/mesh.rs
    Hittable tris;                                      // the `tris` HittableList
    static Mesh new_(Scene& s, const std::vector<Vec3>& positions, const std::vector<uint32_t>& indices, Material m) {
        std::vector<double> p; for (auto& v : positions) { p.push_back(v.e[0]); p.push_back(v.e[1]); p.push_back(v.e[2]); }
        return Mesh{{s.chk(rt_mesh(s.raw(), p.data(), (uint32_t)positions.size(), indices.data(), (uint32_t)indices.size(), m.id))}};
    }
    // Mesh::load_obj(path, offset, scale, material) -> Result<Mesh, String>, src/mesh.rs:33-61 (throws on failure)
    static Mesh load_obj(Scene& s, const std::string& path, Vec3 offset, double scale, Material m) {
        return Mesh{{s.chk(rt_mesh_load_obj(s.raw(), path.c_str(), offset.e, scale, m.id))}};
    }
};

struct Camera {                                         // src/camera.rs:19
    rt_camera c;
    static Camera new_(Point3 lookfrom, Point3 lookat, Vec3 vup, double vfov, double aspect_ratio, double aperture,
                       double focus_dist, double time0, double time1) {
        Camera k;
        for (int i = 0; i < 3; i++) { k.c.lookfrom[i] = lookfrom.e[i]; k.c.lookat[i] = lookat.e[i]; k.c.vup[i] = vup.e[i]; }
        k.c.vfov = vfov; k.c.aspect = aspect_ratio; k.c.aperture = aperture; k.c.focus_dist = focus_dist; k.c.time0 = time0; k.c.time1 = time1;
        return k;
    }
};

// The render loop of src/main.rs:772-833 as one call: per-pixel sums of ray_color in output order.
inline std::vector<double> render(Scene& s, const Camera& cam, Color background, uint32_t W, uint32_t H, uint32_t samples_per_pixel,
                                  uint32_t max_depth, uint64_t seed = 0x5EED, uint32_t flags = RT_F64) {
    std::vector<double> out((size_t)W * H * 3);
    if (rt_render(s.raw(), &cam.c, background.e, W, H, samples_per_pixel, max_depth, seed, flags, out.data()) != 0) throw Error(rt_last_error());
    return out;
}
// The same loop on several GPUs of this node (device_mask: bit d = HIP device d, 0 = all visible): tiles dealt round-robin, one RCCL
// gather to the first device, un-permuted there (rt_render_multi).
inline std::vector<double> render_multi(Scene& s, const Camera& cam, Color background, uint32_t W, uint32_t H, uint32_t samples_per_pixel,
                                        uint32_t max_depth, uint32_t device_mask, uint64_t seed = 0x5EED, uint32_t flags = RT_F64, uint32_t tile_px = 0) {
    std::vector<double> out((size_t)W * H * 3);
    if (rt_render_multi(s.raw(), &cam.c, background.e, W, H, samples_per_pixel, max_depth, seed, flags, device_mask, tile_px, out.data()) != 0) throw Error(rt_last_error());
    return out;
}
// main.rs:767-769,832
inline void write_ppm(const char* path, const std::vector<double>& rgb_sum, uint32_t W, uint32_t H, uint64_t spp) {
    if (rt_write_ppm(path, rgb_sum.data(), W, H, spp) != 0) throw Error(rt_last_error());
}

} // namespace rtr

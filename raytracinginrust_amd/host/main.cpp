// host/main.cpp — the reference's `main` (src/main.rs:577-836) with its render loop replaced by ONE call into
// librt_amd.so.  Scene functions, cameras, the `Scene` enum and the PPM-on-stdout contract follow main.rs; the
// reference is Rust, this environment has no Rust toolchain, so the host is C++ over include/raytracinginrust.hpp
// (same constructor names and argument order).  Usage mirrors `cargo run --release > image.ppm` (README.md:4):
//
//     rtrender [--scene cornell|random|final|teapot|two_sphere|two_perlin|earth|light_room|smoke|progress] [--width W] [--height H] [--spp N] [--depth D]
//              [--seed S] [--obj teapot.obj] [--earth earth.ppm] [--f32] [--fast-bvh] [--gpus N] > image.ppm
//
// The reference hard-codes its settings as consts (main.rs:579-583, :623); they are flags here.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>
#include <string>
#include "../../include/raytracinginrust.hpp"

using namespace rtr;

static const uint32_t STREAM_RANDOM_SCENE = 0, STREAM_FINAL_SCENE = 1, STREAM_TWO_PERLIN = 2;

// src/main.rs:212-227
static void two_spehre(Scene& s) {
    HittableList world(s);
    Material top_mat = Lambertian::new_(s, CheckTexture::new_(s, ConstantTexture::new_(s, Color(1.0, 1.0, 1.0)), ConstantTexture::new_(s, Color(0.3, 0.3, 1.0))));
    Material bottom_mat = Lambertian::new_(s, CheckTexture::new_(s, ConstantTexture::new_(s, Color(1.0, 1.0, 1.0)), ConstantTexture::new_(s, Color(0.3, 0.3, 1.0))));
    world.push(Sphere::new_(s, Point3(0.0, 10.0, 0.0), 10.0, top_mat));
    world.push(Sphere::new_(s, Point3(0.0, -10.0, 0.0), 10.0, bottom_mat));
    s.set(world, {});
}
// src/main.rs:229-245
static void two_perlin_sphere(Scene& s, uint64_t seed) {
    Rng rng(seed, STREAM_TWO_PERLIN);
    HittableList world(s);
    Material top_mat = Lambertian::new_(s, NoiseTexture::new_(s, 2.0, rng));
    Material bottom_mat = Lambertian::new_(s, NoiseTexture::new_(s, 2.0, rng));
    world.push(Sphere::new_(s, Point3(1000.0, 2.0, 1000.0), 2.0, top_mat));
    world.push(Sphere::new_(s, Point3(1000.0, -1000.0, 1000.0), 1000.0, bottom_mat));
    s.set(world, {});
}
// src/main.rs:247-255
static void earth(Scene& s, const std::vector<uint8_t>& data, uint32_t w, uint32_t h) {
    s.set(Sphere::new_(s, Vec3(0.0, 0.0, 0.0), 2.0, Lambertian::new_(s, ImageTexture::new_(s, data, w, h))), {});
}
// src/main.rs:257-276
static void light_room(Scene& s) {
    HittableList world(s);
    Material bottom_mat = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.7, 0.7, 0.7)));
    Material top_mat = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.0, 0.1843, 0.6549)));
    Material emitted = DiffuseLight::new_(s, ConstantTexture::new_(s, Color(4.0, 4.0, 4.0)));
    world.push(Sphere::new_(s, Point3(0.0, -1000.0, 0.0), 1000.0, bottom_mat));
    world.push(Sphere::new_(s, Point3(0.0, 2.0, 0.0), 2.0, top_mat));
    Hittable plane = AARect::new_(s, Plane::XY, 3.0, 5.0, 1.0, 3.0, -2.0, emitted);
    world.push(plane);
    s.set(world, {plane});
}
// src/main.rs:313-346
static void cornell_box_with_smoke(Scene& s) {
    Material red = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.65, 0.05, 0.05)));
    Material white = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.73, 0.73, 0.73)));
    Material green = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.12, 0.45, 0.15)));
    Material light = DiffuseLight::new_(s, ConstantTexture::new_(s, Color(15.0, 15.0, 15.0)));
    Hittable rect_light = FlipNormal::new_(s, AARect::new_(s, Plane::XZ, 213.0, 343.0, 227.0, 332.0, 554.0, light));
    HittableList world(s);
    world.push(AARect::new_(s, Plane::YZ, 0.0, 555.0, 0.0, 555.0, 555.0, green));
    world.push(AARect::new_(s, Plane::YZ, 0.0, 555.0, 0.0, 555.0, 0.0, red));
    world.push(rect_light);
    world.push(AARect::new_(s, Plane::XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white));
    world.push(AARect::new_(s, Plane::XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white));
    world.push(AARect::new_(s, Plane::XY, 0.0, 555.0, 0.0, 555.0, 555.0, white));
    Hittable box1 = Translate::new_(s, Rotate::new_(s, Axis::Y, Cube::new_(s, Point3(0.0, 0.0, 0.0), Point3(165.0, 165.0, 165.0), white), -18.0), Vec3(130.0, 0.0, 65.0));
    Hittable box2 = Translate::new_(s, Rotate::new_(s, Axis::Y, Cube::new_(s, Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), white), 15.0), Vec3(265.0, 0.0, 295.0));
    world.push(ConstantMedium::new_(s, box1, 0.01, ConstantTexture::new_(s, Color(1.0, 1.0, 1.0))));
    world.push(ConstantMedium::new_(s, box2, 0.01, ConstantTexture::new_(s, Color(0.0, 0.0, 0.0))));
    s.set(world, {rect_light});
}
// src/main.rs:515-562: the body is entirely commented out
static void progress_showcase(Scene& s) { HittableList world(s); s.set(world, {}); }

// src/main.rs:153-210
static void random_scene(Scene& s, uint64_t seed) {
    Rng rng(seed, STREAM_RANDOM_SCENE);
    std::vector<Hittable> world;
    Material ground_mat = Lambertian::new_(s, CheckTexture::new_(s, ConstantTexture::new_(s, Color(1.0, 1.0, 1.0)), ConstantTexture::new_(s, Color(0.3, 0.3, 1.0))));
    world.push_back(Sphere::new_(s, Point3(0.0, -1000.0, 0.0), 1000.0, ground_mat));
    for (int a = -11; a <= 11; a++) for (int b = -11; b <= 11; b++) {
        double choose_mat = rng.gen_f64();
        double cx = (double)a + rng.gen_range(0.0, 0.9);
        double cz = (double)b + rng.gen_range(0.0, 0.9);
        Point3 center(cx, 0.2, cz);
        if (choose_mat < 0.8) {
            Color c1 = rng.color_random(0.0, 1.0); Color c2 = rng.color_random(0.0, 1.0);
            Material m = Lambertian::new_(s, ConstantTexture::new_(s, c1 * c2));
            Point3 center1 = center + Vec3(0.0, rng.gen_range(0.0, 0.01), 0.0);
            world.push_back(MovingSphere::new_(s, center, center1, 0.0, 1.0, 0.2, m));
        } else if (choose_mat < 0.95) {
            Color albedo = rng.color_random(0.4, 1.0);
            double fuzz = rng.gen_range(0.0, 0.5);
            world.push_back(Sphere::new_(s, center, 0.2, Metal::new_(s, albedo, fuzz)));
        } else {
            world.push_back(Sphere::new_(s, center, 0.2, Dielectric::new_(s, 1.5)));
        }
    }
    world.push_back(Sphere::new_(s, Point3(0.0, 1.0, 0.0), 1.0, Dielectric::new_(s, 1.5)));
    world.push_back(Sphere::new_(s, Point3(-4.0, 1.0, 0.0), 1.0, Lambertian::new_(s, ConstantTexture::new_(s, Color(0.4, 0.2, 0.1)))));
    world.push_back(Sphere::new_(s, Point3(4.0, 1.0, 0.0), 1.0, Metal::new_(s, Color(0.7, 0.6, 0.5), 0.0)));
    s.set(BVH::new_(s, world, 0.0, 1.0), {});      // `lights` is empty in the reference (main.rs:207): cosine-only fallback, DESIGN.md D2
}

// src/main.rs:278-311
static void cornell_box(Scene& s) {
    Material red = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.65, 0.05, 0.05)));
    Material white = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.73, 0.73, 0.73)));
    Material green = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.12, 0.45, 0.15)));
    Material metal = Metal::new_(s, Color(0.8, 0.85, 0.88), 0.0);
    Material light = DiffuseLight::new_(s, ConstantTexture::new_(s, Color(15.0, 15.0, 15.0)));
    Hittable rect_light = FlipNormal::new_(s, AARect::new_(s, Plane::XZ, 213.0, 343.0, 227.0, 332.0, 554.0, light));
    HittableList world(s);
    world.push(AARect::new_(s, Plane::YZ, 0.0, 555.0, 0.0, 555.0, 555.0, green));
    world.push(AARect::new_(s, Plane::YZ, 0.0, 555.0, 0.0, 555.0, 0.0, red));
    world.push(rect_light);
    world.push(AARect::new_(s, Plane::XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white));
    world.push(AARect::new_(s, Plane::XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white));
    world.push(AARect::new_(s, Plane::XY, 0.0, 555.0, 0.0, 555.0, 555.0, white));
    world.push(Translate::new_(s, Rotate::new_(s, Axis::Y, Cube::new_(s, Point3(0.0, 0.0, 0.0), Point3(165.0, 165.0, 165.0), white), -18.0), Vec3(130.0, 0.0, 65.0)));
    world.push(Translate::new_(s, Rotate::new_(s, Axis::Y, Cube::new_(s, Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), metal), 15.0), Vec3(265.0, 0.0, 295.0)));
    s.set(world, {rect_light});
}

// src/main.rs:348-451 as committed, with teapot.obj standing in for the absent Venus.obj (placement: DESIGN.md)
static void cornell_test(Scene& s, const std::string& obj_path) {
    Material white = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.73, 0.73, 0.73)));
    Material desire = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.922, 0.238, 0.331)));
    Material safety_orange = Lambertian::new_(s, ConstantTexture::new_(s, Color(1.000, 0.471, 0.0)));
    Material color_80cf00 = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.502, 0.812, 0.002)));
    Material light0 = DiffuseLight::new_(s, ConstantTexture::new_(s, Color(1.0, 1.0, 0.88) * 2.2));
    HittableList world(s);
    world.push(AARect::new_(s, Plane::YZ, 0.0, 555.0, 0.0, 555.0, 555.0, desire));
    world.push(AARect::new_(s, Plane::YZ, 0.0, 555.0, 0.0, 555.0, 0.0, safety_orange));
    world.push(AARect::new_(s, Plane::XZ, 0.0, 555.0, 0.0, 555.0, 0.0, white));
    world.push(AARect::new_(s, Plane::XZ, 0.0, 555.0, 0.0, 555.0, 555.0, white));
    world.push(AARect::new_(s, Plane::XY, 0.0, 555.0, 0.0, 555.0, 555.0, white));
    Hittable rect_light0 = FlipNormal::new_(s, AARect::new_(s, Plane::XZ, 128.0, 428.0, 115.0, 270.0, 554.0, light0));
    Mesh obj = Mesh::load_obj(s, obj_path, Vec3(268.0, 340.0, 258.0), 1.5, color_80cf00);
    world.push(rect_light0);
    world.push(BVH::of_list(s, obj.tris, 0.0, 1.0));
    s.set(world, {rect_light0});
}

// src/main.rs:453-513
static void final_scene(Scene& s, uint64_t seed, const std::vector<uint8_t>& earth, uint32_t ew, uint32_t eh) {
    Rng rng(seed, STREAM_FINAL_SCENE);
    HittableList world(s);
    Material ground = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.48, 0.83, 0.53)));
    std::vector<Hittable> box_list1;
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; i++) for (int j = 0; j < boxes_per_side; j++) {
        double w = 100.0;
        double x0 = -1000.0 + (double)i * w, z0 = -1000.0 + (double)j * w, y0 = 0.0;
        double x1 = x0 + w, y1 = 100.0 * (rng.gen_f64() + 0.01), z1 = z0 + w;
        box_list1.push_back(Cube::new_(s, Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
    }
    world.push(BVH::new_(s, box_list1, 0.0, 1.0));
    Material light = DiffuseLight::new_(s, ConstantTexture::new_(s, Color(7.0, 7.0, 7.0)));
    Hittable rect_light = FlipNormal::new_(s, AARect::new_(s, Plane::XZ, 147.0, 412.0, 123.0, 423.0, 554.0, light));
    world.push(rect_light);
    Point3 center(400.0, 400.0, 200.0);
    world.push(MovingSphere::new_(s, center, center + Point3(30.0, 0.0, 0.0), 0.0, 1.0, 50.0, Lambertian::new_(s, ConstantTexture::new_(s, Color(0.7, 0.3, 0.1)))));
    world.push(Sphere::new_(s, Point3(260.0, 150.0, 45.0), 50.0, Dielectric::new_(s, 1.5)));
    world.push(Sphere::new_(s, Point3(0.0, 150.0, 145.0), 50.0, Metal::new_(s, Color(0.8, 0.8, 0.9), 1.0)));
    Hittable boundary = Sphere::new_(s, Point3(360.0, 150.0, 145.0), 70.0, Dielectric::new_(s, 1.5));
    world.push(boundary);
    world.push(ConstantMedium::new_(s, boundary, 0.2, ConstantTexture::new_(s, Color(0.2, 0.4, 0.9))));
    boundary = Sphere::new_(s, Point3(0.0, 0.0, 0.0), 5000.0, Dielectric::new_(s, 1.5));
    world.push(ConstantMedium::new_(s, boundary, 0.0001, ConstantTexture::new_(s, Color(1.0, 1.0, 1.0))));
    world.push(Sphere::new_(s, Point3(400.0, 200.0, 400.0), 100.0, Lambertian::new_(s, ImageTexture::new_(s, earth, ew, eh))));
    world.push(Sphere::new_(s, Point3(220.0, 280.0, 300.0), 80.0, Lambertian::new_(s, NoiseTexture::new_(s, 0.1, rng))));
    Material white = Lambertian::new_(s, ConstantTexture::new_(s, Color(0.73, 0.73, 0.73)));
    std::vector<Hittable> box_list2;
    for (int k = 0; k < 1000; k++) {
        double x = 165.0 * rng.gen_f64(), y = 165.0 * rng.gen_f64(), z = 165.0 * rng.gen_f64();
        box_list2.push_back(Sphere::new_(s, Point3(x, y, z), 10.0, white));
    }
    world.push(Translate::new_(s, Rotate::new_(s, Axis::Y, BVH::new_(s, box_list2, 0.0, 0.1), 15.0), Point3(-100.0, 270.0, 395.0)));
    s.set(world, {rect_light});
}

// binary PPM (P6) reader for the decoded earthmap texels (image::open(..).to_rgb8() in the reference, main.rs:491)
static bool read_p6(const std::string& path, std::vector<uint8_t>& data, uint32_t& w, uint32_t& h) {
    std::ifstream f(path, std::ios::binary);
    std::string magic; int maxv = 0;
    if (!(f >> magic >> w >> h >> maxv) || magic != "P6" || maxv != 255) return false;
    f.get();
    data.resize((size_t)w * h * 3);
    f.read((char*)data.data(), (std::streamsize)data.size());
    return (bool)f;
}

// image::open(path).to_rgb8() (main.rs:248,491): a baseline JPEG goes through the library's decoder, a P6 file is read directly
static bool read_image(const std::string& path, std::vector<uint8_t>& data, uint32_t& w, uint32_t& h) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (bytes.size() > 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) {
        uint8_t* rgb = rt_decode_jpeg_rgb8(bytes.data(), bytes.size(), &w, &h);
        if (!rgb) { std::fprintf(stderr, "%s\n", rt_last_error()); return false; }
        data.assign(rgb, rgb + (size_t)w * h * 3);
        rt_free(rgb);
        return true;
    }
    return read_p6(path, data, w, h);
}

enum class SceneKind { Random, TwoSphere, TwoPerlinSphere, Earth, LightRoom, CornellBox, CornellSmoke, CornellTest, FinalScene, Progress };   // src/main.rs:564-575

int main(int argc, char** argv) {
    SceneKind scene = SceneKind::CornellBox;
    uint32_t image_width = 500, image_height = 500, samples_per_pixel = 800, max_depth = 100;    // main.rs:579-583
    uint64_t seed = 0x5EED; uint32_t flags = RT_F64; bool fast_bvh = false;
    int gpus = 1;                                                           // --gpus N: the first N devices (0 = all) through rt_render_multi
    std::string obj_path = "teapot.obj", earth_path = "earthmap.jpg";       // the reference's asset names (main.rs:248,491)
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--scene") {
            std::string v = next();
            scene = v == "random" ? SceneKind::Random : v == "final" ? SceneKind::FinalScene : v == "teapot" ? SceneKind::CornellTest
                  : v == "two_sphere" ? SceneKind::TwoSphere : v == "two_perlin" ? SceneKind::TwoPerlinSphere : v == "earth" ? SceneKind::Earth
                  : v == "light_room" ? SceneKind::LightRoom : v == "smoke" ? SceneKind::CornellSmoke : v == "progress" ? SceneKind::Progress
                  : SceneKind::CornellBox;
        }
        else if (a == "--isotropic-scatter") flags |= RT_ISOTROPIC_SCATTER;
        else if (a == "--width") image_width = (uint32_t)std::atoi(next());
        else if (a == "--height") image_height = (uint32_t)std::atoi(next());
        else if (a == "--spp") samples_per_pixel = (uint32_t)std::atoi(next());
        else if (a == "--depth") max_depth = (uint32_t)std::atoi(next());
        else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
        else if (a == "--obj") obj_path = next();
        else if (a == "--earth") earth_path = next();
        else if (a == "--f32") flags |= RT_F32;
        else if (a == "--gpus") gpus = std::atoi(next());
        else if (a == "--collective") flags |= RT_MULTI_COLLECTIVE;         // with --gpus 1: run the RCCL gather anyway
        else if (a == "--fast-bvh") { fast_bvh = true; flags |= RT_NEAR_FIRST_BVH; }      // opt-in, not the reference's tree / visiting order
        else { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    const double aspect_ratio = (double)image_width / (double)image_height;
    try {
        Scene s;
        if (fast_bvh) s.set_bvh_builder(RT_BVH_SAH);
        Color background; Camera camera;
        Vec3 vup(0.0, 1.0, 0.0);
        switch (scene) {                                                    // main.rs:624-765
        case SceneKind::Random:
            random_scene(s, seed);
            background = Color(0.7, 0.8, 1.0);
            camera = Camera::new_(Point3(13.0, 2.0, 3.0), Point3(0.0, 0.0, 0.0), vup, 20.0, aspect_ratio, 0.1, 10.0, 0.0, 1.0);
            break;
        case SceneKind::TwoSphere:
            two_spehre(s);
            background = Color(0.7, 0.8, 1.0);
            camera = Camera::new_(Point3(13.0, 2.0, 3.0), Point3(0.0, 0.0, 0.0), vup, 20.0, aspect_ratio, 0.0, 10.0, 0.0, 1.0);
            break;
        case SceneKind::TwoPerlinSphere:
            two_perlin_sphere(s, seed);
            background = Color(0.7, 0.8, 1.0);
            camera = Camera::new_(Point3(1013.0, 2.0, 1003.0), Point3(1000.0, 0.0, 1000.0), vup, 20.0, aspect_ratio, 0.0, 10.0, 0.0, 1.0);
            break;
        case SceneKind::Earth: {
            std::vector<uint8_t> tex; uint32_t ew = 0, eh = 0;
            if (!read_image(earth_path, tex, ew, eh)) throw Error("image not found: " + earth_path);      // main.rs:248
            earth(s, tex, ew, eh);
            background = Color(0.7, 0.8, 1.0);
            camera = Camera::new_(Point3(13.0, 2.0, 3.0), Point3(0.0, 0.0, 0.0), vup, 20.0, aspect_ratio, 0.1, 10.0, 0.0, 1.0);
            break;
        }
        case SceneKind::LightRoom:
            light_room(s);
            background = Color(0.0, 0.0, 0.0);
            camera = Camera::new_(Point3(26.0, 3.0, 6.0), Point3(0.0, 2.0, 0.0), vup, 20.0, aspect_ratio, 0.0, 10.0, 0.0, 1.0);
            break;
        case SceneKind::CornellSmoke:
            cornell_box_with_smoke(s);
            background = Color(0.0, 0.0, 0.0);
            camera = Camera::new_(Point3(278.0, 278.0, -800.0), Point3(278.0, 278.0, 0.0), vup, 40.0, aspect_ratio, 0.05, 10.0, 0.0, 1.0);
            break;
        case SceneKind::Progress:
            progress_showcase(s);
            background = Color(0.0, 0.0, 0.0);
            camera = Camera::new_(Point3(-3.3, 6.8, -9.8), Point3(0.0, 1.0, 0.0), vup, 40.0, aspect_ratio, 0.2, 12.0, 0.0, 1.0);
            break;
        case SceneKind::CornellBox:
            cornell_box(s);
            background = Color(0.0, 0.0, 0.0);
            camera = Camera::new_(Point3(278.0, 278.0, -800.0), Point3(278.0, 278.0, 0.0), vup, 40.0, aspect_ratio, 0.05, 10.0, 0.0, 1.0);
            break;
        case SceneKind::CornellTest:
            cornell_test(s, obj_path);
            background = Color(0.0, 0.0, 0.0);
            camera = Camera::new_(Point3(199.0, 439.0, -200.0), Point3(278.0, 375.0, 258.0), vup, 30.0, aspect_ratio, 0.01, 10.0, 0.0, 1.0);
            break;
        case SceneKind::FinalScene: {
            std::vector<uint8_t> earth; uint32_t ew = 0, eh = 0;
            if (!read_image(earth_path, earth, ew, eh)) throw Error("image not found: " + earth_path);    // main.rs:491 .expect("image not found")
            final_scene(s, seed, earth, ew, eh);
            background = Color(0.0, 0.0, 0.0);
            camera = Camera::new_(Point3(478.0, 278.0, -600.0), Point3(278.0, 278.0, 0.0), vup, 40.0, aspect_ratio, 0.01, 10.0, 0.0, 1.0);
            break;
        }
        }
        // main.rs:772-833: the whole loop nest is this one call
        std::vector<double> pixel_sums;
        if (gpus == 1 && !(flags & RT_MULTI_COLLECTIVE)) {
            pixel_sums = render(s, camera, background, image_width, image_height, samples_per_pixel, max_depth, seed, flags);
        } else {                                                            // every GPU of the node from this one process
            if (gpus < 0 || gpus > 32) throw Error("--gpus must be 0 (all) .. 32");
            const uint32_t mask = gpus == 0 ? 0u : (gpus == 32 ? 0xFFFFFFFFu : ((1u << gpus) - 1u));
            pixel_sums = render_multi(s, camera, background, image_width, image_height, samples_per_pixel, max_depth, mask, seed, flags);
            double ms[4]; rt_last_multi_ms(s.raw(), ms);
            std::fprintf(stderr, "rt_render_multi: slowest kernel %.2f ms, gather %.3f ms, un-permute %.3f ms, call %.2f ms\n", ms[0], ms[1], ms[2], ms[3]);
        }
        write_ppm("-", pixel_sums, image_width, image_height, samples_per_pixel);              // main.rs:767-769,832
        std::fprintf(stderr, "Done.\n");                                                      // main.rs:835
    } catch (const Error& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

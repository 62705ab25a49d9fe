// A C++17 caller of include/pbrt_hip.hpp — the reference's trait surface for the hot path (Primitive, BVHAccel, Scene,
// Integrator, PathIntegrator, Film, PerspectiveCamera) over the C ABI. The same open box as
// examples/render_box.c, written the way a pbrt-rs host would write it:
//     let aggregate = BVHAccel::new(primitives, 4, SplitMethod::SAH);  let scene = Scene::new(aggregate, lights);
//     let mut integrator = PathIntegrator::new(5, camera, sampler, bounds, 1.0, "power");  integrator.render(&scene);
//     camera.film.write_image();
// then Primitive::intersect / intersect_p and Integrator::li called ray by ray, as the reference's render loop calls them.
//
//   g++ -std=c++17 -Wall -Iinclude examples/render_box.cpp -o render_box_cpp -Lpbrt-rs_amd/pbrt_hip -lpbrt_hip -Wl,-rpath,$PWD/pbrt-rs_amd/pbrt_hip
//   ./render_box_cpp out.png [width height spp]
#include <cstdio>
#include <algorithm>
#include <array>
#include <cstdlib>

#include "pbrt_hip.hpp"

using namespace pbrt;

static void add_quad(TriangleMesh& m, const float a[3], const float b[3], const float c[3], const float d[3], int material, bool emitter) {
    const int v0 = m.n_vertices();
    for (const float* v : {a, b, c, d}) m.p.insert(m.p.end(), v, v + 3);
    for (const auto& tri : {std::array<int, 3>{0, 1, 2}, std::array<int, 3>{0, 2, 3}}) {
        for (int k : tri) m.vertex_indices.push_back(v0 + k);
        m.material.push_back(material);
        if (emitter) {  // DiffuseAreaLight on this triangle (src/lights/diffuse.rs)
            PbrtLight l{};
            l.type = PBRT_LIGHT_DIFFUSE_AREA;
            l.L[0] = l.L[1] = l.L[2] = 17.0f;
            l.prim = m.n_triangles() - 1;
            l.two_sided = 1;
            l.n_samples = 1;
            m.area_light.push_back((int32_t)m.lights.size());
            m.lights.push_back(l);
        } else {
            m.area_light.push_back(-1);
        }
    }
}

int main(int argc, char** argv) try {
    const char* out_path = argc > 1 ? argv[1] : "render_box_cpp.png";
    const int width = argc > 4 ? std::atoi(argv[2]) : 128, height = argc > 4 ? std::atoi(argv[3]) : 128, spp = argc > 4 ? std::atoi(argv[4]) : 16;
    TriangleMesh mesh;
    const float p000[3] = {-1, -1, -1}, p100[3] = {1, -1, -1}, p010[3] = {-1, 1, -1}, p110[3] = {1, 1, -1};
    const float p001[3] = {-1, -1, 1}, p101[3] = {1, -1, 1}, p011[3] = {-1, 1, 1}, p111[3] = {1, 1, 1};
    add_quad(mesh, p000, p100, p101, p001, 0, false);  // floor   y = -1
    add_quad(mesh, p010, p011, p111, p110, 0, false);  // ceiling y = +1
    add_quad(mesh, p001, p101, p111, p011, 0, false);  // back    z = +1
    add_quad(mesh, p000, p001, p011, p010, 1, false);  // left    x = -1
    add_quad(mesh, p100, p110, p111, p101, 2, false);  // right   x = +1
    const float e0[3] = {-0.3f, 0.99f, -0.3f}, e1[3] = {0.3f, 0.99f, -0.3f}, e2[3] = {0.3f, 0.99f, 0.3f}, e3[3] = {-0.3f, 0.99f, 0.3f};
    add_quad(mesh, e0, e1, e2, e3, 0, true);
    const float kd[3][3] = {{0.73f, 0.73f, 0.73f}, {0.65f, 0.05f, 0.05f}, {0.12f, 0.45f, 0.15f}};
    for (const auto& k : kd) {
        PbrtMaterial m{};
        m.type = PBRT_MAT_MATTE;
        m.kd[0] = k[0], m.kd[1] = k[1], m.kd[2] = k[2];
        m.eta = 1.0f;
        mesh.materials.push_back(m);
    }

    auto ctx = std::make_shared<Context>(0);  // throws pbrt::Error without a GPU: there is no CPU fallback
    auto aggregate = std::make_shared<BVHAccel>(ctx, mesh, 4, SplitMethod::SAH);
    std::printf("BVH: %d nodes over %d triangles\n", (int)aggregate->n_nodes(), (int)mesh.n_triangles());
    Scene scene(aggregate);

    auto film = std::make_shared<Film>(width, height);
    auto camera = std::make_shared<PerspectiveCamera>(Point3f{0, 0, -3.4f}, Point3f{0, 0, 0}, Vector3f{0, 1, 0}, 40.0f, film);
    PathIntegrator integrator(5, camera, RandomSampler{spp, 0}, Bounds2i(), 1.0f, LightSampleStrategy::Power);
    integrator.render(scene);
    const std::vector<float> rgb = film->rgb();
    double mean = 0.0;
    for (float v : rgb) mean += v;
    mean /= (double)rgb.size();
    std::printf("%llu camera samples, %llu closest-hit + %llu shadow rays in %.2f ms; mean RGB %.4f\n", (unsigned long long)integrator.stats.camera_samples,
                (unsigned long long)integrator.stats.rays_closest, (unsigned long long)integrator.stats.rays_shadow, integrator.stats.ms, mean);
    film->write_image(out_path);

    // Primitive::intersect (mutates ray.t_max) / intersect_p, one ray at a time
    Ray down;
    down.d = {0, -1, 0};
    SurfaceInteraction isect;
    const bool hit = scene.intersect(down, &isect);
    std::printf("intersect: hit %d t %.6f ray.t_max %.6f primitive %d barycentrics %.4f %.4f %.4f\n", (int)hit, isect.t, down.t_max, isect.primitive,
                isect.b0, isect.b1, isect.b2);
    Ray shadow = down;  // t_max is now the hit distance: a shadow ray stopping just short of the floor sees nothing
    shadow.t_max = 0.999f;
    Ray out_of_box;
    out_of_box.d = {0, 0, -1};  // the box is open towards -z
    std::printf("intersect_p: blocked-short %d towards-floor %d through-the-opening %d\n", (int)scene.intersect_p(shadow), (int)scene.intersect_p(down),
                (int)scene.intersect_p(out_of_box));
    // Integrator::li for one camera-like ray and one stream: looking straight at the emitter returns its radiance
    Ray up_ray;
    up_ray.d = {0, 1, 0};
    const Spectrum le = integrator.li(up_ray, scene, 12345);
    std::printf("li towards the emitter: %.4f %.4f %.4f\n", le.c[0], le.c[1], le.c[2]);
    // TransformedPrimitive instances of the same box under a top-level BVHAccel (primitive.rs:105-159): a copy moved to x = +5
    // and one turned a quarter about y at x = -5. From the centre of the moved copy straight down: its floor at t = 1, instance 0.
    {
        TriangleMesh object = mesh;
        object.lights.clear();  // an instanced primitive cannot be an area light: an environment lights this scene
        std::fill(object.area_light.begin(), object.area_light.end(), -1);
        PbrtLight env{};
        env.type = PBRT_LIGHT_INFINITE;
        env.L[0] = env.L[1] = env.L[2] = 1.0f;
        env.prim = -1;
        env.n_samples = 1;
        object.lights.push_back(env);
        const double moved[16] = {1, 0, 0, 5, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, turned[16] = {0, 0, 1, -5, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 0, 1};
        auto instanced = std::make_shared<BVHAccel>(ctx, object, std::vector<PbrtInstance>{TransformedPrimitive(moved), TransformedPrimitive(turned, 1)});
        Scene two(instanced);
        Ray r;
        r.o = {5, 0, 0};
        r.d = {0, -1, 0};
        SurfaceInteraction si;
        const bool h2 = two.intersect(r, &si);
        Ray between;  // from between the two copies towards +z: nothing there
        between.d = {0, 0, 1};
        std::printf("instanced: hit %d t %.6f instance %d world bound x [%.1f, %.1f] miss-between %d\n", (int)h2, si.t, si.instance, two.world_bound().min.x,
                    two.world_bound().max.x, (int)two.intersect_p(between));
    }
    try {
        aggregate->get_material();
    } catch (const Error& e) {
        std::printf("aggregate.get_material(): %s\n", e.what());
    }
    return 0;
} catch (const pbrt::Error& e) {
    std::fprintf(stderr, "pbrt::Error (%d): %s\n", e.status, e.what());
    return 3;
}

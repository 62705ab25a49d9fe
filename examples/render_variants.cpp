// The "next" rows of SURVEY.md 8(f) through include/pbrt_hip.hpp, the way a pbrt-rs host would name them: the reconstruction
// filters of src/filters/, the samplers of src/samplers/ (stratified, (0,2)-sequence, Halton), the orthographic and environment
// cameras, the four integrators, and a frame rendered as the two halves two processes would render (tile_rank of tile_world).
// One line per variant: what was traced and what the film holds, so that a test can set the same job up through another
// binding of the same C ABI and compare (tests/test_cpp_example.py).
//
//   g++ -std=c++17 -Wall -Iinclude examples/render_variants.cpp -o render_variants -Lpbrt-rs_amd/pbrt_hip -lpbrt_hip -Wl,-rpath,$PWD/pbrt-rs_amd/pbrt_hip
//   ./render_variants [width height]
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pbrt_hip.hpp"

using namespace pbrt;

static void add_quad(TriangleMesh& m, const float a[3], const float b[3], const float c[3], const float d[3], int material, bool emitter) {
    const int v0 = m.n_vertices();
    for (const float* v : {a, b, c, d}) m.p.insert(m.p.end(), v, v + 3);
    for (const auto& tri : {std::array<int, 3>{0, 1, 2}, std::array<int, 3>{0, 2, 3}}) {
        for (int k : tri) m.vertex_indices.push_back(v0 + k);
        m.material.push_back(material);
        if (emitter) {
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

static TriangleMesh open_box() {  // the box of examples/render_box.c: open towards -z, an emitter under the ceiling
    TriangleMesh mesh;
    const float p000[3] = {-1, -1, -1}, p100[3] = {1, -1, -1}, p010[3] = {-1, 1, -1}, p110[3] = {1, 1, -1};
    const float p001[3] = {-1, -1, 1}, p101[3] = {1, -1, 1}, p011[3] = {-1, 1, 1}, p111[3] = {1, 1, 1};
    add_quad(mesh, p000, p100, p101, p001, 0, false);
    add_quad(mesh, p010, p011, p111, p110, 0, false);
    add_quad(mesh, p001, p101, p111, p011, 0, false);
    add_quad(mesh, p000, p001, p011, p010, 1, false);
    add_quad(mesh, p100, p110, p111, p101, 2, false);
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
    return mesh;
}

// what the film holds: the sum of every xyz value and of the filter weights (film.rs:9-15), in double
static void report(const char* name, const SamplerIntegrator& integrator, const Film& film) {
    double xyz = 0.0, weight = 0.0;
    for (size_t i = 0; i < film.pixels.size(); i += 4) xyz += (double)film.pixels[i] + film.pixels[i + 1] + film.pixels[i + 2], weight += film.pixels[i + 3];
    std::printf("variant %s: %llu camera samples, %llu closest-hit + %llu shadow rays; film xyz %.9e weight %.9e\n", name,
                (unsigned long long)integrator.stats.camera_samples, (unsigned long long)integrator.stats.rays_closest,
                (unsigned long long)integrator.stats.rays_shadow, xyz, weight);
}

static void print_camera(const char* name, const Camera& c) {
    std::printf("camera %s:", name);
    for (float v : c.cam.camera_to_world) std::printf(" %a", v);
    for (float v : c.cam.raster_to_camera) std::printf(" %a", v);
    std::printf("\n");
}

int main(int argc, char** argv) try {
    const int width = argc > 2 ? std::atoi(argv[1]) : 64, height = argc > 2 ? std::atoi(argv[2]) : 48;
    const TriangleMesh mesh = open_box();
    auto ctx = std::make_shared<Context>(0);  // throws pbrt::Error without a GPU: there is no CPU fallback
    Scene scene(std::make_shared<BVHAccel>(ctx, mesh, 4, SplitMethod::SAH));
    const Point3f eye{0, 0, -3.4f}, look{0, 0, 0};
    const Vector3f up{0, 1, 0};

    // every reconstruction filter of src/filters/ under the path integrator, 16 random samples per pixel
    const struct {
        const char* name;
        Filter filter;
    } filters[] = {{"box", BoxFilter()},
                   {"triangle", TriangleFilter(Vector2f{2.0f, 1.5f})},
                   {"gaussian", GaussianFilter(Vector2f{2.0f, 2.0f}, 2.0f)},
                   {"mitchell", MitchellFilter()},
                   {"lanczos", LanczosSincFilter(Vector2f{3.0f, 3.0f}, 3.0f)}};
    for (const auto& f : filters) {
        auto film = std::make_shared<Film>(width, height, f.filter);
        auto camera = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, film);
        PathIntegrator integrator(5, camera, RandomSampler(16, 7), Bounds2i(), 1.0f, LightSampleStrategy::Power);
        integrator.render(scene);
        report(f.name, integrator, *film);
    }

    // the samplers of src/samplers/
    const struct {
        const char* name;
        Sampler sampler;
    } samplers[] = {{"stratified", StratifiedSampler(4, 4, true, 4, 3)}, {"zerotwo", ZeroTwoSequenceSampler(12, 4, 3)}, {"halton", HaltonSampler(16, 3)}};
    for (const auto& sm : samplers) {
        auto film = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, film);
        PathIntegrator integrator(4, camera, sm.sampler);
        integrator.render(scene);
        report(sm.name, integrator, *film);
    }

    // the other cameras (orthographic.rs, environment.rs), under the other integrators
    {
        auto film = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<OrthographicCamera>(eye, look, up, 1.2f, film);
        print_camera("orthographic", *camera);
        DirectLightingIntegrator integrator(LightStrategy::UniformSampleAll, 3, camera, RandomSampler(8, 1));
        integrator.render(scene);
        report("orthographic-direct", integrator, *film);
    }
    {
        auto film = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<EnvironmentCamera>(Point3f{0, 0, 0}, Point3f{0, 0, 1}, up, film);
        print_camera("environment", *camera);
        WhittedIntegrator whitted(3, camera, RandomSampler(4, 2));
        whitted.render(scene);
        report("environment-whitted", whitted, *film);
        auto film_ao = std::make_shared<Film>(width, height);
        auto camera_ao = std::make_shared<EnvironmentCamera>(Point3f{0, 0, 0}, Point3f{0, 0, 1}, up, film_ao);
        AOIntegrator ao(true, 16, camera_ao, RandomSampler(4, 2));
        ao.render(scene);
        report("environment-ao", ao, *film_ao);
    }

    // the other ways a BVHAccel comes to be: HLBVH built on the device (the same tree as the host's, so the same film), a Sphere
    // beside the triangles (sphere.rs), per-vertex shading normals on the mesh (triangle.rs:252-312)
    {
        auto on_device = std::make_shared<BVHAccel>(ctx, mesh, 4, BuildOnDevice{});
        auto on_host = std::make_shared<BVHAccel>(ctx, mesh, 4, SplitMethod::HLBVH);
        std::shared_ptr<Film> films[2];
        int k = 0;
        for (const auto& aggregate : {on_device, on_host}) {
            films[k] = std::make_shared<Film>(width, height);
            auto camera = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, films[k]);
            PathIntegrator integrator(5, camera, RandomSampler(16, 5));
            integrator.render(Scene(aggregate));
            if (k == 0) report("hlbvh-device", integrator, *films[k]);
            ++k;
        }
        std::printf("hlbvh device against host: films %s, world bound y [%.2f, %.2f] / [%.2f, %.2f]\n", films[0]->pixels == films[1]->pixels ? "equal" : "DIFFER",
                    on_device->world_bound().min.y, on_device->world_bound().max.y, on_host->world_bound().min.y, on_host->world_bound().max.y);
    }
    {
        Sphere ball;
        ball.centre = {0.2f, -0.6f, 0.1f}, ball.radius = 0.4f, ball.material = 1;
        Scene with_ball(std::make_shared<BVHAccel>(ctx, mesh, std::vector<Sphere>{ball}));
        auto film = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, film);
        DirectLightingIntegrator integrator(LightStrategy::UniformSampleOne, 3, camera, RandomSampler(8, 9));
        integrator.render(with_ball);
        report("sphere", integrator, *film);
        Ray at_ball;  // from the opening towards the ball's centre: the sphere's near side, primitive n_triangles + 0
        at_ball.o = {0.2f, -0.6f, -3.0f}, at_ball.d = {0, 0, 1};
        SurfaceInteraction si;
        const bool hit = with_ball.intersect(at_ball, &si);
        std::printf("sphere: hit %d t %.4f primitive %d\n", (int)hit, si.t, si.primitive);
    }
    {
        TriangleMesh smooth = mesh;  // shading normals bent towards the middle of the box
        for (int v = 0; v < smooth.n_vertices(); ++v) {
            const float* q = &smooth.p[3 * (size_t)v];
            const float l = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
            for (int k = 0; k < 3; ++k) smooth.n.push_back(-q[k] / l);
        }
        auto film = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, film);
        PathIntegrator integrator(3, camera, RandomSampler(8, 11));
        integrator.render(Scene(std::make_shared<BVHAccel>(ctx, smooth, 4, SplitMethod::SAH)));
        report("vertex-normals", integrator, *film);
    }

    // the general top level: TransformedPrimitives of two different aggregates and world-space triangles (with the area lights) beside them
    {
        TriangleMesh walls;  // object 0: the five walls of the box; object 1: a square blade in the plane y = 0
        walls.p = mesh.p, walls.vertex_indices.assign(mesh.vertex_indices.begin(), mesh.vertex_indices.begin() + 30);
        walls.material.assign(mesh.material.begin(), mesh.material.begin() + 10);
        TriangleMesh blade;
        blade.p = {-0.5f, 0, -0.5f, 0.5f, 0, -0.5f, 0.5f, 0, 0.5f, -0.5f, 0, 0.5f};
        blade.vertex_indices = {0, 1, 2, 0, 2, 3};
        blade.material = {1, 1};
        TriangleMesh world;  // a floor under everything and an emitter above, facing down
        world.p = {-8, -1.25f, -8, 8, -1.25f, -8, 8, -1.25f, 8, -8, -1.25f, 8, -1, 3, -1, 1, 3, -1, 1, 3, 1, -1, 3, 1};
        world.vertex_indices = {0, 2, 1, 0, 3, 2, 4, 5, 6, 4, 6, 7};
        world.material = {0, 0, 0, 0};
        world.area_light = {-1, -1, 0, 1};
        world.materials = mesh.materials;
        for (int t = 2; t < 4; ++t) {
            PbrtLight l{};
            l.type = PBRT_LIGHT_DIFFUSE_AREA, l.L[0] = l.L[1] = l.L[2] = 30.0f, l.prim = t, l.two_sided = 0, l.n_samples = 1;
            world.lights.push_back(l);
        }
        const double moved[16] = {1, 0, 0, 3, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}, turned[16] = {0, 0, 1, -3, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 0, 1},
                     lowered[16] = {1, 0, 0, 0, 0, 1, 0, -0.5, 0, 0, 1, 0, 0, 0, 0, 1};
        auto top = std::make_shared<BVHAccel>(ctx, std::vector<TriangleMesh>{walls, blade},
                                              std::vector<PbrtInstance>{TransformedPrimitive(moved), TransformedPrimitive(turned, 2), TransformedPrimitive(lowered)},
                                              std::vector<int32_t>{0, 0, 1}, world);
        auto film = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<PerspectiveCamera>(Point3f{0, 1.5f, -9}, look, up, 40.0f, film);
        PathIntegrator integrator(4, camera, RandomSampler(8, 13));
        integrator.render(Scene(top));
        report("two-level", integrator, *film);
        Ray down;  // onto the blade from above: instance 2, its first or second triangle, t = 1.5 from y = 1
        down.o = {0.1f, 1, 0.2f}, down.d = {0, -1, 0};
        SurfaceInteraction si;
        const bool hit = top->intersect(down, &si);
        std::printf("two-level: hit %d t %.4f instance %d primitive %d; world bound x [%.1f, %.1f]\n", (int)hit, si.t, si.instance, si.primitive,
                    top->world_bound().min.x, top->world_bound().max.x);
    }

    // one frame as the two shares two processes (one per GPU) would render: the films add up to the frame
    {
        auto whole = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, whole);
        PathIntegrator all(5, camera, RandomSampler(16, 7), Bounds2i(), 1.0f, LightSampleStrategy::Power);
        all.render(scene);
        std::vector<float> sum(whole->pixels.size(), 0.0f);
        unsigned long long rays = 0;
        for (int rank = 0; rank < 2; ++rank) {
            auto part = std::make_shared<Film>(width, height);
            auto cam = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, part);
            PathIntegrator share(5, cam, RandomSampler(16, 7), Bounds2i(), 1.0f, LightSampleStrategy::Power);
            share.tile_rank = rank, share.tile_world = 2;
            share.render(scene);
            rays += share.stats.rays_closest + share.stats.rays_shadow;
            for (size_t i = 0; i < sum.size(); ++i) sum[i] += part->pixels[i];
        }
        size_t differing = 0;
        for (size_t i = 0; i < sum.size(); ++i) differing += sum[i] != whole->pixels[i];
        std::printf("two shares: %llu rays against %llu of the whole frame, %zu of %zu film values differ\n", rays,
                    (unsigned long long)(all.stats.rays_closest + all.stats.rays_shadow), differing, sum.size());
    }
    // ... and as a job of one process per GPU would run it (here: a world of one): the share rendered into a film on the device,
    // one RCCL reduce, the frame fetched by the root
    {
        auto whole = std::make_shared<Film>(width, height);
        auto camera = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, whole);
        PathIntegrator all(5, camera, RandomSampler(16, 7), Bounds2i(), 1.0f, LightSampleStrategy::Power);
        all.render(scene);
        auto merged = std::make_shared<Film>(width, height);
        auto cam = std::make_shared<PerspectiveCamera>(eye, look, up, 40.0f, merged);
        PathIntegrator rank0(5, cam, RandomSampler(16, 7), Bounds2i(), 1.0f, LightSampleStrategy::Power);
        Comm comm(ctx, 1, 0, Comm::unique_id());
        rank0.render(scene, comm);
        std::printf("world of one over RCCL: film %s\n", merged->pixels == whole->pixels ? "equal" : "DIFFERS");
    }
    return 0;
} catch (const pbrt::Error& e) {
    std::fprintf(stderr, "pbrt::Error (%d): %s\n", e.status, e.what());
    return 3;
}

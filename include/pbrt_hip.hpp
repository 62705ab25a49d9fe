// pbrt_hip.hpp — the host side above the C ABI in C++: a header-only mirror of the trait surface the hot path sits behind
// in lazytiger/pbrt-rs, same names, argument meaning and error behaviour, over include/pbrt_hip.h.
//
// The reference is a Rust crate and rustc is not in this image, so the drop-in a maintainer would write in Rust
// (`impl Integrator for HipIntegrator`, `impl Primitive for HipAggregate`) is shown here in the one compiled
// host language the image has. Nothing in this header computes: every method forwards to a C entry point; no torch types,
// no device pointers in the signatures.
//
//   reference (file:line)                                             here
//   Primitive        src/core/primitive.rs:17-30                      pbrt::Primitive  (intersect mutates ray.t_max; intersect_p; world_bound)
//   BVHAccel::new    src/accelerators/bvh.rs:216-271                  pbrt::BVHAccel   (host build -> resident in HBM; aggregates panic on get_material / get_area_light, bvh.rs:934-953)
//   Scene            src/core/scene.rs:18-46                          pbrt::Scene      (intersect / intersect_p forward to the aggregate)
//   Integrator       src/core/integrator.rs:29-42                     pbrt::Integrator (render(&scene)), pbrt::SamplerIntegrator (li)
//   TransformedPrimitive::new      src/core/primitive.rs:105-123      pbrt::TransformedPrimitive (instances of one or of several aggregates under a top-level BVHAccel, world triangles beside them)
//   TriangleMesh (n, s, uv), Sphere::new   src/shapes/triangle.rs:17-26, sphere.rs:38-62   pbrt::TriangleMesh, pbrt::Sphere
//   PathIntegrator::new            src/integrators/path.rs:31-46      pbrt::PathIntegrator
//   DirectLightingIntegrator::new  src/integrators/directlighting.rs:33-46   pbrt::DirectLightingIntegrator
//   WhittedIntegrator::new / AOIntegrator::new   src/integrators/whitted.rs:22-45, ao.rs:20-34   pbrt::WhittedIntegrator, pbrt::AOIntegrator
//   Film, Film::write_image        src/core/film.rs:30-63, 153-178    pbrt::Film
//   Filter + src/filters/*.rs       src/core/filter.rs:10-15           pbrt::Filter, BoxFilter, TriangleFilter, GaussianFilter, MitchellFilter, LanczosSincFilter
//   Camera, PerspectiveCamera::new src/core/camera.rs:17-60, src/cameras/perspective.rs:34-82   pbrt::Camera, pbrt::PerspectiveCamera
//   OrthographicCamera::new / EnvironmentCamera::new   src/cameras/orthographic.rs:37-80, environment.rs:19-29   pbrt::OrthographicCamera, pbrt::EnvironmentCamera
//   Sampler, RandomSampler::new    src/core/sampler.rs:12-60, src/samplers/random.rs:12-20      pbrt::Sampler, pbrt::RandomSampler
//   StratifiedSampler::new / ZeroTwoSequenceSampler::new / HaltonSampler::new   src/samplers/stratified.rs:22-40, zerotwosequence.rs:17-23, halton.rs:63-98
// Errors: the reference has no error channel — bool / Option for misses, panic! otherwise. A miss is `false` here
// too; what would be a panic there (a failed device call, a bad argument) is a pbrt::Error carrying pbrt_hip_last_error.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pbrt_hip.h"

namespace pbrt {

using Float = float;  // src/core/pbrt.rs:12
struct Point3f {
    Float x = 0, y = 0, z = 0;
};
using Vector3f = Point3f;
struct Bounds3f {
    Point3f min, max;
};
struct Spectrum {  // RGBSpectrum, src/core/spectrum.rs:653-716
    Float c[3] = {0, 0, 0};
    Float y_value() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }
};
// src/core/geometry.rs:757-763 (medium dropped: handle_media is false on every call of this path)
struct Ray {
    Point3f o;
    Vector3f d;
    Float t_max = std::numeric_limits<Float>::infinity();
    Float time = 0;
    Point3f point(Float t) const { return {o.x + d.x * t, o.y + d.y * t, o.z + d.z * t}; }
};
// What Primitive::intersect leaves behind at this boundary: the hit parameter, the watertight barycentrics and the primitive
// (the full SurfaceInteraction of src/core/interaction.rs:224-245 is rebuilt from these inside the shading kernels).
struct SurfaceInteraction {
    Float t = std::numeric_limits<Float>::infinity();
    Float b0 = 0, b1 = 0, b2 = 0;
    int32_t primitive = -1;  // the caller's triangle index
    int32_t instance = -1;   // TransformedPrimitive that was entered, -1 = none
};

class Error : public std::runtime_error {
public:
    Error(const std::string& what, int status) : std::runtime_error(what), status(status) {}
    int status;
};

// one device, one stream; owns nothing the caller can see (include/pbrt_hip.h: context)
class Context {
public:
    explicit Context(int device_id = 0) {
        int rc = pbrt_hip_context_create(device_id, &h_);
        if (rc != PBRT_HIP_OK) throw Error(std::string("pbrt_hip_context_create: ") + pbrt_hip_last_error(nullptr), rc);
    }
    ~Context() { pbrt_hip_context_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    PbrtHipContext* handle() const { return h_; }
    void check(int rc, const char* what) const {
        if (rc != PBRT_HIP_OK) throw Error(std::string(what) + ": " + pbrt_hip_last_error(h_), rc);
    }

private:
    PbrtHipContext* h_ = nullptr;
};

// TriangleMesh (src/shapes/triangle.rs:17-26) + what GeometricPrimitive adds per triangle (material, area light)
struct TriangleMesh {
    std::vector<Float> p;                 // 3 per vertex
    std::vector<int32_t> vertex_indices;  // 3 per triangle
    std::vector<int32_t> material;        // per triangle, index into `materials`
    std::vector<int32_t> area_light;      // per triangle, index into `lights` or -1
    std::vector<PbrtMaterial> materials;
    std::vector<PbrtLight> lights;  // DiffuseAreaLights on triangles, InfiniteAreaLight, point / spot / distant
    std::vector<Float> n, s, uv;    // optional per-vertex shading normals (3), tangents (3), texture coordinates (2): triangle.rs:22-25
    int32_t n_triangles() const { return (int32_t)(vertex_indices.size() / 3); }
    int32_t n_vertices() const { return (int32_t)(p.size() / 3); }
};

// Sphere::new(object_to_world = translate(centre), radius, z_min = -radius, z_max = radius, phi_max = 360) (src/shapes/sphere.rs:38-62)
// as a GeometricPrimitive with a material and, optionally, a DiffuseAreaLight (index into the mesh's lights, whose `prim` is
// n_triangles + the sphere's index)
struct Sphere {
    Point3f centre;
    Float radius = 1;
    int32_t material = 0;
    int32_t area_light = -1;
};
struct BuildOnDevice {};  // tag: BVHAccel::new(HLBVH) built and laid out on the GPU (bvh.rs:475-568), nothing of the tree crosses PCIe

// src/core/primitive.rs:17-30
class Primitive {
public:
    virtual ~Primitive() = default;
    virtual Bounds3f world_bound() const = 0;
    // true on a hit: ray.t_max is lowered to it (primitive.rs:70) and *isect filled
    virtual bool intersect(Ray& ray, SurfaceInteraction* isect) const = 0;
    virtual bool intersect_p(const Ray& ray) const = 0;
    // aggregates do not answer these (bvh.rs:934-953: unimplemented!())
    virtual const PbrtLight* get_area_light() const { throw Error("Primitive::get_area_light: aggregates do not hold one", PBRT_HIP_ERR_INVALID); }
    virtual const PbrtMaterial* get_material() const { throw Error("Primitive::get_material: aggregates do not hold one", PBRT_HIP_ERR_INVALID); }
};

enum class SplitMethod { SAH = 0, HLBVH = 1, Middle = 2, EqualCounts = 3 };  // bvh.rs:30-35

// TransformedPrimitive::new(primitive, primitive_to_world) with a static transform (primitive.rs:105-123): the record the
// instanced BVHAccel constructor takes. to_world row-major 4 x 4 with last row (0, 0, 0, 1); the inverse is formed in double.
// material >= 0 overrides the object's materials for this instance.
inline PbrtInstance TransformedPrimitive(const double to_world[16], int32_t material = -1) {
    PbrtInstance inst;
    std::memset(&inst, 0, sizeof(inst));
    const double* m = to_world;
    const double det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
    if (det == 0.0) throw Error("TransformedPrimitive::new: singular transform", PBRT_HIP_ERR_INVALID);
    double inv[16] = {0};
    const double id = 1.0 / det;
    inv[0] = (m[5] * m[10] - m[6] * m[9]) * id, inv[1] = (m[2] * m[9] - m[1] * m[10]) * id, inv[2] = (m[1] * m[6] - m[2] * m[5]) * id;
    inv[4] = (m[6] * m[8] - m[4] * m[10]) * id, inv[5] = (m[0] * m[10] - m[2] * m[8]) * id, inv[6] = (m[2] * m[4] - m[0] * m[6]) * id;
    inv[8] = (m[4] * m[9] - m[5] * m[8]) * id, inv[9] = (m[1] * m[8] - m[0] * m[9]) * id, inv[10] = (m[0] * m[5] - m[1] * m[4]) * id;
    for (int r = 0; r < 3; ++r) inv[4 * r + 3] = -(inv[4 * r] * m[3] + inv[4 * r + 1] * m[7] + inv[4 * r + 2] * m[11]);
    inv[15] = 1.0;
    for (int k = 0; k < 16; ++k) {
        inst.to_world[k] = (float)to_world[k];
        inst.to_object[k] = (float)inv[k];
    }
    inst.to_world[12] = inst.to_world[13] = inst.to_world[14] = 0.0f, inst.to_world[15] = 1.0f;
    inst.material = material;
    return inst;
}

// BVHAccel over the mesh's triangles as GeometricPrimitives: built on the host in the reference's node order
// (pbrt_hip_bvh_build), resident in HBM (pbrt_hip_scene_create), walked by the HIP traversal kernels.
class BVHAccel : public Primitive {
public:
    BVHAccel(std::shared_ptr<Context> ctx, const TriangleMesh& mesh, int max_prims_in_node = 4, SplitMethod split_method = SplitMethod::SAH)
        : ctx_(std::move(ctx)) {
        PbrtLinearBVHNode* nodes = nullptr;
        int32_t n_nodes = 0;
        int32_t* order = nullptr;
        int rc = pbrt_hip_bvh_build(mesh.p.data(), mesh.n_vertices(), mesh.vertex_indices.data(), mesh.n_triangles(), max_prims_in_node,
                                    (int)split_method, &nodes, &n_nodes, &order);
        if (rc != PBRT_HIP_OK) throw Error("BVHAccel::new: pbrt_hip_bvh_build failed", rc);
        for (int k = 0; k < 3; ++k) {
            (&bound_.min.x)[k] = nodes[0].bounds_min[k];
            (&bound_.max.x)[k] = nodes[0].bounds_max[k];
        }
        n_nodes_ = n_nodes;
        rc = pbrt_hip_scene_create(ctx_->handle(), mesh.p.data(), mesh.n_vertices(), mesh.vertex_indices.data(), mesh.n_triangles(),
                                   mesh.material.data(), mesh.materials.data(), (int32_t)mesh.materials.size(), mesh.area_light.data(),
                                   mesh.lights.data(), (int32_t)mesh.lights.size(), nodes, n_nodes, order, &h_);
        pbrt_hip_free(nodes);
        pbrt_hip_free(order);
        ctx_->check(rc, "BVHAccel::new: pbrt_hip_scene_create");
        set_shading_data(mesh);
    }
    // BVHAccel::new(primitives, max_prims_in_node, SplitMethod::HLBVH) with the tree built on the device; the same tree, byte for
    // byte, as the host builder's (include/pbrt_hip.h: pbrt_hip_scene_create_hlbvh)
    BVHAccel(std::shared_ptr<Context> ctx, const TriangleMesh& mesh, int max_prims_in_node, BuildOnDevice) : ctx_(std::move(ctx)) {
        bound_ = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
        for (int32_t v : mesh.vertex_indices)
            for (int k = 0; k < 3; ++k) {
                (&bound_.min.x)[k] = std::fmin((&bound_.min.x)[k], mesh.p[3 * (size_t)v + k]);
                (&bound_.max.x)[k] = std::fmax((&bound_.max.x)[k], mesh.p[3 * (size_t)v + k]);
            }
        ctx_->check(pbrt_hip_scene_create_hlbvh(ctx_->handle(), mesh.p.data(), mesh.n_vertices(), mesh.vertex_indices.data(), mesh.n_triangles(),
                                                mesh.material.data(), mesh.materials.data(), (int32_t)mesh.materials.size(), mesh.area_light.data(),
                                                mesh.lights.data(), (int32_t)mesh.lights.size(), max_prims_in_node, &h_, &build_ms, &layout_ms),
                    "BVHAccel::new: pbrt_hip_scene_create_hlbvh");
        set_shading_data(mesh);
    }
    // Triangles and Spheres side by side under one BVHAccel (BASELINE config 1's shapes): the tree is built over the primitives'
    // world bounds, triangles first (Triangle::world_bound triangle.rs:175-180, Sphere::world_bound through translate(centre))
    BVHAccel(std::shared_ptr<Context> ctx, const TriangleMesh& mesh, const std::vector<Sphere>& spheres, int max_prims_in_node = 4,
             SplitMethod split_method = SplitMethod::SAH)
        : ctx_(std::move(ctx)) {
        const size_t nt = (size_t)mesh.n_triangles(), ns = spheres.size();
        std::vector<float> lo(3 * (nt + ns)), hi(3 * (nt + ns)), sph(4 * ns);
        std::vector<int32_t> sph_material(ns), sph_light(ns);
        for (size_t t = 0; t < nt; ++t)
            for (int k = 0; k < 3; ++k) {
                const float a = mesh.p[3 * (size_t)mesh.vertex_indices[3 * t] + k], b = mesh.p[3 * (size_t)mesh.vertex_indices[3 * t + 1] + k],
                            c = mesh.p[3 * (size_t)mesh.vertex_indices[3 * t + 2] + k];
                lo[3 * t + k] = std::fmin(a, std::fmin(b, c)), hi[3 * t + k] = std::fmax(a, std::fmax(b, c));
            }
        for (size_t i = 0; i < ns; ++i) {
            const float c[3] = {spheres[i].centre.x, spheres[i].centre.y, spheres[i].centre.z}, r = spheres[i].radius;
            for (int k = 0; k < 3; ++k) lo[3 * (nt + i) + k] = c[k] + (-r), hi[3 * (nt + i) + k] = c[k] + r, sph[4 * i + k] = c[k];
            sph[4 * i + 3] = r, sph_material[i] = spheres[i].material, sph_light[i] = spheres[i].area_light;
        }
        PbrtLinearBVHNode* nodes = nullptr;
        int32_t n_nodes = 0, *order = nullptr;
        int rc = pbrt_hip_bvh_build_boxes(lo.data(), hi.data(), (int32_t)(nt + ns), max_prims_in_node, (int)split_method, &nodes, &n_nodes, &order);
        if (rc != PBRT_HIP_OK) throw Error("BVHAccel::new: pbrt_hip_bvh_build_boxes failed", rc);
        for (int k = 0; k < 3; ++k) (&bound_.min.x)[k] = nodes[0].bounds_min[k], (&bound_.max.x)[k] = nodes[0].bounds_max[k];
        n_nodes_ = n_nodes;
        rc = pbrt_hip_scene_create_with_spheres(ctx_->handle(), mesh.p.data(), mesh.n_vertices(), mesh.vertex_indices.data(), mesh.n_triangles(),
                                                mesh.material.data(), mesh.materials.data(), (int32_t)mesh.materials.size(), mesh.area_light.data(),
                                                mesh.lights.data(), (int32_t)mesh.lights.size(), sph.data(), sph_material.data(), sph_light.data(), (int32_t)ns,
                                                nodes, n_nodes, order, &h_);
        pbrt_hip_free(nodes), pbrt_hip_free(order);
        ctx_->check(rc, "BVHAccel::new: pbrt_hip_scene_create_with_spheres");
    }
    // The top-level aggregate of a scene of TransformedPrimitives (primitive.rs:105-159): BVHAccel::new over the object's
    // triangles (object space), TransformedPrimitive::new(object, to_world) per instance, BVHAccel::new over their world bounds.
    // Lights: non-area lights only (an instanced primitive cannot be an area light).
    BVHAccel(std::shared_ptr<Context> ctx, const TriangleMesh& object, const std::vector<PbrtInstance>& instances, int max_prims_in_node = 4,
             SplitMethod split_method = SplitMethod::SAH)
        : ctx_(std::move(ctx)) {
        PbrtLinearBVHNode *blas = nullptr, *tlas = nullptr;
        int32_t n_blas = 0, n_tlas = 0, *blas_order = nullptr, *tlas_order = nullptr;
        int rc = pbrt_hip_bvh_build(object.p.data(), object.n_vertices(), object.vertex_indices.data(), object.n_triangles(), max_prims_in_node,
                                    (int)split_method, &blas, &n_blas, &blas_order);
        if (rc != PBRT_HIP_OK) throw Error("BVHAccel::new: pbrt_hip_bvh_build failed", rc);
        const int32_t n = (int32_t)instances.size();
        std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
        rc = pbrt_hip_instance_bounds(blas[0].bounds_min, blas[0].bounds_max, instances.data(), n, lo.data(), hi.data());
        if (rc == PBRT_HIP_OK) rc = pbrt_hip_bvh_build_boxes(lo.data(), hi.data(), n, max_prims_in_node, (int)split_method, &tlas, &n_tlas, &tlas_order);
        if (rc == PBRT_HIP_OK) {
            for (int k = 0; k < 3; ++k) {
                (&bound_.min.x)[k] = tlas[0].bounds_min[k];
                (&bound_.max.x)[k] = tlas[0].bounds_max[k];
            }
            n_nodes_ = n_tlas;
            rc = pbrt_hip_scene_create_instanced(ctx_->handle(), object.p.data(), object.n_vertices(), object.vertex_indices.data(), object.n_triangles(),
                                                 object.material.data(), object.materials.data(), (int32_t)object.materials.size(), object.lights.data(),
                                                 (int32_t)object.lights.size(), blas, n_blas, blas_order, instances.data(), n, tlas, n_tlas, tlas_order, &h_);
        }
        pbrt_hip_free(blas), pbrt_hip_free(blas_order), pbrt_hip_free(tlas), pbrt_hip_free(tlas_order);
        ctx_->check(rc, "BVHAccel::new (TransformedPrimitives)");
    }
    // The general top-level aggregate (primitive.rs:105-159 beside :33-103): TransformedPrimitives of SEVERAL object aggregates
    // (instance i shows objects[instance_object[i]]) and plain world-space triangles beside them — the only primitives that can
    // carry area lights here. Materials and lights are the WORLD mesh's (`world.materials`, `world.lights`; an object's triangles
    // index the same material table); `world` may hold no triangles.
    BVHAccel(std::shared_ptr<Context> ctx, const std::vector<TriangleMesh>& objects, const std::vector<PbrtInstance>& instances,
             const std::vector<int32_t>& instance_object, const TriangleMesh& world, int max_prims_in_node = 4, SplitMethod split_method = SplitMethod::SAH)
        : ctx_(std::move(ctx)) {
        struct Tree {
            PbrtLinearBVHNode* nodes = nullptr;
            int32_t n = 0, *order = nullptr;
        };
        std::vector<Tree> trees(objects.size());
        Tree top;
        auto release = [&]() {
            for (Tree& t : trees) pbrt_hip_free(t.nodes), pbrt_hip_free(t.order);
            pbrt_hip_free(top.nodes), pbrt_hip_free(top.order);
        };
        int rc = instances.size() == instance_object.size() ? PBRT_HIP_OK : PBRT_HIP_ERR_INVALID;
        for (size_t k = 0; k < objects.size() && rc == PBRT_HIP_OK; ++k)
            rc = pbrt_hip_bvh_build(objects[k].p.data(), objects[k].n_vertices(), objects[k].vertex_indices.data(), objects[k].n_triangles(), max_prims_in_node,
                                    (int)split_method, &trees[k].nodes, &trees[k].n, &trees[k].order);
        const size_t ni = instances.size(), nw = (size_t)world.n_triangles();
        std::vector<float> lo(3 * (ni + nw)), hi(3 * (ni + nw));
        for (size_t i = 0; i < ni && rc == PBRT_HIP_OK; ++i) {  // TransformedPrimitive::world_bound (primitive.rs:126-134)
            const int32_t o = instance_object[i];
            rc = o >= 0 && (size_t)o < objects.size()
                     ? pbrt_hip_instance_bounds(trees[o].nodes[0].bounds_min, trees[o].nodes[0].bounds_max, &instances[i], 1, &lo[3 * i], &hi[3 * i])
                     : PBRT_HIP_ERR_INVALID;
        }
        for (size_t t = 0; t < nw; ++t)
            for (int k = 0; k < 3; ++k) {
                const float a = world.p[3 * (size_t)world.vertex_indices[3 * t] + k], b = world.p[3 * (size_t)world.vertex_indices[3 * t + 1] + k],
                            c = world.p[3 * (size_t)world.vertex_indices[3 * t + 2] + k];
                lo[3 * (ni + t) + k] = std::fmin(a, std::fmin(b, c)), hi[3 * (ni + t) + k] = std::fmax(a, std::fmax(b, c));
            }
        if (rc == PBRT_HIP_OK) rc = pbrt_hip_bvh_build_boxes(lo.data(), hi.data(), (int32_t)(ni + nw), max_prims_in_node, (int)split_method, &top.nodes, &top.n, &top.order);
        if (rc != PBRT_HIP_OK) {
            release();
            throw Error("BVHAccel::new (two levels): bad arguments or a host build failed", rc);
        }
        for (int k = 0; k < 3; ++k) (&bound_.min.x)[k] = top.nodes[0].bounds_min[k], (&bound_.max.x)[k] = top.nodes[0].bounds_max[k];
        n_nodes_ = top.n;
        std::vector<PbrtObject> objs(objects.size());
        for (size_t k = 0; k < objects.size(); ++k)
            objs[k] = {objects[k].p.data(), objects[k].n_vertices(), objects[k].vertex_indices.data(), objects[k].n_triangles(),
                       objects[k].material.empty() ? nullptr : objects[k].material.data(), trees[k].nodes, trees[k].n, trees[k].order};
        rc = pbrt_hip_scene_create_two_level(ctx_->handle(), objs.data(), (int32_t)objs.size(), instances.data(), instance_object.data(), (int32_t)ni,
                                             nw ? world.p.data() : nullptr, nw ? world.n_vertices() : 0, nw ? world.vertex_indices.data() : nullptr, (int32_t)nw,
                                             nw ? world.material.data() : nullptr, nw ? world.area_light.data() : nullptr, world.materials.data(),
                                             (int32_t)world.materials.size(), world.lights.empty() ? nullptr : world.lights.data(), (int32_t)world.lights.size(),
                                             top.nodes, top.n, top.order, &h_);
        release();
        ctx_->check(rc, "BVHAccel::new: pbrt_hip_scene_create_two_level");
    }
    ~BVHAccel() override { pbrt_hip_scene_destroy(h_); }
    BVHAccel(const BVHAccel&) = delete;
    BVHAccel& operator=(const BVHAccel&) = delete;

    Bounds3f world_bound() const override { return bound_; }
    bool intersect(Ray& ray, SurfaceInteraction* isect) const override {  // a batch of one
        PbrtRay r = to_c(ray);
        PbrtHit hit;
        ctx_->check(pbrt_hip_intersect(h_, &r, 1, &hit), "BVHAccel::intersect");
        if (hit.prim_id < 0) return false;
        ray.t_max = hit.t;
        if (isect) *isect = {hit.t, hit.b0, hit.b1, hit.b2, hit.prim_id, hit.instance_id};
        return true;
    }
    bool intersect_p(const Ray& ray) const override {
        PbrtRay r = to_c(ray);
        uint8_t any = 0;
        ctx_->check(pbrt_hip_intersect_p(h_, &r, 1, &any), "BVHAccel::intersect_p");
        return any != 0;
    }
    // the batch forms the kernels are made for: rays[i].t_max is lowered on a hit exactly as the single call does
    void intersect(std::vector<Ray>& rays, std::vector<SurfaceInteraction>& isects) const {
        std::vector<PbrtRay> r(rays.size());
        std::vector<PbrtHit> h(rays.size());
        for (size_t i = 0; i < rays.size(); ++i) r[i] = to_c(rays[i]);
        ctx_->check(pbrt_hip_intersect(h_, r.data(), (int64_t)r.size(), h.data()), "BVHAccel::intersect");
        isects.resize(rays.size());
        for (size_t i = 0; i < rays.size(); ++i) {
            isects[i] = {h[i].t, h[i].b0, h[i].b1, h[i].b2, h[i].prim_id, h[i].instance_id};
            if (h[i].prim_id >= 0) rays[i].t_max = h[i].t;
        }
    }
    PbrtHipScene* handle() const { return h_; }
    const std::shared_ptr<Context>& context() const { return ctx_; }
    int32_t n_nodes() const { return n_nodes_; }  // 0 for a tree that never left the device
    double build_ms = 0, layout_ms = 0;           // BuildOnDevice: HIP-event times of the build and of the re-layout

private:
    void set_shading_data(const TriangleMesh& mesh) {
        if (mesh.n.empty() && mesh.s.empty() && mesh.uv.empty()) return;
        ctx_->check(pbrt_hip_scene_set_shading_data(h_, mesh.p.data(), mesh.n_vertices(), mesh.vertex_indices.data(), mesh.n_triangles(),
                                                    mesh.n.empty() ? nullptr : mesh.n.data(), mesh.s.empty() ? nullptr : mesh.s.data(),
                                                    mesh.uv.empty() ? nullptr : mesh.uv.data()),
                    "TriangleMesh: pbrt_hip_scene_set_shading_data");
    }
    static PbrtRay to_c(const Ray& ray) { return {{ray.o.x, ray.o.y, ray.o.z}, {ray.d.x, ray.d.y, ray.d.z}, ray.t_max, ray.time}; }
    std::shared_ptr<Context> ctx_;
    PbrtHipScene* h_ = nullptr;
    Bounds3f bound_;
    int32_t n_nodes_ = 0;
};

// src/core/scene.rs:18-46: the aggregate and the lights (which travel with the mesh at this boundary)
class Scene {
public:
    explicit Scene(std::shared_ptr<BVHAccel> aggregate) : aggregate_(std::move(aggregate)) {}
    const Bounds3f world_bound() const { return aggregate_->world_bound(); }
    bool intersect(Ray& ray, SurfaceInteraction* isect) const { return aggregate_->intersect(ray, isect); }
    bool intersect_p(const Ray& ray) const { return aggregate_->intersect_p(ray); }
    const BVHAccel& aggregate() const { return *aggregate_; }

private:
    std::shared_ptr<BVHAccel> aggregate_;
};

struct Vector2f {
    Float x = 0, y = 0;
};
// Filter (src/core/filter.rs:10-15: radius, evaluate) with the five of src/filters/: what Film::new tabulates (film.rs:52-63)
struct Filter {
    PbrtFilterType kind = PBRT_FILTER_BOX;
    Vector2f radius{0.5f, 0.5f};
    Float a = 0, b = 0;  // GaussianFilter alpha / MitchellFilter B, C / LanczosSincFilter tau
};
struct BoxFilter : Filter {  // src/filters/boxf.rs
    explicit BoxFilter(Vector2f r = {0.5f, 0.5f}) { kind = PBRT_FILTER_BOX, radius = r; }
};
struct TriangleFilter : Filter {  // src/filters/triangle.rs
    explicit TriangleFilter(Vector2f r = {2.0f, 2.0f}) { kind = PBRT_FILTER_TRIANGLE, radius = r; }
};
struct GaussianFilter : Filter {  // src/filters/gaussian.rs:12-31
    explicit GaussianFilter(Vector2f r = {2.0f, 2.0f}, Float alpha = 2.0f) { kind = PBRT_FILTER_GAUSSIAN, radius = r, a = alpha; }
};
struct MitchellFilter : Filter {  // src/filters/mitchell.rs:12-41
    explicit MitchellFilter(Vector2f r = {2.0f, 2.0f}, Float B = 1.0f / 3.0f, Float C = 1.0f / 3.0f) { kind = PBRT_FILTER_MITCHELL, radius = r, a = B, b = C; }
};
struct LanczosSincFilter : Filter {  // src/filters/sinc.rs:12-45
    explicit LanczosSincFilter(Vector2f r = {4.0f, 4.0f}, Float tau = 3.0f) { kind = PBRT_FILTER_LANCZOS, radius = r, a = tau; }
};

// Film::new (src/core/film.rs:30-63) with one of src/filters/*.rs; pixels = {xyz[3], filter_weight_sum} (film.rs:9-15)
class Film {
public:
    Film(int width, int height, const Filter& filter, Float max_sample_luminance = 0)
        : width(width), height(height), radius(filter.radius.x), radius_y(filter.radius.y), max_sample_luminance(max_sample_luminance),
          pixels((size_t)width * height * 4, 0.0f) {
        box_ = filter.kind == PBRT_FILTER_BOX && filter.radius.x == 0.5f && filter.radius.y == 0.5f;
        if (!box_ && pbrt_hip_filter_table((int32_t)filter.kind, filter.radius.x, filter.radius.y, filter.a, filter.b, table_) != PBRT_HIP_OK)
            throw Error("Film::new: bad filter", PBRT_HIP_ERR_INVALID);
    }
    Film(int width, int height, PbrtFilterType filter = PBRT_FILTER_BOX, Float radius = 0.5f, Float a = 0, Float b = 0, Float max_sample_luminance = 0)
        : Film(width, height, make_filter(filter, radius, a, b), max_sample_luminance) {}
    // Film::write_image (film.rs:153-178): xyz / weight -> RGB; the writer the reference leaves as todo!() (imageio.rs:3-5)
    void write_image(const std::string& path) const {
        std::vector<float> rgb((size_t)width * height * 3);
        pbrt_hip_film_to_rgb(pixels.data(), (int64_t)width * height, rgb.data());
        const bool pfm = path.size() > 4 && path.substr(path.size() - 4) == ".pfm", exr = path.size() > 4 && path.substr(path.size() - 4) == ".exr";
        int rc = pfm ? pbrt_hip_write_pfm(path.c_str(), rgb.data(), width, height)
                     : exr ? pbrt_hip_write_exr(path.c_str(), rgb.data(), width, height) : pbrt_hip_write_png(path.c_str(), rgb.data(), width, height);
        if (rc != PBRT_HIP_OK) throw Error("Film::write_image: cannot write " + path, rc);
    }
    std::vector<float> rgb() const {
        std::vector<float> out((size_t)width * height * 3);
        pbrt_hip_film_to_rgb(pixels.data(), (int64_t)width * height, out.data());
        return out;
    }
    const float* filter_table() const { return box_ ? nullptr : table_; }
    int width, height;
    Float radius, radius_y, max_sample_luminance;  // filter radius in x (`radius`) and y
    std::vector<float> pixels;

private:
    static Filter make_filter(PbrtFilterType kind, Float r, Float a, Float b) {
        Filter f;
        f.kind = kind, f.radius = {r, r}, f.a = a, f.b = b;
        return f;
    }
    bool box_ = true;
    float table_[256];
};

// Camera (src/core/camera.rs:17-60): camera_to_world, the shutter, the film. look_at as transform.rs:510-545, in double, rounded once
class Camera {
public:
    virtual ~Camera() = default;
    PbrtCamera cam;
    std::shared_ptr<Film> film;

protected:
    Camera(const Point3f& eye, const Point3f& look, const Vector3f& up_in, std::shared_ptr<Film> film_in, PbrtCameraKind kind) : film(std::move(film_in)) {
        std::memset(&cam, 0, sizeof(cam));
        double d[3] = {look.x - eye.x, look.y - eye.y, look.z - eye.z}, up[3] = {up_in.x, up_in.y, up_in.z}, right[3], new_up[3];
        auto norm = [](double v[3]) {
            double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            v[0] /= l, v[1] /= l, v[2] /= l;
        };
        auto cross = [](const double a[3], const double b[3], double o[3]) {
            o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
        };
        norm(d), norm(up);
        cross(up, d, right);
        norm(right);
        cross(d, right, new_up);
        const double e[3] = {eye.x, eye.y, eye.z};
        for (int r = 0; r < 3; ++r) {
            cam.camera_to_world[4 * r + 0] = (float)right[r];
            cam.camera_to_world[4 * r + 1] = (float)new_up[r];
            cam.camera_to_world[4 * r + 2] = (float)d[r];
            cam.camera_to_world[4 * r + 3] = (float)e[r];
        }
        cam.camera_to_world[15] = 1.0f;
        cam.shutter_close = 1.0f;
        cam.kind = kind;
    }
    // inverse(screen_to_raster) for the screen window [sx0, sx1] x [sy0, sy1] (camera.rs:120-135): raster (x, y) -> screen
    void raster_to_screen(double sx0, double sx1, double sy0, double sy1, double out[4]) const {
        out[0] = (sx1 - sx0) / film->width, out[1] = sx0, out[2] = (sy0 - sy1) / film->height, out[3] = sy1;
    }
};

// PerspectiveCamera::new(camera_to_world, screen_window, shutter, lens_radius, focal_distance, fov, film):
// perspective.rs:34-82 with transform.rs:510-566's look_at / perspective, computed in double and rounded once
class PerspectiveCamera : public Camera {
public:
    PerspectiveCamera(const Point3f& eye, const Point3f& look, const Vector3f& up, Float fov_deg, std::shared_ptr<Film> film_in, Float lens_radius = 0,
                      Float focal_distance = 1e6f)
        : Camera(eye, look, up, std::move(film_in), PBRT_CAMERA_PERSPECTIVE) {
        const int w = film->width, h = film->height;
        const double n = 1e-2, f = 1000.0, A = f / (f - n), B = -f * n / (f - n);
        const double inv_tan = 1.0 / std::tan(fov_deg * 3.14159265358979323846 / 360.0), aspect = (double)w / h;
        const double sx0 = aspect > 1.0 ? -aspect : -1.0, sx1 = -sx0, sy0 = aspect > 1.0 ? -1.0 : -1.0 / aspect, sy1 = -sy0;
        float* m = cam.raster_to_camera;  // inverse(camera_to_screen) * raster_to_screen
        m[0] = (float)((sx1 - sx0) / (w * inv_tan));
        m[3] = (float)(sx0 / inv_tan);
        m[5] = (float)((sy0 - sy1) / (h * inv_tan));
        m[7] = (float)(sy1 / inv_tan);
        m[11] = 1.0f;
        m[14] = (float)(1.0 / B);
        m[15] = (float)(-A / B);
        cam.lens_radius = lens_radius;
        cam.focal_distance = focal_distance;
    }
};

// OrthographicCamera::new (orthographic.rs:37-80): camera_to_screen = orthographic(0, 1), the identity in z; the screen window is
// [-aspect, aspect] x [-1, 1] scaled by `half_height` (pbrt's "screenwindow")
class OrthographicCamera : public Camera {
public:
    OrthographicCamera(const Point3f& eye, const Point3f& look, const Vector3f& up, Float half_height, std::shared_ptr<Film> film_in, Float lens_radius = 0,
                       Float focal_distance = 1e6f)
        : Camera(eye, look, up, std::move(film_in), PBRT_CAMERA_ORTHOGRAPHIC) {
        const double aspect = (double)film->width / film->height, hh = half_height;
        double r[4];
        raster_to_screen(-aspect * hh, aspect * hh, -hh, hh, r);
        float* m = cam.raster_to_camera;
        m[0] = (float)r[0], m[3] = (float)r[1], m[5] = (float)r[2], m[7] = (float)r[3];
        m[10] = 1.0f, m[15] = 1.0f;
        cam.lens_radius = lens_radius;
        cam.focal_distance = focal_distance;
    }
};

// EnvironmentCamera::new (environment.rs:19-29): every direction of the sphere, from one point; only camera_to_world is used
class EnvironmentCamera : public Camera {
public:
    EnvironmentCamera(const Point3f& eye, const Point3f& look, const Vector3f& up, std::shared_ptr<Film> film_in)
        : Camera(eye, look, up, std::move(film_in), PBRT_CAMERA_ENVIRONMENT) {
        for (int k = 0; k < 4; ++k) cam.raster_to_camera[5 * k] = 1.0f;
    }
};

// Sampler (src/core/sampler.rs:12-60): samples_per_pixel and what its constructor fixes. A value: the device draws the samples
struct Sampler {
    int samples_per_pixel = 16;
    uint64_t seed = 0;
    PbrtSamplerKind kind = PBRT_SAMPLER_RANDOM;
    int x_pixel_samples = 0, y_pixel_samples = 0;  // StratifiedSampler
    bool jitter_samples = true;
    int n_sampled_dimensions = 0;  // PixelSampler::new (sampler.rs:252-275)
};
// RandomSampler::new(samples_per_pixel, seed) (random.rs:12-20); streams are keyed per (pixel, sample): DESIGN.md section 2
struct RandomSampler : Sampler {
    RandomSampler(int spp = 16, uint64_t seed_in = 0) { samples_per_pixel = spp, seed = seed_in; }
};
// StratifiedSampler::new(x_pixel_samples, y_pixel_samples, jitter_samples, n_sampled_dimensions) (stratified.rs:22-40)
struct StratifiedSampler : Sampler {
    StratifiedSampler(int x, int y, bool jitter, int n_dims, uint64_t seed_in = 0) {
        kind = PBRT_SAMPLER_STRATIFIED, x_pixel_samples = x, y_pixel_samples = y, jitter_samples = jitter, n_sampled_dimensions = n_dims;
        samples_per_pixel = x * y, seed = seed_in;
    }
};
// ZeroTwoSequenceSampler::new(samples_per_pixel, n_sampled_dimensions) (zerotwosequence.rs:17-23: rounded up to a power of two)
struct ZeroTwoSequenceSampler : Sampler {
    ZeroTwoSequenceSampler(int spp, int n_dims = 4, uint64_t seed_in = 0) {
        kind = PBRT_SAMPLER_ZEROTWO, n_sampled_dimensions = n_dims, seed = seed_in, x_pixel_samples = y_pixel_samples = 1;
        samples_per_pixel = 1;
        while (samples_per_pixel < spp) samples_per_pixel *= 2;
    }
};
// HaltonSampler::new(samples_per_pixel, sample_bounds, sample_at_pixel_center = false) (halton.rs:63-98); the bounds are the film's
struct HaltonSampler : Sampler {
    explicit HaltonSampler(int spp, uint64_t seed_in = 0) { kind = PBRT_SAMPLER_HALTON, samples_per_pixel = spp, seed = seed_in, x_pixel_samples = y_pixel_samples = 1; }
};
struct Bounds2i {
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
};
struct RenderStats {
    uint64_t camera_samples = 0, rays_closest = 0, rays_shadow = 0;
    double ms = 0;
};

// The film merge of a job that runs one process per GPU (what replaces parallel_for_2d's join, parallel.rs:4-21,
// across GPUs): rank 0 draws an id (Comm::unique_id) and hands the bytes to the other processes by whatever means the host
// has; every process builds its Comm and renders its share with SamplerIntegrator::render(scene, comm).
class Comm {
public:
    static std::vector<uint8_t> unique_id() {
        std::vector<uint8_t> id(PBRT_HIP_COMM_ID_BYTES);
        const int rc = pbrt_hip_comm_unique_id(id.data());
        if (rc != PBRT_HIP_OK) throw Error(std::string("pbrt_hip_comm_unique_id: ") + pbrt_hip_comm_last_error(), rc);
        return id;
    }
    Comm(std::shared_ptr<Context> ctx, int world, int rank, const std::vector<uint8_t>& id) : world(world), rank(rank), ctx_(std::move(ctx)) {
        if (id.size() != PBRT_HIP_COMM_ID_BYTES) throw Error("Comm: the id is PBRT_HIP_COMM_ID_BYTES bytes", PBRT_HIP_ERR_INVALID);
        const int rc = pbrt_hip_comm_create(ctx_->handle(), world, rank, id.data(), &h_);
        if (rc != PBRT_HIP_OK) throw Error(std::string("pbrt_hip_comm_create: ") + pbrt_hip_comm_last_error(), rc);
    }
    ~Comm() { pbrt_hip_comm_destroy(h_); }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    PbrtHipComm* handle() const { return h_; }
    const int world, rank;

private:
    std::shared_ptr<Context> ctx_;
    PbrtHipComm* h_ = nullptr;
};

// src/core/integrator.rs:29-42
class Integrator {
public:
    virtual ~Integrator() = default;
    virtual void render(const Scene& scene) = 0;  // called once, blocks until the film is complete
};

// SamplerIntegrator (integrator.rs:399-480): camera, sampler, pixel bounds; render = the tile / pixel / sample loop, on the device
class SamplerIntegrator : public Integrator {
public:
    SamplerIntegrator(std::shared_ptr<const Camera> camera, Sampler sampler, Bounds2i pixel_bounds)
        : camera(std::move(camera)), sampler(sampler), pixel_bounds(pixel_bounds) {}
    void render(const Scene& scene) override { render_into(scene, nullptr); }
    // The same frame as one of comm.world processes: this process renders the tiles dealt to comm.rank into a film on its
    // device, the films are summed with one RCCL reduce, and rank `root` (every rank if root < 0) holds the frame in camera->film.
    void render(const Scene& scene, const Comm& comm, int root = 0) {
        const int rank0 = tile_rank, world0 = tile_world;  // the share is the communicator's for this call only
        tile_rank = comm.rank, tile_world = comm.world;
        try {
            render_into(scene, &comm, root);
        } catch (...) {
            tile_rank = rank0, tile_world = world0;
            throw;
        }
        tile_rank = rank0, tile_world = world0;
    }

private:
    void render_into(const Scene& scene, const Comm* comm, int root = 0) {
        PbrtRenderParams rp = params();
        Film& film = *camera->film;
        rp.spp = sampler.samples_per_pixel;
        rp.seed = sampler.seed;
        rp.sampler = sampler.kind, rp.sampler_x = sampler.x_pixel_samples, rp.sampler_y = sampler.y_pixel_samples;
        rp.sampler_jitter = sampler.jitter_samples ? 1 : 0, rp.sampler_dims = sampler.n_sampled_dimensions;
        rp.width = film.width, rp.height = film.height;
        if (pixel_bounds.x1 > pixel_bounds.x0) {
            rp.x0 = pixel_bounds.x0, rp.y0 = pixel_bounds.y0, rp.x1 = pixel_bounds.x1, rp.y1 = pixel_bounds.y1;
        } else {  // Film::get_sample_bounds (film.rs:76-81)
            int32_t b[4];
            pbrt_hip_sample_bounds(film.width, film.height, film.radius, film.radius_y, b);
            rp.x0 = b[0], rp.y0 = b[1], rp.x1 = b[2], rp.y1 = b[3];
        }
        // the 16x16 tiles of the frame (integrator.rs:404-409) this process renders: tile_rank of tile_world, dealt in Morton order
        rp.tile_rank = tile_rank, rp.tile_world = tile_world;
        rp.filter_radius[0] = film.radius, rp.filter_radius[1] = film.radius_y;
        rp.filter_table = film.filter_table();
        rp.max_sample_luminance = film.max_sample_luminance;
        PbrtRenderStats st;
        const auto& ctx = scene.aggregate().context();
        if (!comm) {
            ctx->check(pbrt_hip_render(scene.aggregate().handle(), &camera->cam, &rp, film.pixels.data(), &st), "Integrator::render");
        } else {
            const int64_t n_pixels = (int64_t)film.width * film.height;
            float* d_film = nullptr;
            ctx->check(pbrt_hip_film_create(ctx->handle(), n_pixels, &d_film), "Integrator::render: pbrt_hip_film_create");
            int rc = pbrt_hip_render_device(scene.aggregate().handle(), &camera->cam, &rp, d_film, &st);
            const char* what = "Integrator::render: pbrt_hip_render_device";
            if (rc == PBRT_HIP_OK) rc = pbrt_hip_film_reduce(comm->handle(), d_film, n_pixels, root), what = "Integrator::render: pbrt_hip_film_reduce";
            if (rc == PBRT_HIP_OK && (root < 0 || root == comm->rank))
                rc = pbrt_hip_film_download(ctx->handle(), d_film, n_pixels, film.pixels.data()), what = "Integrator::render: pbrt_hip_film_download";
            pbrt_hip_film_destroy(ctx->handle(), d_film);
            ctx->check(rc, what);
        }
        stats = {st.camera_samples, st.rays_closest, st.rays_shadow, st.total_ms};
    }

public:
    // li(&mut ray, scene, sampler, depth) -> Spectrum (integrator.rs:29-42, :452): the sampler argument is the stream the
    // integrator draws from (RNG::set_sequence(stream_key), rng.rs:21-35). Re-entrant; depth is 0 at this boundary.
    Spectrum li(const Ray& ray, const Scene& scene, uint64_t stream_key) const {
        PbrtRenderParams rp = params();
        PbrtLiParams lp = {rp.integrator, rp.max_depth, rp.rr_threshold, rp.light_strategy, rp.ao_samples, 0};
        PbrtRay r = {{ray.o.x, ray.o.y, ray.o.z}, {ray.d.x, ray.d.y, ray.d.z}, ray.t_max, ray.time};
        Spectrum l;
        scene.aggregate().context()->check(pbrt_hip_li(scene.aggregate().handle(), &lp, &r, &stream_key, 1, l.c, nullptr), "Integrator::li");
        return l;
    }
    std::shared_ptr<const Camera> camera;
    Sampler sampler;
    Bounds2i pixel_bounds;
    RenderStats stats;
    int tile_rank = 0, tile_world = 1;  // one process per GPU: the films of the tile_world processes add up to the frame

protected:
    virtual PbrtRenderParams params() const = 0;
    static PbrtRenderParams zeroed() {
        PbrtRenderParams rp;
        std::memset(&rp, 0, sizeof(rp));
        return rp;
    }
};

enum class LightSampleStrategy { Uniform = 0, Power = 1, Spatial = 2 };  // lightdistrib.rs:222-232

// PathIntegrator::new(max_depth, camera, sampler, pixel_bounds, rr_threshold, light_sample_strategy) (path.rs:31-46)
class PathIntegrator : public SamplerIntegrator {
public:
    PathIntegrator(int max_depth, std::shared_ptr<const Camera> camera, Sampler sampler, Bounds2i pixel_bounds = Bounds2i(),
                   Float rr_threshold = 1.0f, LightSampleStrategy light_sample_strategy = LightSampleStrategy::Spatial)
        : SamplerIntegrator(std::move(camera), sampler, pixel_bounds), max_depth(max_depth), rr_threshold(rr_threshold), strategy(light_sample_strategy) {}
    int max_depth;
    Float rr_threshold;
    LightSampleStrategy strategy;

protected:
    PbrtRenderParams params() const override {
        PbrtRenderParams rp = zeroed();
        rp.integrator = PBRT_INTEGRATOR_PATH, rp.max_depth = max_depth, rp.rr_threshold = rr_threshold, rp.light_strategy = (int)strategy;
        return rp;
    }
};

enum class LightStrategy { UniformSampleAll = 0, UniformSampleOne = 1 };  // directlighting.rs:21-24

// DirectLightingIntegrator::new(strategy, max_depth, camera, sampler, pixel_bounds) (directlighting.rs:33-46)
class DirectLightingIntegrator : public SamplerIntegrator {
public:
    DirectLightingIntegrator(LightStrategy strategy, int max_depth, std::shared_ptr<const Camera> camera, Sampler sampler,
                             Bounds2i pixel_bounds = Bounds2i())
        : SamplerIntegrator(std::move(camera), sampler, pixel_bounds), strategy(strategy), max_depth(max_depth) {}
    LightStrategy strategy;
    int max_depth;

protected:
    PbrtRenderParams params() const override {
        PbrtRenderParams rp = zeroed();
        rp.integrator = PBRT_INTEGRATOR_DIRECT, rp.max_depth = max_depth, rp.light_strategy = (int)strategy;
        return rp;
    }
};

// WhittedIntegrator::new(max_depth, camera, sampler, pixel_bounds) (whitted.rs:22-45)
class WhittedIntegrator : public SamplerIntegrator {
public:
    WhittedIntegrator(int max_depth, std::shared_ptr<const Camera> camera, Sampler sampler, Bounds2i pixel_bounds = Bounds2i())
        : SamplerIntegrator(std::move(camera), sampler, pixel_bounds), max_depth(max_depth) {}
    int max_depth;

protected:
    PbrtRenderParams params() const override {
        PbrtRenderParams rp = zeroed();
        rp.integrator = PBRT_INTEGRATOR_WHITTED, rp.max_depth = max_depth;
        return rp;
    }
};

// AOIntegrator::new(cos_sample, n_samples, camera, sampler, pixel_bounds) (ao.rs:20-34)
class AOIntegrator : public SamplerIntegrator {
public:
    AOIntegrator(bool cos_sample, int n_samples, std::shared_ptr<const Camera> camera, Sampler sampler, Bounds2i pixel_bounds = Bounds2i())
        : SamplerIntegrator(std::move(camera), sampler, pixel_bounds), cos_sample(cos_sample), n_samples(n_samples) {}
    bool cos_sample;
    int n_samples;

protected:
    PbrtRenderParams params() const override {
        PbrtRenderParams rp = zeroed();
        rp.integrator = PBRT_INTEGRATOR_AO, rp.light_strategy = cos_sample ? 1 : 0, rp.ao_samples = n_samples;
        return rp;
    }
};

}  // namespace pbrt

// ORACLE — TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library, and only as the checker (see oracle/README.md).
// Parity unpinned: the reference has no fixtures for this path and cannot be built here.
//
// Flat C entry points (ctypes) over the CPU restatement in o_*.h.
#include <chrono>

#include "o_render.h"
#include "o_sphere.h"

using namespace oracle;

namespace {
struct OracleScene {
    Scene scene;
    std::shared_ptr<TriangleMesh> mesh;
    std::shared_ptr<BVHAccel> blas;  // instanced scenes: the (first) object-space aggregate
    std::vector<std::shared_ptr<BVHAccel>> blases;  // two-level scenes: every object aggregate
    std::vector<Matrix4> to_object;  // per instance
};

struct OrcHit {  // same 32-byte layout as PbrtHit (include/pbrt_hip.h)
    float t, b0, b1, b2;
    int32_t prim_id;
    int32_t pad[3];
};

template <class F>
void parallel_chunks(int64_t n, int n_threads, F f) {
    n_threads = std::max(1, n_threads);
    std::vector<std::thread> th;
    int64_t chunk = (n + n_threads - 1) / n_threads;
    for (int t = 0; t < n_threads; ++t) {
        int64_t b = t * chunk, e = std::min(n, b + chunk);
        if (b >= e) break;
        th.emplace_back(f, t, b, e);
    }
    for (auto& x : th) x.join();
}
}  // namespace

static const int kLightStride = 24;
static std::shared_ptr<Light> make_non_area_light(const float* l) {
    Spectrum L(l[1], l[2], l[3]);
    switch ((int)l[0]) {
        case 2: return std::make_shared<PointLight>(Point3f(l[8], l[9], l[10]), L);
        case 3: return std::make_shared<SpotLight>(Point3f(l[8], l[9], l[10]), L, l[11], l[12], l + 13);
        case 4: return std::make_shared<DistantLight>(L, Vector3f(l[8], l[9], l[10]));
        default: return std::make_shared<InfiniteAreaLight>(L, (int)l[6]);
    }
}

extern "C" {

// materials: n_mat x 8 floats {type, kd.r, kd.g, kd.b, kt.r, kt.g, kt.b, eta}
// lights:    n_light x 24 floats in PbrtLight's field order {type (0 diffuse area, 1 infinite, 2 point, 3 spot,
//            4 distant), L rgb, tri, two_sided, n_samples, 0, pos xyz, cos_total_width, cos_falloff_start,
//            world_to_light 3x3, 0, 0}
void* orc_scene_create_with_spheres(const float* positions, int n_verts, const int32_t* indices, int n_tris,
                                    const float* normals, const float* uvs, const int32_t* tri_material,
                                    const float* materials, int n_mat, const int32_t* tri_light, const float* lights,
                                    int n_light, const float* spheres, int n_spheres, int max_prims_in_node,
                                    int split_method, uint32_t quirks);
void* orc_scene_create(const float* positions, int n_verts, const int32_t* indices, int n_tris,
                       const float* normals, const float* uvs, const int32_t* tri_material,
                       const float* materials, int n_mat, const int32_t* tri_light, const float* lights,
                       int n_light, int max_prims_in_node, int split_method, uint32_t quirks) {
    return orc_scene_create_with_spheres(positions, n_verts, indices, n_tris, normals, uvs, tri_material, materials,
                                         n_mat, tri_light, lights, n_light, nullptr, 0, max_prims_in_node, split_method,
                                         quirks);
}
// spheres: n_spheres x 8 floats {cx, cy, cz, radius, material, light (-1 = none), 0, 0}; sphere i is
// primitive n_tris + i (config 1: src/shapes/sphere.rs). A light's `tri` field may name a sphere primitive.
void* orc_scene_create_with_spheres(const float* positions, int n_verts, const int32_t* indices, int n_tris,
                                    const float* normals, const float* uvs, const int32_t* tri_material,
                                    const float* materials, int n_mat, const int32_t* tri_light, const float* lights,
                                    int n_light, const float* spheres, int n_spheres, int max_prims_in_node,
                                    int split_method, uint32_t quirks) {
    OracleScene* os = new OracleScene();
    auto mesh = std::make_shared<TriangleMesh>();
    mesh->n_vertices = n_verts;
    mesh->n_triangles = n_tris;
    mesh->p.resize(n_verts);
    for (int i = 0; i < n_verts; ++i) mesh->p[i] = Point3f(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
    mesh->vertex_indices.assign(indices, indices + 3 * (size_t)n_tris);
    if (normals) {
        mesh->n.resize(n_verts);
        for (int i = 0; i < n_verts; ++i) mesh->n[i] = Normal3f(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
    }
    if (uvs) {
        mesh->uv.resize(n_verts);
        for (int i = 0; i < n_verts; ++i) mesh->uv[i] = Point2f(uvs[2 * i], uvs[2 * i + 1]);
    }
    os->mesh = mesh;
    Scene& sc = os->scene;
    sc.quirks = quirks;
    for (int i = 0; i < n_mat; ++i) {
        const float* m = materials + 8 * i;
        MaterialDesc d;
        d.type = (int)m[0];
        d.kd = Spectrum(m[1], m[2], m[3]);
        d.kt = Spectrum(m[4], m[5], m[6]);
        d.eta = m[7];
        sc.materials.push_back(d);
    }
    const int n_prims = n_tris + n_spheres;
    std::vector<std::shared_ptr<Shape>> shapes(n_prims);
    for (int i = 0; i < n_tris; ++i) shapes[i] = std::make_shared<Triangle>(mesh, i, false, quirks);
    for (int i = 0; i < n_spheres; ++i) {
        const float* sp = spheres + 8 * i;
        shapes[n_tris + i] = Sphere::at(Point3f(sp[0], sp[1], sp[2]), sp[3]);
    }
    for (int i = 0; i < n_light; ++i) {
        const float* l = lights + kLightStride * i;
        if ((int)l[0] == 0)
            sc.lights.push_back(std::make_shared<DiffuseAreaLight>(Spectrum(l[1], l[2], l[3]), (int)l[6], shapes[(int)l[4]],
                                                                   l[5] != 0.0f));
        else
            sc.lights.push_back(make_non_area_light(l));
    }
    std::vector<std::shared_ptr<Primitive>> prims(n_prims);
    sc.prim_material.resize(n_prims);
    sc.prim_light.resize(n_prims);
    for (int i = 0; i < n_prims; ++i) {
        if (i < n_tris) {
            sc.prim_material[i] = tri_material ? tri_material[i] : 0;
            sc.prim_light[i] = tri_light ? tri_light[i] : -1;
        } else {
            sc.prim_material[i] = (int)spheres[8 * (i - n_tris) + 4];
            sc.prim_light[i] = (int)spheres[8 * (i - n_tris) + 5];
        }
        prims[i] = std::make_shared<GeometricPrimitive>(shapes[i], sc.prim_material[i], sc.prim_light[i], i);
    }
    sc.aggregate = std::make_shared<BVHAccel>(prims, max_prims_in_node, (SplitMethod)split_method, quirks);
    sc.finish();
    return os;
}
// Instanced scene (config 5): one base mesh in object space inside a BVHAccel (BLAS), n_inst
// TransformedPrimitives of it (primitive.rs:105-177) inside a top-level BVHAccel.
// instances: n_inst x 32 floats {to_world[16], to_object[16]} row-major; inst_material[i] = material of instance i.
// tri_material: material of every base triangle (nullptr = material 0), used where an instance carries no override (-1)
void* orc_scene_create_instanced(const float* positions, int n_verts, const int32_t* indices, int n_tris, const int32_t* tri_material,
                                 const float* instances, const int32_t* inst_material, int n_inst,
                                 const float* materials, int n_mat, const float* lights, int n_light,
                                 int max_prims_in_node, int split_method, uint32_t quirks) {
    OracleScene* os = new OracleScene();
    auto mesh = std::make_shared<TriangleMesh>();
    mesh->n_vertices = n_verts;
    mesh->n_triangles = n_tris;
    mesh->p.resize(n_verts);
    for (int i = 0; i < n_verts; ++i) mesh->p[i] = Point3f(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
    mesh->vertex_indices.assign(indices, indices + 3 * (size_t)n_tris);
    os->mesh = mesh;
    Scene& sc = os->scene;
    sc.quirks = quirks;
    for (int i = 0; i < n_mat; ++i) {
        const float* m = materials + 8 * i;
        MaterialDesc d;
        d.type = (int)m[0];
        d.kd = Spectrum(m[1], m[2], m[3]);
        d.kt = Spectrum(m[4], m[5], m[6]);
        d.eta = m[7];
        sc.materials.push_back(d);
    }
    for (int i = 0; i < n_light; ++i) {
        const float* l = lights + kLightStride * i;
        if ((int)l[0] == 0) continue;  // instanced primitives cannot be area lights
        sc.lights.push_back(make_non_area_light(l));
    }
    std::vector<std::shared_ptr<Primitive>> prims(n_tris);
    sc.prim_material.assign(n_tris, 0);
    sc.prim_light.assign(n_tris, -1);
    for (int i = 0; i < n_tris; ++i) {
        const int mat = tri_material ? tri_material[i] : 0;
        sc.prim_material[i] = mat;
        prims[i] = std::make_shared<GeometricPrimitive>(std::make_shared<Triangle>(mesh, i, false, quirks), mat, -1, i);
    }
    os->blas = std::make_shared<BVHAccel>(prims, max_prims_in_node, (SplitMethod)split_method, quirks);
    os->blas->counts_rays = false;
    std::vector<std::shared_ptr<Primitive>> insts(n_inst);
    sc.instance_material.resize(n_inst);
    os->to_object.resize(n_inst);
    for (int i = 0; i < n_inst; ++i) {
        Matrix4 tw, to;
        std::memcpy(tw.m, instances + 32 * (size_t)i, 64);
        std::memcpy(to.m, instances + 32 * (size_t)i + 16, 64);
        os->to_object[i] = to;
        insts[i] = std::make_shared<TransformedPrimitive>(os->blas, tw, to, i);
        sc.instance_material[i] = inst_material ? inst_material[i] : -1;
    }
    sc.aggregate = std::make_shared<BVHAccel>(insts, max_prims_in_node, (SplitMethod)split_method, quirks);
    sc.aggregate->leaves_are_instances = true;
    sc.finish();
    return os;
}
// The general two-level scene (primitive.rs:105-159, 33-103). The mesh is the COMBINED one: every object's triangles,
// object after object (obj_tri_offset[k] .. obj_tri_offset[k+1]), then n_world_tris world-space triangles; indices already
// point into the combined vertex list. Top-level primitives: the instances (instance i of object inst_object[i]), then the
// world-space triangles; only those may carry area lights (tri_light). prim ids are combined-mesh triangle indices.
void* orc_scene_create_two_level(const float* positions, int n_verts, const int32_t* indices, int n_tris,
                                 const int32_t* obj_tri_offset, int n_objects, int n_world_tris, const int32_t* tri_material,
                                 const int32_t* tri_light, const float* instances, const int32_t* inst_object,
                                 const int32_t* inst_material, int n_inst, const float* materials, int n_mat, const float* lights,
                                 int n_light, int max_prims_in_node, int split_method, uint32_t quirks) {
    OracleScene* os = new OracleScene();
    auto mesh = std::make_shared<TriangleMesh>();
    mesh->n_vertices = n_verts;
    mesh->n_triangles = n_tris;
    mesh->p.resize(n_verts);
    for (int i = 0; i < n_verts; ++i) mesh->p[i] = Point3f(positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]);
    mesh->vertex_indices.assign(indices, indices + 3 * (size_t)n_tris);
    os->mesh = mesh;
    Scene& sc = os->scene;
    sc.quirks = quirks;
    for (int i = 0; i < n_mat; ++i) {
        const float* m = materials + 8 * i;
        MaterialDesc d;
        d.type = (int)m[0];
        d.kd = Spectrum(m[1], m[2], m[3]);
        d.kt = Spectrum(m[4], m[5], m[6]);
        d.eta = m[7];
        sc.materials.push_back(d);
    }
    const int world0 = obj_tri_offset[n_objects];
    sc.prim_material.assign(n_tris, 0);
    sc.prim_light.assign(n_tris, -1);
    std::vector<std::shared_ptr<Shape>> shapes(n_tris);
    for (int i = 0; i < n_tris; ++i) {
        shapes[i] = std::make_shared<Triangle>(mesh, i, false, quirks);
        sc.prim_material[i] = tri_material ? tri_material[i] : 0;
        if (i >= world0 && tri_light) sc.prim_light[i] = tri_light[i];
    }
    for (int i = 0; i < n_light; ++i) {
        const float* l = lights + kLightStride * i;
        if ((int)l[0] == 0) {  // DiffuseAreaLight on world triangle l.prim
            int prim = world0 + (int)l[4];
            sc.lights.push_back(std::make_shared<DiffuseAreaLight>(Spectrum(l[1], l[2], l[3]), (int)l[6], shapes[prim], l[5] != 0.0f));
        } else {
            sc.lights.push_back(make_non_area_light(l));
        }
    }
    os->blases.resize(n_objects);
    for (int k = 0; k < n_objects; ++k) {
        std::vector<std::shared_ptr<Primitive>> prims;
        for (int i = obj_tri_offset[k]; i < obj_tri_offset[k + 1]; ++i)
            prims.push_back(std::make_shared<GeometricPrimitive>(shapes[i], sc.prim_material[i], -1, i));
        os->blases[k] = std::make_shared<BVHAccel>(prims, max_prims_in_node, (SplitMethod)split_method, quirks);
        os->blases[k]->counts_rays = false;
    }
    os->blas = os->blases[0];
    std::vector<std::shared_ptr<Primitive>> top;
    sc.instance_material.resize(n_inst);
    os->to_object.resize(n_inst);
    for (int i = 0; i < n_inst; ++i) {
        Matrix4 tw, to;
        std::memcpy(tw.m, instances + 32 * (size_t)i, 64);
        std::memcpy(to.m, instances + 32 * (size_t)i + 16, 64);
        os->to_object[i] = to;
        top.push_back(std::make_shared<TransformedPrimitive>(os->blases[inst_object ? inst_object[i] : 0], tw, to, i));
        sc.instance_material[i] = inst_material ? inst_material[i] : -1;
    }
    for (int i = world0; i < n_tris; ++i)
        top.push_back(std::make_shared<GeometricPrimitive>(shapes[i], sc.prim_material[i], sc.prim_light[i], i));
    sc.aggregate = std::make_shared<BVHAccel>(top, max_prims_in_node, (SplitMethod)split_method, quirks);
    sc.aggregate->leaves_are_instances = true;  // per primitive: TransformedPrimitives count as instance tests
    sc.finish();
    return os;
}
int orc_scene_num_object_nodes(void* h, int k) { return (int)((OracleScene*)h)->blases[k]->nodes.size(); }
void orc_scene_get_object(void* h, int k, void* nodes_out, int32_t* order_out) {
    auto& b = ((OracleScene*)h)->blases[k];
    std::memcpy(nodes_out, b->nodes.data(), b->nodes.size() * sizeof(LinearBVHNode));
    std::memcpy(order_out, b->ordered_prims.data(), b->ordered_prims.size() * sizeof(int32_t));
}
int orc_scene_num_blas_nodes(void* h) {
    OracleScene* os = (OracleScene*)h;
    return os->blas ? (int)os->blas->nodes.size() : 0;
}
void orc_scene_get_blas(void* h, void* nodes_out, int32_t* order_out) {
    OracleScene* os = (OracleScene*)h;
    std::memcpy(nodes_out, os->blas->nodes.data(), os->blas->nodes.size() * sizeof(LinearBVHNode));
    std::memcpy(order_out, os->blas->ordered_prims.data(), os->blas->ordered_prims.size() * sizeof(int32_t));
}
void orc_scene_destroy(void* h) { delete (OracleScene*)h; }
int orc_scene_num_nodes(void* h) { return (int)((OracleScene*)h)->scene.aggregate->nodes.size(); }
void orc_scene_get_nodes(void* h, void* out) {
    auto& n = ((OracleScene*)h)->scene.aggregate->nodes;
    std::memcpy(out, n.data(), n.size() * sizeof(LinearBVHNode));
}
void orc_scene_get_prim_order(void* h, int32_t* out) {
    auto& o = ((OracleScene*)h)->scene.aggregate->ordered_prims;
    std::memcpy(out, o.data(), o.size() * sizeof(int32_t));
}

// rays: n x 8 floats {o.xyz, d.xyz, t_max, time}; counters: {rays, node_tests, prim_tests}
void orc_intersect(void* h, const float* rays, int64_t n, void* out_hits, uint64_t* counters, int n_threads) {
    const Scene& sc = ((OracleScene*)h)->scene;
    OrcHit* out = (OrcHit*)out_hits;
    std::vector<TraversalCounters> tc(std::max(1, n_threads));
    parallel_chunks(n, n_threads, [&](int tid, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            const float* r = rays + 8 * i;
            Ray ray(Point3f(r[0], r[1], r[2]), Vector3f(r[3], r[4], r[5]), r[6], r[7]);
            // re-run the winning triangle's test to report its barycentrics (the SurfaceInteraction
            // does not keep them)
            SurfaceInteraction si;
            OrcHit hit = {FLOAT_INF, 0, 0, 0, -1, {-1, 0, 0}};
            if (sc.intersect(ray, &si, &tc[tid])) {
                hit.t = ray.t_max;
                hit.prim_id = si.prim_id;
                hit.pad[0] = si.instance_id;
                const TriangleMesh& m = *((OracleScene*)h)->mesh;
                if (si.prim_id >= m.n_triangles) {  // a sphere: no barycentrics
                    out[i] = hit;
                    continue;
                }
                const int32_t* v = &m.vertex_indices[3 * (size_t)si.prim_id];
                Ray r0(Point3f(r[0], r[1], r[2]), Vector3f(r[3], r[4], r[5]), r[6], r[7]);
                if (si.instance_id >= 0) r0 = xform_ray(((OracleScene*)h)->to_object[si.instance_id], r0);
                TriHit th = triangle_intersect_test(m.p[v[0]], m.p[v[1]], m.p[v[2]], r0, sc.quirks);
                hit.b0 = th.b0;
                hit.b1 = th.b1;
                hit.b2 = th.b2;
            }
            out[i] = hit;
        }
    });
    if (counters) {
        TraversalCounters s;
        for (auto& c : tc) s.add(c);
        counters[0] = s.rays;
        counters[1] = s.node_tests;
        counters[2] = s.prim_tests;
        counters[3] = s.inst_tests;
    }
}
// Closest hit WITHOUT an aggregate: every triangle of the mesh (of every instance, if `to_object` holds n_instances 4 x 4
// world-to-object matrices) is put to Triangle::intersect's test with the ray's OWN t_max, and the smallest t wins — the
// order-free definition BVHAccel::intersect (bvh.rs:828-879) must agree with, whatever its tree and its walk look like.
// out_t[i] (inf: miss), out_prim[i] / out_inst[i] = the first triangle / instance reaching it, out_ties[i] = how many reach exactly it.
void orc_brute_force(const float* positions, const int32_t* indices, int n_tris, const float* to_object, int n_instances, const float* rays,
                     int64_t n, float* out_t, int32_t* out_prim, int32_t* out_inst, int32_t* out_ties, int n_threads) {
    parallel_chunks(n, n_threads, [&](int, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            const float* r = rays + 8 * i;
            const Ray world(Point3f(r[0], r[1], r[2]), Vector3f(r[3], r[4], r[5]), r[6], r[7]);
            float best = FLOAT_INF;
            int32_t prim = -1, inst = -1, ties = 0;
            for (int k = 0; k < std::max(1, n_instances); ++k) {
                Ray ray = world;
                if (n_instances > 0) {
                    Matrix4 m;
                    std::memcpy(m.m, to_object + 16 * (size_t)k, sizeof(m.m));
                    ray = xform_ray(m, world);  // TransformedPrimitive::intersect (primitive.rs:136-145): t is kept
                }
                for (int t = 0; t < n_tris; ++t) {
                    const int32_t* v = indices + 3 * (size_t)t;
                    auto P = [&](int32_t a) { return Point3f(positions[3 * (size_t)a], positions[3 * (size_t)a + 1], positions[3 * (size_t)a + 2]); };
                    const TriHit h = triangle_intersect_test(P(v[0]), P(v[1]), P(v[2]), ray, 0);
                    if (!h.hit) continue;
                    if (h.t < best) best = h.t, prim = t, inst = n_instances > 0 ? k : -1, ties = 1;
                    else if (h.t == best) ++ties;
                }
            }
            out_t[i] = best, out_prim[i] = prim, out_inst[i] = inst, out_ties[i] = ties;
        }
    });
}
// Instrumentation for the layout study of the wide records (tools/wide_fill_study.py): for a single-level scene, how many of
// these rays pass the box test of every node of BVHAccel::intersect's walk (any == 0) or intersect_p's (any != 0).
void orc_node_visits(void* h, const float* rays, int64_t n, int any, uint64_t* entered_per_node) {
    const Scene& sc = ((OracleScene*)h)->scene;
    TraversalCounters tc;
    tc.node_entered = entered_per_node;
    for (int64_t i = 0; i < n; ++i) {
        const float* r = rays + 8 * i;
        Ray ray(Point3f(r[0], r[1], r[2]), Vector3f(r[3], r[4], r[5]), r[6], r[7]);
        SurfaceInteraction si;
        if (any)
            (void)sc.intersect_p(ray, &tc);
        else
            (void)sc.intersect(ray, &si, &tc);
    }
}
void orc_intersect_p(void* h, const float* rays, int64_t n, uint8_t* out, uint64_t* counters, int n_threads) {
    const Scene& sc = ((OracleScene*)h)->scene;
    std::vector<TraversalCounters> tc(std::max(1, n_threads));
    parallel_chunks(n, n_threads, [&](int tid, int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i) {
            const float* r = rays + 8 * i;
            Ray ray(Point3f(r[0], r[1], r[2]), Vector3f(r[3], r[4], r[5]), r[6], r[7]);
            out[i] = sc.intersect_p(ray, &tc[tid]) ? 1 : 0;
        }
    });
    if (counters) {
        TraversalCounters s;
        for (auto& c : tc) s.add(c);
        counters[0] = s.rays;
        counters[1] = s.node_tests;
        counters[2] = s.prim_tests;
        counters[3] = s.inst_tests;
    }
}

// camera: 37 floats {camera_to_world[16] row-major, raster_to_camera[16], lens_radius, focal_distance, shutter_open,
// shutter_close, kind (0 perspective, 1 orthographic, 2 environment)}
// integrator: 0 = path, 1 = direct lighting, 2 = Whitted, 3 = ambient occlusion.
// light_strategy: path {0 uniform, 1 power, 2 spatial}; direct {0 all, 1 one}; AO {0 uniform hemisphere, 1 cosine};
// for AO max_depth carries n_samples.
// stats: {rays, node_tests, prim_tests, camera_samples, nanoseconds, inst_tests}
void orc_render_filtered(void* h, const float* cam, int integrator, int max_depth, float rr_threshold,
                         int light_strategy, int spp, uint64_t seed, int width, int height, int x0, int y0, int x1,
                         int y1, int n_threads, float filter_rx, float filter_ry, const float* filter_table256,
                         const int32_t* sampler5, float* film_out, uint64_t* stats);
void orc_render(void* h, const float* cam, int integrator, int max_depth, float rr_threshold, int light_strategy,
                int spp, uint64_t seed, int width, int height, int x0, int y0, int x1, int y1, int n_threads,
                float* film_out, uint64_t* stats) {
    orc_render_filtered(h, cam, integrator, max_depth, rr_threshold, light_strategy, spp, seed, width, height, x0, y0,
                        x1, y1, n_threads, 0.5f, 0.5f, nullptr, nullptr, film_out, stats);
}
// filter_table256 = Film::filter_table (film.rs:52-63), nullptr = box
// sampler5 = {kind (0 random, 1 stratified, 2 (0,2)-sequence, 3 Halton), x_samples, y_samples, jitter,
// n_sampled_dimensions, float bits of Film::max_sample_luminance (0 = infinity)}
// or nullptr = random; the samples per pixel become x*y (stratified) / the next power of two ((0,2)).
void orc_render_filtered(void* h, const float* cam, int integrator, int max_depth, float rr_threshold,
                         int light_strategy, int spp, uint64_t seed, int width, int height, int x0, int y0, int x1,
                         int y1, int n_threads, float filter_rx, float filter_ry, const float* filter_table256,
                         const int32_t* sampler5, float* film_out, uint64_t* stats) {
    const Scene& sc = ((OracleScene*)h)->scene;
    PerspectiveCamera camera;
    std::memcpy(camera.camera_to_world.m, cam, 64);
    std::memcpy(camera.raster_to_camera.m, cam + 16, 64);
    camera.lens_radius = cam[32];
    camera.focal_distance = cam[33];
    camera.shutter_open = cam[34];
    camera.shutter_close = cam[35];
    camera.kind = (int)cam[36];  // 0 perspective, 1 orthographic, 2 environment
    camera.film_width = width;
    camera.film_height = height;
    Film film(width, height);
    film.filter_radius_x = filter_rx;
    film.filter_radius_y = filter_ry;
    if (filter_table256) std::memcpy(film.filter_table, filter_table256, sizeof(film.filter_table));
    RenderParams rp;
    rp.spp = spp;
    rp.seed = seed;
    rp.x0 = x0;
    rp.y0 = y0;
    rp.x1 = x1;
    rp.y1 = y1;
    rp.n_threads = n_threads;
    if (sampler5) {
        rp.sampler.kind = sampler5[0];
        rp.sampler.nx = sampler5[1];
        rp.sampler.ny = sampler5[2];
        rp.sampler.jitter = sampler5[3] != 0;
        rp.sampler.n_dims = sampler5[4];
        float max_lum;  // sixth entry: the bits of Film::max_sample_luminance, 0 = infinity
        std::memcpy(&max_lum, &sampler5[5], 4);
        if (max_lum > 0.0f) film.max_sample_luminance = max_lum;
    }
    RenderStats st;
    std::unique_ptr<Integrator> integ;
    if (integrator == 0)
        integ.reset(new PathIntegrator(max_depth, rr_threshold,
                                       light_strategy == 0 ? "uniform" : (light_strategy == 1 ? "power" : "spatial")));
    else if (integrator == 1)
        integ.reset(new DirectLightingIntegrator((LightStrategy)light_strategy, max_depth));
    else if (integrator == 2)
        integ.reset(new WhittedIntegrator(max_depth));
    else
        integ.reset(new AOIntegrator(light_strategy != 0, max_depth));  // AO: max_depth carries n_samples
    auto t0 = std::chrono::steady_clock::now();
    render(sc, camera, *integ, film, rp, &st);
    auto t1 = std::chrono::steady_clock::now();
    std::memcpy(film_out, film.pixels.data(), film.pixels.size() * sizeof(float));
    if (stats) {
        stats[0] = st.ctr.rays;
        stats[1] = st.ctr.node_tests;
        stats[2] = st.ctr.prim_tests;
        stats[3] = st.camera_samples;
        stats[4] = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
        stats[5] = st.ctr.inst_tests;
    }
}

// Integrator::li (integrator.rs:29-42) for n caller-supplied rays: rays = n x {o.xyz, d.xyz, t_max, time}, keys[i] =
// RNG::set_sequence argument of ray i's RandomSampler (rng.rs:21-35), `skip` values already drawn from it before li
// (render draws the CameraSample first, integrator.rs:430). rgb = n x 3, the returned Spectrum unguarded.
// stats: {rays, node_tests, prim_tests}
void orc_li(void* h, int integrator, int max_depth, float rr_threshold, int light_strategy, const float* rays,
            const uint64_t* keys, int64_t n, int skip, float* rgb, uint64_t* stats) {
    const Scene& sc = ((OracleScene*)h)->scene;
    std::unique_ptr<Integrator> integ;
    if (integrator == 0)
        integ.reset(new PathIntegrator(max_depth, rr_threshold,
                                       light_strategy == 0 ? "uniform" : (light_strategy == 1 ? "power" : "spatial")));
    else if (integrator == 1)
        integ.reset(new DirectLightingIntegrator((LightStrategy)light_strategy, max_depth));
    else if (integrator == 2)
        integ.reset(new WhittedIntegrator(max_depth));
    else
        integ.reset(new AOIntegrator(light_strategy != 0, max_depth));  // AO: max_depth carries n_samples
    integ->pre_process(sc);
    SamplerSpec spec;  // RandomSampler
    integ->request_samples(sc, spec);
    RenderCtx rc;
    rc.sampler.spec = &spec;
    for (int64_t i = 0; i < n; ++i) {
        const float* r = rays + 8 * i;
        Ray ray;
        ray.o = Point3f(r[0], r[1], r[2]);
        ray.d = Vector3f(r[3], r[4], r[5]);
        ray.t_max = r[6];
        ray.time = r[7];
        // one stream per call: start_sample's set_sequence with the caller's key
        rc.sampler.start_sample(keys[i], 0, 1, 0);
        for (int k = 0; k < skip; ++k) (void)rc.sampler.get_1d();
        Spectrum l = integ->li(ray, sc, rc, 0);
        for (int k = 0; k < 3; ++k) rgb[3 * i + k] = l.c[k];
    }
    if (stats) {
        stats[0] = rc.ctr.rays;
        stats[1] = rc.ctr.node_tests;
        stats[2] = rc.ctr.prim_tests;
    }
}

// TriangleMesh.s (triangle.rs:21): per-vertex tangents, n_verts x 3; call before rendering
void orc_scene_set_tangents(void* h, const float* tangents, int n_verts) {
    OracleScene* os = (OracleScene*)h;
    os->mesh->s.resize(n_verts);
    for (int i = 0; i < n_verts; ++i) os->mesh->s[i] = Vector3f(tangents[3 * i], tangents[3 * i + 1], tangents[3 * i + 2]);
}

// HaltonSampler probe: out = {index of pixel (px, py)'s sample_num-th sample as double, sample_dimension(index, dim),
// unscrambled radical inverse of the index in base 2 and in base 3, base_scales x, y}
void orc_halton_probe(int res_x, int res_y, int px, int py, int sample_num, int dim, double* out6) {
    HaltonSetup h;
    h.init(res_x, res_y);
    int64_t index = h.offset_for_pixel(px, py) + (int64_t)sample_num * h.sample_stride;
    out6[0] = (double)index;
    out6[1] = h.sample_dimension(index, dim);
    out6[2] = radical_inverse(0, (uint64_t)index);
    out6[3] = radical_inverse(1, (uint64_t)index);
    out6[4] = h.base_scales[0];
    out6[5] = h.base_scales[1];
}
// PixelSampler tables of ONE pixel after start_pixel (stratified.rs:44-104, zerotwosequence.rs:28-60): out_1d[d * spp + s],
// out_2d[(d * spp + s) * 2 + c] for the n_dims tabulated dimensions; spec5 as oracle.sampler_spec. Returns the samples per pixel
// the sampler takes (nx * ny / the next power of two), 0 for a sampler without tables.
int orc_sampler_tables(const int32_t* spec5, int spp_requested, uint64_t seed, int64_t pixel_index, float* out_1d, float* out_2d, int cap_spp) {
    SamplerSpec spec;
    spec.kind = spec5[0], spec.nx = spec5[1], spec.ny = spec5[2], spec.jitter = spec5[3] != 0, spec.n_dims = spec5[4];
    if (spec.kind != SAMPLER_STRATIFIED && spec.kind != SAMPLER_ZEROTWO) return 0;
    const int spp = (int)spec.samples_per_pixel(spp_requested);
    if (spp > cap_spp) return spp;
    Sampler smp;
    smp.spec = &spec;
    smp.start_pixel(seed, pixel_index, spp);
    for (int d = 0; d < spec.n_dims; ++d)
        for (int k = 0; k < spp; ++k) {
            out_1d[(size_t)d * spp + k] = smp.samples_1d[d][k];
            out_2d[((size_t)d * spp + k) * 2] = smp.samples_2d[d][k].x;
            out_2d[((size_t)d * spp + k) * 2 + 1] = smp.samples_2d[d][k].y;
        }
    return spp;
}
// digit permutation of the base-th prime (compute_radical_inverse_permutations), n = that prime
int orc_halton_permutation(int base_index, int32_t* out, int cap) {
    const PrimeTables& t = prime_tables();
    int n = (int)t.primes[base_index];
    for (int i = 0; i < n && i < cap; ++i) out[i] = t.perms[t.prime_sums[base_index] + i];
    return n;
}

void orc_filter_table(int type, float rx, float ry, float a, float b, float* table256) {
    filter_table(type, rx, ry, a, b, table256);
}
void orc_sample_bounds(int width, int height, float rx, float ry, int32_t* out4) {
    Film f(width, height);
    f.filter_radius_x = rx;
    f.filter_radius_y = ry;
    int x0, y0, x1, y1;
    f.sample_bounds(&x0, &y0, &x1, &y1);
    out4[0] = x0; out4[1] = y0; out4[2] = x1; out4[3] = y1;
}

// ---- unit entry points for known-answer tests ----
void orc_triangle_test(const float* p0, const float* p1, const float* p2, const float* ray8, uint32_t quirks,
                       float* out5) {
    Ray ray(Point3f(ray8[0], ray8[1], ray8[2]), Vector3f(ray8[3], ray8[4], ray8[5]), ray8[6], ray8[7]);
    TriHit h = triangle_intersect_test(Point3f(p0[0], p0[1], p0[2]), Point3f(p1[0], p1[1], p1[2]),
                                       Point3f(p2[0], p2[1], p2[2]), ray, quirks);
    out5[0] = h.hit ? 1.0f : 0.0f;
    out5[1] = h.b0;
    out5[2] = h.b1;
    out5[3] = h.b2;
    out5[4] = h.t;
}
int orc_bounds_intersect_p(const float* b6, const float* ray8, uint32_t quirks) {
    Bounds3f b;
    b.min = Point3f(b6[0], b6[1], b6[2]);
    b.max = Point3f(b6[3], b6[4], b6[5]);
    Ray ray(Point3f(ray8[0], ray8[1], ray8[2]), Vector3f(ray8[3], ray8[4], ray8[5]), ray8[6], ray8[7]);
    Vector3f inv_dir(1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z);
    int dir_is_neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
    return bounds_intersect_p(b, ray, inv_dir, dir_is_neg, quirks) ? 1 : 0;
}
void orc_offset_ray_origin(const float* p, const float* err, const float* n, const float* w, float* out3) {
    Point3f r = offset_ray_origin(Point3f(p[0], p[1], p[2]), Vector3f(err[0], err[1], err[2]),
                                  Normal3f(n[0], n[1], n[2]), Vector3f(w[0], w[1], w[2]));
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}
float orc_next_float_up(float v) { return next_float_up(v); }
float orc_next_float_down(float v) { return next_float_down(v); }
float orc_gamma(float n) { return gamma(n); }
void orc_pcg32(uint64_t sequence, int n, uint32_t* out_u32, float* out_f32) {
    RNG a(sequence), b(sequence);
    for (int i = 0; i < n; ++i) {
        if (out_u32) out_u32[i] = a.uniform_u32();
        if (out_f32) out_f32[i] = b.uniform_float();
    }
}
// elementary functions, batch form: op 0 sin, 1 cos, 2 acos, 3 atan2(x = y-arg, y = x-arg)
void orc_elementary(int op, const float* x, const float* y, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        switch (op) {
            case 0: out[i] = det_sin(x[i]); break;
            case 1: out[i] = det_cos(x[i]); break;
            case 2: out[i] = det_acos(x[i]); break;
            default: out[i] = det_atan2(x[i], y[i]); break;
        }
    }
}
// op 0 concentric_sample_disk (2), 1 cosine_sample_hemisphere (3), 2 uniform_sample_triangle (2)
void orc_sample(int op, const float* u2, int64_t n, uint32_t quirks, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        Point2f u(u2[2 * i], u2[2 * i + 1]);
        if (op == 0) {
            Point2f d = concentric_sample_disk(u);
            out[2 * i] = d.x;
            out[2 * i + 1] = d.y;
        } else if (op == 1) {
            Vector3f w = cosine_sample_hemisphere(u, quirks);
            out[3 * i] = w.x;
            out[3 * i + 1] = w.y;
            out[3 * i + 2] = w.z;
        } else {
            Point2f b = uniform_sample_triangle(u);
            out[2 * i] = b.x;
            out[2 * i + 1] = b.y;
        }
    }
}
// Sphere::intersect_test (sphere.rs:228-284) of a full sphere placed by a translation: out[0] = hit, out[1..3] = t.v, t.low,
// t.high of the EFloat root it chose, out[4..10] = the object-space ray it solved for (o, d, t_max)
void orc_sphere_test(const float* centre, float radius, const float* ray8, float* out11) {
    auto sp = Sphere::at(Point3f(centre[0], centre[1], centre[2]), radius);
    Ray r(Point3f(ray8[0], ray8[1], ray8[2]), Vector3f(ray8[3], ray8[4], ray8[5]), ray8[6], 0.0f);
    Point3f ph;
    Float phi = 0.0f;
    Ray ro;
    EFloat t;
    const bool hit = sp->intersect_test(r, &ph, &phi, &ro, &t);
    for (int i = 0; i < 11; ++i) out11[i] = 0.0f;
    Vector3f oe, de;
    Ray rx = xform_ray_err(sp->world_to_object, r, &oe, &de);  // the object-space ray, hit or not
    out11[4] = rx.o.x, out11[5] = rx.o.y, out11[6] = rx.o.z;
    out11[7] = rx.d.x, out11[8] = rx.d.y, out11[9] = rx.d.z;
    out11[10] = rx.t_max;
    if (!hit) return;
    out11[0] = 1.0f;
    out11[1] = t.v;
    out11[2] = t.lower_bound();
    out11[3] = t.upper_bound();
}
float orc_fr_dielectric(float cos_theta_i, float eta_i, float eta_t) { return fr_dielectric(cos_theta_i, eta_i, eta_t); }
int orc_refract(const float* wi, const float* n, float eta, uint32_t quirks, float* wt) {
    Vector3f t;
    bool ok = refract(Vector3f(wi[0], wi[1], wi[2]), Normal3f(n[0], n[1], n[2]), eta, &t, quirks);
    wt[0] = t.x;
    wt[1] = t.y;
    wt[2] = t.z;
    return ok ? 1 : 0;
}
// BSDF local_to_world with an explicit frame (D36 KAT)
void orc_local_to_world(const float* ss, const float* ts, const float* ns, const float* v, uint32_t quirks,
                        float* out3) {
    SurfaceInteraction si;
    BSDF b(si, 1.0f, quirks);
    b.ss = Vector3f(ss[0], ss[1], ss[2]);
    b.ts = Vector3f(ts[0], ts[1], ts[2]);
    b.ns = Vector3f(ns[0], ns[1], ns[2]);
    Vector3f r = b.local_to_world(Vector3f(v[0], v[1], v[2]));
    out3[0] = r.x;
    out3[1] = r.y;
    out3[2] = r.z;
}

}  // extern "C"

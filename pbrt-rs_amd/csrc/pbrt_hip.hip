// pbrt_hip.hip — C ABI (include/pbrt_hip.h) over the gfx950 kernels.
//
// Stands in for the reference's Integrator / Primitive / Scene trait objects:
//   pbrt_hip_scene_create   Scene::new, src/core/scene.rs:18-34
//   pbrt_hip_intersect[_p]  Scene::intersect / intersect_p, src/core/scene.rs:40-46
//   pbrt_hip_render         SamplerIntegrator::render, src/core/integrator.rs:399-480
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scene.h"
#include "trace.h"
#include "wavefront.h"

using namespace pb;

static std::string g_create_error;

#define HIP_TRY(ctx, call)                                        \
    do {                                                          \
        if (!hip_ok((ctx), (call), #call)) return PBRT_HIP_ERR_DEVICE; \
    } while (0)

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
extern "C" int pbrt_hip_context_create(int device_id, PbrtHipContext** out) {
    if (!out) return PBRT_HIP_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_create_error = "no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback)";
        return PBRT_HIP_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n) {
        g_create_error = "device_id out of range";
        return PBRT_HIP_ERR_INVALID;
    }
    PbrtHipContext* ctx = new PbrtHipContext();
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) {
        g_create_error = "hipSetDevice failed";
        delete ctx;
        return PBRT_HIP_ERR_DEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) {
        g_create_error = "hipGetDeviceProperties failed";
        delete ctx;
        return PBRT_HIP_ERR_DEVICE;
    }
    ctx->n_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc((void**)&ctx->d_counters, 4 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_work_counter, kQueueSegments * sizeof(unsigned int)) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_counts, 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_sync, hipEventDisableTiming) != hipSuccess ||
        hipMemset(ctx->d_counters, 0, 4 * sizeof(unsigned long long)) != hipSuccess) {
        g_create_error = "stream / event creation failed";
        delete ctx;
        return PBRT_HIP_ERR_DEVICE;
    }
    *out = ctx;
    return PBRT_HIP_OK;
}

extern "C" void pbrt_hip_context_destroy(PbrtHipContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_work_counter) (void)hipFree(ctx->d_work_counter);
    for (auto& b : ctx->block_cache)
        if (b.ptr) (void)hipFree(b.ptr);
    ctx->block_cache.clear();
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->ev_sync) (void)hipEventDestroy(ctx->ev_sync);
    if (ctx->d_halton_primes) (void)hipFree(ctx->d_halton_primes);
    if (ctx->d_halton_perms) (void)hipFree(ctx->d_halton_perms);
    delete ctx;
}

extern "C" const char* pbrt_hip_last_error(const PbrtHipContext* ctx) {
    return ctx ? ctx->last_error.c_str() : g_create_error.c_str();
}

extern "C" int pbrt_hip_synchronize(PbrtHipContext* ctx) {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PBRT_HIP_OK;
}

extern "C" int pbrt_hip_trace_timing(PbrtHipContext* ctx, int reset, double* total_ms, uint64_t* launches) {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(ctx);
    if (total_ms) *total_ms = ctx->trace_ms;
    if (launches) *launches = ctx->trace_launches;
    if (reset) {
        ctx->trace_ms = 0.0;
        ctx->trace_launches = 0;
    }
    return PBRT_HIP_OK;
}

extern "C" int pbrt_hip_set_counting(PbrtHipContext* ctx, int enable) {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(ctx);
    ctx->count_traversal = enable != 0;
    return PBRT_HIP_OK;
}

extern "C" int pbrt_hip_get_counters(PbrtHipContext* ctx, int reset, uint64_t counters[4]) {
    if (!ctx || !counters) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(h, ctx->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    counters[0] = h[2];
    counters[1] = h[0];
    counters[2] = h[1];
    counters[3] = h[3];
    if (reset) {
        HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, sizeof(h)));
    }
    return PBRT_HIP_OK;
}

// ------------------------------------------------------------------------------------
// scene
// ------------------------------------------------------------------------------------
static float tri_area(const float* a, const float* b, const float* c) {  // triangle.rs:323-328
    float e1[3], e2[3];
    for (int k = 0; k < 3; ++k) {
        e1[k] = b[k] - a[k];
        e2[k] = c[k] - a[k];
    }
    float cx = e1[1] * e2[2] - e1[2] * e2[1];
    float cy = e1[2] * e2[0] - e1[0] * e2[2];
    float cz = e1[0] * e2[1] - e1[1] * e2[0];
    return std::sqrt(cx * cx + cy * cy + cz * cz) * 0.5f;
}

// Distribution1D::new (sampling.rs:69-93), intended (D40)
static void make_distribution(const std::vector<float>& f, std::vector<float>* cdf, float* func_int) {
    int n = (int)f.size();
    cdf->assign(n + 1, 0.0f);
    for (int i = 1; i < n + 1; ++i) (*cdf)[i] = (*cdf)[i - 1] + f[i - 1] / (float)n;
    *func_int = (*cdf)[n];
    if (*func_int == 0.0f) {
        for (int i = 1; i < n + 1; ++i) (*cdf)[i] = (float)i / (float)n;
    } else {
        for (int i = 1; i < n + 1; ++i) (*cdf)[i] /= *func_int;
    }
}

// host replica of det_sincos (dev_math.h) for the env light's sin-weighted table
static float host_det_sin(float x) {
    float q = x * 0.63661977236758134308f;
    float k = __builtin_rintf(q);
    float r = __builtin_fmaf(-k, 1.5707397460937500f, x);
    r = __builtin_fmaf(-k, 5.6579709053039550781e-05f, r);
    r = __builtin_fmaf(-k, 9.9209362947050294680e-10f, r);
    float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    float sr = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    float cr = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    int ki = (int)k & 3;
    return (ki == 0) ? sr : (ki == 1) ? cr : (ki == 2) ? -sr : -cr;
}

// Validates one reference-order LinearBVHNode array and converts it to the device records
// (trace.h): interior nodes get indices base + 0.. in DFS order, leaves are encoded with count_bits.
struct TreeLayout {
    std::vector<float> inodes;   // 16 floats per interior node
    int n_interior = 0, count_bits = 0, depth = 0;
    int32_t root_ref = 0;
};
static const char* convert_tree(const PbrtLinearBVHNode* nodes, int32_t n_nodes, int32_t n_prims, int32_t base,
                                TreeLayout* out) {
    std::vector<int32_t> interior_index(n_nodes, -1);
    int n_interior = 0, max_count = 1;
    int64_t leaf_prims = 0;
    for (int32_t i = 0; i < n_nodes; ++i) {
        const PbrtLinearBVHNode& nd = nodes[i];
        if (nd.n_primitives > 0) {
            if (nd.offset < 0 || (int64_t)nd.offset + nd.n_primitives > n_prims) return "leaf range outside the primitive list";
            max_count = std::max<int>(max_count, nd.n_primitives);
            leaf_prims += nd.n_primitives;
        } else {
            if (nd.axis > 2) return "interior node axis > 2";
            if (nd.offset <= i + 1 || nd.offset >= n_nodes || i + 1 >= n_nodes) return "second-child offset out of range";
            interior_index[i] = n_interior++;
        }
    }
    if (leaf_prims != n_prims) return "leaves do not cover the primitive list exactly once";
    {
        // the traversal stack holds one entry per level, 64 in all (bvh.rs:839 `nodes_to_visit = [0; 64]`, where the
        // reference would index out of bounds): refuse deeper trees instead of overrunning the spill slab
        std::vector<std::pair<int32_t, int>> st;  // (node, depth)
        st.emplace_back(0, 1);
        int max_depth = 0;
        int64_t visited = 0;
        while (!st.empty()) {
            auto [node, depth] = st.back();
            st.pop_back();
            if (++visited > n_nodes) return "node array is not a tree";
            max_depth = std::max(max_depth, depth);
            if (nodes[node].n_primitives == 0) {
                st.emplace_back(node + 1, depth + 1);
                st.emplace_back(nodes[node].offset, depth + 1);
            }
        }
        out->depth = max_depth;
    }
    if (n_interior != (n_nodes - 1) / 2 || (n_nodes & 1) == 0) return "node array is not a full binary tree";
    int count_bits = 0;
    while ((1 << count_bits) < max_count) ++count_bits;
    if (((int64_t)n_prims << count_bits) >= (1ll << 31)) return "scene too large for 31-bit leaf references";
    auto child_ref = [&](int32_t node) -> int32_t {
        const PbrtLinearBVHNode& nd = nodes[node];
        if (nd.n_primitives > 0) return ~(int32_t)(((uint32_t)nd.offset << count_bits) | (uint32_t)(nd.n_primitives - 1));
        return base + interior_index[node];
    };
    out->inodes.assign((size_t)n_interior * 16, 0.0f);
    for (int32_t i = 0; i < n_nodes; ++i) {
        if (interior_index[i] < 0) continue;
        const PbrtLinearBVHNode& c0 = nodes[i + 1];
        const PbrtLinearBVHNode& c1 = nodes[nodes[i].offset];
        float* r = &out->inodes[(size_t)interior_index[i] * 16];
        r[0] = c0.bounds_min[0]; r[1] = c0.bounds_min[1]; r[2] = c0.bounds_min[2];
        r[3] = c0.bounds_max[0]; r[4] = c0.bounds_max[1]; r[5] = c0.bounds_max[2];
        r[6] = c1.bounds_min[0]; r[7] = c1.bounds_min[1]; r[8] = c1.bounds_min[2];
        r[9] = c1.bounds_max[0]; r[10] = c1.bounds_max[1]; r[11] = c1.bounds_max[2];
        int32_t refs[4] = {child_ref(i + 1), child_ref(nodes[i].offset), (int32_t)nodes[i].axis, 0};
        std::memcpy(r + 12, refs, 16);
    }
    out->n_interior = n_interior;
    out->count_bits = count_bits;
    out->root_ref = child_ref(0);
    return nullptr;
}

// Spheres next to the triangles (src/shapes/sphere.rs; BASELINE config 1): full spheres placed by a translation.
// Primitive ids n_tris .. n_tris + n - 1 in prim_order; not usable as area lights on the device.
struct SphereArgs {
    const float* spheres = nullptr;  // n x {centre.xyz, radius}
    const int32_t* material = nullptr;
    const int32_t* light = nullptr;  // per sphere: index of its DiffuseAreaLight or -1
    int32_t n = 0;
};
struct InstancingArgs {
    const PbrtInstance* instances = nullptr;
    int32_t n_instances = 0;
    const PbrtLinearBVHNode* tlas_nodes = nullptr;
    int32_t n_tlas_nodes = 0;
    const int32_t* tlas_order = nullptr;
};

static int scene_create_impl(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                             int32_t n_tris, const int32_t* tri_material, const PbrtMaterial* materials,
                             int32_t n_materials, const int32_t* tri_light, const PbrtLight* lights, int32_t n_lights,
                             const PbrtLinearBVHNode* nodes, int32_t n_nodes, const int32_t* prim_order,
                             const InstancingArgs& ia, PbrtHipScene** out, pb::DeviceTree* dt = nullptr,
                             const SphereArgs& sa = SphereArgs());

extern "C" int pbrt_hip_scene_create(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                     const int32_t* indices, int32_t n_tris, const int32_t* tri_material,
                                     const PbrtMaterial* materials, int32_t n_materials, const int32_t* tri_light,
                                     const PbrtLight* lights, int32_t n_lights, const PbrtLinearBVHNode* nodes,
                                     int32_t n_nodes, const int32_t* prim_order, PbrtHipScene** out) {
    return scene_create_impl(ctx, positions, n_verts, indices, n_tris, tri_material, materials, n_materials, tri_light,
                             lights, n_lights, nodes, n_nodes, prim_order, InstancingArgs(), out);
}

namespace pb {
int hlbvh_build_scene_tree(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                           int32_t n_tris, const int32_t* tri_material, const int32_t* tri_light, const PbrtLight* lights,
                           int32_t n_lights, int32_t max_prims_in_node, DeviceTree* out);
}

// Scene::new with BVHAccel::new(HLBVH) built and laid out on the device: only the mesh goes up.
extern "C" int pbrt_hip_scene_create_hlbvh(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                           const int32_t* indices, int32_t n_tris, const int32_t* tri_material,
                                           const PbrtMaterial* materials, int32_t n_materials, const int32_t* tri_light,
                                           const PbrtLight* lights, int32_t n_lights, int32_t max_prims_in_node,
                                           PbrtHipScene** out, double* build_ms, double* layout_ms) {
    if (!ctx || !out) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(ctx);
    *out = nullptr;
    auto fail = [&](const char* msg) {
        ctx->last_error = msg;
        return PBRT_HIP_ERR_INVALID;
    };
    // the checks scene_create_impl makes before it touches the device, needed here before the build
    if (n_tris <= 0 || n_verts <= 0) return fail("empty scene: n_tris and n_verts must be > 0");
    if (!positions || !indices) return fail("null geometry pointer");
    if (n_materials <= 0 || !materials) return fail("at least one material is required");
    if (n_lights < 0 || (n_lights > 0 && !lights)) return fail("bad light table");
    for (int64_t i = 0; i < 3 * (int64_t)n_tris; ++i)
        if (indices[i] < 0 || indices[i] >= n_verts) return fail("vertex index out of range");
    for (int32_t i = 0; i < n_lights; ++i)
        if (lights[i].type == PBRT_LIGHT_DIFFUSE_AREA && (lights[i].prim < 0 || lights[i].prim >= n_tris))
            return fail("area light triangle out of range");
    pb::DeviceTree dt;
    int rc = pb::hlbvh_build_scene_tree(ctx, positions, n_verts, indices, n_tris, tri_material, tri_light, lights, n_lights,
                                        max_prims_in_node, &dt);
    if (rc != PBRT_HIP_OK) return rc;
    if (build_ms) *build_ms = dt.build_ms;
    if (layout_ms) *layout_ms = dt.convert_ms;
    rc = scene_create_impl(ctx, positions, n_verts, indices, n_tris, tri_material, materials, n_materials, tri_light, lights,
                           n_lights, nullptr, 0, nullptr, InstancingArgs(), out, &dt);
    if (dt.inodes) {  // scene_create_impl failed before taking ownership
        (void)hipFree(dt.inodes);
        (void)hipFree(dt.tris);
        (void)hipFree(dt.slot_prim);
    }
    return rc;
}

extern "C" int pbrt_hip_scene_create_with_spheres(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                                  const int32_t* indices, int32_t n_tris, const int32_t* tri_material,
                                                  const PbrtMaterial* materials, int32_t n_materials, const int32_t* tri_light,
                                                  const PbrtLight* lights, int32_t n_lights, const float* spheres,
                                                  const int32_t* sphere_material, const int32_t* sphere_light, int32_t n_spheres,
                                                  const PbrtLinearBVHNode* nodes, int32_t n_nodes, const int32_t* prim_order,
                                                  PbrtHipScene** out) {
    SphereArgs sa;
    sa.spheres = spheres;
    sa.material = sphere_material;
    sa.light = sphere_light;
    sa.n = n_spheres;
    return scene_create_impl(ctx, positions, n_verts, indices, n_tris, tri_material, materials, n_materials, tri_light,
                             lights, n_lights, nodes, n_nodes, prim_order, InstancingArgs(), out, nullptr, sa);
}

extern "C" int pbrt_hip_scene_create_instanced(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                               const int32_t* indices, int32_t n_tris, const int32_t* tri_material,
                                               const PbrtMaterial* materials, int32_t n_materials,
                                               const PbrtLight* lights, int32_t n_lights,
                                               const PbrtLinearBVHNode* blas_nodes, int32_t n_blas_nodes,
                                               const int32_t* blas_order, const PbrtInstance* instances,
                                               int32_t n_instances, const PbrtLinearBVHNode* tlas_nodes,
                                               int32_t n_tlas_nodes, const int32_t* tlas_order, PbrtHipScene** out) {
    if (!ctx || !out) return PBRT_HIP_ERR_INVALID;
    if (!instances || n_instances <= 0 || !tlas_nodes || n_tlas_nodes <= 0 || !tlas_order) {
        ctx->last_error = "instanced scene needs instances and a top-level node array";
        return PBRT_HIP_ERR_INVALID;
    }
    for (int32_t i = 0; i < n_lights; ++i)
        if (lights[i].type == PBRT_LIGHT_DIFFUSE_AREA) {
            ctx->last_error = "instanced primitives cannot be area lights: only infinite and delta lights are accepted";
            return PBRT_HIP_ERR_INVALID;
        }
    InstancingArgs ia;
    ia.instances = instances;
    ia.n_instances = n_instances;
    ia.tlas_nodes = tlas_nodes;
    ia.n_tlas_nodes = n_tlas_nodes;
    ia.tlas_order = tlas_order;
    return scene_create_impl(ctx, positions, n_verts, indices, n_tris, tri_material, materials, n_materials, nullptr,
                             lights, n_lights, blas_nodes, n_blas_nodes, blas_order, ia, out);
}

static int scene_create_impl(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                             int32_t n_tris, const int32_t* tri_material, const PbrtMaterial* materials,
                             int32_t n_materials, const int32_t* tri_light, const PbrtLight* lights, int32_t n_lights,
                             const PbrtLinearBVHNode* nodes, int32_t n_nodes, const int32_t* prim_order,
                             const InstancingArgs& ia, PbrtHipScene** out, pb::DeviceTree* dt, const SphereArgs& sa) {
    // dt != nullptr: the tree, the triangle records and the leaf order are already on the device
    // (pbrt_hip_scene_create_hlbvh); nodes / prim_order are then unused.
    if (!ctx || !out) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(ctx);
    *out = nullptr;
    if (dt) n_nodes = dt->n_nodes;
    auto fail = [&](const char* msg) {
        ctx->last_error = msg;
        return PBRT_HIP_ERR_INVALID;
    };
    if (n_tris <= 0 || n_verts <= 0 || n_nodes <= 0) return fail("empty scene: n_tris, n_verts and n_nodes must be > 0");
    if (!positions || !indices || (!dt && (!nodes || !prim_order))) return fail("null geometry / BVH pointer");
    if (n_materials <= 0 || !materials) return fail("at least one material is required");
    if (n_lights < 0 || (n_lights > 0 && !lights)) return fail("bad light table");
    for (int64_t i = 0; i < 3 * (int64_t)n_tris; ++i)
        if (indices[i] < 0 || indices[i] >= n_verts) return fail("vertex index out of range");
    const int32_t n_prims = n_tris + sa.n;
    if (sa.n < 0 || (sa.n > 0 && (!sa.spheres || dt || ia.n_instances > 0))) return fail("bad sphere arguments");
    for (int32_t i = 0; i < sa.n; ++i) {
        if (!(sa.spheres[4 * i + 3] > 0.0f)) return fail("sphere radius must be positive");
        if (sa.material && (sa.material[i] < 0 || sa.material[i] >= n_materials)) return fail("sphere material out of range");
        if (sa.light && (sa.light[i] < -1 || sa.light[i] >= n_lights)) return fail("sphere light out of range");
    }
    for (int32_t i = 0; i < n_prims && !dt; ++i)
        if (prim_order[i] < 0 || prim_order[i] >= n_prims) return fail("prim_order entry out of range");
    for (int32_t i = 0; i < n_tris; ++i) {
        if (tri_material && (tri_material[i] < 0 || tri_material[i] >= n_materials)) return fail("tri_material out of range");
        if (tri_light && (tri_light[i] < -1 || tri_light[i] >= n_lights)) return fail("tri_light out of range");
    }
    for (int32_t i = 0; i < n_materials; ++i) {
        if (materials[i].type < PBRT_MAT_NONE || materials[i].type > PBRT_MAT_GLASS) return fail("unknown material type");
        if (materials[i].type == PBRT_MAT_GLASS && !(materials[i].eta > 0.0f)) return fail("glass needs eta > 0");
    }
    for (int32_t i = 0; i < n_lights; ++i) {
        if (lights[i].type < PBRT_LIGHT_DIFFUSE_AREA || lights[i].type > PBRT_LIGHT_DISTANT) return fail("unknown light type");
        if (lights[i].type == PBRT_LIGHT_DIFFUSE_AREA && (lights[i].prim < 0 || lights[i].prim >= n_tris + sa.n))
            return fail("area light primitive out of range");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- validate the tree(s) and lay out the interior records ----
    // single level: one tree over the triangles. Two levels (instancing): top-level tree over the
    // instances first, then the object-level tree over the triangles, in one record array.
    const bool instanced = ia.n_instances > 0;
    TreeLayout top, obj;
    if (instanced) {
        for (int32_t i = 0; i < ia.n_instances; ++i) {
            if (ia.tlas_order[i] < 0 || ia.tlas_order[i] >= ia.n_instances) return fail("tlas_order entry out of range");
            const float* m = ia.instances[i].to_world;
            const float* mi = ia.instances[i].to_object;
            if (m[12] != 0.0f || m[13] != 0.0f || m[14] != 0.0f || m[15] != 1.0f || mi[12] != 0.0f || mi[13] != 0.0f ||
                mi[14] != 0.0f || mi[15] != 1.0f)
                return fail("instance transforms must be affine (last row 0 0 0 1)");
            if (ia.instances[i].material >= n_materials) return fail("instance material out of range");
        }
        if (const char* e = convert_tree(ia.tlas_nodes, ia.n_tlas_nodes, ia.n_instances, 0, &top)) return fail(e);
        if (const char* e = convert_tree(nodes, n_nodes, n_tris, top.n_interior, &obj)) return fail(e);
    } else if (dt) {
        top.n_interior = dt->n_interior;
        top.root_ref = dt->root_ref;
        top.count_bits = dt->count_bits;
    } else {
        if (const char* e = convert_tree(nodes, n_nodes, n_prims, 0, &top)) return fail(e);
    }
    const int n_interior = top.n_interior + obj.n_interior;
    if (top.depth + obj.depth > 64) return fail("BVH deeper than the 64-entry traversal stack (bvh.rs:839)");

    PbrtHipScene* s = new PbrtHipScene();
    s->ctx = ctx;
    s->n_tris = n_tris;
    s->n_nodes = n_nodes;
    s->n_interior = n_interior;
    bool ok = true;

    // ---- interior records: both children's boxes + references + axis (64 B) ----
    std::vector<float> inodes(top.inodes);
    inodes.insert(inodes.end(), obj.inodes.begin(), obj.inodes.end());
    if (inodes.empty()) inodes.assign(16, 0.0f);
    // ---- triangles in leaf order (48 B) ----
    std::vector<float> tris(dt ? 0 : (size_t)n_prims * 12);
    std::vector<int32_t> prim_slot(dt ? 0 : n_prims, -1);
    for (int32_t slot = 0; slot < n_prims && !dt; ++slot) {
        int32_t prim = prim_order[slot];
        if (prim_slot[prim] != -1) {
            delete s;
            return fail("prim_order is not a permutation");
        }
        prim_slot[prim] = slot;
        if (prim >= n_tris) {  // sphere record: (centre.xyz, radius) | - | (-, prim, material, kPrimSphere)
            const float* sp = sa.spheres + 4 * (size_t)(prim - n_tris);
            float* t = &tris[(size_t)slot * 12];
            t[0] = sp[0]; t[1] = sp[1]; t[2] = sp[2]; t[3] = sp[3];
            for (int k = 4; k < 9; ++k) t[k] = 0.0f;
            int32_t meta[3] = {prim, sa.material ? sa.material[prim - n_tris] : 0,
                               (sa.light ? sa.light[prim - n_tris] + 1 : 0) | kPrimSphere};
            std::memcpy(t + 9, meta, 12);
            continue;
        }
        const float* a = positions + 3 * (size_t)indices[3 * (size_t)prim];
        const float* b = positions + 3 * (size_t)indices[3 * (size_t)prim + 1];
        const float* c = positions + 3 * (size_t)indices[3 * (size_t)prim + 2];
        float* t = &tris[(size_t)slot * 12];
        t[0] = a[0]; t[1] = a[1]; t[2] = a[2];
        t[3] = b[0]; t[4] = b[1]; t[5] = b[2];
        t[6] = c[0]; t[7] = c[1]; t[8] = c[2];
        int32_t meta[3];
        meta[0] = prim;
        meta[1] = tri_material ? tri_material[prim] : 0;
        meta[2] = (tri_light ? tri_light[prim] + 1 : 0) | (triangle_rejected_by_intersect(a, b, c) ? kTriDegenerate : 0);
        std::memcpy(t + 9, meta, 12);
    }
    // ---- lights / materials ----
    std::vector<DevLight> dl(std::max(1, n_lights));
    std::vector<int> infinite_ids;
    float world_center[3], world_radius;
    {
        // Bounds3::bounding_sphere of the root bounds (infinite.rs:135-139)
        const float* mn = dt ? dt->root_min : (ia.n_instances > 0 ? ia.tlas_nodes[0] : nodes[0]).bounds_min;
        const float* mx = dt ? dt->root_max : (ia.n_instances > 0 ? ia.tlas_nodes[0] : nodes[0]).bounds_max;
        float dx[3];
        for (int k = 0; k < 3; ++k) {
            world_center[k] = (mn[k] + mx[k]) / 2.0f;
            dx[k] = world_center[k] - mx[k];
        }
        bool inside = true;
        for (int k = 0; k < 3; ++k) inside = inside && world_center[k] >= mn[k] && world_center[k] <= mx[k];
        world_radius = inside ? std::sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]) : 0.0f;
    }
    std::vector<float> power_y(n_lights);
    for (int32_t i = 0; i < n_lights; ++i) {
        DevLight& l = dl[i];
        l.type = lights[i].type;
        std::memcpy(l.L, lights[i].L, 12);
        l.two_sided = lights[i].two_sided;
        l.slot = -1;
        l.area = 0.0f;
        float scale;
        std::memcpy(l.pos, lights[i].pos, 12);
        l.cos_total_width = lights[i].cos_total_width;
        l.cos_falloff_start = lights[i].cos_falloff_start;
        std::memcpy(l.w2l, lights[i].world_to_light, 36);
        l.delta = (l.type == PBRT_LIGHT_POINT || l.type == PBRT_LIGHT_SPOT || l.type == PBRT_LIGHT_DISTANT) ? 1 : 0;
        if (l.type == PBRT_LIGHT_DIFFUSE_AREA) {
            int32_t prim = lights[i].prim;
            l.slot = dt ? dt->light_slot[i] : prim_slot[prim];
            if (prim >= n_tris) {
                float r = sa.spheres[4 * (size_t)(prim - n_tris) + 3];
                l.area = (360.0f * (kPi / 180.0f)) * r * (r - (-r));  // Sphere::area (sphere.rs:99-101)
            } else {
                const float* a = positions + 3 * (size_t)indices[3 * (size_t)prim];
                const float* b = positions + 3 * (size_t)indices[3 * (size_t)prim + 1];
                const float* c = positions + 3 * (size_t)indices[3 * (size_t)prim + 2];
                l.area = tri_area(a, b, c);
            }
            scale = (l.two_sided ? 2.0f : 1.0f) * l.area * kPi;  // diffuse.rs:83-85
        } else if (l.type == PBRT_LIGHT_POINT) {
            scale = 4.0f * kPi;  // point.rs:65-67
        } else if (l.type == PBRT_LIGHT_SPOT) {
            scale = 2.0f * kPi * (1.0f - 0.5f * (l.cos_falloff_start + l.cos_total_width));  // spot.rs:90-92
        } else if (l.type == PBRT_LIGHT_DISTANT) {
            scale = kPi * world_radius * world_radius;  // distant.rs:76-78
        } else {
            infinite_ids.push_back(i);
            scale = kPi * world_radius * world_radius;  // infinite.rs:131-133
        }
        float p[3] = {l.L[0] * scale, l.L[1] * scale, l.L[2] * scale};
        l.power_y = 0.212671f * p[0] + 0.715160f * p[1] + 0.072169f * p[2];  // spectrum.rs:679-682
        power_y[i] = l.power_y;
    }
    s->h_lights = dl;
    s->light_samples.resize(n_lights);
    for (int32_t i = 0; i < n_lights; ++i) s->light_samples[i] = std::max(1, lights[i].n_samples);
    std::vector<DevMaterial> dm(n_materials);
    for (int32_t i = 0; i < n_materials; ++i) {
        dm[i].type = materials[i].type;
        std::memcpy(dm[i].kd, materials[i].kd, 12);
        std::memcpy(dm[i].kt, materials[i].kt, 12);
        dm[i].eta = materials[i].eta;
    }

    DevSceneData& d = s->d;
    std::memset(&d, 0, sizeof(d));
    if (dt) {
        d.bvh.inodes = dt->inodes;
        d.bvh.tris = dt->tris;
        d.slot_prim = dt->slot_prim;
        s->allocs.push_back(dt->inodes);
        s->allocs.push_back(dt->tris);
        s->allocs.push_back(dt->slot_prim);
        dt->inodes = dt->tris = nullptr;  // owned by the scene from here on
        dt->slot_prim = nullptr;
        std::memcpy(d.bvh.root_min, dt->root_min, 12);
        std::memcpy(d.bvh.root_max, dt->root_max, 12);
    } else {
        d.bvh.inodes = (const float4*)dev_upload(s, inodes.data(), inodes.size(), &ok);
        d.bvh.tris = (const float4*)dev_upload(s, tris.data(), tris.size(), &ok);
        const PbrtLinearBVHNode& root = instanced ? ia.tlas_nodes[0] : nodes[0];
        std::memcpy(d.bvh.root_min, root.bounds_min, 12);
        std::memcpy(d.bvh.root_max, root.bounds_max, 12);
    }
    d.bvh.root_ref = top.root_ref;
    d.bvh.count_bits = top.count_bits;
    d.bvh.n_slots = n_prims;
    d.bvh.has_spheres = sa.n > 0 ? 1 : 0;
    d.bvh.instanced = instanced ? 1 : 0;
    if (instanced) {
        std::memcpy(d.bvh.blas_root_min, nodes[0].bounds_min, 12);
        std::memcpy(d.bvh.blas_root_max, nodes[0].bounds_max, 12);
        d.bvh.blas_root_ref = obj.root_ref;
        d.bvh.blas_count_bits = obj.count_bits;
        // instance records in top-level leaf order: to_object rows 0-2, to_world rows 0-2, (material, id)
        std::vector<float> inst((size_t)ia.n_instances * 28, 0.0f);
        std::vector<int32_t> slot_inst(ia.n_instances);
        std::vector<char> seen(ia.n_instances, 0);
        for (int32_t slot = 0; slot < ia.n_instances; ++slot) {
            int32_t id = ia.tlas_order[slot];
            if (seen[id]) ok = false;
            seen[id] = 1;
            slot_inst[slot] = id;
            float* r = &inst[(size_t)slot * 28];
            std::memcpy(r, ia.instances[id].to_object, 48);
            std::memcpy(r + 12, ia.instances[id].to_world, 48);
            int32_t meta[4] = {ia.instances[id].material, id, 0, 0};
            std::memcpy(r + 24, meta, 16);
        }
        if (!ok) ctx->last_error = "tlas_order is not a permutation";
        d.bvh.instances = (const float4*)dev_upload(s, inst.data(), inst.size(), &ok);
        d.slot_instance = dev_upload(s, slot_inst.data(), slot_inst.size(), &ok);
        s->n_instances = ia.n_instances;
    }
    // spill slab for the deepest 40 stack entries of every resident lane of the traversal grid
    s->spill_lanes = ctx->n_cus * 2048;
    {
        void* p = nullptr;
        if (!hip_ok(ctx, hipMalloc(&p, (size_t)s->spill_lanes * kStackSpill * sizeof(uint2)), "hipMalloc spill")) ok = false;
        else s->allocs.push_back(p);
        d.bvh.spill = (uint2*)p;
        d.bvh.spill_stride = s->spill_lanes;
    }
    if (!dt) d.slot_prim = dev_upload(s, prim_order, n_prims, &ok);
    d.materials = dev_upload(s, dm.data(), dm.size(), &ok);
    d.lights = dev_upload(s, dl.data(), dl.size(), &ok);
    d.n_lights = n_lights;
    d.n_materials = n_materials;
    d.n_infinite = (int)infinite_ids.size();
    d.infinite_ids = dev_upload(s, infinite_ids.data(), infinite_ids.size(), &ok);
    std::memcpy(d.world_center, world_center, 12);
    d.world_radius = world_radius;
    // InfiniteAreaLight's Distribution2D over a 2x2 sin-weighted image of the constant map
    // (infinite.rs:59-73); all infinite lights here are constant so one table per light would be
    // identical up to the luminance scale -- built for the first infinite light, rescaled per use.
    {
        for (int v = 0; v < 2; ++v) {
            float vp = ((float)v + 0.5f) / 2.0f;
            float sin_theta = host_det_sin(kPi * vp);
            std::vector<float> row(2);
            for (int u = 0; u < 2; ++u) {
                float y = 1.0f;
                if (!infinite_ids.empty()) {
                    const float* L = dl[infinite_ids[0]].L;
                    y = 0.212671f * L[0] + 0.715160f * L[1] + 0.072169f * L[2];
                }
                row[u] = y;
                row[u] *= sin_theta;
            }
            std::vector<float> cdf;
            float fi;
            make_distribution(row, &cdf, &fi);
            for (int u = 0; u < 2; ++u) d.env_cond_func[v][u] = row[u];
            for (int u = 0; u < 3; ++u) d.env_cond_cdf[v][u] = cdf[u];
            d.env_cond_int[v] = fi;
        }
        std::vector<float> mf = {d.env_cond_int[0], d.env_cond_int[1]}, cdf;
        float fi;
        make_distribution(mf, &cdf, &fi);
        for (int u = 0; u < 2; ++u) d.env_marg_func[u] = mf[u];
        for (int u = 0; u < 3; ++u) d.env_marg_cdf[u] = cdf[u];
        d.env_marg_int = fi;
    }
    // light sampling distributions (lightdistrib.rs:21-69, integrator.rs:268-277)
    if (n_lights > 0) {
        std::vector<float> uni(n_lights, 1.0f), cdf;
        float fi;
        make_distribution(uni, &cdf, &fi);
        d.light_distrib_uniform.func = dev_upload(s, uni.data(), uni.size(), &ok);
        d.light_distrib_uniform.cdf = dev_upload(s, cdf.data(), cdf.size(), &ok);
        d.light_distrib_uniform.func_int = fi;
        d.light_distrib_uniform.n = n_lights;
        make_distribution(power_y, &cdf, &fi);
        d.light_distrib_power.func = dev_upload(s, power_y.data(), power_y.size(), &ok);
        d.light_distrib_power.cdf = dev_upload(s, cdf.data(), cdf.size(), &ok);
        d.light_distrib_power.func_int = fi;
        d.light_distrib_power.n = n_lights;
    }
    if (!ok) {
        pbrt_hip_scene_destroy(s);
        return PBRT_HIP_ERR_DEVICE;
    }
    *out = s;
    return PBRT_HIP_OK;
}

extern "C" void pbrt_hip_scene_destroy(PbrtHipScene* s) {
    if (!s) return;
    PB_LOCK(s->ctx);
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    for (void* p : s->allocs) (void)hipFree(p);
    delete s;
}

// ------------------------------------------------------------------------------------
// batch intersect
// ------------------------------------------------------------------------------------
// Persistent-threads traversal (trace_persistent.h) over a caller-supplied ray batch
template <bool ANY>
struct BatchRayIO {
    const int* __restrict__ slot_prim;
    const int* __restrict__ slot_instance;
    const PbrtRay* __restrict__ rays;
    uint32_t count;
    PbrtHit* __restrict__ out_hits;
    uint8_t* __restrict__ out_flags;
    PB_DEV uint32_t n() const { return count; }
    PB_DEV int segments() const { return 1; }
    PB_DEV bool load(uint32_t i, TravRay* r, bool* any) const {
        const float4* rp = reinterpret_cast<const float4*>(rays + i);
        float4 a = rp[0], b = rp[1];
        *r = TravRay{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
        *any = ANY;
        return true;
    }
    PB_DEV void store(uint32_t i, bool any, bool found, float t, float b0, float b1, float b2, int slot, int inst) const {
        if (ANY) {
            out_flags[i] = found ? 1 : 0;
        } else {
            float4 o0 = make_float4(found ? t : kInf, found ? b0 : 0.0f, found ? b1 : 0.0f, found ? b2 : 0.0f);
            int prim = found ? slot_prim[slot] : -1;
            int instance = (found && inst >= 0) ? slot_instance[inst] : -1;
            float4* op = reinterpret_cast<float4*>(out_hits + i);
            op[0] = o0;
            op[1] = make_float4(__int_as_float(prim), __int_as_float(instance), 0.0f, 0.0f);
        }
    }
};
template <bool ANY, bool COUNT, bool INST, bool SPH = false>
__global__ void __launch_bounds__(kTraceBlock, (COUNT || SPH) ? 4 : (INST ? PB_INST_WAVES : PB_TRACE_WAVES))
    k_intersect_batch(DevBVH bvh, BatchRayIO<ANY> io, unsigned int* work_counter, unsigned long long* counters) {
    __shared__ uint2 lds_stack[kStackLds * kTraceBlock];
    trace_persistent<BatchRayIO<ANY>, COUNT, INST, SPH>(bvh, io, work_counter, lds_stack + threadIdx.x,
                                                        blockIdx.x * kTraceBlock + threadIdx.x, counters);
}

// persistent kernels: enough blocks to fill every CU (LDS: 160 KiB / (kStackLds * 2 KiB) blocks of 256, at most 8)
static int persistent_grid(PbrtHipScene* s) {
    int per_cu = std::min(PB_TRACE_WAVES, (160 * 1024) / (kStackLds * kTraceBlock * (int)sizeof(uint2)));
    return std::min(s->ctx->n_cus * per_cu, s->spill_lanes / kTraceBlock);
}

template <bool ANY>
static int launch_batch(PbrtHipScene* s, const PbrtRay* d_rays, int64_t n, PbrtHit* d_hits, uint8_t* d_flags) {
    PbrtHipContext* ctx = s->ctx;
    if (n == 0) return PBRT_HIP_OK;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_work_counter, 0, kQueueSegments * sizeof(unsigned int), ctx->stream));
    if (ctx->time_trace) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    if (n >= (1ll << 32) - kChunk * 8192ll) {
        ctx->last_error = "batch too large for one launch (split it below 2^32 rays)";
        return PBRT_HIP_ERR_INVALID;
    }
    {
        BatchRayIO<ANY> io{s->d.slot_prim, s->d.slot_instance, d_rays, (uint32_t)n, d_hits, d_flags};
        dim3 grid(persistent_grid(s)), block(kTraceBlock);
        const bool inst = s->d.bvh.instanced != 0;
        if (s->d.bvh.has_spheres) {  // single-level scenes only (checked at creation)
            if (ctx->count_traversal)
                hipLaunchKernelGGL((k_intersect_batch<ANY, true, false, true>), grid, block, 0, ctx->stream, s->d.bvh, io,
                                   ctx->d_work_counter, ctx->d_counters);
            else
                hipLaunchKernelGGL((k_intersect_batch<ANY, false, false, true>), grid, block, 0, ctx->stream, s->d.bvh, io,
                                   ctx->d_work_counter, ctx->d_counters);
        } else if (ctx->count_traversal) {
            if (inst)
                hipLaunchKernelGGL((k_intersect_batch<ANY, true, true>), grid, block, 0, ctx->stream, s->d.bvh, io,
                                   ctx->d_work_counter, ctx->d_counters);
            else
                hipLaunchKernelGGL((k_intersect_batch<ANY, true, false>), grid, block, 0, ctx->stream, s->d.bvh, io,
                                   ctx->d_work_counter, ctx->d_counters);
        } else {
            if (inst)
                hipLaunchKernelGGL((k_intersect_batch<ANY, false, true>), grid, block, 0, ctx->stream, s->d.bvh, io,
                                   ctx->d_work_counter, ctx->d_counters);
            else
                hipLaunchKernelGGL((k_intersect_batch<ANY, false, false>), grid, block, 0, ctx->stream, s->d.bvh, io,
                                   ctx->d_work_counter, ctx->d_counters);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->time_trace) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        ctx->trace_ms += ms;
        ctx->trace_launches += 1;
    }
    return PBRT_HIP_OK;
}

extern "C" int pbrt_hip_intersect_device(PbrtHipScene* s, const PbrtRay* d_rays, int64_t n, PbrtHit* d_out) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_out))) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(s->ctx);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    return launch_batch<false>(s, d_rays, n, d_out, nullptr);
}
extern "C" int pbrt_hip_intersect_p_device(PbrtHipScene* s, const PbrtRay* d_rays, int64_t n, uint8_t* d_out) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_out))) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(s->ctx);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    return launch_batch<true>(s, d_rays, n, nullptr, d_out);
}

template <bool ANY>
static int intersect_host(PbrtHipScene* s, const PbrtRay* rays, int64_t n, void* out) {
    if (!s || n < 0 || (n > 0 && (!rays || !out))) return PBRT_HIP_ERR_INVALID;
    if (n == 0) return PBRT_HIP_OK;
    PbrtHipContext* ctx = s->ctx;
    PB_LOCK(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PbrtRay* d_rays = nullptr;
    void* d_out = nullptr;
    size_t out_bytes = ANY ? (size_t)n : (size_t)n * sizeof(PbrtHit);
    HIP_TRY(ctx, hipMalloc((void**)&d_rays, (size_t)n * sizeof(PbrtRay)));
    if (!hip_ok(ctx, hipMalloc(&d_out, out_bytes), "hipMalloc")) {
        (void)hipFree(d_rays);
        return PBRT_HIP_ERR_DEVICE;
    }
    int rc = PBRT_HIP_OK;
    if (!hip_ok(ctx, hipMemcpyAsync(d_rays, rays, (size_t)n * sizeof(PbrtRay), hipMemcpyHostToDevice, ctx->stream), "H2D"))
        rc = PBRT_HIP_ERR_DEVICE;
    if (rc == PBRT_HIP_OK) rc = launch_batch<ANY>(s, d_rays, n, (PbrtHit*)d_out, (uint8_t*)d_out);
    if (rc == PBRT_HIP_OK &&
        !hip_ok(ctx, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream), "D2H"))
        rc = PBRT_HIP_ERR_DEVICE;
    if (!hip_ok(ctx, hipStreamSynchronize(ctx->stream), "sync") && rc == PBRT_HIP_OK) rc = PBRT_HIP_ERR_DEVICE;
    (void)hipFree(d_rays);
    (void)hipFree(d_out);
    return rc;
}
extern "C" int pbrt_hip_intersect(PbrtHipScene* s, const PbrtRay* rays, int64_t n, PbrtHit* out) {
    return intersect_host<false>(s, rays, n, out);
}
extern "C" int pbrt_hip_intersect_p(PbrtHipScene* s, const PbrtRay* rays, int64_t n, uint8_t* out) {
    return intersect_host<true>(s, rays, n, out);
}

// Film::get_sample_bounds (film.rs:76-81, D42 intended) of the whole film
extern "C" int pbrt_hip_sample_bounds(int32_t width, int32_t height, float rx, float ry, int32_t b[4]) {
    if (!b || width <= 0 || height <= 0 || !(rx > 0.0f) || !(ry > 0.0f)) return PBRT_HIP_ERR_INVALID;
    b[0] = (int32_t)std::floor(0.0f + 0.5f - rx);
    b[1] = (int32_t)std::floor(0.0f + 0.5f - ry);
    b[2] = (int32_t)std::ceil((float)width - 0.5f + rx);
    b[3] = (int32_t)std::ceil((float)height - 0.5f + ry);
    return PBRT_HIP_OK;
}

// Filter::evaluate of src/filters/*.rs tabulated as Film::new does (film.rs:52-63)
extern "C" int pbrt_hip_filter_table(int32_t type, float rx, float ry, float a, float b, float table[256]) {
    if (!table || !(rx > 0.0f) || !(ry > 0.0f) || type < 0 || type > 4) return PBRT_HIP_ERR_INVALID;
    auto mitchell = [&](float x) {  // mitchell.rs:31-47
        x = std::fabs(2.0f * x);
        if (x > 1.0f)
            return ((-a - 6.0f * b) * x * x * x + (6.0f * a + 30.0f * b) * x * x + (-12.0f * a - 48.0f * b) * x +
                    (8.0f * a + 24.0f * b)) * (1.0f / 6.0f);
        return ((12.0f - 9.0f * a - 6.0f * b) * x * x * x + (-18.0f + 12.0f * a + 6.0f * b) * x * x + (6.0f - 2.0f * a)) *
               (1.0f / 6.0f);
    };
    auto sinc = [](float x) {  // sinc.rs:28-35
        x = std::fabs(x);
        return x < 1e-5f ? 1.0f : std::sin(kPi * x) / (kPi * x);
    };
    auto wsinc = [&](float x, float radius) {  // sinc.rs:37-45
        x = std::fabs(x);
        return x > radius ? 0.0f : sinc(x) * sinc(x / a);
    };
    int k = 0;
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) {
            float px = ((float)x + 0.5f) * rx / 16.0f, py = ((float)y + 0.5f) * ry / 16.0f, v = 1.0f;
            switch (type) {
                case PBRT_FILTER_GAUSSIAN: {  // gaussian.rs:17-39
                    float ex = std::exp(-a * rx * rx), ey = std::exp(-a * ry * ry);
                    float gx = std::exp(-a * px * px) - ex, gy = std::exp(-a * py * py) - ey;
                    v = (gx > 0.0f ? gx : 0.0f) * (gy > 0.0f ? gy : 0.0f);
                    break;
                }
                case PBRT_FILTER_MITCHELL: v = mitchell(px * (1.0f / rx)) * mitchell(py * (1.0f / ry)); break;
                case PBRT_FILTER_LANCZOS: v = wsinc(px, rx) * wsinc(py, ry); break;
                case PBRT_FILTER_TRIANGLE: {  // triangle.rs:21-27
                    float tx = rx - std::fabs(px), ty = ry - std::fabs(py);
                    v = (tx > 0.0f ? tx : 0.0f) * (ty > 0.0f ? ty : 0.0f);
                    break;
                }
                default: v = 1.0f;  // boxf.rs:25-27
            }
            table[k++] = v;
        }
    return PBRT_HIP_OK;
}

extern "C" int pbrt_hip_write_pfm(const char* path, const float* rgb, int32_t width, int32_t height) {
    if (!path || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) return PBRT_HIP_ERR_INVALID;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);  // negative scale = little endian
    for (int32_t y = height - 1; y >= 0; --y)               // PFM stores the bottom row first
        std::fwrite(rgb + (size_t)y * width * 3, sizeof(float), (size_t)width * 3, f);
    bool ok = std::fclose(f) == 0;
    return ok ? PBRT_HIP_OK : PBRT_HIP_ERR_INVALID;
}

// OpenEXR, the format pbrt-v3 writes by default: single-part scanline file, three 32-bit float channels (stored in
// alphabetical order B, G, R), no compression, one scanline per chunk. Linear values, nothing clamped.
namespace {
struct ExrBuf {
    std::vector<unsigned char> b;
    void bytes(const void* p, size_t n) { b.insert(b.end(), (const unsigned char*)p, (const unsigned char*)p + n); }
    void str(const char* s) { bytes(s, std::strlen(s) + 1); }
    void i32(int32_t v) { bytes(&v, 4); }  // little-endian host (x86-64)
    void f32(float v) { bytes(&v, 4); }
    void u8(unsigned char v) { b.push_back(v); }
    void attr(const char* name, const char* type, int32_t size) {
        str(name);
        str(type);
        i32(size);
    }
};
}  // namespace

extern "C" int pbrt_hip_write_exr(const char* path, const float* rgb, int32_t width, int32_t height) {
    if (!path || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
    ExrBuf h;
    h.i32(20000630);  // magic 0x76 0x2f 0x31 0x01
    h.i32(2);         // version 2, no flags: single-part scanline
    h.attr("channels", "chlist", 3 * (2 + 4 + 4 + 4 + 4) + 1);
    for (const char* c : {"B", "G", "R"}) {
        h.str(c);
        h.i32(2);  // FLOAT
        h.u8(0);   // pLinear
        h.u8(0);
        h.u8(0);
        h.u8(0);
        h.i32(1);  // xSampling
        h.i32(1);  // ySampling
    }
    h.u8(0);
    h.attr("compression", "compression", 1);
    h.u8(0);  // NO_COMPRESSION
    for (const char* name : {"dataWindow", "displayWindow"}) {
        h.attr(name, "box2i", 16);
        h.i32(0);
        h.i32(0);
        h.i32(width - 1);
        h.i32(height - 1);
    }
    h.attr("lineOrder", "lineOrder", 1);
    h.u8(0);  // INCREASING_Y
    h.attr("pixelAspectRatio", "float", 4);
    h.f32(1.0f);
    h.attr("screenWindowCenter", "v2f", 8);
    h.f32(0.0f);
    h.f32(0.0f);
    h.attr("screenWindowWidth", "float", 4);
    h.f32(1.0f);
    h.u8(0);  // end of header
    FILE* f = std::fopen(path, "wb");
    if (!f) return PBRT_HIP_ERR_INVALID;
    const uint64_t row_bytes = (uint64_t)width * 3 * sizeof(float), chunk = 8 + row_bytes;
    const uint64_t first = h.b.size() + (uint64_t)height * 8;
    std::fwrite(h.b.data(), 1, h.b.size(), f);
    for (int32_t y = 0; y < height; ++y) {  // offset table
        uint64_t off = first + (uint64_t)y * chunk;
        std::fwrite(&off, 8, 1, f);
    }
    std::vector<float> line((size_t)width * 3);
    for (int32_t y = 0; y < height; ++y) {
        const float* src = rgb + (size_t)y * width * 3;
        for (int32_t x = 0; x < width; ++x) {
            line[x] = src[3 * x + 2];                      // B
            line[(size_t)width + x] = src[3 * x + 1];      // G
            line[2 * (size_t)width + x] = src[3 * x];      // R
        }
        int32_t head[2] = {y, (int32_t)row_bytes};
        std::fwrite(head, 4, 2, f);
        std::fwrite(line.data(), sizeof(float), line.size(), f);
    }
    bool ok = std::fclose(f) == 0;
    return ok ? PBRT_HIP_OK : PBRT_HIP_ERR_INVALID;
}

// 8-bit sRGB PNG (what pbrt-v3's WriteImage does for ".png": gamma_correct, 255 * v + 0.5 clamped to [0, 255]);
// the reference's own writer is todo!() (src/core/imageio.rs:3-5). zlib stream of stored (uncompressed) deflate
// blocks: no dependency, every PNG reader accepts it.
namespace {
struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const unsigned char* p, size_t n) const {
        for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
        return crc;
    }
};
void png_chunk(FILE* f, const Crc32& crc, const char* type, const std::vector<unsigned char>& data) {
    unsigned char len[4] = {(unsigned char)(data.size() >> 24), (unsigned char)(data.size() >> 16),
                            (unsigned char)(data.size() >> 8), (unsigned char)data.size()};
    std::fwrite(len, 1, 4, f);
    std::fwrite(type, 1, 4, f);
    if (!data.empty()) std::fwrite(data.data(), 1, data.size(), f);
    uint32_t c = crc.run(0xffffffffu, (const unsigned char*)type, 4);
    if (!data.empty()) c = crc.run(c, data.data(), data.size());
    c ^= 0xffffffffu;
    unsigned char out[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c};
    std::fwrite(out, 1, 4, f);
}
float gamma_correct(float v) {  // pbrt.rs GammaCorrect: sRGB transfer curve
    if (v <= 0.0031308f) return 12.92f * v;
    return 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
}
}  // namespace

extern "C" int pbrt_hip_write_png(const char* path, const float* rgb, int32_t width, int32_t height) {
    if (!path || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
    const size_t row = (size_t)width * 3 + 1;  // filter byte + pixels
    std::vector<unsigned char> raw(row * height);
    for (int32_t y = 0; y < height; ++y) {
        unsigned char* o = &raw[row * y];
        *o++ = 0;  // filter type None
        for (int32_t x = 0; x < width * 3; ++x) {
            float v = 255.0f * gamma_correct(rgb[(size_t)y * width * 3 + x]) + 0.5f;
            *o++ = (unsigned char)(v < 0.0f || v != v ? 0.0f : (v > 255.0f ? 255.0f : v));
        }
    }
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;  // Adler-32
    for (size_t pos = 0; pos < raw.size();) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back((unsigned char)(n & 0xff));
        z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff));
        z.push_back((unsigned char)((~n >> 8) & 0xff));
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    }
    uint32_t adler = (b << 16) | a;
    for (int k = 3; k >= 0; --k) z.push_back((unsigned char)(adler >> (8 * k)));
    FILE* f = std::fopen(path, "wb");
    if (!f) return PBRT_HIP_ERR_INVALID;
    static const Crc32 crc;
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    std::vector<unsigned char> ihdr = {(unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8),
                                       (unsigned char)width, (unsigned char)(height >> 24), (unsigned char)(height >> 16),
                                       (unsigned char)(height >> 8), (unsigned char)height, 8, 2, 0, 0, 0};  // 8-bit RGB
    png_chunk(f, crc, "IHDR", ihdr);
    png_chunk(f, crc, "IDAT", z);
    png_chunk(f, crc, "IEND", {});
    bool ok = std::fclose(f) == 0;
    return ok ? PBRT_HIP_OK : PBRT_HIP_ERR_INVALID;
}

extern "C" void pbrt_hip_film_to_rgb(const float* film, int64_t n_pixels, float* rgb) {
    for (int64_t i = 0; i < n_pixels; ++i) {
        const float* p = film + 4 * i;
        float r = 3.240479f * p[0] - 1.537150f * p[1] - 0.498535f * p[2];
        float g = -0.969256f * p[0] + 1.875991f * p[1] + 0.041556f * p[2];
        float b = 0.055648f * p[0] - 0.204043f * p[1] + 1.057311f * p[2];
        if (p[3] != 0.0f) {
            float inv = 1.0f / p[3];
            r = std::max(r * inv, 0.0f);
            g = std::max(g * inv, 0.0f);
            b = std::max(b * inv, 0.0f);
        }
        rgb[3 * i] = r;
        rgb[3 * i + 1] = g;
        rgb[3 * i + 2] = b;
    }
}

// ------------------------------------------------------------------------------------
// render: see wavefront.h
// ------------------------------------------------------------------------------------
extern "C" int pbrt_hip_render_device(PbrtHipScene* s, const PbrtCamera* camera, const PbrtRenderParams* params,
                                      float* d_film, PbrtRenderStats* stats) {
    if (!s || !camera || !params || !d_film) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(s->ctx);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    return wavefront_render(s, *camera, *params, d_film, stats);
}

extern "C" int pbrt_hip_render(PbrtHipScene* s, const PbrtCamera* camera, const PbrtRenderParams* params,
                               float* film_xyzw, PbrtRenderStats* stats) {
    if (!s || !camera || !params || !film_xyzw) return PBRT_HIP_ERR_INVALID;
    PbrtHipContext* ctx = s->ctx;
    PB_LOCK(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (params->width <= 0 || params->height <= 0) return PBRT_HIP_ERR_INVALID;
    size_t bytes = (size_t)params->width * params->height * 4 * sizeof(float);
    float* d_film = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_film, bytes));
    int rc = wavefront_render(s, *camera, *params, d_film, stats);
    if (rc == PBRT_HIP_OK && !hip_ok(ctx, hipMemcpy(film_xyzw, d_film, bytes, hipMemcpyDeviceToHost), "film D2H"))
        rc = PBRT_HIP_ERR_DEVICE;
    (void)hipFree(d_film);
    return rc;
}

extern "C" int pbrt_hip_tile_partition(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t rank, int32_t world,
                                       int32_t* origins_xy, int32_t capacity, int32_t* n_out) {
    if (!n_out || x0 > x1 || y0 > y1 || world <= 0 || rank < 0 || rank >= world) return PBRT_HIP_ERR_INVALID;
    int ntx = (x1 - x0 + kTile - 1) / kTile, nty = (y1 - y0 + kTile - 1) / kTile;
    int n = 0;
    for (int t = 0; t < ntx * nty; ++t) {
        if (t % world != rank) continue;
        if (origins_xy && n < capacity) {
            origins_xy[2 * n] = x0 + (t % ntx) * kTile;
            origins_xy[2 * n + 1] = y0 + (t / ntx) * kTile;
        }
        ++n;
    }
    *n_out = n;
    return (origins_xy && n > capacity) ? PBRT_HIP_ERR_INVALID : PBRT_HIP_OK;
}

// ------------------------------------------------------------------------------------
// wavefront_render — host driver of the kernels in wavefront.h
// ------------------------------------------------------------------------------------
namespace {
// Device buffers of one call. Blocks come from (and return to) the context's cache, so that repeated renders of the
// same size do not pay hipMalloc / hipFree again.
struct DevBuf {
    std::vector<size_t> taken;  // indices into ctx->block_cache
    PbrtHipContext* ctx;
    explicit DevBuf(PbrtHipContext* c) : ctx(c) {}
    ~DevBuf() {
        for (size_t i : taken) ctx->block_cache[i].in_use = false;
        size_t idle = 0;
        for (const auto& b : ctx->block_cache)
            if (!b.in_use) idle += b.bytes;
        if (idle > ((size_t)160 << 30)) trim(ctx);  // renders of many different sizes: do not sit on most of the HBM
    }
    static void trim(PbrtHipContext* ctx) {  // release every block no call is using (slots stay, indices are stable)
        for (auto& b : ctx->block_cache)
            if (!b.in_use && b.ptr) {
                (void)hipFree(b.ptr);
                b.ptr = nullptr;
                b.bytes = 0;
            }
    }
    template <class T>
    T* alloc(size_t n, bool* ok) {
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        // best fit among the free cached blocks that are not wastefully large
        size_t best = SIZE_MAX;
        for (size_t i = 0; i < ctx->block_cache.size(); ++i) {
            const auto& b = ctx->block_cache[i];
            if (!b.in_use && b.ptr && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + 4096 &&
                (best == SIZE_MAX || b.bytes < ctx->block_cache[best].bytes))
                best = i;
        }
        if (best != SIZE_MAX) {
            ctx->block_cache[best].in_use = true;
            taken.push_back(best);
            return (T*)ctx->block_cache[best].ptr;
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {  // out of memory with idle cached blocks around: drop them and retry once
            (void)hipGetLastError();
            trim(ctx);
            e = hipMalloc(&p, bytes);
        }
        if (!hip_ok(ctx, e, "hipMalloc (path state)")) {
            *ok = false;
            return nullptr;
        }
        size_t slot = ctx->block_cache.size();
        for (size_t i = 0; i < ctx->block_cache.size(); ++i)
            if (!ctx->block_cache[i].ptr && !ctx->block_cache[i].in_use) {
                slot = i;
                break;
            }
        if (slot == ctx->block_cache.size()) ctx->block_cache.push_back({nullptr, 0, false});
        ctx->block_cache[slot] = {p, bytes, true};
        taken.push_back(slot);
        return (T*)p;
    }
};
}  // namespace

// ---- optional per-vertex normals / uvs of the TriangleMesh (triangle.rs:17-26, used at :60-72, 252-312, 337-341) ----
__global__ void k_set_shading(const int* __restrict__ slot_prim, const int* __restrict__ idx, const float* __restrict__ pos,
                              const float* __restrict__ normals, const float* __restrict__ tangents,
                              const float* __restrict__ uvs, int n, float4* __restrict__ out, float4* __restrict__ tris) {
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    int prim = slot_prim[slot];
    int v[3] = {idx[3 * (size_t)prim], idx[3 * (size_t)prim + 1], idx[3 * (size_t)prim + 2]};
    float nn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, uv[6] = {0.0f, 0.0f, 1.0f, 0.0f, 1.0f, 1.0f};
    for (int k = 0; k < 3; ++k) {
        if (normals)
            for (int c = 0; c < 3; ++c) nn[3 * k + c] = normals[3 * (size_t)v[k] + c];
        if (tangents)
            for (int c = 0; c < 3; ++c) tt[3 * k + c] = tangents[3 * (size_t)v[k] + c];
        if (uvs)
            for (int c = 0; c < 2; ++c) uv[2 * k + c] = uvs[2 * (size_t)v[k] + c];
    }
    float4* o = out + 6 * (size_t)slot;
    o[0] = make_float4(nn[0], nn[1], nn[2], nn[3]);
    o[1] = make_float4(nn[4], nn[5], nn[6], nn[7]);
    o[2] = make_float4(nn[8], tt[0], tt[1], tt[2]);
    o[3] = make_float4(tt[3], tt[4], tt[5], tt[6]);
    o[4] = make_float4(tt[7], tt[8], uv[0], uv[1]);
    o[5] = make_float4(uv[2], uv[3], uv[4], uv[5]);
    // "Triangle::intersect returns false" depends on the uvs (triangle.rs:197-216): refresh the flag
    float a[3], b[3], c[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = pos[3 * (size_t)v[0] + k];
        b[k] = pos[3 * (size_t)v[1] + k];
        c[k] = pos[3 * (size_t)v[2] + k];
    }
    float4 t2 = tris[3 * (size_t)slot + 2];
    int w = __float_as_int(t2.w) & ~kTriDegenerate;
    if (triangle_rejected_by_intersect(a, b, c, uv)) w |= kTriDegenerate;
    t2.w = __int_as_float(w);
    tris[3 * (size_t)slot + 2] = t2;
}

extern "C" int pbrt_hip_scene_set_shading_data(PbrtHipScene* s, const float* positions, int32_t n_verts,
                                               const int32_t* indices, int32_t n_tris, const float* normals,
                                               const float* tangents, const float* uvs) {
    if (!s) return PBRT_HIP_ERR_INVALID;
    PbrtHipContext* ctx = s->ctx;
    PB_LOCK(ctx);
    auto fail = [&](const char* msg) {
        ctx->last_error = msg;
        return PBRT_HIP_ERR_INVALID;
    };
    if (!positions || !indices || n_tris != s->n_tris || n_verts <= 0) return fail("mesh does not match the scene");
    if (!normals && !tangents && !uvs) return fail("no normals, tangents or uvs given");
    if (s->d.bvh.tri_shading) return fail("shading data already set");
    if (s->d.bvh.has_spheres) return fail("per-vertex shading data is not supported for scenes with spheres");
    for (int64_t i = 0; i < 3 * (int64_t)n_tris; ++i)
        if (indices[i] < 0 || indices[i] >= n_verts) return fail("vertex index out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf tmp(ctx);
    bool ok = true;
    int* d_idx = tmp.alloc<int>(3 * (size_t)n_tris, &ok);
    float* d_pos = tmp.alloc<float>(3 * (size_t)n_verts, &ok);
    float* d_n = normals ? tmp.alloc<float>(3 * (size_t)n_verts, &ok) : nullptr;
    float* d_t = tangents ? tmp.alloc<float>(3 * (size_t)n_verts, &ok) : nullptr;
    float* d_uv = uvs ? tmp.alloc<float>(2 * (size_t)n_verts, &ok) : nullptr;
    void* out = nullptr;
    if (!ok || !hip_ok(ctx, hipMalloc(&out, (size_t)n_tris * 96), "hipMalloc shading data")) return PBRT_HIP_ERR_OOM;
    s->allocs.push_back(out);
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_idx, indices, 3 * (size_t)n_tris * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_pos, positions, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    if (d_n) HIP_TRY(ctx, hipMemcpyAsync(d_n, normals, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    if (d_t) HIP_TRY(ctx, hipMemcpyAsync(d_t, tangents, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    if (d_uv) HIP_TRY(ctx, hipMemcpyAsync(d_uv, uvs, 2 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_set_shading, dim3((n_tris + 255) / 256), dim3(256), 0, st, s->d.slot_prim, d_idx, d_pos, d_n, d_t, d_uv,
                       n_tris, (float4*)out, const_cast<float4*>(s->d.bvh.tris));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    s->d.bvh.tri_shading = (const float4*)out;
    s->d.bvh.has_normals = normals ? 1 : 0;
    s->d.bvh.has_tangents = tangents ? 1 : 0;
    s->d.bvh.has_uvs = uvs ? 1 : 0;
    return PBRT_HIP_OK;
}

namespace pb {
int sort_pairs_u32(hipStream_t st, void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                   const uint32_t* vals_in, uint32_t* vals_out, size_t n, int bits);
}

// ---- HaltonSampler host side: prime tables and compute_radical_inverse_permutations (lowdiscrepancy.rs:11-170,
// 333-349: RNG::default + shuffle per prime), HaltonSampler::new constants (halton.rs:40-98) ----
namespace {
struct HostPcg {  // rng.rs
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t u32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27), rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((~rot + 1u) & 31u));
    }
    uint32_t bounded(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        for (;;) {
            uint32_t r = u32();
            if (r >= threshold) return r % b;
        }
    }
};
void halton_host_tables(std::vector<uint32_t>* primes_and_sums, std::vector<uint16_t>* perms) {
    std::vector<uint32_t> primes;
    for (uint32_t v = 2; primes.size() < 1000; ++v) {
        bool is_prime = true;
        for (uint32_t q : primes) {
            if (q * q > v) break;
            if (v % q == 0) {
                is_prime = false;
                break;
            }
        }
        if (is_prime) primes.push_back(v);
    }
    primes_and_sums->assign(2000, 0);
    uint32_t acc = 0;
    for (int i = 0; i < 1000; ++i) {
        (*primes_and_sums)[i] = primes[i];
        (*primes_and_sums)[1000 + i] = acc;
        acc += primes[i];
    }
    perms->resize(acc);
    HostPcg rng;
    uint16_t* p = perms->data();
    for (int i = 0; i < 1000; ++i) {
        int count = (int)primes[i];
        for (int j = 0; j < count; ++j) p[j] = (uint16_t)j;
        for (int j = 0; j < count; ++j) std::swap(p[j], p[j + (int)rng.bounded((uint32_t)(count - j))]);  // sampling.rs:280-287
        p += count;
    }
}
void extended_gcd(uint64_t a, uint64_t b, int64_t* x, int64_t* y) {  // halton.rs:51-61
    if (b == 0) {
        *x = 1;
        *y = 0;
        return;
    }
    int64_t d = (int64_t)(a / b), xp = 0, yp = 0;
    extended_gcd(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - d * yp;
}
uint64_t multiplicative_inverse(int64_t a, int64_t n) {  // halton.rs:40-49
    int64_t x = 0, y = 0;
    extended_gcd((uint64_t)a, (uint64_t)n, &x, &y);
    int64_t r = x - (x / n) * n;
    return (uint64_t)(r < 0 ? r + n : r);
}
}  // namespace

static int round_up_pow2(int v) {  // pbrt.rs:174-182
    v -= 1;
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    return v + 1;
}

int wavefront_render(PbrtHipScene* s, const PbrtCamera& camera, const PbrtRenderParams& rp_in, float* d_film,
                     PbrtRenderStats* stats) {
    PbrtHipContext* ctx = s->ctx;
    auto invalid = [&](const char* m) {
        ctx->last_error = m;
        return PBRT_HIP_ERR_INVALID;
    };
    PbrtRenderParams rp = rp_in;
    if (rp.width <= 0 || rp.height <= 0 || rp.spp <= 0) return invalid("width, height and spp must be positive");
    // ---- sampler: samples per pixel and Sampler::round_count (stratified.rs:30-33, zerotwosequence.rs:20, 62-64) ----
    if (rp.sampler < PBRT_SAMPLER_RANDOM || rp.sampler > PBRT_SAMPLER_HALTON) return invalid("unknown sampler");
    const bool tabulated = rp.sampler != PBRT_SAMPLER_RANDOM;
    if (tabulated && (rp.sampler_dims < 0 || rp.sampler_dims > 63)) return invalid("sampler_dims must be in [0, 63]");
    if (rp.sampler == PBRT_SAMPLER_STRATIFIED) {
        if (rp.sampler_x < 1 || rp.sampler_y < 1 || (int64_t)rp.sampler_x * rp.sampler_y > 65536)
            return invalid("stratified sampler: sampler_x * sampler_y must be in [1, 65536]");
        rp.spp = rp.sampler_x * rp.sampler_y;
    } else if (rp.sampler == PBRT_SAMPLER_ZEROTWO) {
        if (rp.spp > 65536) return invalid("(0,2)-sequence sampler: spp too large");
        rp.spp = round_up_pow2(rp.spp);
    }
    auto round_count = [&](int n) { return rp.sampler == PBRT_SAMPLER_ZEROTWO ? round_up_pow2(n) : n; };
    if (rp.integrator == PBRT_INTEGRATOR_AO && rp.ao_samples >= 1 && rp.ao_samples <= 65535 && tabulated)
        rp.ao_samples = round_count(rp.ao_samples);  // ao.rs:36
    float frx = rp.filter_radius[0] > 0.0f ? rp.filter_radius[0] : 0.5f;
    float fry = rp.filter_radius[1] > 0.0f ? rp.filter_radius[1] : 0.5f;
    bool box = true;  // the 0.5 box filter: exact in-order accumulation (k_film_accumulate)
    if (rp.filter_table) {
        if (frx != 0.5f || fry != 0.5f) box = false;
        for (int i = 0; i < 256; ++i)
            if (rp.filter_table[i] != 1.0f) box = false;
    } else {
        frx = fry = 0.5f;
    }
    if (frx > 64.0f || fry > 64.0f) return invalid("filter radius too large");
    {
        int32_t sb[4];
        pbrt_hip_sample_bounds(rp.width, rp.height, frx, fry, sb);
        if (rp.x0 < sb[0] || rp.y0 < sb[1] || rp.x1 > sb[2] || rp.y1 > sb[3] || rp.x0 > rp.x1 || rp.y0 > rp.y1)
            return invalid("pixel bounds outside the film's sample bounds");
    }
    if (rp.integrator < PBRT_INTEGRATOR_PATH || rp.integrator > PBRT_INTEGRATOR_AO)
        return invalid("integrator must be a PbrtIntegratorKind");
    // DirectLighting, Whitted and AO share the per-vertex stage machine of k_shade_direct
    const bool direct = rp.integrator != PBRT_INTEGRATOR_PATH;
    if (direct && rp.max_depth > 64) return invalid("direct lighting / Whitted: max_depth > 64 (frame stack)");
    if (rp.integrator == PBRT_INTEGRATOR_AO && (rp.ao_samples < 1 || rp.ao_samples > 65535))
        return invalid("ambient occlusion: ao_samples must be in [1, 65535]");
    if (rp.max_depth < 0 || rp.max_depth > 1 << 20) return invalid("bad max_depth");
    int world = rp.tile_world <= 0 ? 1 : rp.tile_world;
    int rank = rp.tile_rank;
    if (rank < 0 || rank >= world) return invalid("tile_rank outside [0, tile_world)");
    hipStream_t st = ctx->stream;

    size_t film_bytes = (size_t)rp.width * rp.height * 4 * sizeof(float);
    HIP_TRY(ctx, hipMemsetAsync(d_film, 0, film_bytes, st));

    // tiles of the sample bounds (integrator.rs:402-409), dealt round-robin to the GPUs
    std::vector<int2> origins;
    {
        int32_t n_mine = 0;
        pbrt_hip_tile_partition(rp.x0, rp.y0, rp.x1, rp.y1, rank, world, nullptr, 0, &n_mine);
        origins.resize(n_mine);
        pbrt_hip_tile_partition(rp.x0, rp.y0, rp.x1, rp.y1, rank, world, (int32_t*)origins.data(), n_mine, &n_mine);
    }
    PbrtRenderStats local{};
    if (origins.empty()) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (stats) *stats = local;
        return PBRT_HIP_OK;
    }
    const int n_pix = (int)origins.size() * kTile * kTile;
    int64_t valid_pixels = 0;
    for (const int2& o : origins)
        valid_pixels += (int64_t)(std::min(o.x + kTile, rp.x1) - o.x) * (std::min(o.y + kTile, rp.y1) - o.y);
    int spp_pass = rp.spp_per_pass > 0 ? rp.spp_per_pass : 0;
    if (spp_pass == 0) {
        // As many samples of a pixel in flight as half of the free HBM holds (about 400 B of path state, queue and
        // sort slots per path; the stage machine of the other integrators adds its frame stack): a whole 64-spp
        // 1080p frame is 133 M paths = 50 GB of the 288 GB and runs 6 large wavefronts instead of 48 small ones
        // (+10 % on config 3: fewer launches and host round trips, shorter tails).
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        for (const auto& b : ctx->block_cache)
            if (!b.in_use) free_b += b.bytes;  // blocks kept from the previous render are available to this one
        const int64_t per_path = 400 + (rp.integrator != PBRT_INTEGRATOR_PATH ? 24 + 48ll * std::max(1, rp.max_depth) : 0);
        const int64_t target_paths = std::max<int64_t>(1ll << 20, std::min<int64_t>(1ll << 28, (int64_t)(free_b / 2) / per_path));
        spp_pass = (int)std::max<int64_t>(1, std::min<int64_t>(rp.spp, target_paths / n_pix));
    }
    spp_pass = std::min(spp_pass, rp.spp);
    const size_t N = (size_t)n_pix * spp_pass;
    if (N * 4 >= (1ull << 32)) return invalid("too many concurrent paths for 32-bit queue entries; lower spp_per_pass");

    DevBuf buf(ctx);
    bool ok = true;
    PathState ps;
    ps.n_paths = N;
    ps.ray = buf.alloc<float4>(N * 6, &ok);
    ps.hit = buf.alloc<float4>(N * 6, &ok);
    ps.rng = buf.alloc<uint64_t>(N, &ok);
    ps.L = buf.alloc<float4>(N, &ok);
    ps.beta = buf.alloc<float4>(N, &ok);
    ps.nee_a = buf.alloc<float4>(N, &ok);
    ps.nee_f = buf.alloc<float4>(N, &ok);
    ps.nee_b = buf.alloc<float4>(N, &ok);
    ps.nee_light = buf.alloc<int>(N, &ok);
    ps.pfilm = buf.alloc<float2>(N, &ok);
    Queues q[2];
    for (int k = 0; k < 2; ++k) {
        q[k].trace = buf.alloc<uint32_t>(N * 3, &ok);
        q[k].shade = buf.alloc<uint32_t>(N, &ok);
        q[k].counts64 = buf.alloc<unsigned long long>(2, &ok);
        q[k].keys = nullptr;
        for (int a = 0; a < 3; ++a) {
            const float lo = s->d.bvh.root_min[a], hi = s->d.bvh.root_max[a];
            q[k].key_lo[a] = lo;
            q[k].key_inv[a] = hi > lo ? 1.0f / (hi - lo) : 0.0f;
        }
    }
    // ray-queue sort (spatial order for the bounce rays): keys in / out, sorted entries, rocPRIM scratch
    const char* sort_env = std::getenv("PBRT_HIP_SORT_RAYS");
    // only the path integrator's later bounces are incoherent; the stage machine of the other integrators keeps
    // shooting from the camera rays' hit points, which are already in pixel order (AO: -6 % with the sort)
    const bool sort_rays = !(sort_env && sort_env[0] == '0') && rp.integrator == PBRT_INTEGRATOR_PATH;
    const char* sort_from_env = std::getenv("PBRT_HIP_SORT_FROM");  // first sorted wavefront (development knob)
    const int sort_from = sort_from_env ? std::atoi(sort_from_env) : 2;
    // the keys are written by k_shade together with the queue entries; PBRT_HIP_SORT_FUSED=0 (and the builds with
    // direction-octant bits) compute them in a pass of their own
    const char* fused_env = std::getenv("PBRT_HIP_SORT_FUSED");
    const bool fused_keys = !(fused_env && fused_env[0] == '0') && PB_SORT_OCTANT == 0;
    uint32_t *sort_keys[2] = {nullptr, nullptr}, *sort_vals = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    if (sort_rays) {
        sort_keys[0] = buf.alloc<uint32_t>(N * 3, &ok);
        sort_keys[1] = buf.alloc<uint32_t>(N * 3, &ok);
        sort_vals = buf.alloc<uint32_t>(N * 3, &ok);
        if (pb::sort_pairs_u32(st, nullptr, &sort_tmp_bytes, sort_keys[0], sort_keys[1], sort_vals, sort_vals, N * 3, kSortKeyBits) != 0)
            return invalid("rocPRIM radix sort: size query failed");
        sort_tmp = buf.alloc<char>(sort_tmp_bytes, &ok);
    }
    DirectState ds{};
    std::vector<int> prefix(s->d.n_lights + 1, 0);
    for (int i = 0; i < s->d.n_lights; ++i)  // directlighting.rs:58-62: round_count with a tabulating sampler
        prefix[i + 1] = prefix[i] + (tabulated ? round_count(s->light_samples[i]) : s->light_samples[i]);
    // ---- PixelSampler tables: n_dims 1D + n_dims 2D dimensions and the requested 2D arrays, per pixel ----
    SamplerParams smp{};
    smp.kind = rp.sampler;
    smp.n_dims = rp.sampler_dims;
    smp.nx = rp.sampler_x;
    smp.ny = rp.sampler_y;
    smp.jitter = rp.sampler_jitter;
    ps.samp = buf.alloc<int>(N, &ok);
    if (rp.sampler == PBRT_SAMPLER_HALTON) {
        smp.n_dims = 0;  // nothing is tabulated per pixel: every value is a function of (sample index, dimension)
        if (!ctx->d_halton_primes) {
            std::vector<uint32_t> primes;
            std::vector<uint16_t> perms;
            halton_host_tables(&primes, &perms);
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_halton_primes, primes.size() * sizeof(uint32_t)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_halton_perms, perms.size() * sizeof(uint16_t)));
            HIP_TRY(ctx, hipMemcpy(ctx->d_halton_primes, primes.data(), primes.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            HIP_TRY(ctx, hipMemcpy(ctx->d_halton_perms, perms.data(), perms.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        }
        smp.primes = ctx->d_halton_primes;
        smp.perms = ctx->d_halton_perms;
        // HaltonSampler::new(spp, film.get_sample_bounds(), false) (halton.rs:63-98)
        int32_t sb[4];
        pbrt_hip_sample_bounds(rp.width, rp.height, frx, fry, sb);
        const int res[2] = {sb[2] - sb[0], sb[3] - sb[1]};
        for (int i = 0; i < 2; ++i) {
            int base = i == 0 ? 2 : 3, scale = 1, exp = 0;
            while (scale < std::min(128, res[i])) {
                scale *= base;
                ++exp;
            }
            smp.h_scale[i] = scale;
            smp.h_exp[i] = exp;
        }
        smp.h_stride = smp.h_scale[0] * smp.h_scale[1];
        smp.h_minv[0] = (unsigned int)multiplicative_inverse(smp.h_scale[1], smp.h_scale[0]);
        smp.h_minv[1] = (unsigned int)multiplicative_inverse(smp.h_scale[0], smp.h_scale[1]);
    }
    if (tabulated) {
        std::vector<int2> arrays;
        int64_t elems = (int64_t)smp.n_dims * rp.spp * 3;
        smp.off2 = smp.n_dims * rp.spp;
        auto request_2d_array = [&](int n) {  // sampler.rs:41-46
            arrays.push_back(make_int2(n, (int)elems));
            elems += (int64_t)n * rp.spp * 2;
        };
        if (rp.integrator == PBRT_INTEGRATOR_DIRECT && rp.light_strategy == 0) {
            for (int i = 0; i < rp.max_depth; ++i)  // directlighting.rs:64-75
                for (int j = 0; j < s->d.n_lights; ++j) {
                    request_2d_array(prefix[j + 1] - prefix[j]);
                    request_2d_array(prefix[j + 1] - prefix[j]);
                }
        } else if (rp.integrator == PBRT_INTEGRATOR_AO) {
            request_2d_array(rp.ao_samples);  // ao.rs:37
        }
        if (arrays.size() > 0x7fff) return invalid("too many sample arrays (max_depth x lights)");
        if (elems * n_pix * 4 > (64ll << 30) || elems >= (1ll << 31))
            return invalid("sampler tables exceed 64 GB: lower spp, sampler_dims or the light sample counts");
        smp.n_arrays = (int)arrays.size();
        smp.array_end_dim = 5 + 2 * smp.n_arrays;  // sampler.rs:344-345
        if (rp.sampler == PBRT_SAMPLER_HALTON) {
            if (smp.array_end_dim > 990) return invalid("Halton sampler: too many sample arrays for 1000 dimensions");
            elems = 0;
        }
        smp.n_elems = (int)elems;
        smp.tables = buf.alloc<float>((size_t)elems * n_pix, &ok);
        int2* d_arrays = buf.alloc<int2>(arrays.size(), &ok);
        smp.arrays = d_arrays;
        if (ok && !arrays.empty())
            HIP_TRY(ctx, hipMemcpyAsync(d_arrays, arrays.data(), arrays.size() * sizeof(int2), hipMemcpyHostToDevice, st));
        if (ok) HIP_TRY(ctx, hipStreamSynchronize(st));  // `arrays` leaves scope
    }
    int* d_prefix = buf.alloc<int>(prefix.size(), &ok);
    if (direct) {
        ds.stage = buf.alloc<int>(N, &ok);
        ds.ld_acc = buf.alloc<float4>(N, &ok);
        ds.frames = buf.alloc<float4>(N * (size_t)std::max(1, rp.max_depth) * 3, &ok);
        ds.light_strategy = rp.light_strategy;
        ds.mode = rp.integrator;
        ds.ao_samples = rp.ao_samples;
        ds.ao_cos_sample = rp.light_strategy != 0;
        if (prefix.back() >= 0xfff0) return invalid("too many light samples per vertex");
    }
    float* d_filter = nullptr;
    if (!box) {
        d_filter = buf.alloc<float>(256, &ok);
        if (ok) HIP_TRY(ctx, hipMemcpyAsync(d_filter, rp.filter_table, 256 * sizeof(float), hipMemcpyHostToDevice, st));
    }
    float4* accum = buf.alloc<float4>(n_pix, &ok);
    int2* d_origins = buf.alloc<int2>(origins.size(), &ok);
    if (!ok) return PBRT_HIP_ERR_OOM;
    HIP_TRY(ctx, hipMemcpyAsync(d_origins, origins.data(), origins.size() * sizeof(int2), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(accum, 0, (size_t)n_pix * sizeof(float4), st));
    HIP_TRY(ctx, hipMemcpyAsync(d_prefix, prefix.data(), prefix.size() * sizeof(int), hipMemcpyHostToDevice, st));
    TileList tiles{d_origins, (int)origins.size()};

    DevCamera cam;
    std::memcpy(cam.c2w, camera.camera_to_world, 64);
    std::memcpy(cam.r2c, camera.raster_to_camera, 64);
    cam.lens_radius = camera.lens_radius;
    cam.focal_distance = camera.focal_distance;
    cam.shutter_open = camera.shutter_open;
    cam.shutter_close = camera.shutter_close;
    cam.kind = camera.kind;
    if (camera.kind < PBRT_CAMERA_PERSPECTIVE || camera.kind > PBRT_CAMERA_ENVIRONMENT) return invalid("unknown camera kind");

    ShadeConsts sc;
    sc.bvh = s->d.bvh;
    sc.materials = s->d.materials;
    sc.lights = s->d.lights;
    sc.n_lights = s->d.n_lights;
    sc.n_infinite = s->d.n_infinite;
    sc.infinite_ids = s->d.infinite_ids;
    // create_light_sample_distribution (lightdistrib.rs:222-232): "uniform", or one light -> uniform
    sc.distrib = (rp.light_strategy == 0 || s->d.n_lights == 1) ? s->d.light_distrib_uniform : s->d.light_distrib_power;
    if (rp.integrator == PBRT_INTEGRATOR_PATH && (rp.light_strategy < 0 || rp.light_strategy > 2))
        return invalid("path: light_strategy must be 0 (uniform), 1 (power) or 2 (spatial)");
    std::memcpy(sc.env_cond_func, s->d.env_cond_func, sizeof(sc.env_cond_func));
    std::memcpy(sc.env_cond_cdf, s->d.env_cond_cdf, sizeof(sc.env_cond_cdf));
    std::memcpy(sc.env_cond_int, s->d.env_cond_int, sizeof(sc.env_cond_int));
    std::memcpy(sc.env_marg_func, s->d.env_marg_func, sizeof(sc.env_marg_func));
    std::memcpy(sc.env_marg_cdf, s->d.env_marg_cdf, sizeof(sc.env_marg_cdf));
    sc.env_marg_int = s->d.env_marg_int;
    sc.world_radius = s->d.world_radius;
    sc.light_sample_prefix = d_prefix;
    sc.total_light_samples = prefix.back();
    sc.spatial = nullptr;
    sc.n_voxel[0] = sc.n_voxel[1] = sc.n_voxel[2] = 1;
    if (rp.integrator == PBRT_INTEGRATOR_PATH && rp.light_strategy == 2 && s->d.n_lights > 1) {
        // create_light_sample_distribution("spatial") -> SpatialLightDistribution::new(scene, 64) (lightdistrib.rs:85-107, 228)
        if (!s->d_spatial) {
            const float* mn = s->d.bvh.root_min;
            const float* mx = s->d.bvh.root_max;
            float diag[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
            int ext = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);
            for (int i = 0; i < 3; ++i) s->spatial_voxels[i] = std::max(1, (int)std::round(diag[i] / diag[ext] * 64.0f));
            size_t n_voxels = (size_t)s->spatial_voxels[0] * s->spatial_voxels[1] * s->spatial_voxels[2];
            size_t floats = n_voxels * (size_t)(2 * s->d.n_lights + 2);
            if (floats > (1ull << 30)) return invalid("spatial light distribution: voxels x lights exceed 4 GB; use \"power\"");
            void* p = nullptr;
            HIP_TRY(ctx, hipMalloc(&p, floats * sizeof(float)));
            s->allocs.push_back(p);
            for (int i = 0; i < 3; ++i) sc.n_voxel[i] = s->spatial_voxels[i];
            hipLaunchKernelGGL(k_spatial_light_tables, dim3((unsigned)((n_voxels + 63) / 64)), dim3(64), 0, st, sc, (float*)p);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(st));
            s->d_spatial = (float*)p;
        }
        sc.spatial = s->d_spatial;
        for (int i = 0; i < 3; ++i) sc.n_voxel[i] = s->spatial_voxels[i];
    }

    hipEvent_t e_begin, e_end, e_t0, e_t1;
    HIP_TRY(ctx, hipEventCreate(&e_begin));
    HIP_TRY(ctx, hipEventCreate(&e_end));
    HIP_TRY(ctx, hipEventCreate(&e_t0));
    HIP_TRY(ctx, hipEventCreate(&e_t1));
    auto cleanup_events = [&]() {
        (void)hipEventDestroy(e_begin);
        (void)hipEventDestroy(e_end);
        (void)hipEventDestroy(e_t0);
        (void)hipEventDestroy(e_t1);
    };
    int rc = PBRT_HIP_OK;
#define RENDER_TRY(call)                                  \
    if (rc == PBRT_HIP_OK && !hip_ok(ctx, (call), #call)) rc = PBRT_HIP_ERR_DEVICE;

    RENDER_TRY(hipEventRecord(e_begin, st));
    bool tables_ready = !tabulated || rp.sampler == PBRT_SAMPLER_HALTON;
    for (int s0 = 0; s0 < rp.spp && rc == PBRT_HIP_OK; s0 += spp_pass) {
        PassParams pp;
        pp.smp = smp;
        pp.n_pix = n_pix;
        pp.n_samples = std::min(spp_pass, rp.spp - s0);
        pp.sample0 = s0;
        pp.spp = rp.spp;
        pp.width = rp.width;
        pp.height = rp.height;
        pp.x0 = rp.x0;
        pp.y0 = rp.y0;
        pp.x1 = rp.x1;
        pp.y1 = rp.y1;
        pp.seed = rp.seed;
        pp.max_depth = rp.max_depth;
        pp.rr_threshold = rp.rr_threshold;
        pp.light_strategy = rp.light_strategy;
        pp.filter_rx = frx;
        pp.filter_ry = fry;
        pp.filter_table = d_filter;
        pp.max_sample_luminance = rp.max_sample_luminance > 0.0f ? rp.max_sample_luminance : INFINITY;
        uint32_t n_paths = (uint32_t)n_pix * pp.n_samples;
        int cur = 0;
        if (!tables_ready) {  // Sampler::start_pixel for every pixel of this GPU, once per render
            hipLaunchKernelGGL(k_sampler_tables, dim3((n_pix + 63) / 64), dim3(64), 0, st, pp, tiles);
            RENDER_TRY(hipGetLastError());
            tables_ready = true;
        }
        hipLaunchKernelGGL(k_generate, dim3((n_paths + 255) / 256), dim3(256), 0, st, ps, q[cur], pp, cam, tiles);
        RENDER_TRY(hipGetLastError());
        if (direct) {
            RENDER_TRY(hipMemsetAsync(ds.stage, 0, (size_t)n_paths * sizeof(int), st));
            RENDER_TRY(hipMemsetAsync(ds.ld_acc, 0, (size_t)n_paths * sizeof(float4), st));
        }
        // the first wavefront is the identity: every path traces its camera ray and is shaded
        unsigned long long counts[2] = {n_paths, n_paths};
        local.camera_samples += (uint64_t)valid_pixels * pp.n_samples;
        local.rays_closest += (uint64_t)valid_pixels * pp.n_samples;
        bool first = true;
        int wavefront = 0;  // 0 = camera rays, 1 = first bounce + its shadow rays, ...
        while (rc == PBRT_HIP_OK && counts[1] > 0) {
            uint32_t n_trace = (uint32_t)counts[0], n_shade = (uint32_t)counts[1];
            if (n_trace > 0) {
                RENDER_TRY(hipMemsetAsync(ctx->d_work_counter, 0, kQueueSegments * sizeof(unsigned int), st));
                const uint32_t* trace_queue = q[cur].trace;
                if (sort_rays && wavefront >= sort_from && n_trace >= (1u << 20)) {
                    // from the second bounce on the rays of a wavefront start all over the scene (the first bounce still
                    // follows the pixel order of its camera rays): trace them in Morton order of their origins
                    if (!fused_keys) {
                        float3 lo = make_float3(q[cur].key_lo[0], q[cur].key_lo[1], q[cur].key_lo[2]);
                        float3 inv = make_float3(q[cur].key_inv[0], q[cur].key_inv[1], q[cur].key_inv[2]);
                        hipLaunchKernelGGL(k_ray_sort_keys, dim3((n_trace + 255) / 256), dim3(256), 0, st, ps, q[cur].trace, n_trace, lo,
                                           inv, sort_keys[0]);
                    }
                    size_t tb = sort_tmp_bytes;
                    if (pb::sort_pairs_u32(st, sort_tmp, &tb, sort_keys[0], sort_keys[1], q[cur].trace, sort_vals, n_trace, kSortKeyBits) != 0 &&
                        rc == PBRT_HIP_OK) {
                        ctx->last_error = "rocPRIM radix sort failed";
                        rc = PBRT_HIP_ERR_DEVICE;
                    }
                    trace_queue = sort_vals;
                }
                RENDER_TRY(hipEventRecord(e_t0, st));
                {
                    dim3 grid(persistent_grid(s)), block(kTraceBlock);
                    const bool inst = s->d.bvh.instanced != 0;
                    const int segments = (wavefront == 0 && !inst) ? kQueueSegments : 1;  // see trace.h
                    if (s->d.bvh.has_spheres) {
                        if (ctx->count_traversal)
                            hipLaunchKernelGGL((k_trace<true, false, true>), grid, block, 0, st, s->d.bvh, ps, trace_queue,
                                               n_trace, ctx->d_work_counter, ctx->d_counters, segments);
                        else
                            hipLaunchKernelGGL((k_trace<false, false, true>), grid, block, 0, st, s->d.bvh, ps, trace_queue,
                                               n_trace, ctx->d_work_counter, ctx->d_counters, segments);
                    } else if (ctx->count_traversal) {
                        if (inst)
                            hipLaunchKernelGGL((k_trace<true, true>), grid, block, 0, st, s->d.bvh, ps, trace_queue,
                                               n_trace, ctx->d_work_counter, ctx->d_counters, segments);
                        else
                            hipLaunchKernelGGL((k_trace<true, false>), grid, block, 0, st, s->d.bvh, ps, trace_queue,
                                               n_trace, ctx->d_work_counter, ctx->d_counters, segments);
                    } else {
                        if (inst)
                            hipLaunchKernelGGL((k_trace<false, true>), grid, block, 0, st, s->d.bvh, ps, trace_queue,
                                               n_trace, ctx->d_work_counter, ctx->d_counters, segments);
                        else
                            hipLaunchKernelGGL((k_trace<false, false>), grid, block, 0, st, s->d.bvh, ps, trace_queue,
                                               n_trace, ctx->d_work_counter, ctx->d_counters, segments);
                    }
                }
                RENDER_TRY(hipGetLastError());
                RENDER_TRY(hipEventRecord(e_t1, st));
            }
            int nxt = cur ^ 1;
            RENDER_TRY(hipMemsetAsync(q[nxt].counts64, 0, 2 * sizeof(unsigned long long), st));
            // the wavefront this launch of k_shade fills is traced in Morton order: have it write the keys as well
            q[nxt].keys = (sort_rays && fused_keys && wavefront + 1 >= sort_from) ? sort_keys[0] : nullptr;
            if (direct)
{
                dim3 sg((n_shade + 255) / 256), sb(256);
                if (rp.integrator == PBRT_INTEGRATOR_DIRECT)
                    hipLaunchKernelGGL((k_shade_direct<PBRT_INTEGRATOR_DIRECT>), sg, sb, 0, st, sc, ps, ds, q[cur], q[nxt], pp, tiles, n_shade);
                else if (rp.integrator == PBRT_INTEGRATOR_WHITTED)
                    hipLaunchKernelGGL((k_shade_direct<PBRT_INTEGRATOR_WHITTED>), sg, sb, 0, st, sc, ps, ds, q[cur], q[nxt], pp, tiles, n_shade);
                else
                    hipLaunchKernelGGL((k_shade_direct<PBRT_INTEGRATOR_AO>), sg, sb, 0, st, sc, ps, ds, q[cur], q[nxt], pp, tiles, n_shade);
            }
            else
                hipLaunchKernelGGL(k_shade, dim3((n_shade + 255) / 256), dim3(256), 0, st, sc, ps, q[cur], q[nxt], pp,
                                   tiles, n_shade);
            RENDER_TRY(hipGetLastError());
            RENDER_TRY(hipMemcpyAsync(ctx->h_counts, q[nxt].counts64, sizeof(counts), hipMemcpyDeviceToHost, st));
            RENDER_TRY(hipEventRecord(ctx->ev_sync, st));
            if (rc == PBRT_HIP_OK) {
                hipError_t qe;
                while ((qe = hipEventQuery(ctx->ev_sync)) == hipErrorNotReady) {
                }
                RENDER_TRY(qe);
            }
            counts[0] = ctx->h_counts[0];
            counts[1] = ctx->h_counts[1];
            if (n_trace > 0 && rc == PBRT_HIP_OK) {
                float ms = 0.0f;
                RENDER_TRY(hipEventElapsedTime(&ms, e_t0, e_t1));
                local.trace_ms += ms;
                local.trace_launches += 1;
                if (std::getenv("PBRT_HIP_TRACE_LOG"))
                    std::fprintf(stderr, "[pbrt_hip] k_trace: %u rays %.3f ms (%.0f Mrays/s)\n", n_trace, ms, n_trace / ms * 1e-3);
                ctx->trace_ms += ms;
                ctx->trace_launches += 1;
            }
            {
                uint64_t n_rays = counts[0] & 0xffffffffull, n_shadow = counts[0] >> 32;
                local.rays_closest += n_rays - n_shadow;
                local.rays_shadow += n_shadow;
                counts[0] = n_rays;
            }
            (void)first;
            first = false;
            wavefront += 1;
            cur = nxt;
        }
        if (!box) {
            hipLaunchKernelGGL(k_film_splat, dim3((n_paths + 255) / 256), dim3(256), 0, st, ps, pp, tiles, d_film);
            RENDER_TRY(hipGetLastError());
            continue;
        }
        hipLaunchKernelGGL(k_film_accumulate, dim3((n_pix + 255) / 256), dim3(256), 0, st, ps, pp, tiles, accum, d_film);
        RENDER_TRY(hipGetLastError());
        if (s0 + spp_pass >= rp.spp) {
            hipLaunchKernelGGL(k_film_merge, dim3((n_pix + 255) / 256), dim3(256), 0, st, pp, tiles, accum, d_film);
            RENDER_TRY(hipGetLastError());
        }
    }
    RENDER_TRY(hipEventRecord(e_end, st));
    RENDER_TRY(hipStreamSynchronize(st));
    if (rc == PBRT_HIP_OK) {
        float ms = 0.0f;
        RENDER_TRY(hipEventElapsedTime(&ms, e_begin, e_end));
        local.total_ms = ms;
    }
#undef RENDER_TRY
    cleanup_events();
    if (stats) *stats = local;
    return rc;
}

#ifdef PB_LANE_STATS
extern "C" int pbrt_hip_debug_lane_stats(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pb::g_lane_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pb::g_lane_stats), z, sizeof(z)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

// pbrt_hip.hip — C ABI (include/pbrt_hip.h) over the gfx950 kernels.
//
// Stands in for the reference's Integrator / Primitive / Scene trait objects:
//   pbrt_hip_scene_create   Scene::new, src/core/scene.rs:18-34
//   pbrt_hip_intersect[_p]  Scene::intersect / intersect_p, src/core/scene.rs:40-46
//   pbrt_hip_render         SamplerIntegrator::render, src/core/integrator.rs:399-480
#include <hip/hip_runtime.h>
#include "abi_guard.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_wide.h"
#include "scene.h"
#include "trace.h"
#include "trace_persistent.h"
#include "trace_wide.h"
#include "trace_stackless.h"

using namespace pb;

// why the calling THREAD's last pbrt_hip_context_create failed (pbrt_hip_last_error(NULL)): contexts may be created from several
// host threads at once, and a failed creation has no context to carry its message
static thread_local std::string g_create_error;



// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
extern "C" int pbrt_hip_context_create(int device_id, PbrtHipContext** out) try {
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
    ctx->trace_log = std::getenv("PBRT_HIP_TRACE_LOG") != nullptr;  // read once, here: the render loop asks per wavefront
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        hipMalloc((void**)&ctx->d_counters, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&ctx->d_work_counter, kWorkCounters * sizeof(unsigned int)) != hipSuccess ||
        hipHostMalloc((void**)&ctx->h_counts, 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_sync, hipEventDisableTiming) != hipSuccess ||
        hipMemset(ctx->d_counters, 0, 8 * sizeof(unsigned long long)) != hipSuccess) {
        g_create_error = "stream / event creation failed";
        delete ctx;
        return PBRT_HIP_ERR_DEVICE;
    }
    *out = ctx;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" void pbrt_hip_context_destroy(PbrtHipContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->lost) {
        // a kernel of an abandoned call may never finish: waiting on the stream or freeing what it writes (hipFree waits
        // for the device) could block for ever. The device memory of a lost context goes back with the process.
        delete ctx;
        return;
    }
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->d_work_counter) (void)hipFree(ctx->d_work_counter);
    if (ctx->d_special_list) (void)hipFree(ctx->d_special_list);
    for (auto& b : ctx->block_cache)
        if (b.ptr) (void)hipFree(b.ptr);
    ctx->block_cache.clear();
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->ev_sync) (void)hipEventDestroy(ctx->ev_sync);
    if (ctx->d_halton_primes) (void)hipFree(ctx->d_halton_primes);
    if (ctx->d_halton_perms) (void)hipFree(ctx->d_halton_perms);
    delete ctx;
}

extern "C" int pbrt_hip_context_is_lost(const PbrtHipContext* ctx) try {
    return ctx ? (ctx->lost.load(std::memory_order_acquire) ? 1 : 0) : -1;  // no lock: `lost` is atomic (polled from other threads)
}
PB_ABI_CATCH

extern "C" int pbrt_hip_context_set_deadline(PbrtHipContext* ctx, double seconds) try {
    if (!ctx || !(seconds > 0.0)) return PBRT_HIP_ERR_INVALID;
    PB_LOCK(ctx);
    ctx->wavefront_deadline_s = seconds;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" const char* pbrt_hip_last_error(const PbrtHipContext* ctx) {
    return ctx ? ctx->last_error.c_str() : g_create_error.c_str();
}

extern "C" int pbrt_hip_synchronize(PbrtHipContext* ctx) try {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_trace_timing(PbrtHipContext* ctx, int reset, double* total_ms, uint64_t* launches) try {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    if (total_ms) *total_ms = ctx->trace_ms;
    if (launches) *launches = ctx->trace_launches;
    if (reset) {
        ctx->trace_ms = 0.0;
        ctx->trace_launches = 0;
    }
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_set_counting(PbrtHipContext* ctx, int enable) try {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    ctx->count_traversal = (enable == 1 || enable == 2) ? enable : 0;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_context_set_wide_build(PbrtHipContext* ctx, int where) try {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    if (where < PBRT_WIDE_BUILD_DEVICE || where > PBRT_WIDE_BUILD_NONE) {
        ctx->last_error = "wide build must be PBRT_WIDE_BUILD_DEVICE, _HOST or _NONE";
        return PBRT_HIP_ERR_INVALID;
    }
    ctx->wide_build = where;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_context_set_wide_layout(PbrtHipContext* ctx, int layout) try {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    if (layout < PBRT_WIDE_LAYOUT_AUTO || layout > PBRT_WIDE_LAYOUT_LINES) {
        ctx->last_error = "wide layout must be PBRT_WIDE_LAYOUT_AUTO, _PACKED or _LINES";
        return PBRT_HIP_ERR_INVALID;
    }
    ctx->wide_layout = layout;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_scene_wide_stride(const PbrtHipScene* s) try {
    return s ? (s->has_wide ? s->wide.vec_stride * 16 : 0) : -1;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_context_set_traversal(PbrtHipContext* ctx, int traversal) try {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    if (traversal < PBRT_TRAVERSAL_AUTO || traversal > PBRT_TRAVERSAL_STACKLESS) {
        ctx->last_error = "traversal must be PBRT_TRAVERSAL_AUTO, _STACK or _STACKLESS";
        return PBRT_HIP_ERR_INVALID;
    }
    ctx->traversal = traversal;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_get_counters(PbrtHipContext* ctx, int reset, uint64_t counters[4]) try {
    if (!ctx || !counters) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
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
PB_ABI_CATCH

extern "C" int pbrt_hip_get_wide_counters(PbrtHipContext* ctx, int reset, uint64_t counters[4]) try {
    if (!ctx || !counters) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long h[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpy(h, ctx->d_counters + 4, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) counters[k] = h[k];
    if (reset) {
        HIP_TRY(ctx, hipMemset(ctx->d_counters + 4, 0, sizeof(h)));
    }
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

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
    int n_interior = 0, count_bits = 0, depth = 0, max_count = 1;
    int32_t root_ref = 0;
};
// slot_base: leaf slot of the tree's first primitive in the scene's slot numbering; force_count_bits >= 0: the width
// of (n_primitives - 1) in the leaf references (several object trees share one width).
static const char* convert_tree(const PbrtLinearBVHNode* nodes, int32_t n_nodes, int32_t n_prims, int32_t base,
                                TreeLayout* out, int32_t slot_base = 0, int force_count_bits = -1, int64_t n_slots_total = -1) {
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
    if (force_count_bits >= 0) {
        if (force_count_bits < count_bits) return "leaf larger than the shared leaf-reference width";
        count_bits = force_count_bits;
    }
    if (((n_slots_total >= 0 ? n_slots_total : (int64_t)n_prims) << count_bits) >= (1ll << 31)) return "scene too large for 31-bit leaf references";
    auto child_ref = [&](int32_t node) -> int32_t {
        const PbrtLinearBVHNode& nd = nodes[node];
        if (nd.n_primitives > 0)
            return ~(int32_t)(((uint32_t)(nd.offset + slot_base) << count_bits) | (uint32_t)(nd.n_primitives - 1));
        return base + interior_index[node];
    };
    out->inodes.assign((size_t)n_interior * 16, 0.0f);
    std::vector<int32_t> parent_record(n_nodes, -1);  // of interior nodes: the record holding them as a child (trace_stackless.h)
    for (int32_t i = 0; i < n_nodes; ++i) {
        if (interior_index[i] < 0) continue;
        parent_record[i + 1] = parent_record[nodes[i].offset] = base + interior_index[i];
    }
    for (int32_t i = 0; i < n_nodes; ++i) {
        if (interior_index[i] < 0) continue;
        const PbrtLinearBVHNode& c0 = nodes[i + 1];
        const PbrtLinearBVHNode& c1 = nodes[nodes[i].offset];
        float* r = &out->inodes[(size_t)interior_index[i] * 16];
        r[0] = c0.bounds_min[0]; r[1] = c0.bounds_min[1]; r[2] = c0.bounds_min[2];
        r[3] = c0.bounds_max[0]; r[4] = c0.bounds_max[1]; r[5] = c0.bounds_max[2];
        r[6] = c1.bounds_min[0]; r[7] = c1.bounds_min[1]; r[8] = c1.bounds_min[2];
        r[9] = c1.bounds_max[0]; r[10] = c1.bounds_max[1]; r[11] = c1.bounds_max[2];
        int32_t refs[4] = {child_ref(i + 1), child_ref(nodes[i].offset), (int32_t)nodes[i].axis, parent_record[i]};
        std::memcpy(r + 12, refs, 16);
    }
    out->n_interior = n_interior;
    out->count_bits = count_bits;
    out->max_count = max_count;
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
// Two-level scenes: the main geometry arguments of scene_create_impl are then the COMBINED mesh (every object's
// triangles, object after object, then the world-space triangles) and prim_order the combined leaf order
// (object k's leaf order at obj_tri_offset[k], world triangles in caller order at world_tri_offset).
struct InstancingArgs {
    const PbrtInstance* instances = nullptr;
    int32_t n_instances = 0;
    const PbrtLinearBVHNode* tlas_nodes = nullptr;
    int32_t n_tlas_nodes = 0;
    const int32_t* tlas_order = nullptr;
    const PbrtObject* objects = nullptr;
    int32_t n_objects = 0;
    const int32_t* instance_object = nullptr;  // nullptr = all instances of object 0
    std::vector<int32_t> obj_tri_offset;       // n_objects + 1
    int32_t n_world_tris = 0, world_tri_offset = 0;
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
                                     int32_t n_nodes, const int32_t* prim_order, PbrtHipScene** out) try {
    return scene_create_impl(ctx, positions, n_verts, indices, n_tris, tri_material, materials, n_materials, tri_light,
                             lights, n_lights, nodes, n_nodes, prim_order, InstancingArgs(), out);
}
PB_ABI_CATCH

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
                                           PbrtHipScene** out, double* build_ms, double* layout_ms) try {
    if (!ctx || !out) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
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
    if (dt.wide.nodes) {
        (void)hipFree(dt.wide.nodes);
        (void)hipFree(dt.wide.tris);
        (void)hipFree(dt.wide.leaf_boxes);
    }
    return rc;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_scene_create_with_spheres(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                                  const int32_t* indices, int32_t n_tris, const int32_t* tri_material,
                                                  const PbrtMaterial* materials, int32_t n_materials, const int32_t* tri_light,
                                                  const PbrtLight* lights, int32_t n_lights, const float* spheres,
                                                  const int32_t* sphere_material, const int32_t* sphere_light, int32_t n_spheres,
                                                  const PbrtLinearBVHNode* nodes, int32_t n_nodes, const int32_t* prim_order,
                                                  PbrtHipScene** out) try {
    SphereArgs sa;
    sa.spheres = spheres;
    sa.material = sphere_material;
    sa.light = sphere_light;
    sa.n = n_spheres;
    return scene_create_impl(ctx, positions, n_verts, indices, n_tris, tri_material, materials, n_materials, tri_light,
                             lights, n_lights, nodes, n_nodes, prim_order, InstancingArgs(), out, nullptr, sa);
}
PB_ABI_CATCH

extern "C" int pbrt_hip_scene_create_two_level(PbrtHipContext* ctx, const PbrtObject* objects, int32_t n_objects,
                                               const PbrtInstance* instances, const int32_t* instance_object, int32_t n_instances,
                                               const float* world_positions, int32_t n_world_verts, const int32_t* world_indices,
                                               int32_t n_world_tris, const int32_t* world_tri_material,
                                               const int32_t* world_tri_light, const PbrtMaterial* materials, int32_t n_materials,
                                               const PbrtLight* lights, int32_t n_lights, const PbrtLinearBVHNode* tlas_nodes,
                                               int32_t n_tlas_nodes, const int32_t* tlas_order, PbrtHipScene** out) try {
    if (!ctx || !out) return PBRT_HIP_ERR_INVALID;
    auto fail = [&](const char* msg) {
        ctx->last_error = msg;
        return PBRT_HIP_ERR_INVALID;
    };
    if (!objects || n_objects <= 0 || !instances || n_instances <= 0 || !tlas_nodes || n_tlas_nodes <= 0 || !tlas_order)
        return fail("two-level scene needs objects, instances and a top-level node array");
    if (n_world_tris < 0 || n_world_verts < 0 || (n_world_tris > 0 && (!world_positions || !world_indices || n_world_verts <= 0)))
        return fail("bad world-space triangle arguments");
    // ---- the combined mesh: object after object, then the world-space triangles ----
    InstancingArgs ia;
    ia.instances = instances;
    ia.n_instances = n_instances;
    ia.tlas_nodes = tlas_nodes;
    ia.n_tlas_nodes = n_tlas_nodes;
    ia.tlas_order = tlas_order;
    ia.objects = objects;
    ia.n_objects = n_objects;
    ia.instance_object = instance_object;
    ia.n_world_tris = n_world_tris;
    int64_t n_verts = 0, n_tris = 0, n_nodes = 0;
    ia.obj_tri_offset.assign(n_objects + 1, 0);
    for (int32_t k = 0; k < n_objects; ++k) {
        const PbrtObject& o = objects[k];
        if (!o.positions || !o.indices || !o.nodes || !o.prim_order || o.n_verts <= 0 || o.n_tris <= 0 || o.n_nodes <= 0)
            return fail("object aggregate with a null pointer or an empty mesh / tree");
        for (int64_t i = 0; i < 3 * (int64_t)o.n_tris; ++i)
            if (o.indices[i] < 0 || o.indices[i] >= o.n_verts) return fail("object vertex index out of range");
        for (int32_t i = 0; i < o.n_tris; ++i)
            if (o.prim_order[i] < 0 || o.prim_order[i] >= o.n_tris) return fail("object prim_order entry out of range");
        n_verts += o.n_verts;
        n_tris += o.n_tris;
        n_nodes += o.n_nodes;
        ia.obj_tri_offset[k + 1] = (int32_t)n_tris;
    }
    ia.world_tri_offset = (int32_t)n_tris;
    for (int64_t i = 0; i < 3 * (int64_t)n_world_tris; ++i)
        if (world_indices[i] < 0 || world_indices[i] >= n_world_verts) return fail("world triangle vertex index out of range");
    n_verts += n_world_verts;
    n_tris += n_world_tris;
    if (n_tris >= (1ll << 30) || n_verts >= (1ll << 31)) return fail("two-level scene too large");
    for (int32_t i = 0; i < n_instances; ++i)
        if (instance_object && (instance_object[i] < 0 || instance_object[i] >= n_objects)) return fail("instance_object out of range");
    for (int32_t i = 0; i < n_lights; ++i)
        if (lights[i].type == PBRT_LIGHT_DIFFUSE_AREA && (lights[i].prim < 0 || lights[i].prim >= n_world_tris))
            return fail("area lights sit on world-space triangles (instanced primitives cannot be area lights): prim out of range");
    std::vector<float> pos((size_t)n_verts * 3);
    std::vector<int32_t> idx((size_t)n_tris * 3), mat((size_t)n_tris, 0), lgt((size_t)n_tris, -1), order((size_t)n_tris);
    int64_t v0 = 0, t0 = 0;
    for (int32_t k = 0; k < n_objects; ++k) {
        const PbrtObject& o = objects[k];
        std::memcpy(&pos[(size_t)v0 * 3], o.positions, (size_t)o.n_verts * 12);
        for (int64_t i = 0; i < 3 * (int64_t)o.n_tris; ++i) idx[(size_t)t0 * 3 + i] = o.indices[i] + (int32_t)v0;
        for (int32_t i = 0; i < o.n_tris; ++i) {
            mat[(size_t)t0 + i] = o.tri_material ? o.tri_material[i] : 0;
            order[(size_t)t0 + i] = o.prim_order[i] + (int32_t)t0;
        }
        v0 += o.n_verts;
        t0 += o.n_tris;
    }
    if (n_world_tris > 0) {
        std::memcpy(&pos[(size_t)v0 * 3], world_positions, (size_t)n_world_verts * 12);
        for (int64_t i = 0; i < 3 * (int64_t)n_world_tris; ++i) idx[(size_t)t0 * 3 + i] = world_indices[i] + (int32_t)v0;
        for (int32_t i = 0; i < n_world_tris; ++i) {
            mat[(size_t)t0 + i] = world_tri_material ? world_tri_material[i] : 0;
            lgt[(size_t)t0 + i] = world_tri_light ? world_tri_light[i] : -1;
            order[(size_t)t0 + i] = (int32_t)t0 + i;
        }
    }
    return scene_create_impl(ctx, pos.data(), (int32_t)n_verts, idx.data(), (int32_t)n_tris, mat.data(), materials, n_materials,
                             lgt.data(), lights, n_lights, nullptr, (int32_t)n_nodes, order.data(), ia, out);
}
PB_ABI_CATCH

extern "C" int pbrt_hip_scene_create_instanced(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                               const int32_t* indices, int32_t n_tris, const int32_t* tri_material,
                                               const PbrtMaterial* materials, int32_t n_materials,
                                               const PbrtLight* lights, int32_t n_lights,
                                               const PbrtLinearBVHNode* blas_nodes, int32_t n_blas_nodes,
                                               const int32_t* blas_order, const PbrtInstance* instances,
                                               int32_t n_instances, const PbrtLinearBVHNode* tlas_nodes,
                                               int32_t n_tlas_nodes, const int32_t* tlas_order, PbrtHipScene** out) try {
    if (!ctx || !out) return PBRT_HIP_ERR_INVALID;
    PbrtObject obj;
    obj.positions = positions;
    obj.n_verts = n_verts;
    obj.indices = indices;
    obj.n_tris = n_tris;
    obj.tri_material = tri_material;
    obj.nodes = blas_nodes;
    obj.n_nodes = n_blas_nodes;
    obj.prim_order = blas_order;
    return pbrt_hip_scene_create_two_level(ctx, &obj, 1, instances, nullptr, n_instances, nullptr, 0, nullptr, 0, nullptr, nullptr,
                                           materials, n_materials, lights, n_lights, tlas_nodes, n_tlas_nodes, tlas_order, out);
}
PB_ABI_CATCH

static int scene_create_impl(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                             int32_t n_tris, const int32_t* tri_material, const PbrtMaterial* materials,
                             int32_t n_materials, const int32_t* tri_light, const PbrtLight* lights, int32_t n_lights,
                             const PbrtLinearBVHNode* nodes, int32_t n_nodes, const int32_t* prim_order,
                             const InstancingArgs& ia, PbrtHipScene** out, pb::DeviceTree* dt, const SphereArgs& sa) {
    // dt != nullptr: the tree, the triangle records and the leaf order are already on the device
    // (pbrt_hip_scene_create_hlbvh); nodes / prim_order are then unused.
    if (!ctx || !out) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    *out = nullptr;
    if (dt) n_nodes = dt->n_nodes;
    auto fail = [&](const char* msg) {
        ctx->last_error = msg;
        return PBRT_HIP_ERR_INVALID;
    };
    if (n_tris <= 0 || n_verts <= 0 || n_nodes <= 0) return fail("empty scene: n_tris, n_verts and n_nodes must be > 0");
    if (!positions || !indices || (!dt && ((!nodes && ia.n_instances == 0) || !prim_order))) return fail("null geometry / BVH pointer");
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
        if (lights[i].type == PBRT_LIGHT_DIFFUSE_AREA &&
            (lights[i].prim < 0 || lights[i].prim >= (ia.n_instances > 0 ? ia.n_world_tris : n_tris + sa.n)))
            return fail("area light primitive out of range");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- validate the tree(s) and lay out the interior records ----
    // single level: one tree over the triangles. Two levels (instancing): top-level tree over the
    // instances first, then the object-level tree over the triangles, in one record array.
    const bool instanced = ia.n_instances > 0;
    const int32_t n_top = ia.n_instances + ia.n_world_tris;  // primitives of the top-level aggregate
    TreeLayout top, obj;
    std::vector<TreeLayout> obj_trees;
    if (instanced) {
        for (int32_t i = 0; i < n_top; ++i)
            if (ia.tlas_order[i] < 0 || ia.tlas_order[i] >= n_top) return fail("tlas_order entry out of range");
        for (int32_t i = 0; i < ia.n_instances; ++i) {
            const float* m = ia.instances[i].to_world;
            const float* mi = ia.instances[i].to_object;
            if (m[12] != 0.0f || m[13] != 0.0f || m[14] != 0.0f || m[15] != 1.0f || mi[12] != 0.0f || mi[13] != 0.0f ||
                mi[14] != 0.0f || mi[15] != 1.0f)
                return fail("instance transforms must be affine (last row 0 0 0 1)");
            if (ia.instances[i].material >= n_materials) return fail("instance material out of range");
        }
        if (const char* e = convert_tree(ia.tlas_nodes, ia.n_tlas_nodes, n_top, 0, &top)) return fail(e);
        // the object-level trees behind it in one record array; one leaf-reference width for all of them
        int max_count = 1;
        for (int32_t k = 0; k < ia.n_objects; ++k)
            for (int32_t i = 0; i < ia.objects[k].n_nodes; ++i) max_count = std::max<int>(max_count, ia.objects[k].nodes[i].n_primitives);
        int bits = 0;
        while ((1 << bits) < max_count) ++bits;
        int base = top.n_interior, deepest = 0;
        obj_trees.resize(ia.n_objects);
        for (int32_t k = 0; k < ia.n_objects; ++k) {
            const PbrtObject& o = ia.objects[k];
            if (const char* e = convert_tree(o.nodes, o.n_nodes, o.n_tris, base, &obj_trees[k], ia.obj_tri_offset[k], bits, n_tris)) return fail(e);
            base += obj_trees[k].n_interior;
            deepest = std::max(deepest, obj_trees[k].depth);
            obj.inodes.insert(obj.inodes.end(), obj_trees[k].inodes.begin(), obj_trees[k].inodes.end());
        }
        obj.n_interior = base - top.n_interior;
        obj.count_bits = bits;
        obj.depth = deepest;
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
            int32_t prim = lights[i].prim + (instanced ? ia.world_tri_offset : 0);  // two-level: prim counts the world-space triangles
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
    std::vector<float> top_slot_records;  // two-level scenes: the 112-B records of the top-level leaf slots
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
        d.bvh.blas_count_bits = obj.count_bits;
        d.bvh.general_top = (ia.n_objects > 1 || ia.n_world_tris > 0) ? 1 : 0;
        d.bvh.blas_root_ref = obj_trees[0].root_ref;
        std::memcpy(d.bvh.blas_root_min, ia.objects[0].nodes[0].bounds_min, 12);
        std::memcpy(d.bvh.blas_root_max, ia.objects[0].nodes[0].bounds_max, 12);
        // object table: (root box min, root reference) (root box max, -)
        std::vector<float> objs((size_t)ia.n_objects * 8, 0.0f);
        for (int32_t k = 0; k < ia.n_objects; ++k) {
            const PbrtLinearBVHNode& root = ia.objects[k].nodes[0];
            float* r = &objs[(size_t)k * 8];
            std::memcpy(r, root.bounds_min, 12);
            std::memcpy(r + 3, &obj_trees[k].root_ref, 4);
            std::memcpy(r + 4, root.bounds_max, 12);
        }
        d.bvh.objects = (const float4*)dev_upload(s, objs.data(), objs.size(), &ok);
        // records of the top-level leaf slots: to_object rows 0-2, to_world rows 0-2, (material, instance id, object, kind);
        // a world-space triangle beside the instances: kind 1, its leaf slot in the third field
        std::vector<float>& inst = top_slot_records;
        inst.assign((size_t)n_top * 28, 0.0f);
        std::vector<int32_t> slot_inst(n_top);
        std::vector<char> seen(n_top, 0);
        for (int32_t slot = 0; slot < n_top; ++slot) {
            int32_t id = ia.tlas_order[slot];
            if (seen[id]) ok = false;
            seen[id] = 1;
            float* r = &inst[(size_t)slot * 28];
            if (id < ia.n_instances) {
                slot_inst[slot] = id;
                std::memcpy(r, ia.instances[id].to_object, 48);
                std::memcpy(r + 12, ia.instances[id].to_world, 48);
                int32_t meta[4] = {ia.instances[id].material, id, ia.instance_object ? ia.instance_object[id] : 0, 0};
                std::memcpy(r + 24, meta, 16);
            } else {
                slot_inst[slot] = -1;
                int32_t meta[4] = {-1, -1, prim_slot[ia.world_tri_offset + (id - ia.n_instances)], 1};
                std::memcpy(r + 24, meta, 16);
            }
        }
        if (!ok) ctx->last_error = "tlas_order is not a permutation";
        d.bvh.instances = (const float4*)dev_upload(s, inst.data(), inst.size(), &ok);
        d.slot_instance = dev_upload(s, slot_inst.data(), slot_inst.size(), &ok);
        s->n_instances = ia.n_instances;
    }
    // ---- 4-wide quantised records over the same tree (wide_bvh.h): single-level triangle scenes whose tree came
    // from the host; PBRT_WIDE_BUILD_NONE keeps the scene on the binary records ----
    int spill_entries = kStackSpill;
    {
        if (ctx->wide_build == PBRT_WIDE_BUILD_NONE) {
            s->wide_reason = "disabled by PBRT_WIDE_BUILD_NONE";
        } else if (sa.n > 0) {
            s->wide_reason = "scene with spheres";
        } else if (dt && dt->wide.n_records >= 0) {
            // laid out on the device beside the tree (wide_gpu.hip): the scene takes the arrays over
            s->wide.nodes = (const uint4*)dt->wide.nodes;
            s->wide.tris = (const float4*)dt->wide.tris;
            s->wide.leaf_boxes = (const float4*)dt->wide.leaf_boxes;
            s->allocs.push_back(dt->wide.nodes);
            s->allocs.push_back(dt->wide.tris);
            s->allocs.push_back(dt->wide.leaf_boxes);
            dt->wide.nodes = nullptr;
            dt->wide.tris = dt->wide.leaf_boxes = nullptr;
            s->wide.root_ref = dt->wide.root_ref;
            s->n_wide_records = dt->wide.n_records;
            s->has_wide = true;
            spill_entries = std::max(spill_entries, dt->wide.stack_need + 1 - kWideStackLds);
        } else if (dt && dt->h_nodes.empty()) {
            s->wide_reason = dt->wide_reason ? dt->wide_reason : "tree built on the device: no wide records";
        } else if (instanced) {
            // two levels: the top-level tree's records first (its leaves keep their exact boxes and name top-level
            // primitives in wide order), then every object aggregate's; one triangle / leaf-box array for all objects
            pb::WideTree top;
            const char* why = pb::build_wide_tree(ia.tlas_nodes, ia.n_tlas_nodes, nullptr, n_top, &top);
            std::vector<pb::WideTree> wobj(ia.n_objects);
            int record_base = why ? 0 : top.n_records, deepest = 0;
            for (int32_t k = 0; k < ia.n_objects && !why; ++k) {
                pb::WideBase b;
                b.record = record_base;
                b.tri = b.slot = ia.obj_tri_offset[k];
                why = pb::build_wide_tree(ia.objects[k].nodes, ia.objects[k].n_nodes, tris.data() + 12 * (size_t)ia.obj_tri_offset[k],
                                          ia.objects[k].n_tris, &wobj[k], b);
                if (!why) {
                    record_base += wobj[k].n_records;
                    deepest = std::max(deepest, wobj[k].stack_need);
                }
            }
            if (why) {
                s->wide_reason = why;
            } else {
                std::vector<uint32_t> wn;
                wn.insert(wn.end(), top.nodes.begin(), top.nodes.begin() + (size_t)top.n_records * kWideNodeDwords);
                std::vector<float> wt((size_t)n_prims * 12, 0.0f), wb((size_t)n_prims * 8, 0.0f), wo((size_t)ia.n_objects * 8, 0.0f);
                for (int32_t k = 0; k < ia.n_objects; ++k) {
                    wn.insert(wn.end(), wobj[k].nodes.begin(), wobj[k].nodes.begin() + (size_t)wobj[k].n_records * kWideNodeDwords);
                    std::memcpy(&wt[12 * (size_t)ia.obj_tri_offset[k]], wobj[k].tris.data(), wobj[k].tris.size() * 4);
                    std::memcpy(&wb[8 * (size_t)ia.obj_tri_offset[k]], wobj[k].leaf_boxes.data(), wobj[k].leaf_boxes.size() * 4);
                    const PbrtLinearBVHNode& root = ia.objects[k].nodes[0];
                    float* r = &wo[(size_t)k * 8];
                    std::memcpy(r, root.bounds_min, 12);
                    std::memcpy(r + 3, &wobj[k].root_ref, 4);
                    std::memcpy(r + 4, root.bounds_max, 12);
                }
                if (wn.empty()) wn.assign(kWideNodeDwords, 0u);
                // the top-level entries as the wide kernel reads them, in wide order, 20 floats each: the exact box of the leaf
                // the entry belongs to with two words in the w fields — the binary-layout top slot (hit records name it) and
                // the object index (an instance) or 0x40000000 | leaf slot (a world-space triangle) — then the three
                // world-to-object rows
                std::vector<float> ws((size_t)n_top * 20);
                for (int32_t pos = 0; pos < n_top; ++pos) {
                    const int32_t slot = top.order[pos];
                    const float* src = &top_slot_records[(size_t)slot * 28];
                    float* dst = &ws[(size_t)pos * 20];
                    int32_t meta[4];
                    std::memcpy(meta, src + 24, 16);  // (material, id, object | leaf slot, kind)
                    const int32_t word = meta[3] == 1 ? (0x40000000 | meta[2]) : meta[2];
                    std::memcpy(dst, &top.leaf_boxes[(size_t)pos * 8], 12);
                    std::memcpy(dst + 3, &slot, 4);
                    std::memcpy(dst + 4, &top.leaf_boxes[(size_t)pos * 8 + 4], 12);
                    std::memcpy(dst + 7, &word, 4);
                    std::memcpy(dst + 8, src, 48);
                }
                s->wide.nodes = (const uint4*)dev_upload(s, wn.data(), wn.size(), &ok);
                s->wide.tris = (const float4*)dev_upload(s, wt.data(), wt.size(), &ok);
                s->wide.leaf_boxes = (const float4*)dev_upload(s, wb.data(), wb.size(), &ok);
                s->wide.top_slots = (const float4*)dev_upload(s, ws.data(), ws.size(), &ok);
                s->wide.top_boxes = (const float4*)dev_upload(s, top.leaf_boxes.data(), top.leaf_boxes.size(), &ok);
                s->wide.objects = (const float4*)dev_upload(s, wo.data(), wo.size(), &ok);
                s->wide.slot_tris = d.bvh.tris;
                s->wide.root_ref = top.root_ref;
                s->wide.general_top = d.bvh.general_top;
                s->wide.obj0_root = wobj[0].root_ref;
                std::memcpy(s->wide.obj0_min, ia.objects[0].nodes[0].bounds_min, 12);
                std::memcpy(s->wide.obj0_max, ia.objects[0].nodes[0].bounds_max, 12);
                s->n_wide_records = record_base;
                s->has_wide = true;
                spill_entries = std::max(spill_entries, top.stack_need + deepest + 2 - pb::wide_stack_lds(1));
            }
        } else if (!dt && ok && ctx->wide_build != PBRT_WIDE_BUILD_HOST) {
            // a tree from the host: its flat nodes go up once, the records are laid out on the device from them and from
            // the triangle records already there (wide_gpu.hip; the host builder produces the same bytes, tens of ms per
            // million triangles slower: PBRT_WIDE_BUILD_HOST)
            bool up = true;
            PbrtLinearBVHNode* d_flat = dev_upload(s, nodes, (size_t)n_nodes, &up);
            pb::WideDeviceTree wd;
            const char* why = nullptr;
            if (!up || !pb::build_wide_tree_device(ctx, d_flat, n_nodes, (const float*)d.bvh.tris, n_prims, nodes[0], &wd, &why)) {
                ok = false;
            } else if (why) {
                s->wide_reason = why;
            } else {
                s->wide.nodes = (const uint4*)wd.nodes;
                s->wide.tris = (const float4*)wd.tris;
                s->wide.leaf_boxes = (const float4*)wd.leaf_boxes;
                s->allocs.push_back(wd.nodes);
                s->allocs.push_back(wd.tris);
                s->allocs.push_back(wd.leaf_boxes);
                s->wide.root_ref = wd.root_ref;
                s->n_wide_records = wd.n_records;
                s->has_wide = true;
                spill_entries = std::max(spill_entries, wd.stack_need + 1 - kWideStackLds);
            }
            if (d_flat) {  // only the builder needed it
                (void)hipFree(d_flat);
                s->allocs.erase(std::remove(s->allocs.begin(), s->allocs.end(), (void*)d_flat), s->allocs.end());
            }
        } else {
            pb::WideTree wt;
            const char* why = dt ? pb::build_wide_tree(dt->h_nodes.data(), n_nodes, dt->h_tris.data(), n_prims, &wt)
                                 : pb::build_wide_tree(nodes, n_nodes, tris.data(), n_prims, &wt);
            if (why) {
                s->wide_reason = why;
            } else {
                s->wide.nodes = (const uint4*)dev_upload(s, wt.nodes.data(), wt.nodes.size(), &ok);
                s->wide.tris = (const float4*)dev_upload(s, wt.tris.data(), wt.tris.size(), &ok);
                s->wide.leaf_boxes = (const float4*)dev_upload(s, wt.leaf_boxes.data(), wt.leaf_boxes.size(), &ok);
                s->wide.root_ref = wt.root_ref;
                s->n_wide_records = wt.n_records;
                s->has_wide = true;
                spill_entries = std::max(spill_entries, wt.stack_need + 1 - kWideStackLds);
            }
        }
    }
    // one 64-byte line per record / wide-order triangle where the tree is too large for L2 (wide_bvh.h: kWideLineAlignBytes;
    // pbrt_hip_context_set_wide_layout overrides): the builders' packed arrays are spread on the device and released
    s->wide.vec_stride = 3;
    if (s->has_wide && ok) {
        const size_t packed_bytes = ((size_t)std::max(s->n_wide_records, 1) + (size_t)n_prims) * 48;
        // (two-level scenes stay packed: the trees of a scene of instances are small, and trace_wide<.., INST> keeps its compile-time stride)
        const bool lines = !instanced && (ctx->wide_layout == PBRT_WIDE_LAYOUT_LINES || (ctx->wide_layout == PBRT_WIDE_LAYOUT_AUTO && packed_bytes > pb::kWideLineAlignBytes));
        auto spread = [&](const void* packed, size_t count) -> void* {
            void* out = nullptr;
            count = count ? count : 1;
            if (!hip_ok(ctx, hipMalloc(&out, count * 64), "hipMalloc wide lines")) return nullptr;
            if (!hip_ok(ctx, hipMemset(out, 0, count * 64), "hipMemset") ||
                !hip_ok(ctx, hipMemcpy2D(out, 64, packed, 48, 48, count, hipMemcpyDeviceToDevice), "hipMemcpy2D")) {
                (void)hipFree(out);
                return nullptr;
            }
            (void)hipFree(const_cast<void*>(packed));
            std::replace(s->allocs.begin(), s->allocs.end(), const_cast<void*>(packed), out);
            return out;
        };
        if (lines) {
            void* n4 = spread(s->wide.nodes, (size_t)std::max(s->n_wide_records, 1));
            void* t4 = n4 ? spread(s->wide.tris, (size_t)n_prims) : nullptr;
            if (!n4 || !t4) {
                ok = false;
            } else {
                s->wide.nodes = (const uint4*)n4;
                s->wide.tris = (const float4*)t4;
                s->wide.vec_stride = 4;
            }
        }
    }
    // spill slab for the deepest stack entries of every resident lane of the traversal grid
    s->spill_lanes = ctx->n_cus * 2048;
    {
        void* p = nullptr;
        if (!hip_ok(ctx, hipMalloc(&p, (size_t)s->spill_lanes * spill_entries * sizeof(uint2)), "hipMalloc spill")) ok = false;
        else s->allocs.push_back(p);
        d.bvh.spill = (uint2*)p;
        d.bvh.spill_stride = s->spill_lanes;
        s->wide.spill = (uint2*)p;
        s->wide.spill_stride = s->spill_lanes;
    }
    if (!dt && !instanced) d.slot_prim = dev_upload(s, prim_order, n_prims, &ok);
    if (instanced) {  // hit records name a triangle by its index inside its own object (world triangles: their own list)
        std::vector<int32_t> local(n_prims);
        int32_t k = 0;
        for (int32_t slot = 0; slot < n_prims; ++slot) {
            while (k < ia.n_objects && slot >= ia.obj_tri_offset[k + 1]) ++k;
            local[slot] = prim_order[slot] - (k < ia.n_objects ? ia.obj_tri_offset[k] : ia.world_tri_offset);
        }
        d.slot_prim = dev_upload(s, local.data(), local.size(), &ok);
    }
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

extern "C" int pbrt_hip_scene_wide_records(const PbrtHipScene* s, int32_t* n_records, const char** reason) try {
    if (!s) return PBRT_HIP_ERR_INVALID;
    if (n_records) *n_records = s->has_wide ? s->n_wide_records : -1;
    if (reason) *reason = s->wide_reason.c_str();
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

// Copies a single-level scene's wide records back: n_records x 12 dwords, n_slots x 12 floats (wide-order triangles),
// n_slots x 8 floats (leaf boxes). A diagnostic entry point (declared as such in include/pbrt_hip.h): the test that the device
// builder (wide_gpu.hip) and the host builder (host_wide.cpp) produce the same bytes reads the arrays through it.
extern "C" int pbrt_hip_debug_wide_export(PbrtHipScene* s, uint32_t* nodes, float* tris, float* boxes, int32_t n_slots) try {
    if (!s || !s->has_wide || s->d.bvh.instanced || n_slots != s->d.bvh.n_slots) return PBRT_HIP_ERR_INVALID;
    PbrtHipContext* ctx = s->ctx;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // the packed form, whatever the stride on the device (WideTrees::vec_stride)
    if (nodes && s->n_wide_records > 0)
        HIP_TRY(ctx, hipMemcpy2D(nodes, 48, s->wide.nodes, (size_t)s->wide.vec_stride * 16, 48, (size_t)s->n_wide_records, hipMemcpyDeviceToHost));
    if (tris) HIP_TRY(ctx, hipMemcpy2D(tris, 48, s->wide.tris, (size_t)s->wide.vec_stride * 16, 48, (size_t)n_slots, hipMemcpyDeviceToHost));
    if (boxes) HIP_TRY(ctx, hipMemcpy(boxes, s->wide.leaf_boxes, (size_t)n_slots * 32, hipMemcpyDeviceToHost));
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" void pbrt_hip_scene_destroy(PbrtHipScene* s) {
    if (!s) return;
    PB_LOCK(s->ctx);
    (void)hipSetDevice(s->ctx->device);
    if (!s->ctx->lost) {  // a lost context: an abandoned kernel may still read the scene, and waiting for it may never end
        (void)hipStreamSynchronize(s->ctx->stream);
        for (void* p : s->allocs) (void)hipFree(p);
    }
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
    PB_DEV uint32_t token(uint32_t i) const { return i; }  // a batch ray's token is its position
    PB_DEV bool strict(uint32_t) const { return false; }   // pbrt_hip_intersect_p is Primitive::intersect_p
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
template <bool ANY, bool COUNT, int INST, bool SPH = false>
__global__ void __launch_bounds__(kTraceBlock, (COUNT || SPH) ? 4 : (INST ? PB_INST_WAVES : PB_TRACE_WAVES))
    k_intersect_batch(DevBVH bvh, BatchRayIO<ANY> io, unsigned int* work_counter, unsigned long long* counters) {
    __shared__ uint2 lds_stack[kStackLds * kTraceBlock];
    trace_persistent<BatchRayIO<ANY>, COUNT, INST, SPH>(bvh, io, work_counter, lds_stack + threadIdx.x,
                                                        blockIdx.x * kTraceBlock + threadIdx.x, counters);
}

// the binary records without a stack (trace_stackless.h)
template <bool ANY>
__global__ void __launch_bounds__(kTraceBlock, PB_STACKLESS_WAVES)
    k_intersect_batch_stackless(DevBVH bvh, BatchRayIO<ANY> io, unsigned int* work_counter) {
    trace_stackless<BatchRayIO<ANY>>(bvh, io, work_counter);
}

// the same batch over the 4-wide records (trace_wide.h), and the follow-up over the rays that kernel left out
template <bool ANY, bool COUNT, int INST = 0>
__global__ void __launch_bounds__(kTraceBlock, (COUNT || INST) ? PB_WIDE_INST_WAVES : PB_WIDE_WAVES)
    k_intersect_batch_wide(WideTrees wt, BatchRayIO<ANY> io, unsigned int* work_counter, unsigned long long* counters) {
    __shared__ uint2 lds_stack[wide_stack_lds(INST) * kTraceBlock];
    __shared__ float lds_world[INST ? kWideWorldFloats * kTraceBlock : 1];
    trace_wide<BatchRayIO<ANY>, COUNT, INST>(wt, io, work_counter, lds_stack + threadIdx.x, blockIdx.x * kTraceBlock + threadIdx.x,
                                             counters, lds_world + (INST ? threadIdx.x : 0));
}
template <bool ANY, int INST = 0>
__global__ void __launch_bounds__(kTraceBlock, INST ? PB_INST_WAVES : PB_TRACE_WAVES)
    k_intersect_batch_special(DevBVH bvh, SpecialListIO<BatchRayIO<ANY>> io, unsigned int* work_counter) {
    if ((unsigned long long)blockIdx.x * (kTraceBlock / 64) * kChunk >= (unsigned long long)io.n()) return;  // see k_trace_special
    __shared__ uint2 lds_stack[kStackLds * kTraceBlock];
    trace_persistent<SpecialListIO<BatchRayIO<ANY>>, false, INST, false>(bvh, io, work_counter, lds_stack + threadIdx.x,
                                                                         blockIdx.x * kTraceBlock + threadIdx.x, nullptr);
}

template <bool ANY>
static int launch_batch(PbrtHipScene* s, const PbrtRay* d_rays, int64_t n, PbrtHit* d_hits, uint8_t* d_flags) {
    PbrtHipContext* ctx = s->ctx;
    if (n == 0) return PBRT_HIP_OK;
    if (n >= (1ll << 32) - kChunk * 8192ll) {
        ctx->last_error = "batch too large for one launch (split it below 2^32 rays)";
        return PBRT_HIP_ERR_INVALID;
    }
    const bool wide = s->has_wide && ctx->count_traversal != 1 && ctx->traversal == PBRT_TRAVERSAL_AUTO;
    const bool stackless = ctx->traversal == PBRT_TRAVERSAL_STACKLESS;
    if (stackless && !stackless_applies(s)) return PBRT_HIP_ERR_INVALID;
    if (wide && ctx->special_capacity < (size_t)n) {  // room for the queue positions of the rays the wide kernel leaves out
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_special_list) (void)hipFree(ctx->d_special_list);
        ctx->d_special_list = nullptr;
        ctx->special_capacity = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_special_list, (size_t)n * sizeof(uint32_t)));
        ctx->special_capacity = (size_t)n;
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_work_counter, 0, kWorkCounters * sizeof(unsigned int), ctx->stream));
    if (ctx->time_trace) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    {
        BatchRayIO<ANY> io{s->d.slot_prim, s->d.slot_instance, d_rays, (uint32_t)n, d_hits, d_flags};
        dim3 grid(persistent_grid(s)), block(kTraceBlock);
        const int inst = s->d.bvh.instanced ? (s->d.bvh.general_top ? 2 : 1) : 0;  // trace_persistent.h: INST
        const bool count_ref = ctx->count_traversal == 1, count_wide = ctx->count_traversal == 2;
#define PB_LAUNCH_BINARY(COUNT, INST, SPH) \
    hipLaunchKernelGGL((k_intersect_batch<ANY, COUNT, INST, SPH>), grid, block, 0, ctx->stream, s->d.bvh, io, ctx->d_work_counter, ctx->d_counters)
#define PB_LAUNCH_WIDE(COUNT, INST)                                                                                                  \
    hipLaunchKernelGGL((k_intersect_batch_wide<ANY, COUNT, INST>),                                                                   \
                       dim3(persistent_grid(s, (COUNT || INST) ? PB_WIDE_INST_WAVES : PB_WIDE_WAVES, wide_stack_lds(INST), wide_world_lds_bytes(INST))), block, 0, ctx->stream, \
                       wt, io, ctx->d_work_counter, ctx->d_counters)
#define PB_LAUNCH_SPECIAL(INST) \
    hipLaunchKernelGGL((k_intersect_batch_special<ANY, INST>), grid, block, 0, ctx->stream, s->d.bvh, sio, ctx->d_work_counter + kFollowUpCounter)
        if (stackless) {
            hipLaunchKernelGGL((k_intersect_batch_stackless<ANY>), dim3(stackless_grid(s)), block, 0, ctx->stream, s->d.bvh, io, ctx->d_work_counter);
        } else if (wide) {
            WideTrees wt = s->wide;
            wt.special_list = ctx->d_special_list;
            wt.special_count = ctx->d_work_counter + kSpecialCount;
            SpecialListIO<BatchRayIO<ANY>> sio{io, ctx->d_special_list, ctx->d_work_counter + kSpecialCount};
            if (inst == 2) {
                if (count_wide) PB_LAUNCH_WIDE(true, 2); else PB_LAUNCH_WIDE(false, 2);
                PB_LAUNCH_SPECIAL(2);
            } else if (inst == 1) {
                if (count_wide) PB_LAUNCH_WIDE(true, 1); else PB_LAUNCH_WIDE(false, 1);
                PB_LAUNCH_SPECIAL(1);
            } else {
                if (count_wide) PB_LAUNCH_WIDE(true, 0); else PB_LAUNCH_WIDE(false, 0);
                PB_LAUNCH_SPECIAL(0);
            }
        } else if (s->d.bvh.has_spheres) {  // single-level scenes only (checked at creation)
            if (count_ref) PB_LAUNCH_BINARY(true, 0, true); else PB_LAUNCH_BINARY(false, 0, true);
        } else if (inst == 2) {
            if (count_ref) PB_LAUNCH_BINARY(true, 2, false); else PB_LAUNCH_BINARY(false, 2, false);
        } else if (inst == 1) {
            if (count_ref) PB_LAUNCH_BINARY(true, 1, false); else PB_LAUNCH_BINARY(false, 1, false);
        } else {
            if (count_ref) PB_LAUNCH_BINARY(true, 0, false); else PB_LAUNCH_BINARY(false, 0, false);
        }
#undef PB_LAUNCH_BINARY
#undef PB_LAUNCH_WIDE
#undef PB_LAUNCH_SPECIAL
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

extern "C" int pbrt_hip_intersect_device(PbrtHipScene* s, const PbrtRay* d_rays, int64_t n, PbrtHit* d_out) try {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_out))) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(s->ctx);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    return launch_batch<false>(s, d_rays, n, d_out, nullptr);
}
PB_ABI_CATCH
extern "C" int pbrt_hip_intersect_p_device(PbrtHipScene* s, const PbrtRay* d_rays, int64_t n, uint8_t* d_out) try {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_out))) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(s->ctx);
    HIP_TRY(s->ctx, hipSetDevice(s->ctx->device));
    return launch_batch<true>(s, d_rays, n, nullptr, d_out);
}
PB_ABI_CATCH

template <bool ANY>
static int intersect_host(PbrtHipScene* s, const PbrtRay* rays, int64_t n, void* out) {
    if (!s || n < 0 || (n > 0 && (!rays || !out))) return PBRT_HIP_ERR_INVALID;
    if (n == 0) return PBRT_HIP_OK;
    PbrtHipContext* ctx = s->ctx;
    PB_ENTER(ctx);
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
extern "C" int pbrt_hip_intersect(PbrtHipScene* s, const PbrtRay* rays, int64_t n, PbrtHit* out) try {
    return intersect_host<false>(s, rays, n, out);
}
PB_ABI_CATCH
extern "C" int pbrt_hip_intersect_p(PbrtHipScene* s, const PbrtRay* rays, int64_t n, uint8_t* out) try {
    return intersect_host<true>(s, rays, n, out);
}
PB_ABI_CATCH

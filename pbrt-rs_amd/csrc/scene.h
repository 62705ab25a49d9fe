// scene.h — context / scene objects behind the C ABI and the host->device re-layout.
//
// Scene::new (src/core/scene.rs:18-34) + GeometricPrimitive (src/core/primitive.rs:33-55) +
// BVHAccel's flat node array (src/accelerators/bvh.rs:129-135, 774-811), re-laid out for the
// gfx950 traversal kernel (layout described in trace.h).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pbrt_hip.h"
#include "trace.h"
#include "wide_bvh.h"

// Calls that reach the device through one context are serialised: the reference's Primitive is Sync + Send and li() is
// re-entrant (src/core/primitive.rs:179, integrator.rs:412-452), so a drop-in may be entered from several host threads.
#define PB_LOCK(ctx) std::lock_guard<std::recursive_mutex> pb_lock_((ctx)->mu)
// ... and refused once the context is lost (a wavefront ran into the deadline: a kernel of that call may still be running
// on the stream, over buffers the call has abandoned; nothing else may be queued behind it or handed its memory)
#define PB_ENTER(ctx)                                                                                              \
    PB_LOCK(ctx);                                                                                                  \
    if ((ctx)->lost) {                                                                                             \
        (ctx)->last_error = "context lost: an earlier call ran into the wavefront deadline; destroy the context";  \
        return PBRT_HIP_ERR_DEVICE;                                                                                \
    }

struct PbrtHipContext {
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cus = 0;
    std::string last_error;
    // sticky: set when a call gave up on a kernel that did not finish (render.hip); every entry point then fails
    // (atomic: pbrt_hip_context_is_lost polls it from other threads without the lock)
    std::atomic<bool> lost{false};
    bool trace_log = false;  // PBRT_HIP_TRACE_LOG was set when the context was created: one stderr line per traversal launch
    double wavefront_deadline_s = 120.0;  // no wavefront of a render takes this long (pbrt_hip_context_set_deadline)
    // traversal-kernel timing (HIP events on `stream`)
    double trace_ms = 0.0;
    uint64_t trace_launches = 0;
    bool time_trace = true;
    // instrumented traversal (roofline accounting): device counters {node_tests, prim_tests}
    // 0 = off; 1 = the binary kernels count the reference's box / triangle tests; 2 = the wide kernels count their own
    // record / leaf / triangle fetches (pbrt_hip_set_counting)
    int count_traversal = 0;
    int traversal = 0;  // PBRT_TRAVERSAL_*: pbrt_hip_context_set_traversal
    int wide_build = 0;  // PBRT_WIDE_BUILD_*: pbrt_hip_context_set_wide_build
    int wide_layout = 0;  // PBRT_WIDE_LAYOUT_*: pbrt_hip_context_set_wide_layout
    unsigned long long* d_counters = nullptr;  // [0..3] mode 1: node, prim, rays, instance tests; [4..7] mode 2
    uint64_t counted_rays = 0;
    // ray-queue heads of the persistent traversal kernels: [0, kQueueSegments) the launch's segments,
    // [kFollowUpCounter] the follow-up launch over the rays the wide kernel left out, [kSpecialCount] their number
    unsigned int* d_work_counter = nullptr;
    uint32_t* d_special_list = nullptr;  // batch calls: tokens (IO::token) of those rays (grown on demand)
    size_t special_capacity = 0;
    // the render loop reads the queue lengths back once per wavefront: pinned landing buffer + an event the host
    // spins on (a blocking stream sync costs a scheduler wake-up per wavefront, milliseconds on a busy host)
    unsigned long long* h_counts = nullptr;
    hipEvent_t ev_sync = nullptr;
    // Device blocks of finished renders, kept for the next one: a 64-spp 1080p frame holds 50 GB of path state and
    // hipMalloc / hipFree of that much costs seconds. Freed with the context (or when an allocation fails).
    struct CachedBlock {
        void* ptr;
        size_t bytes;
        bool in_use;
    };
    std::vector<CachedBlock> block_cache;
    // HaltonSampler tables, uploaded on first use: primes + prime sums, radical-inverse digit permutations
    uint32_t* d_halton_primes = nullptr;
    uint16_t* d_halton_perms = nullptr;
};

namespace pb {

constexpr int kWorkCounters = 16, kFollowUpCounter = 8, kSpecialCount = 12;

// device-side light / material tables
struct DevLight {
    int type;       // PbrtLightType
    float L[3];
    int slot;       // area light: leaf slot of its triangle
    int two_sided;
    float area;     // Triangle::area (triangle.rs:323-328)
    float power_y;  // Light::power().y_value() for the power distribution
    // delta lights (lights/point.rs, spot.rs, distant.rs): p_light or w_light, spot cone, world_to_light 3x3
    float pos[3];
    float cos_total_width, cos_falloff_start;
    float w2l[9];
    int delta;      // is_delta_light (light.rs:28-31): no BSDF-sampling half, no MIS weight
};
struct DevMaterial {
    int type;
    float kd[3], kt[3];
    float eta;
};

// Distribution1D (src/core/sampling.rs:62-154) flattened: func[n], cdf[n+1]
struct DevDistribution1D {
    const float* func;
    const float* cdf;
    float func_int;
    int n;
};

struct DevSceneData {
    DevBVH bvh;
    const int* slot_prim;       // leaf slot -> caller's triangle index
    const int* slot_instance;   // top-level leaf slot -> caller's instance index (instanced scenes)
    const DevMaterial* materials;
    const DevLight* lights;
    int n_lights;
    int n_materials;
    int n_infinite;             // number of lights with the INFINITE flag (scene.rs:29-31, D29 intended)
    const int* infinite_ids;
    // InfiniteAreaLight's 2x2 sin-weighted Distribution2D (lights/infinite.rs:59-73): conditional rows + marginal
    float env_cond_func[2][2], env_cond_cdf[2][3], env_cond_int[2];
    float env_marg_func[2], env_marg_cdf[3], env_marg_int;
    float world_center[3], world_radius;  // lights/infinite.rs:135-139
    DevDistribution1D light_distrib_uniform, light_distrib_power;
};

}  // namespace pb

struct PbrtHipScene {
    PbrtHipContext* ctx = nullptr;
    pb::DevSceneData d;          // device pointers inside
    std::vector<void*> allocs;   // everything to hipFree
    int n_tris = 0, n_nodes = 0, n_interior = 0, n_instances = 0;
    int spill_lanes = 0;
    std::vector<pb::DevLight> h_lights;
    std::vector<int> light_samples;  // max(1, n_samples) per light (light.rs:76)
    // SpatialLightDistribution tables (lightdistrib.rs:76-220), built on first use by a render with that strategy
    float* d_spatial = nullptr;
    int spatial_voxels[3] = {0, 0, 0};
    // 4-wide quantised records over the same tree (wide_bvh.h): the traversal kernels' fast path when present
    bool has_wide = false;
    pb::WideTrees wide{};
    int n_wide_records = 0;
    std::string wide_reason;  // why the scene has none
};

namespace pb {

inline bool hip_ok(PbrtHipContext* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    if (ctx) ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

template <class T>
inline T* dev_upload(PbrtHipScene* s, const T* src, size_t n, bool* ok) {
    void* p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    if (!hip_ok(s->ctx, hipMalloc(&p, bytes), "hipMalloc")) {
        *ok = false;
        return nullptr;
    }
    s->allocs.push_back(p);
    if (n && !hip_ok(s->ctx, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D")) *ok = false;
    return (T*)p;
}

// A single-level scene tree built and re-laid out on the device (hlbvh_gpu.hip): the scene takes
// ownership of the three device arrays.
// device arrays of the 4-wide records of one tree, built on the device (wide_gpu.hip); the caller owns them
struct WideDeviceTree {
    uint32_t* nodes = nullptr;    // 12 dwords per record
    float* tris = nullptr;        // 12 floats per wide-order triangle
    float* leaf_boxes = nullptr;  // 8 floats per wide-order triangle position
    int32_t root_ref = 0;
    int n_records = -1;  // -1: not built
    int stack_need = 0;
};
bool build_wide_tree_device(PbrtHipContext* ctx, const PbrtLinearBVHNode* d_nodes, int32_t n_nodes, const float* d_tris, int32_t n_slots,
                            const PbrtLinearBVHNode& root, WideDeviceTree* out, const char** reason);

struct DeviceTree {
    float4* inodes = nullptr;   // 64-B child-pair records
    float4* tris = nullptr;     // 48-B triangles in leaf order
    int* slot_prim = nullptr;   // leaf slot -> caller's triangle index
    float root_min[3], root_max[3];
    int root_ref = 0, count_bits = 0, n_interior = 0, n_nodes = 0;
    std::vector<int32_t> light_slot;  // per light: leaf slot of an area light's triangle, -1 otherwise
    // the 4-wide records (wide_bvh.h) laid over the device-built tree, on the device as well (wide_gpu.hip);
    // wide.n_records < 0: not built (wide_reason says why: PBRT_WIDE_BUILD_NONE, or the tree does not qualify).
    // PBRT_WIDE_BUILD_HOST: the host builder instead, on copies of the tree (the test of host == device).
    WideDeviceTree wide;
    const char* wide_reason = nullptr;
    std::vector<PbrtLinearBVHNode> h_nodes;
    std::vector<float> h_tris;
    double build_ms = 0.0;            // tree build; convert_ms: re-layout into the traversal format
    double convert_ms = 0.0;
};

// Evaluation of "Triangle::intersect returns false" (triangle.rs:197-216) with the default
// uv set (triangle.rs:66-70): degenerate uv frame or dpdu x dpdv == 0, and a zero geometric
// normal. Ray independent, so it is a per-triangle flag. Same float operation order as the kernel
// side (this file is compiled with -ffp-contract=off).
__host__ __device__ inline bool triangle_rejected_by_intersect(const float* a, const float* b, const float* c,
                                                              const float* uv6 = nullptr) {
    float uv[3][2] = {{0.0f, 0.0f}, {1.0f, 0.0f}, {1.0f, 1.0f}};
    if (uv6)  // per-vertex uvs of the mesh (triangle.rs:60-65)
        for (int k = 0; k < 6; ++k) uv[k / 2][k % 2] = uv6[k];
    float duv02[2] = {uv[0][0] - uv[2][0], uv[0][1] - uv[2][1]};
    float duv12[2] = {uv[1][0] - uv[2][0], uv[1][1] - uv[2][1]};
    float dp02[3], dp12[3];
    for (int k = 0; k < 3; ++k) {
        dp02[k] = a[k] - c[k];
        dp12[k] = b[k] - c[k];
    }
    float determinant = duv02[0] * duv12[1] - duv02[1] * duv12[0];
    bool degenerate_uv = __builtin_fabsf(determinant) < 1e-8f;
    float dpdu[3] = {0, 0, 0}, dpdv[3] = {0, 0, 0};
    if (!degenerate_uv) {
        float inv_det = 1.0f / determinant;
        for (int k = 0; k < 3; ++k) {
            dpdu[k] = (dp02[k] * duv12[1] - dp12[k] * duv02[1]) * inv_det;
            dpdv[k] = (dp02[k] * -duv12[0] + dp12[k] * duv02[0]) * inv_det;
        }
    }
    float cx = dpdu[1] * dpdv[2] - dpdu[2] * dpdv[1];
    float cy = dpdu[2] * dpdv[0] - dpdu[0] * dpdv[2];
    float cz = dpdu[0] * dpdv[1] - dpdu[1] * dpdv[0];
    if (degenerate_uv || (cx * cx + cy * cy + cz * cz) == 0.0f) {
        float e1[3], e2[3];
        for (int k = 0; k < 3; ++k) {
            e1[k] = c[k] - a[k];
            e2[k] = b[k] - a[k];
        }
        float nx = e1[1] * e2[2] - e1[2] * e2[1];
        float ny = e1[2] * e2[0] - e1[0] * e2[2];
        float nz = e1[0] * e2[1] - e1[1] * e2[0];
        if ((nx * nx + ny * ny + nz * nz) == 0.0f) return true;
    }
    return false;
}

// persistent kernels: enough blocks to fill every CU (LDS: 160 KiB / (kStackLds * 2 KiB) blocks of 256, at most 8)
inline int persistent_grid(const PbrtHipScene* s, int waves = PB_TRACE_WAVES, int stack_lds = kStackLds, int other_lds_bytes = 0) {
    int per_cu = std::min(waves, (160 * 1024) / (stack_lds * kTraceBlock * (int)sizeof(uint2) + other_lds_bytes));
    return std::min(s->ctx->n_cus * per_cu, s->spill_lanes / kTraceBlock);
}

// the stackless walk (trace_stackless.h) keeps no LDS: its grid is set by the registers alone
#ifndef PB_STACKLESS_WAVES
#define PB_STACKLESS_WAVES 5  // 80 registers: 6 waves spill 11 of them (config 3 trace-only: 5 -> 1137, 6 -> 1110, 8 -> 963 Mrays/s)
#endif
inline int stackless_grid(const PbrtHipScene* s) { return s->ctx->n_cus * PB_STACKLESS_WAVES; }
// PBRT_TRAVERSAL_STACKLESS covers single-level triangle scenes; sets last_error otherwise
inline bool stackless_applies(const PbrtHipScene* s) {
    if (!s->d.bvh.instanced && !s->d.bvh.has_spheres && s->ctx->count_traversal == 0) return true;
    s->ctx->last_error = s->ctx->count_traversal ? "PBRT_TRAVERSAL_STACKLESS has no counting variant (pbrt_hip_set_counting)"
                                                 : "PBRT_TRAVERSAL_STACKLESS: single-level triangle scenes only";
    return false;
}

}  // namespace pb

#define HIP_TRY(ctx, call)                                                 \
    do {                                                                   \
        if (!pb::hip_ok((ctx), (call), #call)) return PBRT_HIP_ERR_DEVICE; \
    } while (0)

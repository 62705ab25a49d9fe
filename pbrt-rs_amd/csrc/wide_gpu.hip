// wide_gpu.hip — the 4-wide quantised records (wide_bvh.h) laid over a reference-order LinearBVHNode array that is
// already on the device (the GPU HLBVH build, hlbvh_gpu.hip): the tree never visits the host.
//
// Same result as host_wide.cpp's breadth-first build, byte for byte (tests/test_gpu_wide.py): the arithmetic is the
// same source (wide_build.h) and the order is the same — records level by level, inside a level in the order of their
// parents, a record's interior children consecutive, its leaf children's triangles consecutive. Per level of RECORDS (two
// levels of the binary tree):
//   k_wide_count   per record of the level: (#triangles of its leaf children) << 32 | #interior children
//   rocPRIM        exclusive scan of that 64-bit pair -> where each record's children and triangles start
//   k_wide_emit    the record (wide_make_record), its triangles and leaf boxes (wide_emit_leaf), its interior children
//                  appended to the next level's list at the scanned position
// The host reads one 16-byte total per level (a few tens of levels). Afterwards one kernel per level, bottom-up, gives
// the exact worst-case stack depth (the spill slab is sized from it).
#include <hip/hip_runtime.h>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include "scene.h"
#include "wide_build.h"

namespace pb {

namespace {

constexpr int kWB = 256;
inline int wb_blocks(int64_t n) { return (int)((n + kWB - 1) / kWB); }

__global__ void __launch_bounds__(kWB) k_wide_check(const PbrtLinearBVHNode* __restrict__ nodes, int n_nodes,
                                                    const float* __restrict__ tris, int* __restrict__ error) {
    int i = blockIdx.x * kWB + threadIdx.x;
    if (i >= n_nodes) return;
    int code = wide_check_node(nodes, i, tris);
    if (code) atomicMax(error, code);
}

__global__ void __launch_bounds__(kWB) k_wide_count(const PbrtLinearBVHNode* __restrict__ nodes, const int* __restrict__ level, int n,
                                                    unsigned long long* __restrict__ counts) {
    int j = blockIdx.x * kWB + threadIdx.x;
    if (j >= n) return;
    int32_t slot_node[4];
    int axis_c[2];
    wide_slots_of(nodes, level[j], slot_node, axis_c);
    unsigned long long c = 0;
    for (int s = 0; s < 4; ++s) {
        if (slot_node[s] < 0) continue;
        const int np = nodes[slot_node[s]].n_primitives;
        c += np > 0 ? ((unsigned long long)np << 32) : 1ull;
    }
    counts[j] = c;
}

__global__ void __launch_bounds__(kWB) k_wide_emit(const PbrtLinearBVHNode* __restrict__ nodes, const float* __restrict__ tris,
                                                   const int* __restrict__ level, int n, const unsigned long long* __restrict__ offsets,
                                                   uint32_t level_base, uint32_t next_base, uint32_t tri_base, int* __restrict__ next_level,
                                                   uint32_t* __restrict__ out_nodes, float* __restrict__ out_tris,
                                                   float* __restrict__ out_boxes, int* __restrict__ error) {
    int j = blockIdx.x * kWB + threadIdx.x;
    if (j >= n) return;
    const int32_t i = level[j];
    int32_t slot_node[4];
    int axis_c[2], tri_off[4];
    wide_slots_of(nodes, i, slot_node, axis_c);
    const unsigned long long off = offsets[j];
    const uint32_t child_off = (uint32_t)off, first_tri = tri_base + (uint32_t)(off >> 32);
    uint32_t rec[kWideNodeDwords];
    for (int k = 0; k < kWideNodeDwords; ++k) rec[k] = 0u;
    int coarse = 0;
    int code = wide_make_record(nodes, i, slot_node, axis_c, next_base + child_off, first_tri, rec, tri_off, &coarse);
    if (code) atomicMax(error, code);
    if (coarse) atomicAdd(error + 1, 1);  // records whose planes do not filter (wide_build.h): counted, judged by the host
    uint32_t* dst = out_nodes + (size_t)(level_base + (uint32_t)j) * kWideNodeDwords;
    for (int k = 0; k < kWideNodeDwords; ++k) dst[k] = rec[k];
    int n_interior = 0;
    for (int s = 0; s < 4; ++s) {
        if (slot_node[s] < 0) continue;
        const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
        if (ch.n_primitives > 0)
            wide_emit_leaf(ch, tris, 0, (size_t)first_tri + (size_t)tri_off[s], out_tris, out_boxes);
        else
            next_level[child_off + (uint32_t)(n_interior++)] = slot_node[s];
    }
}

// a record with k children leaves at most k - 1 of them on the stack while the first is being walked
__global__ void __launch_bounds__(kWB) k_wide_stack_need(const uint32_t* __restrict__ out_nodes, uint32_t level_base, int n,
                                                         int* __restrict__ need) {
    int j = blockIdx.x * kWB + threadIdx.x;
    if (j >= n) return;
    const uint32_t w = level_base + (uint32_t)j;
    const uint32_t* rec = out_nodes + (size_t)w * kWideNodeDwords;
    const uint32_t m[4] = {rec[0] & 0xffu, rec[1] & 0xffu, rec[2] & 0xffu, rec[3] >> 24};
    int k = 0, deepest = 0;
    for (int s = 0; s < 4; ++s) {
        if (m[s] == 0xffu) continue;
        ++k;
        if (m[s] & 0x80u) deepest = max(deepest, need[rec[10] + (m[s] & 3u)]);
    }
    need[w] = k - 1 + deepest;
}

}  // namespace

// d_nodes: n_nodes LinearBVHNodes that have ALREADY passed convert_tree's validation of offsets and counts (k_wide_count /
// k_wide_emit follow child offsets before k_wide_check's verdict is read back), d_tris: n_slots 48-B leaf-order triangle
// records, both on the device. On success the
// three output arrays are hipMalloc'ed (the caller owns them) and *reason is nullptr; when the tree does not qualify
// (*reason says why) or a HIP call fails (returns false, ctx->last_error set) nothing is left allocated.
bool build_wide_tree_device(PbrtHipContext* ctx, const PbrtLinearBVHNode* d_nodes, int32_t n_nodes, const float* d_tris,
                            int32_t n_slots, const PbrtLinearBVHNode& root, WideDeviceTree* out, const char** reason) {
    *reason = nullptr;
    *out = WideDeviceTree();
    hipStream_t st = ctx->stream;
    if (n_nodes <= 0 || n_slots <= 0) {
        *reason = "empty tree";
        return true;
    }
    if ((int64_t)n_slots >= (1ll << 29) - 4) {  // the five lowest integers are lane codes of the traversal kernel (trace_wide.h)
        *reason = "too many triangles for 30-bit wide references";
        return true;
    }
    const int max_records = std::max(1, n_nodes / 2);  // records stand for interior nodes, at most every second node
    std::vector<void*> tmp;
    void *p_nodes = nullptr, *p_tris = nullptr, *p_boxes = nullptr;
    auto release = [&](bool all) {
        for (void* p : tmp) (void)hipFree(p);
        tmp.clear();
        if (all) {
            if (p_nodes) (void)hipFree(p_nodes);
            if (p_tris) (void)hipFree(p_tris);
            if (p_boxes) (void)hipFree(p_boxes);
            p_nodes = p_tris = p_boxes = nullptr;
        }
    };
    auto fail = [&](hipError_t e, const char* what) {
        release(true);
        pb::hip_ok(ctx, e, what);
        return false;
    };
    auto alloc = [&](void** p, size_t bytes, bool temporary) -> hipError_t {
        hipError_t e = hipMalloc(p, std::max<size_t>(bytes, 16));
        if (e == hipSuccess && temporary) tmp.push_back(*p);
        return e;
    };
    // Out of device memory for the (optional) records is not a failure of the scene: it keeps the binary records, and
    // pbrt_hip_scene_wide_records says why. Anything else that goes wrong on the device is.
#define WB_TRY(call, what)                                         \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ == hipErrorOutOfMemory) {                           \
            (void)hipGetLastError();                               \
            (void)hipStreamSynchronize(st);                        \
            release(true);                                         \
            *reason = "out of device memory for the wide records"; \
            return true;                                           \
        }                                                          \
        if (e_ != hipSuccess) return fail(e_, what);               \
    } while (0)
    int* d_error = nullptr;
    WB_TRY(alloc((void**)&d_error, 2 * sizeof(int), true), "wide build: alloc");
    WB_TRY(hipMemsetAsync(d_error, 0, 2 * sizeof(int), st), "wide build: memset");
    hipLaunchKernelGGL(k_wide_check, dim3(wb_blocks(n_nodes)), dim3(kWB), 0, st, d_nodes, n_nodes, d_tris, d_error);
    WB_TRY(alloc(&p_tris, (size_t)n_slots * 48, false), "wide build: alloc triangles");
    WB_TRY(alloc(&p_boxes, (size_t)n_slots * 32, false), "wide build: alloc leaf boxes");
    WB_TRY(hipMemsetAsync(p_boxes, 0, (size_t)n_slots * 32, st), "wide build: memset");
    if (root.n_primitives > 0) {  // the whole tree is one leaf: one thread lays it out
        int h_err = 0;
        WB_TRY(hipMemcpyAsync(&h_err, d_error, sizeof(int), hipMemcpyDeviceToHost, st), "wide build: check");
        WB_TRY(hipStreamSynchronize(st), "wide build: check");
        if (h_err) {
            release(true);
            *reason = wide_error_text(h_err);
            return true;
        }
        std::vector<float> t((size_t)root.n_primitives * 12), wt((size_t)n_slots * 12, 0.0f), wb((size_t)n_slots * 8, 0.0f);
        WB_TRY(hipMemcpy(t.data(), d_tris + 12 * (size_t)root.offset, t.size() * 4, hipMemcpyDeviceToHost), "wide build: leaf");
        PbrtLinearBVHNode lf = root;
        lf.offset = 0;
        wide_emit_leaf(lf, t.data(), root.offset, 0, wt.data(), wb.data());
        WB_TRY(hipMemcpy(p_tris, wt.data(), wt.size() * 4, hipMemcpyHostToDevice), "wide build: leaf");
        WB_TRY(hipMemcpy(p_boxes, wb.data(), wb.size() * 4, hipMemcpyHostToDevice), "wide build: leaf");
        WB_TRY(alloc(&p_nodes, kWideNodeDwords * 4, false), "wide build: alloc records");
        WB_TRY(hipMemset(p_nodes, 0, kWideNodeDwords * 4), "wide build: memset");
        release(false);
        out->nodes = (uint32_t*)p_nodes;
        out->tris = (float*)p_tris;
        out->leaf_boxes = (float*)p_boxes;
        out->root_ref = ~(int32_t)((0u << 2) | (uint32_t)(root.n_primitives - 1));
        out->n_records = 0;
        out->stack_need = 0;
        return true;
    }
    int *d_level[2] = {nullptr, nullptr}, *d_need = nullptr;
    unsigned long long *d_counts = nullptr, *d_offsets = nullptr;
    void* d_scan_tmp = nullptr;
    size_t scan_bytes = 0;
    WB_TRY(alloc(&p_nodes, (size_t)max_records * kWideNodeDwords * 4, false), "wide build: alloc records");
    WB_TRY(alloc((void**)&d_level[0], (size_t)max_records * sizeof(int), true), "wide build: alloc");
    WB_TRY(alloc((void**)&d_level[1], (size_t)max_records * sizeof(int), true), "wide build: alloc");
    WB_TRY(alloc((void**)&d_counts, ((size_t)max_records + 1) * 8, true), "wide build: alloc");
    WB_TRY(alloc((void**)&d_offsets, ((size_t)max_records + 1) * 8, true), "wide build: alloc");
    WB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_counts, d_offsets, 0ull, (size_t)max_records + 1, rocprim::plus<unsigned long long>(), st),
           "wide build: scan size");
    WB_TRY(alloc(&d_scan_tmp, scan_bytes, true), "wide build: alloc");
    WB_TRY(hipMemsetAsync(d_level[0], 0, sizeof(int), st), "wide build: memset");  // level 0 = {node 0}
    std::vector<std::pair<uint32_t, int>> levels;  // (first record, records) per level
    uint32_t level_base = 0, tri_base = 0;
    int n_level = 1, cur = 0;
    while (n_level > 0) {
        if ((size_t)level_base + (size_t)n_level > (size_t)max_records) {
            release(true);
            ctx->last_error = "wide build: more records than interior nodes (corrupt tree)";
            return false;
        }
        levels.emplace_back(level_base, n_level);
        hipLaunchKernelGGL(k_wide_count, dim3(wb_blocks(n_level)), dim3(kWB), 0, st, d_nodes, d_level[cur], n_level, d_counts);
        // n_level + 1 items: the scanned value at index n_level is the level's total (the extra count is never read)
        WB_TRY(hipMemsetAsync(d_counts + n_level, 0, 8, st), "wide build: memset");
        WB_TRY(rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_counts, d_offsets, 0ull, (size_t)n_level + 1,
                                       rocprim::plus<unsigned long long>(), st),
               "wide build: scan");
        unsigned long long total = 0;
        WB_TRY(hipMemcpyAsync(&total, d_offsets + n_level, 8, hipMemcpyDeviceToHost, st), "wide build: level total");
        WB_TRY(hipStreamSynchronize(st), "wide build: level total");
        const uint32_t n_children = (uint32_t)total, n_tris = (uint32_t)(total >> 32);
        const uint32_t next_base = level_base + (uint32_t)n_level;
        if ((size_t)next_base + n_children > (size_t)max_records || (size_t)tri_base + n_tris > (size_t)n_slots) {
            release(true);
            ctx->last_error = "wide build: children or triangles beyond the tree's size (corrupt tree)";
            return false;
        }
        hipLaunchKernelGGL(k_wide_emit, dim3(wb_blocks(n_level)), dim3(kWB), 0, st, d_nodes, d_tris, d_level[cur], n_level, d_offsets,
                           level_base, next_base, tri_base, d_level[cur ^ 1], (uint32_t*)p_nodes, (float*)p_tris, (float*)p_boxes, d_error);
        level_base = next_base;
        tri_base += n_tris;
        n_level = (int)n_children;
        cur ^= 1;
    }
    const int n_records = (int)level_base;
    WB_TRY(alloc((void**)&d_need, (size_t)n_records * sizeof(int), true), "wide build: alloc");
    for (size_t l = levels.size(); l-- > 0;)
        hipLaunchKernelGGL(k_wide_stack_need, dim3(wb_blocks(levels[l].second)), dim3(kWB), 0, st, (const uint32_t*)p_nodes, levels[l].first,
                           levels[l].second, d_need);
    int h_err2[2] = {0, 0}, h_need = 0;
    WB_TRY(hipGetLastError(), "wide build: launch");
    WB_TRY(hipMemcpyAsync(h_err2, d_error, 2 * sizeof(int), hipMemcpyDeviceToHost, st), "wide build: result");
    WB_TRY(hipMemcpyAsync(&h_need, d_need, sizeof(int), hipMemcpyDeviceToHost, st), "wide build: result");
    WB_TRY(hipStreamSynchronize(st), "wide build: result");
    int h_err = h_err2[0];
    if (!h_err && (long long)h_err2[1] * 8 > (long long)n_records) h_err = kWideErrCoarse;
    if (h_err || tri_base != (uint32_t)n_slots) {
        release(true);
        *reason = h_err ? wide_error_text(h_err) : "leaves do not cover the triangle list";
        return true;
    }
    if ((size_t)n_records * 4 < (size_t)max_records * 3) {  // give back what the bound over-allocated
        void* exact = nullptr;
        WB_TRY(hipMalloc(&exact, (size_t)n_records * kWideNodeDwords * 4), "wide build: alloc records");
        hipError_t e = hipMemcpy(exact, p_nodes, (size_t)n_records * kWideNodeDwords * 4, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(exact);
            return fail(e, "wide build: compact");
        }
        (void)hipFree(p_nodes);
        p_nodes = exact;
    }
    release(false);
#undef WB_TRY
    out->nodes = (uint32_t*)p_nodes;
    out->tris = (float*)p_tris;
    out->leaf_boxes = (float*)p_boxes;
    out->root_ref = 0;
    out->n_records = n_records;
    out->stack_need = h_need;
    return true;
}

}  // namespace pb

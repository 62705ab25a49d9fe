// hlbvh_gpu.hip — BVHAccel::new with SplitMethod::HLBVH on the GPU.
//
// Reference: src/accelerators/bvh.rs — primitive bounds / centroids :26-41, Morton encode :137-156
// (the loop the reference marks "TODO parallel", :489), radix sort :158-197, treelets :509-528,
// emit_lbvh :570-676, build_upper_sah :678-772, flatten :774-811. The output (flat pre-order node array
// + leaf order) is byte-identical to the host builder (host_bvh.cpp, split_method 1) and to the oracle.
//
// Device formulation (no recursion, no per-treelet serial loop):
//   1. k_prim_bounds   one thread per triangle: bounds, centroid; centroid bounds by wave/block reduction and
//                      one ordered-integer atomic per block.
//   2. k_morton        30-bit Morton code of the centroid offset (10 bits per axis).
//   3. rocprim radix_sort_pairs on bits [0, 30): any stable sort equals the reference's 5 x 6-bit LSD passes.
//   4. k_treelet_*     one treelet per distinct top-12-bit prefix, in ascending order.
//   5. k_level x 19    level-synchronous emit_lbvh: a work item is a (range, bit) pair; each thread skips the
//                      bits on which its range does not split, then makes a leaf (bit == -1 or n < max_prims)
//                      or an interior node and two items for the next level. Node ids and queue slots are
//                      allocated with one atomic per wave.
//   6. k_fit x 19      bottom-up by level: child-box unions and subtree sizes.
//   7. k_local_index   pre-order index of every node inside its treelet from the subtree sizes (walk to the root).
//   8. host            SAH over the <= 4096 treelet roots (host_bvh.cpp: hlbvh_upper_tree) -> global pre-order
//                      offset of every treelet and the upper interior nodes. ~100 KB crosses PCIe.
//   9. k_emit          LinearBVHNode records at (treelet offset + local index). The leaf order is the sorted
//                      primitive list (the reference's ordered_prims_offset advances in DFS = Morton order).
#include <hip/hip_runtime.h>
#include "abi_guard.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cfloat>
#include <cstdlib>
#include <vector>

#include "scene.h"

namespace pb {
void hlbvh_upper_tree(const float* boxes6, const int32_t* sizes, int32_t n, std::vector<int32_t>& treelet_offset,
                      std::vector<int32_t>& upper_index, std::vector<PbrtLinearBVHNode>& upper_nodes,
                      int32_t* n_nodes_total, int32_t* upper_depth);
}

namespace {

constexpr int kBlock = 256;
constexpr int kMaxTreelets = 4096;   // 12 Morton bits
constexpr int kFirstBit = 29 - 12;   // bvh.rs:521
constexpr int kLevels = kFirstBit + 2;  // bit 17 .. -1

// monotone float <-> uint map for atomicMin / atomicMax
__device__ inline uint32_t f2ord(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ inline float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ inline float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) {
        float w = __shfl_xor(v, o);
        v = w < v ? w : v;
    }
    return v;
}
__device__ inline float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) {
        float w = __shfl_xor(v, o);
        v = w > v ? w : v;
    }
    return v;
}

// Triangle::world_bound (triangle.rs:175-180), centroid = 0.5 * min + 0.5 * max (bvh.rs:38)
__global__ void __launch_bounds__(kBlock) k_prim_bounds(const float* __restrict__ pos, const int* __restrict__ idx, int n,
                                                        float* __restrict__ lo, float* __restrict__ hi,
                                                        uint32_t* __restrict__ cb_ord) {
    float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    // grid-stride: same-address atomics retire at ~100 per microsecond, so the grid is kept small
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const float* a = pos + 3 * (size_t)idx[3 * (size_t)i];
        const float* b = pos + 3 * (size_t)idx[3 * (size_t)i + 1];
        const float* c = pos + 3 * (size_t)idx[3 * (size_t)i + 2];
        for (int k = 0; k < 3; ++k) {
            float av = a[k], bv = b[k], cv = c[k];
            float mn = av < bv ? av : bv;
            float mx = av > bv ? av : bv;
            mn = mn < cv ? mn : cv;
            mx = mx > cv ? mx : cv;
            lo[3 * (size_t)i + k] = mn;
            hi[3 * (size_t)i + k] = mx;
            float ctr = mn * 0.5f + mx * 0.5f;
            cmn[k] = ctr < cmn[k] ? ctr : cmn[k];
            cmx[k] = ctr > cmx[k] ? ctr : cmx[k];
        }
    }
    __shared__ float sh[kBlock / 64][6];
    int wave = threadIdx.x >> 6;
    for (int k = 0; k < 3; ++k) {
        float mn = wave_min(cmn[k]), mx = wave_max(cmx[k]);
        if ((threadIdx.x & 63) == 0) {
            sh[wave][k] = mn;
            sh[wave][3 + k] = mx;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int k = threadIdx.x;
        float v = sh[0][k];
        for (int w = 1; w < kBlock / 64; ++w) v = (k < 3) ? (sh[w][k] < v ? sh[w][k] : v) : (sh[w][k] > v ? sh[w][k] : v);
        if (k < 3)
            atomicMin(&cb_ord[k], f2ord(v));
        else
            atomicMax(&cb_ord[k], f2ord(v));
    }
}

// left_shift3 / encode_morton3 (bvh.rs:137-156), offset of the centroid in the centroid bounds (bvh.rs:491-497)
__device__ inline uint32_t spread3(uint32_t x) {
    if (x == (1u << 10)) x -= 1;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}
__global__ void __launch_bounds__(kBlock) k_morton(const float* __restrict__ lo, const float* __restrict__ hi, int n,
                                                   const uint32_t* __restrict__ cb_ord, uint32_t* __restrict__ code,
                                                   int* __restrict__ prim) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float mn = ord2f(cb_ord[k]), mx = ord2f(cb_ord[3 + k]);
        float ctr = lo[3 * (size_t)i + k] * 0.5f + hi[3 * (size_t)i + k] * 0.5f;
        float o = ctr - mn;
        if (mx > mn) o /= mx - mn;
        q[k] = (uint32_t)(o * 1024.0f);
    }
    code[i] = (spread3(q[2]) << 2) | (spread3(q[1]) << 1) | spread3(q[0]);
    prim[i] = i;
}

struct Item {
    int begin, n;
    int bit_treelet;  // (bit + 1) | treelet << 8
    int parent;       // node id | which << 31; -1 = treelet root
};

// bvh.rs:509-528: a treelet starts wherever the top 12 Morton bits change
__global__ void __launch_bounds__(kBlock) k_treelet_flags(const uint32_t* __restrict__ code, int n, int* __restrict__ first) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t p = code[i] >> 18;
    if (i == 0 || p != (code[i - 1] >> 18)) first[p] = i;
}
// single block: compact the present prefixes in ascending order into level-0 work items
__global__ void __launch_bounds__(kBlock) k_treelet_compact(const int* __restrict__ first, int n, Item* __restrict__ items,
                                                            int* __restrict__ level_count, int* __restrict__ n_treelets) {
    __shared__ int sh_count[kBlock];
    __shared__ int sh_start[kMaxTreelets + 1];
    constexpr int per = kMaxTreelets / kBlock;
    int t = threadIdx.x;
    int cnt = 0;
    for (int j = 0; j < per; ++j) cnt += first[t * per + j] >= 0;
    sh_count[t] = cnt;
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int j = 0; j < kBlock; ++j) {
            int c = sh_count[j];
            sh_count[j] = acc;
            acc += c;
        }
        *n_treelets = acc;
        level_count[0] = acc;
        sh_start[acc] = n;
    }
    __syncthreads();
    int at = sh_count[t];
    for (int j = 0; j < per; ++j) {
        int f = first[t * per + j];
        if (f >= 0) sh_start[at++] = f;
    }
    __syncthreads();
    at = sh_count[t];
    for (int j = 0; j < per; ++j)
        if (first[t * per + j] >= 0) {
            Item it;
            it.begin = sh_start[at];
            it.n = sh_start[at + 1] - sh_start[at];
            it.bit_treelet = (kFirstBit + 1) | (at << 8);
            it.parent = -1;
            items[at] = it;
            ++at;
        }
}

struct Nodes {
    int4* info;     // (begin, n, parent | which << 31, (bit + 1) | treelet << 8); bit = split bit, -1.. for leaves unused
    int* child;     // [2 * id + which], -1 = leaf
    float* box;     // [6 * id]: min xyz, max xyz
    int* size;      // subtree node count
    int* local;     // pre-order index inside the treelet
    int* treelet_root;
};

// emit_lbvh (bvh.rs:570-676), one level. Every work item becomes exactly one node, so node ids need no
// allocation: id = level_start[level] + item index. Queue slots for the two children of the interior nodes
// are handed out with one atomic per block (wave ballots + an LDS prefix over the waves).
constexpr int kLevelBlock = 1024;
__global__ void __launch_bounds__(kLevelBlock) k_level(const uint32_t* __restrict__ code, const int* __restrict__ prim,
                                                       const float* __restrict__ lo, const float* __restrict__ hi,
                                                       const Item* __restrict__ in, const int* __restrict__ count_in,
                                                       Item* __restrict__ out, int* __restrict__ count_out, Nodes nd,
                                                       int* __restrict__ level_start, int max_prims) {
    const int count = *count_in;
    const int first_id = level_start[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) level_start[1] = first_id + count;
    __shared__ int sh_wave[kLevelBlock / 64];
    __shared__ int sh_base;
    const int stride = gridDim.x * kLevelBlock;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // block-uniform trip count: threads past the end stay in the loop with valid = false
    for (int base = blockIdx.x * kLevelBlock; base < count; base += stride) {
        int i = base + threadIdx.x;
        bool valid = i < count;
        Item it = valid ? in[i] : Item{0, 0, 0, -1};
        int begin = it.begin, n = it.n, bit = (it.bit_treelet & 0xff) - 1, treelet = it.bit_treelet >> 8;
        int split = 0;
        bool leaf = true;
        if (valid) {
            // ranges that do not split on this bit move on to the next one without a node (bvh.rs:607-620)
            while (bit >= 0 && n >= max_prims) {
                uint32_t mask = 1u << bit;
                if ((code[begin] & mask) != (code[begin + n - 1] & mask)) break;
                --bit;
            }
            leaf = (bit < 0) || (n < max_prims);
            if (!leaf) {
                // first index whose bit differs from the first element's (bvh.rs:622-639)
                uint32_t mask = 1u << bit;
                uint32_t b0 = code[begin] & mask;
                int a = 0, b = n - 1;
                while (a + 1 != b) {
                    int m = (a + b) >> 1;
                    if ((code[begin + m] & mask) == b0)
                        a = m;
                    else
                        b = m;
                }
                split = b;
            }
        }
        unsigned long long mask = __ballot(valid && !leaf);
        if (lane == 0) sh_wave[wave] = __popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
            for (int w = 0; w < kLevelBlock / 64; ++w) {
                int c = sh_wave[w];
                sh_wave[w] = total;
                total += c;
            }
            sh_base = total ? atomicAdd(count_out, 2 * total) : 0;
        }
        __syncthreads();
        int slot = sh_base + 2 * (sh_wave[wave] + __popcll(mask & ((1ull << lane) - 1ull)));
        __syncthreads();  // sh_wave / sh_base are rewritten by the next iteration
        if (!valid) continue;
        int id = first_id + i;
        nd.info[id] = make_int4(begin, n, it.parent, (bit + 1) | (treelet << 8));
        if (it.parent == -1)
            nd.treelet_root[treelet] = id;
        else
            nd.child[2 * (size_t)(it.parent & 0x7fffffff) + ((unsigned)it.parent >> 31)] = id;
        if (leaf) {
            float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int j = 0; j < n; ++j) {
                size_t p = (size_t)prim[begin + j];
                for (int k = 0; k < 3; ++k) {
                    float l = lo[3 * p + k], h = hi[3 * p + k];
                    mn[k] = l < mn[k] ? l : mn[k];
                    mx[k] = h > mx[k] ? h : mx[k];
                }
            }
            for (int k = 0; k < 3; ++k) {
                nd.box[6 * (size_t)id + k] = mn[k];
                nd.box[6 * (size_t)id + 3 + k] = mx[k];
            }
            nd.size[id] = 1;
        } else {
            Item c0{begin, split, bit | (treelet << 8), id};                       // (bit - 1) + 1
            Item c1{begin + split, n - split, bit | (treelet << 8), (int)((unsigned)id | 0x80000000u)};
            out[slot] = c0;
            out[slot + 1] = c1;
        }
    }
}

// interior nodes of one level: bounds = union(child 0, child 1) in that order, subtree size
__global__ void __launch_bounds__(kBlock) k_fit(Nodes nd, const int* __restrict__ level_start, int level) {
    int id = level_start[level] + blockIdx.x * kBlock + threadIdx.x;
    if (id >= level_start[level + 1]) return;
    int c0 = nd.child[2 * (size_t)id], c1 = nd.child[2 * (size_t)id + 1];
    if (c0 < 0) return;
    for (int k = 0; k < 3; ++k) {
        float mn = FLT_MAX, mx = -FLT_MAX;
        float l0 = nd.box[6 * (size_t)c0 + k], h0 = nd.box[6 * (size_t)c0 + 3 + k];
        float l1 = nd.box[6 * (size_t)c1 + k], h1 = nd.box[6 * (size_t)c1 + 3 + k];
        mn = l0 < mn ? l0 : mn;
        mn = l1 < mn ? l1 : mn;
        mx = h0 > mx ? h0 : mx;
        mx = h1 > mx ? h1 : mx;
        nd.box[6 * (size_t)id + k] = mn;
        nd.box[6 * (size_t)id + 3 + k] = mx;
    }
    nd.size[id] = 1 + nd.size[c0] + nd.size[c1];
}

// pre-order index inside the treelet: each step to the parent adds 1, plus the first sibling's subtree for a second child
__global__ void __launch_bounds__(kBlock) k_local_index(Nodes nd, const int* __restrict__ node_count) {
    int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= *node_count) return;
    int local = 0;
    int parent = nd.info[id].z;
    while (parent != -1) {
        int pid = parent & 0x7fffffff;
        local += 1;
        if ((unsigned)parent >> 31) local += nd.size[nd.child[2 * (size_t)pid]];
        parent = nd.info[pid].z;
    }
    nd.local[id] = local;
}

struct RootSummary {
    int n_treelets, n_nodes;
    int level_start[kLevels + 1];
    int size[kMaxTreelets];
    float box[kMaxTreelets * 6];
};
__global__ void __launch_bounds__(kBlock) k_root_summary(Nodes nd, const int* __restrict__ n_treelets,
                                                         const int* __restrict__ node_count,
                                                         const int* __restrict__ level_start, RootSummary* out) {
    int t = blockIdx.x * kBlock + threadIdx.x;
    if (t == 0) {
        out->n_treelets = *n_treelets;
        out->n_nodes = *node_count;
        for (int l = 0; l <= kLevels; ++l) out->level_start[l] = level_start[l];
    }
    if (t >= *n_treelets) return;
    int id = nd.treelet_root[t];
    out->size[t] = nd.size[id];
    for (int k = 0; k < 6; ++k) out->box[6 * t + k] = nd.box[6 * (size_t)id + k];
}

// flatten_bvh_tree (bvh.rs:774-811) for the treelet nodes
__global__ void __launch_bounds__(kBlock) k_emit(Nodes nd, const int* __restrict__ node_count,
                                                 const int* __restrict__ treelet_offset, PbrtLinearBVHNode* __restrict__ out) {
    int id = blockIdx.x * kBlock + threadIdx.x;
    if (id >= *node_count) return;
    int4 info = nd.info[id];
    int base = treelet_offset[info.w >> 8];
    PbrtLinearBVHNode ln;
    for (int k = 0; k < 3; ++k) {
        ln.bounds_min[k] = nd.box[6 * (size_t)id + k];
        ln.bounds_max[k] = nd.box[6 * (size_t)id + 3 + k];
    }
    int c1 = nd.child[2 * (size_t)id + 1];
    if (c1 < 0) {
        ln.offset = info.x;  // leaves take the sorted primitive list in order
        ln.n_primitives = (uint16_t)info.y;
        ln.axis = 0;
    } else {
        ln.offset = base + nd.local[c1];
        ln.n_primitives = 0;
        ln.axis = (uint8_t)(((info.w & 0xff) - 1) % 3);
    }
    ln.pad = 0;
    out[base + nd.local[id]] = ln;
}
__global__ void __launch_bounds__(kBlock) k_scatter_upper(const int* __restrict__ index, const PbrtLinearBVHNode* __restrict__ src,
                                                          int n, PbrtLinearBVHNode* __restrict__ out) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[index[i]] = src[i];
}

struct DeviceArena {
    std::vector<void*> ptrs;
    bool ok = true;
    hipError_t err = hipSuccess;
    template <class T>
    T* alloc(size_t n) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (n ? n : 1) * sizeof(T));
        if (e != hipSuccess) {
            ok = false;
            err = e;
            return nullptr;
        }
        ptrs.push_back(p);
        return (T*)p;
    }
    ~DeviceArena() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};

inline int blocks_for(size_t n) { return (int)((n + kBlock - 1) / kBlock); }

}  // namespace

#define HL_TRY(call)                                                   \
    do {                                                               \
        if (!pb::hip_ok(ctx, (call), #call)) return PBRT_HIP_ERR_DEVICE; \
    } while (0)

namespace {

// Steps 1-9 of the header comment over a mesh that is already on the device. Leaves the flat node array
// (d_nodes, n_nodes) and the leaf order (d_order) in `mem`; ev0/ev1 of the context bracket the build.
int hlbvh_build_core(PbrtHipContext* ctx, DeviceArena& mem, const float* d_pos, const int* d_idx, int n, int max_prims,
                     PbrtLinearBVHNode** d_nodes_out, int32_t* n_nodes_out, const int** d_order_out) {
    hipStream_t st = ctx->stream;
    float* d_lo = mem.alloc<float>(3 * (size_t)n);
    float* d_hi = mem.alloc<float>(3 * (size_t)n);
    uint32_t* d_code[2] = {mem.alloc<uint32_t>(n), mem.alloc<uint32_t>(n)};
    int* d_prim[2] = {mem.alloc<int>(n), mem.alloc<int>(n)};
    Item* d_items[2] = {mem.alloc<Item>(std::max(n, kMaxTreelets)), mem.alloc<Item>(std::max(n, kMaxTreelets))};
    // small state: centroid bounds [6], level counters [kLevels + 1], level starts [kLevels + 1], node counter, treelet count
    constexpr int kSmall = 6 + 2 * (kLevels + 1) + 2;
    uint32_t* d_small = mem.alloc<uint32_t>(kSmall);
    int* d_first = mem.alloc<int>(kMaxTreelets);
    const size_t cap = 2 * (size_t)n;  // leaves <= n, interior nodes < leaves
    Nodes nd;
    nd.info = mem.alloc<int4>(cap);
    nd.child = mem.alloc<int>(2 * cap);
    nd.box = mem.alloc<float>(6 * cap);
    nd.size = mem.alloc<int>(cap);
    nd.local = mem.alloc<int>(cap);
    nd.treelet_root = mem.alloc<int>(kMaxTreelets);
    int* d_treelet_offset = mem.alloc<int>(kMaxTreelets);
    int* d_upper_index = mem.alloc<int>(kMaxTreelets);
    PbrtLinearBVHNode* d_upper_nodes = mem.alloc<PbrtLinearBVHNode>(kMaxTreelets);
    RootSummary* d_summary = mem.alloc<RootSummary>(1);
    size_t sort_bytes = 0;
    HL_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, d_code[0], d_code[1], d_prim[0], d_prim[1], (size_t)n, 0u, 30u, st));
    void* d_sort_tmp = mem.alloc<char>(sort_bytes);
    if (!mem.ok) {
        ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(mem.err);
        return PBRT_HIP_ERR_OOM;
    }
    uint32_t* d_cb = d_small;
    int* d_level_count = (int*)d_small + 6;
    int* d_level_start = d_level_count + (kLevels + 1);
    int* d_n_treelets = d_level_start + (kLevels + 1);

    HL_TRY(hipEventRecord(ctx->ev0, st));
    HL_TRY(hipMemsetAsync(d_small, 0, kSmall * sizeof(uint32_t), st));
    HL_TRY(hipMemsetAsync(d_small, 0xff, 3 * sizeof(uint32_t), st));  // running minima of the centroid bounds
    HL_TRY(hipMemsetAsync(d_first, 0xff, kMaxTreelets * sizeof(int), st));
    HL_TRY(hipMemsetAsync(nd.child, 0xff, 2 * cap * sizeof(int), st));
    hipLaunchKernelGGL(k_prim_bounds, dim3(std::min(blocks_for(n), 2 * std::max(1, ctx->n_cus))), dim3(kBlock), 0, st, d_pos,
                       d_idx, n, d_lo, d_hi, d_cb);
    hipLaunchKernelGGL(k_morton, dim3(blocks_for(n)), dim3(kBlock), 0, st, d_lo, d_hi, n, d_cb, d_code[0], d_prim[0]);
    HL_TRY(rocprim::radix_sort_pairs(d_sort_tmp, sort_bytes, d_code[0], d_code[1], d_prim[0], d_prim[1], (size_t)n, 0u, 30u, st));
    const uint32_t* code = d_code[1];
    const int* prim = d_prim[1];
    hipLaunchKernelGGL(k_treelet_flags, dim3(blocks_for(n)), dim3(kBlock), 0, st, code, n, d_first);
    hipLaunchKernelGGL(k_treelet_compact, dim3(1), dim3(kBlock), 0, st, d_first, n, d_items[0], d_level_count, d_n_treelets);
    const int level_grid = std::max(1, std::min((n + kLevelBlock - 1) / kLevelBlock, 2 * std::max(1, ctx->n_cus)));
    for (int l = 0; l < kLevels; ++l)  // k_level l also writes level_start[l + 1]; level_start[0] = 0 from the memset
        hipLaunchKernelGGL(k_level, dim3(level_grid), dim3(kLevelBlock), 0, st, code, prim, d_lo, d_hi, d_items[l & 1],
                           d_level_count + l, d_items[(l + 1) & 1], d_level_count + l + 1, nd, d_level_start + l, max_prims);
    const int* d_node_count = d_level_start + kLevels;
    // The level sizes are only known on the device: size every k_fit launch for the largest possible level.
    // A level holds disjoint ranges, so at most n nodes.
    for (int l = kLevels - 1; l >= 0; --l)
        hipLaunchKernelGGL(k_fit, dim3(blocks_for(n)), dim3(kBlock), 0, st, nd, d_level_start, l);
    hipLaunchKernelGGL(k_local_index, dim3(blocks_for(cap)), dim3(kBlock), 0, st, nd, d_node_count);
    hipLaunchKernelGGL(k_root_summary, dim3(kMaxTreelets / kBlock), dim3(kBlock), 0, st, nd, d_n_treelets, d_node_count,
                       d_level_start, d_summary);
    HL_TRY(hipGetLastError());
    std::vector<RootSummary> summary(1);
    HL_TRY(hipMemcpyAsync(summary.data(), d_summary, sizeof(RootSummary), hipMemcpyDeviceToHost, st));
    HL_TRY(hipStreamSynchronize(st));
    const RootSummary& rs = summary[0];
    if (rs.n_treelets < 1 || rs.n_treelets > kMaxTreelets || rs.n_nodes < rs.n_treelets || (size_t)rs.n_nodes > cap) {
        ctx->last_error = "hlbvh: inconsistent treelet summary";
        return PBRT_HIP_ERR_DEVICE;
    }

    // SAH over the treelet roots + global pre-order numbering (host, <= 4096 entries)
    std::vector<int32_t> treelet_offset, upper_index;
    std::vector<PbrtLinearBVHNode> upper_nodes;
    int32_t n_nodes = 0;
    int32_t upper_depth = 0;
    pb::hlbvh_upper_tree(rs.box, rs.size, rs.n_treelets, treelet_offset, upper_index, upper_nodes, &n_nodes, &upper_depth);
    if (upper_depth + kLevels - 1 > 64) {  // a treelet adds at most kLevels - 1 levels below its root
        ctx->last_error = "hlbvh: tree deeper than the 64-entry traversal stack (bvh.rs:839)";
        return PBRT_HIP_ERR_INVALID;
    }
    if ((size_t)n_nodes != (size_t)rs.n_nodes + upper_nodes.size()) {
        ctx->last_error = "hlbvh: node count mismatch";
        return PBRT_HIP_ERR_DEVICE;
    }
    PbrtLinearBVHNode* d_out = mem.alloc<PbrtLinearBVHNode>(n_nodes);
    if (!mem.ok) {
        ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(mem.err);
        return PBRT_HIP_ERR_OOM;
    }
    HL_TRY(hipMemcpyAsync(d_treelet_offset, treelet_offset.data(), treelet_offset.size() * sizeof(int), hipMemcpyHostToDevice, st));
    if (!upper_nodes.empty()) {
        HL_TRY(hipMemcpyAsync(d_upper_index, upper_index.data(), upper_index.size() * sizeof(int), hipMemcpyHostToDevice, st));
        HL_TRY(hipMemcpyAsync(d_upper_nodes, upper_nodes.data(), upper_nodes.size() * sizeof(PbrtLinearBVHNode),
                              hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_scatter_upper, dim3(blocks_for(upper_nodes.size())), dim3(kBlock), 0, st, d_upper_index,
                           d_upper_nodes, (int)upper_nodes.size(), d_out);
    }
    hipLaunchKernelGGL(k_emit, dim3(blocks_for(rs.n_nodes)), dim3(kBlock), 0, st, nd, d_node_count, d_treelet_offset, d_out);
    HL_TRY(hipGetLastError());
    HL_TRY(hipEventRecord(ctx->ev1, st));
    HL_TRY(hipStreamSynchronize(st));  // the host vectors above feed async copies
    *d_nodes_out = d_out;
    *n_nodes_out = n_nodes;
    *d_order_out = prim;
    return PBRT_HIP_OK;
}

// ---- re-layout for the traversal kernel (what convert_tree and the triangle loop of pbrt_hip.hip do on the host) ----
__global__ void __launch_bounds__(kBlock) k_mark_interior(const PbrtLinearBVHNode* __restrict__ nodes, int n_nodes,
                                                          int* __restrict__ flag, int* __restrict__ max_count) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    int cnt = 0;
    if (i < n_nodes) {
        cnt = nodes[i].n_primitives;
        flag[i] = cnt == 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt = max(cnt, __shfl_xor(cnt, o));
    // the maximum saturates after a few blocks: read first so that nearly every wave skips the atomic
    if ((threadIdx.x & 63) == 0 && cnt > *(volatile int*)max_count) atomicMax(max_count, cnt);
}
__device__ inline int child_ref(const PbrtLinearBVHNode* nodes, const int* interior_index, int node, int count_bits) {
    PbrtLinearBVHNode nd = nodes[node];
    if (nd.n_primitives > 0) return ~(int)(((uint32_t)nd.offset << count_bits) | (uint32_t)(nd.n_primitives - 1));
    return interior_index[node];
}
__global__ void __launch_bounds__(kBlock) k_convert(const PbrtLinearBVHNode* __restrict__ nodes, int n_nodes,
                                                    const int* __restrict__ interior_index, int count_bits,
                                                    float4* __restrict__ inodes) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    PbrtLinearBVHNode nd = nodes[i];
    if (nd.n_primitives > 0) return;
    PbrtLinearBVHNode c0 = nodes[i + 1], c1 = nodes[nd.offset];
    float4* r = inodes + 4 * (size_t)interior_index[i];
    r[0] = make_float4(c0.bounds_min[0], c0.bounds_min[1], c0.bounds_min[2], c0.bounds_max[0]);
    r[1] = make_float4(c0.bounds_max[1], c0.bounds_max[2], c1.bounds_min[0], c1.bounds_min[1]);
    r[2] = make_float4(c1.bounds_min[2], c1.bounds_max[0], c1.bounds_max[1], c1.bounds_max[2]);
    r[3] = make_float4(__int_as_float(child_ref(nodes, interior_index, i + 1, count_bits)),
                       __int_as_float(child_ref(nodes, interior_index, nd.offset, count_bits)), __int_as_float((int)nd.axis),
                       __int_as_float(-1));  // parent link: k_parent_links
}
// the record an interior node is a child of, into the fourth field of its last float4 (trace_stackless.h); the root keeps -1.
// After k_convert: the two kernels write different words of a child's record, but k_convert writes the whole float4.
__global__ void __launch_bounds__(kBlock) k_parent_links(const PbrtLinearBVHNode* __restrict__ nodes, int n_nodes,
                                                         const int* __restrict__ interior_index, float4* __restrict__ inodes) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    PbrtLinearBVHNode nd = nodes[i];
    if (nd.n_primitives > 0) return;
    const int me = interior_index[i];
    if (nodes[i + 1].n_primitives == 0) ((int*)(inodes + 4 * (size_t)interior_index[i + 1] + 3))[3] = me;
    if (nodes[nd.offset].n_primitives == 0) ((int*)(inodes + 4 * (size_t)interior_index[nd.offset] + 3))[3] = me;
}
__global__ void __launch_bounds__(kBlock) k_tri_records(const float* __restrict__ pos, const int* __restrict__ idx,
                                                        const int* __restrict__ order, const int* __restrict__ tri_material,
                                                        const int* __restrict__ tri_light, int n, float4* __restrict__ tris,
                                                        int* __restrict__ prim_slot) {
    int slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= n) return;
    int prim = order[slot];
    prim_slot[prim] = slot;
    float a[3], b[3], c[3];
    for (int k = 0; k < 3; ++k) {
        a[k] = pos[3 * (size_t)idx[3 * (size_t)prim] + k];
        b[k] = pos[3 * (size_t)idx[3 * (size_t)prim + 1] + k];
        c[k] = pos[3 * (size_t)idx[3 * (size_t)prim + 2] + k];
    }
    int material = tri_material ? tri_material[prim] : 0;
    int light = (tri_light ? tri_light[prim] + 1 : 0) | (pb::triangle_rejected_by_intersect(a, b, c) ? pb::kTriDegenerate : 0);
    float4* t = tris + 3 * (size_t)slot;
    t[0] = make_float4(a[0], a[1], a[2], b[0]);
    t[1] = make_float4(b[1], b[2], c[0], c[1]);
    t[2] = make_float4(c[2], __int_as_float(prim), __int_as_float(material), __int_as_float(light));
}
__global__ void __launch_bounds__(kBlock) k_light_slots(const int* __restrict__ light_prim, int n_lights,
                                                        const int* __restrict__ prim_slot, int* __restrict__ light_slot) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_lights) light_slot[i] = light_prim[i] >= 0 ? prim_slot[light_prim[i]] : -1;
}

}  // namespace

extern "C" int pbrt_hip_bvh_build_hlbvh_device(PbrtHipContext* ctx, const float* positions, int32_t n_verts,
                                               const int32_t* indices, int32_t n_tris, int32_t max_prims_in_node,
                                               PbrtLinearBVHNode** nodes_out, int32_t* n_nodes_out,
                                               int32_t** prim_order_out, double* build_ms) try {
    if (!ctx) return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    auto invalid = [&](const char* m) {
        ctx->last_error = m;
        return PBRT_HIP_ERR_INVALID;
    };
    if (!nodes_out || !n_nodes_out || !prim_order_out) return invalid("null output pointer");
    *nodes_out = nullptr;
    *prim_order_out = nullptr;
    *n_nodes_out = 0;
    if (build_ms) *build_ms = 0.0;
    if (n_tris < 0 || n_verts < 0 || (n_tris > 0 && (!positions || !indices))) return invalid("bad mesh arguments");
    if (n_tris == 0) return PBRT_HIP_OK;  // bvh.rs:228-230
    for (int64_t i = 0; i < 3 * (int64_t)n_tris; ++i)
        if (indices[i] < 0 || indices[i] >= n_verts) return invalid("vertex index out of range");
    const int n = n_tris;
    HL_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DeviceArena mem;
    float* d_pos = mem.alloc<float>(3 * (size_t)n_verts);
    int* d_idx = mem.alloc<int>(3 * (size_t)n);
    if (!mem.ok) {
        ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(mem.err);
        return PBRT_HIP_ERR_OOM;
    }
    HL_TRY(hipMemcpyAsync(d_pos, positions, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    HL_TRY(hipMemcpyAsync(d_idx, indices, 3 * (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    PbrtLinearBVHNode* d_out = nullptr;
    const int* d_order = nullptr;
    int32_t n_nodes = 0;
    int rc = hlbvh_build_core(ctx, mem, d_pos, d_idx, n, std::min(max_prims_in_node, 255) /* bvh.rs:222 */, &d_out, &n_nodes,
                              &d_order);
    if (rc != PBRT_HIP_OK) return rc;
    if (build_ms) {
        float ms = 0.0f;
        HL_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *build_ms = ms;
    }
    PbrtLinearBVHNode* nodes = (PbrtLinearBVHNode*)std::malloc((size_t)n_nodes * sizeof(PbrtLinearBVHNode));
    int32_t* order = (int32_t*)std::malloc((size_t)n * sizeof(int32_t));
    if (!nodes || !order) {
        std::free(nodes);
        std::free(order);
        return PBRT_HIP_ERR_OOM;
    }
    hipError_t ce = hipMemcpyAsync(nodes, d_out, (size_t)n_nodes * sizeof(PbrtLinearBVHNode), hipMemcpyDeviceToHost, st);
    if (ce == hipSuccess) ce = hipMemcpyAsync(order, d_order, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (ce == hipSuccess) ce = hipStreamSynchronize(st);
    if (ce != hipSuccess) {
        std::free(nodes);
        std::free(order);
        pb::hip_ok(ctx, ce, "hlbvh: copy back");
        return PBRT_HIP_ERR_DEVICE;
    }
    *nodes_out = nodes;
    *n_nodes_out = n_nodes;
    *prim_order_out = order;
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

// Stable radix sort of (key, value) pairs on `bits` key bits (rocPRIM), used by the render loop to put the ray queue
// into spatial order. temp == nullptr: only reports the scratch size.
namespace pb {
int sort_pairs_u32(hipStream_t st, void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                   const uint32_t* vals_in, uint32_t* vals_out, size_t n, int bits) {
    hipError_t e = rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, (unsigned)bits, st);
    return e == hipSuccess ? 0 : 1;
}
}  // namespace pb

// Tree + triangle records + leaf order for a single-level scene, all produced on the device. Inputs are
// validated by the caller (scene_create_impl's checks on indices / tri_material / tri_light / lights).
namespace pb {
int hlbvh_build_scene_tree(PbrtHipContext* ctx, const float* positions, int32_t n_verts, const int32_t* indices,
                           int32_t n_tris, const int32_t* tri_material, const int32_t* tri_light, const PbrtLight* lights,
                           int32_t n_lights, int32_t max_prims_in_node, DeviceTree* out) {
    const int n = n_tris;
    HL_TRY(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    DeviceArena mem;
    float* d_pos = mem.alloc<float>(3 * (size_t)n_verts);
    int* d_idx = mem.alloc<int>(3 * (size_t)n);
    int* d_mat = tri_material ? mem.alloc<int>(n) : nullptr;
    int* d_tl = tri_light ? mem.alloc<int>(n) : nullptr;
    int* d_prim_slot = mem.alloc<int>(n);
    int* d_light_prim = mem.alloc<int>(std::max(1, n_lights));
    int* d_light_slot = mem.alloc<int>(std::max(1, n_lights));
    int* d_max_count = mem.alloc<int>(1);
    if (!mem.ok) {
        ctx->last_error = std::string("hipMalloc: ") + hipGetErrorString(mem.err);
        return PBRT_HIP_ERR_OOM;
    }
    std::vector<int> light_prim(std::max(1, n_lights), -1);
    for (int32_t i = 0; i < n_lights; ++i)
        if (lights[i].type == PBRT_LIGHT_DIFFUSE_AREA) light_prim[i] = lights[i].prim;
    HL_TRY(hipMemcpyAsync(d_pos, positions, 3 * (size_t)n_verts * sizeof(float), hipMemcpyHostToDevice, st));
    HL_TRY(hipMemcpyAsync(d_idx, indices, 3 * (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    if (d_mat) HL_TRY(hipMemcpyAsync(d_mat, tri_material, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    if (d_tl) HL_TRY(hipMemcpyAsync(d_tl, tri_light, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    HL_TRY(hipMemcpyAsync(d_light_prim, light_prim.data(), light_prim.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HL_TRY(hipMemsetAsync(d_max_count, 0, sizeof(int), st));
    PbrtLinearBVHNode* d_nodes = nullptr;
    const int* d_order = nullptr;
    int32_t n_nodes = 0;
    int rc = hlbvh_build_core(ctx, mem, d_pos, d_idx, n, std::min(max_prims_in_node, 255), &d_nodes, &n_nodes, &d_order);
    if (rc != PBRT_HIP_OK) return rc;
    {
        float ms = 0.0f;
        HL_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        out->build_ms = ms;
    }
    // ---- re-layout ----
    HL_TRY(hipEventRecord(ctx->ev0, st));
    const int n_interior = (n_nodes - 1) / 2;
    int* d_flag = mem.alloc<int>(n_nodes);
    int* d_interior_index = mem.alloc<int>(n_nodes);
    size_t scan_bytes = 0;
    HL_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, d_flag, d_interior_index, 0, (size_t)n_nodes, rocprim::plus<int>(), st));
    void* d_scan_tmp = mem.alloc<char>(scan_bytes);
    // the three arrays the scene keeps
    void *p_inodes = nullptr, *p_tris = nullptr, *p_slot_prim = nullptr;
    auto release = [&]() {
        (void)hipFree(p_inodes);
        (void)hipFree(p_tris);
        (void)hipFree(p_slot_prim);
    };
    if (!mem.ok || hipMalloc(&p_inodes, (size_t)std::max(1, n_interior) * 64) != hipSuccess ||
        hipMalloc(&p_tris, (size_t)n * 48) != hipSuccess || hipMalloc(&p_slot_prim, (size_t)n * sizeof(int)) != hipSuccess) {
        release();
        ctx->last_error = "hipMalloc: scene tree";
        return PBRT_HIP_ERR_OOM;
    }
    hipLaunchKernelGGL(k_mark_interior, dim3(blocks_for(n_nodes)), dim3(kBlock), 0, st, d_nodes, n_nodes, d_flag, d_max_count);
    hipError_t e = rocprim::exclusive_scan(d_scan_tmp, scan_bytes, d_flag, d_interior_index, 0, (size_t)n_nodes,
                                           rocprim::plus<int>(), st);
    int max_count = 0;
    PbrtLinearBVHNode root;
    if (e == hipSuccess) e = hipMemcpyAsync(&max_count, d_max_count, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&root, d_nodes, sizeof(root), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        release();
        pb::hip_ok(ctx, e, "hlbvh: re-layout");
        return PBRT_HIP_ERR_DEVICE;
    }
    int count_bits = 0;
    while ((1 << count_bits) < std::max(1, max_count)) ++count_bits;
    if (((int64_t)n << count_bits) >= (1ll << 31)) {
        release();
        ctx->last_error = "scene too large for 31-bit leaf references";
        return PBRT_HIP_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_convert, dim3(blocks_for(n_nodes)), dim3(kBlock), 0, st, d_nodes, n_nodes, d_interior_index, count_bits,
                       (float4*)p_inodes);
    hipLaunchKernelGGL(k_parent_links, dim3(blocks_for(n_nodes)), dim3(kBlock), 0, st, d_nodes, n_nodes, d_interior_index,
                       (float4*)p_inodes);
    hipLaunchKernelGGL(k_tri_records, dim3(blocks_for(n)), dim3(kBlock), 0, st, d_pos, d_idx, d_order, d_mat, d_tl, n,
                       (float4*)p_tris, d_prim_slot);
    hipLaunchKernelGGL(k_light_slots, dim3(blocks_for(std::max(1, n_lights))), dim3(kBlock), 0, st, d_light_prim, n_lights,
                       d_prim_slot, d_light_slot);
    out->light_slot.assign(std::max(1, n_lights), -1);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(p_slot_prim, d_order, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess && n_lights > 0)
        e = hipMemcpyAsync(out->light_slot.data(), d_light_slot, (size_t)n_lights * sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev1, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        release();
        pb::hip_ok(ctx, e, "hlbvh: re-layout");
        return PBRT_HIP_ERR_DEVICE;
    }
    {
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        out->convert_ms = ms;
    }
    {
        // The wide records are laid over the flat tree while it is still on the device (wide_gpu.hip: a few ms per 10 M
        // triangles). PBRT_WIDE_BUILD_HOST (pbrt_hip_context_set_wide_build) brings the flat tree and the triangle records back
        // instead and lets the host builder do it (host_wide.cpp; 80 B per triangle over PCIe + tens of ms per million
        // triangles): same bytes, kept for the test of exactly that. PBRT_WIDE_BUILD_NONE: binary records only.
        if (ctx->wide_build == PBRT_WIDE_BUILD_NONE) {
            out->wide_reason = "disabled by PBRT_WIDE_BUILD_NONE";
        } else if (ctx->wide_build != PBRT_WIDE_BUILD_HOST) {
            if (!pb::build_wide_tree_device(ctx, d_nodes, n_nodes, (const float*)p_tris, n, root, &out->wide, &out->wide_reason)) {
                release();
                return PBRT_HIP_ERR_DEVICE;
            }
        } else {
            out->h_nodes.resize(n_nodes);
            out->h_tris.resize((size_t)n * 12);
            e = hipMemcpy(out->h_nodes.data(), d_nodes, (size_t)n_nodes * sizeof(PbrtLinearBVHNode), hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(out->h_tris.data(), p_tris, (size_t)n * 48, hipMemcpyDeviceToHost);
            if (e != hipSuccess) {
                release();
                pb::hip_ok(ctx, e, "hlbvh: copy of the tree for the wide records");
                return PBRT_HIP_ERR_DEVICE;
            }
        }
    }
    out->inodes = (float4*)p_inodes;
    out->tris = (float4*)p_tris;
    out->slot_prim = (int*)p_slot_prim;
    std::memcpy(out->root_min, root.bounds_min, 12);
    std::memcpy(out->root_max, root.bounds_max, 12);
    out->n_nodes = n_nodes;
    out->n_interior = n_interior;
    out->count_bits = count_bits;
    out->root_ref = root.n_primitives > 0
                        ? ~(int32_t)(((uint32_t)root.offset << count_bits) | (uint32_t)(root.n_primitives - 1))
                        : 0;
    return PBRT_HIP_OK;
}
}  // namespace pb

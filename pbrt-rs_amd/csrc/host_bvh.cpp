// host_bvh.cpp — host-side BVH construction for the HIP path.
//
// Mirrors BVHAccel::new of the reference (src/accelerators/bvh.rs:216-271): primitive bounds and
// centroids (:26-41), recursive_build (:273-473) with the SAH 12-bucket split (:373-448), Middle
// (:329-347) and EqualCounts (:348-359) splits, and the depth-first flattening of :774-811 (first
// child = self + 1, `offset` = second child; leaves index the reordered primitive list). The
// north_star keeps BVH build on the host; the kernels consume the flat array.
//
// Differences from the reference's text, all "intended pbrt-v3" dispositions of SURVEY.md §2.3:
// union results are kept (D16), the partition range is [start, end) (D17), the bucket index is
// floor(12 * offset) (D18), the SAH prefix covers buckets 0..=i (D19), and the partition
// predicate is `bucket <= best` (bvh.rs:424-431 compares the untruncated float with `<`).
//
// Implementation: the node array is emitted directly in pre-order (no build-node arena), over a
// permutation of primitive indices with SoA bounds/centroids.
#include <algorithm>
#include "abi_guard.h"
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/pbrt_hip.h"

namespace {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; ++a) {
            mn[a] = FLT_MAX;   // Bounds3::default, src/core/geometry.rs:439-448
            mx[a] = -FLT_MAX;
        }
    }
    void grow(const float* lo, const float* hi) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = lo[a] < mn[a] ? lo[a] : mn[a];
            mx[a] = hi[a] > mx[a] ? hi[a] : mx[a];
        }
    }
    void grow(const Box& b) { grow(b.mn, b.mx); }
    float area() const {  // src/core/geometry.rs:667-670
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2.0f * (dx * dy + dx * dz + dy * dz);
    }
    int widest() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx > dy && dx > dz) return 0;
        return dy > dz ? 1 : 2;
    }
};

struct Builder {
    const float* lo;   // n x 3 primitive bounds min
    const float* hi;   // n x 3 primitive bounds max
    const float* ctr;  // n x 3 centroids
    int32_t* perm = nullptr;            // working permutation of primitive indices (shared by the parallel sub-builders)
    std::vector<PbrtLinearBVHNode> out;
    std::vector<int32_t> order;         // leaf slots -> primitive index
    int max_prims;
    int method;

    int leaf(int self, int start, int end, const Box& b) {
        PbrtLinearBVHNode& n = out[self];
        std::memcpy(n.bounds_min, b.mn, 12);
        std::memcpy(n.bounds_max, b.mx, 12);
        n.offset = (int32_t)order.size();
        n.n_primitives = (uint16_t)(end - start);
        n.axis = 0;
        n.pad = 0;
        for (int i = start; i < end; ++i) order.push_back(perm[i]);
        return self;
    }

    // Two-pointer partition: first false from the front swaps with last true from the back.
    template <class Pred>
    int partition(int start, int end, Pred pred) {
        int i = start, j = end;
        for (;;) {
            while (i < j && pred(perm[i])) ++i;
            if (i == j) return i;
            --j;
            while (i < j && !pred(perm[j])) --j;
            if (i == j) return i;
            std::swap(perm[i], perm[j]);
            ++i;
        }
    }

    float centroid_offset(const Box& cb, int prim, int dim) const {  // Bounds3::offset
        float o = ctr[3 * prim + dim] - cb.mn[dim];
        if (cb.mx[dim] > cb.mn[dim]) o /= cb.mx[dim] - cb.mn[dim];
        return o;
    }

    // par_depth > 0: the two children of a large range are built concurrently (they own disjoint ranges of `perm`);
    // the second child goes into a builder of its own and is appended afterwards, so the output is the serial one.
    int build(int start, int end, int par_depth = 0) {
        int self = (int)out.size();
        out.emplace_back();
        Box b;
        b.reset();
        for (int i = start; i < end; ++i) b.grow(lo + 3 * perm[i], hi + 3 * perm[i]);
        int n = end - start;
        if (n == 1) return leaf(self, start, end, b);
        Box cb;
        cb.reset();
        for (int i = start; i < end; ++i) cb.grow(ctr + 3 * perm[i], ctr + 3 * perm[i]);
        int dim = cb.widest();
        if (cb.mx[dim] == cb.mn[dim]) return leaf(self, start, end, b);

        int mid = (start + end) / 2;
        bool median_split = false;
        if (method == 2) {  // Middle
            float p_mid = (cb.mn[dim] + cb.mx[dim]) / 2.0f;
            mid = partition(start, end, [&](int p) { return ctr[3 * p + dim] < p_mid; });
            if (mid == start || mid == end) median_split = true;
        } else if (method == 3 || n <= 2) {  // EqualCounts, or SAH on <= 2 primitives
            median_split = true;
        } else {  // SAH
            const int NB = 12;
            int count[NB] = {0};
            Box bb[NB];
            for (int k = 0; k < NB; ++k) bb[k].reset();
            for (int i = start; i < end; ++i) {
                int k = (int)((float)NB * centroid_offset(cb, perm[i], dim));
                if (k == NB) k = NB - 1;
                count[k]++;
                bb[k].grow(lo + 3 * perm[i], hi + 3 * perm[i]);
            }
            float best_cost = FLT_MAX;
            int best = 0;
            float parent_area = b.area();
            for (int s = 0; s < NB - 1; ++s) {
                Box b0, b1;
                b0.reset();
                b1.reset();
                int c0 = 0, c1 = 0;
                for (int j = 0; j <= s; ++j) {
                    b0.grow(bb[j]);
                    c0 += count[j];
                }
                for (int j = s + 1; j < NB; ++j) {
                    b1.grow(bb[j]);
                    c1 += count[j];
                }
                float cost = 1.0f + ((float)c0 * b0.area() + (float)c1 * b1.area()) / parent_area;
                if (cost < best_cost) {
                    best_cost = cost;
                    best = s;
                }
            }
            if (n > max_prims || best_cost < (float)n) {
                mid = partition(start, end, [&](int p) {
                    int k = (int)((float)NB * centroid_offset(cb, p, dim));
                    if (k == NB) k = NB - 1;
                    return k <= best;
                });
            } else {
                return leaf(self, start, end, b);
            }
        }
        if (median_split) {
            mid = (start + end) / 2;
            std::nth_element(perm + start, perm + mid, perm + end,
                             [&](int a, int c) { return ctr[3 * a + dim] < ctr[3 * c + dim]; });
        }
        int c0, c1;
        if (par_depth > 0 && n >= (1 << 14)) {
            Builder right;
            right.lo = lo;
            right.hi = hi;
            right.ctr = ctr;
            right.perm = perm;
            right.max_prims = max_prims;
            right.method = method;
            right.out.reserve(2 * (size_t)(end - mid));
            right.order.reserve(end - mid);
            std::thread worker([&]() { right.build(mid, end, par_depth - 1); });
            c0 = build(start, mid, par_depth - 1);
            worker.join();
            c1 = (int)out.size();
            const int32_t order_base = (int32_t)order.size();
            for (PbrtLinearBVHNode nd : right.out) {
                nd.offset += nd.n_primitives > 0 ? order_base : c1;  // leaf: first slot; interior: second child
                out.push_back(nd);
            }
            order.insert(order.end(), right.order.begin(), right.order.end());
        } else {
            c0 = build(start, mid, 0);
            c1 = build(mid, end, 0);
        }
        Box u;
        u.reset();
        u.grow(out[c0].bounds_min, out[c0].bounds_max);
        u.grow(out[c1].bounds_min, out[c1].bounds_max);
        PbrtLinearBVHNode& nd = out[self];
        std::memcpy(nd.bounds_min, u.mn, 12);
        std::memcpy(nd.bounds_max, u.mx, 12);
        nd.offset = c1;
        nd.n_primitives = 0;
        nd.axis = (uint8_t)dim;
        nd.pad = 0;
        return self;
    }
};

// ---------------------------------------------------------------------------------------------
// HLBVH (src/accelerators/bvh.rs:475-772): Morton codes of the centroids (:137-156, :490-502), LSD
// radix sort 6 bits x 5 passes (:158-197), one LBVH treelet per distinct top-12-bit prefix
// (:509-528, emit_lbvh :570-676), SAH over the treelet roots (:678-772), DFS flattening (:774-811).
// Dispositions (SURVEY.md D20 and the D18/D45 analogues in build_upper_sah): treelet ranges advance,
// treelets are emitted in order so leaf offsets are deterministic, bucket = floor(12 * offset),
// partition keeps buckets <= best; coincident treelet centroids fall back to the median.
// ---------------------------------------------------------------------------------------------
struct HlbvhBuilder {
    struct Node {
        Box box;
        int child[2] = {-1, -1};
        int axis = 0, first = 0, count = 0;
    };
    const float* lo;
    const float* hi;
    const float* ctr;
    int max_prims;
    std::vector<Node> pool;
    std::vector<int32_t> order;
    std::vector<uint32_t> code;
    std::vector<int32_t> prim;  // primitive index, sorted by Morton code

    static uint32_t spread3(uint32_t x) {
        if (x == (1u << 10)) x -= 1;
        x = (x | (x << 16)) & 0x30000ffu;
        x = (x | (x << 8)) & 0x300f00fu;
        x = (x | (x << 4)) & 0x30c30c3u;
        x = (x | (x << 2)) & 0x9249249u;
        return x;
    }
    void sort_by_code() {
        size_t n = code.size();
        std::vector<uint32_t> code2(n);
        std::vector<int32_t> prim2(n);
        for (int pass = 0; pass < 5; ++pass) {
            int shift = 6 * pass;
            size_t start[65] = {0};
            for (size_t i = 0; i < n; ++i) start[((code[i] >> shift) & 63u) + 1]++;
            for (int b = 0; b < 64; ++b) start[b + 1] += start[b];
            for (size_t i = 0; i < n; ++i) {
                size_t dst = start[(code[i] >> shift) & 63u]++;
                code2[dst] = code[i];
                prim2[dst] = prim[i];
            }
            code.swap(code2);
            prim.swap(prim2);
        }
    }
    int emit(size_t begin, int n, int bit) {
        if (bit == -1 || n < max_prims) {
            int self = (int)pool.size();
            pool.emplace_back();
            Node& nd = pool.back();
            nd.box.reset();
            nd.first = (int)order.size();
            nd.count = n;
            for (int i = 0; i < n; ++i) {
                int32_t p = prim[begin + i];
                order.push_back(p);
                nd.box.grow(lo + 3 * (size_t)p, hi + 3 * (size_t)p);
            }
            return self;
        }
        uint32_t mask = 1u << bit;
        if ((code[begin] & mask) == (code[begin + n - 1] & mask)) return emit(begin, n, bit - 1);
        int a = 0, b = n - 1;  // last index with the first element's bit / first index with the other
        while (a + 1 != b) {
            int m = (a + b) / 2;
            if ((code[begin + a] & mask) == (code[begin + m] & mask)) a = m;
            else b = m;
        }
        int self = (int)pool.size();
        pool.emplace_back();
        int c0 = emit(begin, b, bit - 1);
        int c1 = emit(begin + b, n - b, bit - 1);
        Node& nd = pool[self];
        nd.child[0] = c0;
        nd.child[1] = c1;
        nd.box.reset();
        nd.box.grow(pool[c0].box);
        nd.box.grow(pool[c1].box);
        nd.axis = bit % 3;
        return self;
    }
    int upper(std::vector<int>& roots, int start, int end) {
        if (end - start == 1) return roots[start];
        int self = (int)pool.size();
        pool.emplace_back();
        Box b, cb;
        b.reset();
        cb.reset();
        for (int i = start; i < end; ++i) b.grow(pool[roots[i]].box);
        for (int i = start; i < end; ++i) {
            const Box& r = pool[roots[i]].box;
            float c[3];
            for (int k = 0; k < 3; ++k) c[k] = (r.mn[k] + r.mx[k]) * 0.5f;
            cb.grow(c, c);
        }
        int dim = cb.widest();
        const int NB = 12;
        auto bucket = [&](int root) {
            const Box& r = pool[root].box;
            float c = (r.mn[dim] + r.mx[dim]) * 0.5f;
            int k = (int)((float)NB * ((c - cb.mn[dim]) / (cb.mx[dim] - cb.mn[dim])));
            return k == NB ? NB - 1 : k;
        };
        int mid = start;
        if (cb.mx[dim] != cb.mn[dim]) {
            int count[NB] = {0};
            Box bb[NB];
            for (int k = 0; k < NB; ++k) bb[k].reset();
            for (int i = start; i < end; ++i) {
                int k = bucket(roots[i]);
                count[k]++;
                bb[k].grow(pool[roots[i]].box);
            }
            float best_cost = FLT_MAX, area = b.area();
            int best = 0;
            for (int s = 0; s < NB - 1; ++s) {
                Box b0, b1;
                b0.reset();
                b1.reset();
                int c0 = 0, c1 = 0;
                for (int j = 0; j <= s; ++j) {
                    b0.grow(bb[j]);
                    c0 += count[j];
                }
                for (int j = s + 1; j < NB; ++j) {
                    b1.grow(bb[j]);
                    c1 += count[j];
                }
                float cost = 0.125f + ((float)c0 * b0.area() + (float)c1 * b1.area()) / area;
                if (cost < best_cost) {
                    best_cost = cost;
                    best = s;
                }
            }
            int i = start, j = end;  // two-pointer partition on `bucket <= best`
            for (;;) {
                while (i < j && bucket(roots[i]) <= best) ++i;
                if (i == j) break;
                --j;
                while (i < j && !(bucket(roots[j]) <= best)) --j;
                if (i == j) break;
                std::swap(roots[i], roots[j]);
                ++i;
            }
            mid = i;
        }
        if (mid == start || mid == end) mid = (start + end) / 2;
        int c0 = upper(roots, start, mid);
        int c1 = upper(roots, mid, end);
        Node& nd = pool[self];
        nd.child[0] = c0;
        nd.child[1] = c1;
        nd.box.reset();
        nd.box.grow(pool[c0].box);
        nd.box.grow(pool[c1].box);
        nd.axis = dim;
        return self;
    }
    int flatten(int node, std::vector<PbrtLinearBVHNode>& out) {
        int self = (int)out.size();
        out.emplace_back();
        const Node& nd = pool[node];
        PbrtLinearBVHNode ln;
        std::memcpy(ln.bounds_min, nd.box.mn, 12);
        std::memcpy(ln.bounds_max, nd.box.mx, 12);
        ln.pad = 0;
        if (nd.count > 0) {
            ln.offset = nd.first;
            ln.n_primitives = (uint16_t)nd.count;
            ln.axis = 0;
            out[self] = ln;
        } else {
            ln.n_primitives = 0;
            ln.axis = (uint8_t)nd.axis;
            ln.offset = 0;
            out[self] = ln;
            flatten(nd.child[0], out);
            out[self].offset = flatten(nd.child[1], out);
        }
        return self;
    }
    void run(int32_t n, std::vector<PbrtLinearBVHNode>& out) {
        Box cb;
        cb.reset();
        for (int32_t i = 0; i < n; ++i) cb.grow(ctr + 3 * (size_t)i, ctr + 3 * (size_t)i);
        code.resize(n);
        prim.resize(n);
        for (int32_t i = 0; i < n; ++i) {
            uint32_t q[3];
            for (int k = 0; k < 3; ++k) {
                float o = ctr[3 * (size_t)i + k] - cb.mn[k];
                if (cb.mx[k] > cb.mn[k]) o /= cb.mx[k] - cb.mn[k];
                q[k] = (uint32_t)(o * 1024.0f);
            }
            code[i] = (spread3(q[2]) << 2) | (spread3(q[1]) << 1) | spread3(q[0]);
            prim[i] = i;
        }
        sort_by_code();
        std::vector<int> roots;
        const uint32_t prefix = 0x3ffc0000u;
        for (size_t start = 0, end = 1; end <= (size_t)n; ++end)
            if (end == (size_t)n || (code[start] & prefix) != (code[end] & prefix)) {
                roots.push_back(emit(start, (int)(end - start), 29 - 12));
                start = end;
            }
        int root = upper(roots, 0, (int)roots.size());
        out.reserve(pool.size());
        flatten(root, out);
    }
};

}  // namespace

// Upper levels of the HLBVH for the device builder (hlbvh_gpu.hip): the SAH tree over the treelet
// roots (bvh.rs:678-772) and the pre-order numbering of bvh.rs:774-811, with every treelet standing in
// for its `sizes[t]` nodes. boxes6 = {min xyz, max xyz} per treelet root.
namespace pb {
void hlbvh_upper_tree(const float* boxes6, const int32_t* sizes, int32_t n, std::vector<int32_t>& treelet_offset,
                      std::vector<int32_t>& upper_index, std::vector<PbrtLinearBVHNode>& upper_nodes,
                      int32_t* n_nodes_total, int32_t* upper_depth) {
    HlbvhBuilder hb;
    hb.lo = hb.hi = hb.ctr = nullptr;
    hb.max_prims = 0;
    hb.pool.resize(n);
    hb.pool.reserve(2 * (size_t)n);
    std::vector<int> roots(n);
    for (int32_t t = 0; t < n; ++t) {
        std::memcpy(hb.pool[t].box.mn, boxes6 + 6 * (size_t)t, 12);
        std::memcpy(hb.pool[t].box.mx, boxes6 + 6 * (size_t)t + 3, 12);
        roots[t] = t;
    }
    int root = hb.upper(roots, 0, n);
    treelet_offset.assign(n, 0);
    upper_index.clear();
    upper_nodes.clear();
    int32_t next = 0;
    // explicit stack: (node, slot of the parent's record waiting for its second-child offset or -1)
    struct Item { int node; int parent_slot; int depth; };
    std::vector<Item> stack;
    stack.push_back({root, -1, 1});
    int max_depth = 0;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        max_depth = std::max(max_depth, it.depth);
        if (it.parent_slot >= 0) upper_nodes[it.parent_slot].offset = next;  // this subtree is a second child
        if (it.node < n) {
            treelet_offset[it.node] = next;
            next += sizes[it.node];
            continue;
        }
        const HlbvhBuilder::Node& nd = hb.pool[it.node];
        PbrtLinearBVHNode ln;
        std::memcpy(ln.bounds_min, nd.box.mn, 12);
        std::memcpy(ln.bounds_max, nd.box.mx, 12);
        ln.offset = 0;
        ln.n_primitives = 0;
        ln.axis = (uint8_t)nd.axis;
        ln.pad = 0;
        upper_index.push_back(next++);
        upper_nodes.push_back(ln);
        int slot = (int)upper_nodes.size() - 1;
        stack.push_back({nd.child[1], slot, it.depth + 1});   // popped after the whole first subtree
        stack.push_back({nd.child[0], -1, it.depth + 1});
    }
    *n_nodes_total = next;
    if (upper_depth) *upper_depth = max_depth;  // levels down to and including the treelet roots
}
}  // namespace pb

static int build_from_boxes(std::vector<float>& lo, std::vector<float>& hi, int32_t n, int32_t max_prims_in_node,
                            int32_t split_method, PbrtLinearBVHNode** nodes_out, int32_t* n_nodes_out,
                            int32_t** prim_order_out) {
    std::vector<float> ctr(3 * (size_t)n);
    for (size_t i = 0; i < 3 * (size_t)n; ++i) ctr[i] = lo[i] * 0.5f + hi[i] * 0.5f;  // bvh.rs:38
    Builder bl;
    if (split_method == 1) {  // HLBVH
        HlbvhBuilder hb;
        hb.lo = lo.data();
        hb.hi = hi.data();
        hb.ctr = ctr.data();
        hb.max_prims = std::min(max_prims_in_node, 255);
        hb.order.reserve(n);
        hb.pool.reserve(2 * (size_t)n);
        hb.run(n, bl.out);
        bl.order.swap(hb.order);
    } else {
        bl.lo = lo.data();
        bl.hi = hi.data();
        bl.ctr = ctr.data();
        std::vector<int32_t> perm(n);
        for (int32_t i = 0; i < n; ++i) perm[i] = i;
        bl.perm = perm.data();
        bl.max_prims = std::min(max_prims_in_node, 255);  // bvh.rs:222
        bl.method = split_method;
        bl.out.reserve(2 * (size_t)n);
        bl.order.reserve(n);
        bl.build(0, n, 3);  // up to 8 concurrent subtrees
    }

    size_t nn = bl.out.size();
    PbrtLinearBVHNode* nodes = (PbrtLinearBVHNode*)std::malloc(nn * sizeof(PbrtLinearBVHNode));
    int32_t* order = (int32_t*)std::malloc((size_t)n * sizeof(int32_t));
    if (!nodes || !order) {
        std::free(nodes);
        std::free(order);
        return PBRT_HIP_ERR_OOM;
    }
    std::memcpy(nodes, bl.out.data(), nn * sizeof(PbrtLinearBVHNode));
    std::memcpy(order, bl.order.data(), (size_t)n * sizeof(int32_t));
    *nodes_out = nodes;
    *n_nodes_out = (int32_t)nn;
    *prim_order_out = order;
    return PBRT_HIP_OK;
}

extern "C" int pbrt_hip_bvh_build(const float* positions, int32_t n_verts, const int32_t* indices, int32_t n_tris,
                                  int32_t max_prims_in_node, int32_t split_method, PbrtLinearBVHNode** nodes_out,
                                  int32_t* n_nodes_out, int32_t** prim_order_out) try {
    if (!nodes_out || !n_nodes_out || !prim_order_out) return PBRT_HIP_ERR_INVALID;
    *nodes_out = nullptr;
    *prim_order_out = nullptr;
    *n_nodes_out = 0;
    if (n_tris < 0 || n_verts < 0 || (n_tris > 0 && (!positions || !indices))) return PBRT_HIP_ERR_INVALID;
    if (split_method < 0 || split_method > 3) return PBRT_HIP_ERR_INVALID;
    if (n_tris == 0) return PBRT_HIP_OK;  // bvh.rs:228-230: empty aggregate, no nodes
    for (int64_t i = 0; i < 3 * (int64_t)n_tris; ++i)
        if (indices[i] < 0 || indices[i] >= n_verts) return PBRT_HIP_ERR_INVALID;

    std::vector<float> lo(3 * (size_t)n_tris), hi(3 * (size_t)n_tris);
    for (int32_t t = 0; t < n_tris; ++t) {
        // Triangle::world_bound, src/shapes/triangle.rs:175-180
        const float* a = positions + 3 * (size_t)indices[3 * (size_t)t];
        const float* b = positions + 3 * (size_t)indices[3 * (size_t)t + 1];
        const float* c = positions + 3 * (size_t)indices[3 * (size_t)t + 2];
        for (int k = 0; k < 3; ++k) {
            float mn = a[k] < b[k] ? a[k] : b[k];
            float mx = a[k] > b[k] ? a[k] : b[k];
            mn = mn < c[k] ? mn : c[k];
            mx = mx > c[k] ? mx : c[k];
            lo[3 * (size_t)t + k] = mn;
            hi[3 * (size_t)t + k] = mx;
        }
    }
    return build_from_boxes(lo, hi, n_tris, max_prims_in_node, split_method, nodes_out, n_nodes_out, prim_order_out);
}
PB_ABI_CATCH

extern "C" int pbrt_hip_bvh_build_boxes(const float* bounds_min, const float* bounds_max, int32_t n,
                                        int32_t max_prims_in_node, int32_t split_method, PbrtLinearBVHNode** nodes_out,
                                        int32_t* n_nodes_out, int32_t** prim_order_out) try {
    if (!nodes_out || !n_nodes_out || !prim_order_out) return PBRT_HIP_ERR_INVALID;
    *nodes_out = nullptr;
    *prim_order_out = nullptr;
    *n_nodes_out = 0;
    if (n < 0 || (n > 0 && (!bounds_min || !bounds_max))) return PBRT_HIP_ERR_INVALID;
    if (split_method < 0 || split_method > 3) return PBRT_HIP_ERR_INVALID;
    if (n == 0) return PBRT_HIP_OK;
    std::vector<float> lo(bounds_min, bounds_min + 3 * (size_t)n), hi(bounds_max, bounds_max + 3 * (size_t)n);
    return build_from_boxes(lo, hi, n, max_prims_in_node, split_method, nodes_out, n_nodes_out, prim_order_out);
}
PB_ABI_CATCH

// Transform * Bounds3f (src/core/transform.rs:568-607): union of the 8 transformed corners
extern "C" int pbrt_hip_instance_bounds(const float object_min[3], const float object_max[3],
                                        const PbrtInstance* instances, int32_t n_instances, float* bounds_min,
                                        float* bounds_max) try {
    if (!object_min || !object_max || n_instances < 0 || (n_instances > 0 && (!instances || !bounds_min || !bounds_max)))
        return PBRT_HIP_ERR_INVALID;
    for (int32_t i = 0; i < n_instances; ++i) {
        const float* m = instances[i].to_world;
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int c = 0; c < 8; ++c) {
            float x = (c & 1) ? object_max[0] : object_min[0];
            float y = (c & 2) ? object_max[1] : object_min[1];
            float z = (c & 4) ? object_max[2] : object_min[2];
            float p[3];
            for (int r = 0; r < 3; ++r) p[r] = m[4 * r] * x + m[4 * r + 1] * y + m[4 * r + 2] * z + m[4 * r + 3];
            float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
            if (wp != 1.0f)
                for (int r = 0; r < 3; ++r) p[r] = p[r] / wp;
            for (int r = 0; r < 3; ++r) {
                mn[r] = p[r] < mn[r] ? p[r] : mn[r];
                mx[r] = p[r] > mx[r] ? p[r] : mx[r];
            }
        }
        std::memcpy(bounds_min + 3 * (size_t)i, mn, 12);
        std::memcpy(bounds_max + 3 * (size_t)i, mx, 12);
    }
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" void pbrt_hip_free(void* p) { std::free(p); }

// host_wide.cpp — lays the 4-wide quantised records (wide_bvh.h) over a reference-order LinearBVHNode array.
//
// Host only. Two levels of the reference's tree (src/accelerators/bvh.rs:129-135, 774-811) become one record; the
// tree itself, its boxes and its visiting order are untouched (the 64-B child-pair records stay on the device beside
// these for the instrumented kernel and for rays the wide path does not take). Every rounding here is outward: a
// quantised box contains the float box it stands for in exact arithmetic, so the traversal kernel's filter can only
// pass more, never less (wide_bvh.h).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "host_wide.h"
#include "wide_build.h"

namespace pb {


namespace {

// runs fn(begin, end) over [0, n) on the host threads this process may use (PBRT_HIP_HOST_THREADS, at most 16);
// fn returns an error text or nullptr, the first error in index order wins
template <class F>
const char* parallel_chunks(size_t n, size_t min_chunk, F fn) {
    unsigned hw = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("PBRT_HIP_HOST_THREADS")) hw = (unsigned)std::max(1, std::atoi(e));
    size_t n_threads = std::min<size_t>(std::min<unsigned>(hw ? hw : 1u, 16u), std::max<size_t>(1, n / std::max<size_t>(1, min_chunk)));
    if (n_threads <= 1) return fn((size_t)0, n);
    std::vector<const char*> err(n_threads, nullptr);
    std::vector<std::thread> pool;
    for (size_t t = 0; t < n_threads; ++t)
        pool.emplace_back([&, t] { err[t] = fn(n * t / n_threads, n * (t + 1) / n_threads); });
    for (auto& th : pool) th.join();
    for (const char* e : err)
        if (e) return e;
    return nullptr;
}

}  // namespace

// Three passes. (1) the properties the exactness argument needs, checked per node (parallel). (2) the structure, one
// sequential walk over node headers: which binary node every record stands for, where its interior children and its leaf
// children's triangles start (the interior children of a record are consecutive records, the triangles of its leaf
// children consecutive wide-order positions). (3) the records themselves — bases, exponents, the 24 planes — and the
// triangle copies, independent per record (parallel). 1 M triangles: 230 ms -> a few tens of ms on 16 threads.
const char* build_wide_tree(const PbrtLinearBVHNode* nodes, int32_t n_nodes, const float* tris, int32_t n_slots,
                            WideTree* out, const WideBase& base_of) {
    if (n_nodes <= 0 || n_slots <= 0) return "empty tree";
    const bool opaque = tris == nullptr;
    // ---- (1) checked, not assumed ----
    if (const char* why = parallel_chunks((size_t)n_nodes, 1 << 15, [&](size_t i0, size_t i1) -> const char* {
            for (size_t i = i0; i < i1; ++i)
                if (int code = wide_check_node(nodes, (int32_t)i, tris)) return wide_error_text(code);
            return nullptr;
        }))
        return why;
    auto leaf_ref = [&](int64_t first_wide, int n) -> int32_t { return ~(int32_t)(((first_wide + base_of.tri) << 2) | (int64_t)(n - 1)); };
    if (!opaque) out->tris.assign((size_t)n_slots * 12, 0.0f);
    if (opaque) out->order.assign((size_t)n_slots, -1);
    out->leaf_boxes.assign((size_t)n_slots * 8, 0.0f);
    // copies a leaf's triangles to wide-order position `first` (and its exact box, where the kernel reads one)
    auto emit_leaf = [&](const PbrtLinearBVHNode& lf, int64_t first) {
        if (opaque) {
            for (int j = 0; j < lf.n_primitives; ++j) {  // the leaf's exact box at every one of its positions
                out->order[(size_t)(first + j)] = lf.offset + j + base_of.slot;
                float* b = &out->leaf_boxes[8 * (size_t)(first + j)];
                std::memcpy(b, lf.bounds_min, 12);
                std::memcpy(b + 4, lf.bounds_max, 12);
            }
            return;
        }
        wide_emit_leaf(lf, tris, base_of.slot, (size_t)first, out->tris.data(), out->leaf_boxes.data());
    };
    if (nodes[0].n_primitives > 0) {  // the whole tree is one leaf
        emit_leaf(nodes[0], 0);
        out->root_ref = leaf_ref(0, nodes[0].n_primitives);
        out->nodes.assign(kWideNodeDwords, 0u);
        out->n_records = 0;
        return nullptr;
    }
    // (the five lowest integers are lane codes of the traversal kernel, trace_wide.h: leaf references stay above them)
    if ((int64_t)n_slots + base_of.tri >= (1ll << 29) - 4) return "too many triangles for 30-bit wide references";
    // ---- (2) structure ----
    // A record's index is fixed when its parent is laid out (the parent's interior children take the next free indices,
    // contiguously); the ORDER in which records are laid out decides which records and triangles end up near each other:
    // first-in-first-out = breadth-first (levels contiguous), which is also what the device builder emits (depth-first order
    // measured no different for the traversal, profiles/r02_scene_rates.txt).
    struct Rec {
        int32_t node;         // the binary node this record stands for
        int32_t first_child;  // record index of its first interior child
        int64_t first_tri;    // wide-order position of its first leaf child's first triangle
    };
    std::vector<Rec> recs;
    recs.reserve((size_t)n_nodes / 3 + 16);
    recs.push_back(Rec{0, 0, 0});
    std::vector<size_t> work{0};
    size_t work_head = 0;
    int64_t tri_cursor = 0;
    while (work_head < work.size()) {
        const size_t w = work[work_head++];
        int32_t slot_node[4];
        int axis_c[2];
        wide_slots_of(nodes, recs[w].node, slot_node, axis_c);
        recs[w].first_child = (int32_t)recs.size();
        recs[w].first_tri = tri_cursor;
        for (int s = 0; s < 4; ++s) {
            if (slot_node[s] < 0) continue;
            const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
            if (ch.n_primitives > 0) {
                tri_cursor += ch.n_primitives;
            } else {
                recs.push_back(Rec{slot_node[s], 0, 0});
                work.push_back(recs.size() - 1);
            }
        }
        if (recs.size() >= (1u << 31)) return "too many records";
    }
    if (tri_cursor != n_slots) return "leaves do not cover the triangle list";
    out->nodes.assign(recs.size() * kWideNodeDwords, 0u);
    // ---- (3) the records ----
    std::atomic<long long> n_coarse{0};
    if (const char* why = parallel_chunks(recs.size(), 1 << 13, [&](size_t w0, size_t w1) -> const char* {
            long long coarse_here = 0;
            for (size_t w = w0; w < w1; ++w) {
                const int32_t i = recs[w].node;
                int32_t slot_node[4];
                int axis_c[2], tri_off[4];
                wide_slots_of(nodes, i, slot_node, axis_c);
                uint32_t rec[kWideNodeDwords] = {0};
                int coarse = 0;
                if (int code = wide_make_record(nodes, i, slot_node, axis_c, (uint32_t)((int64_t)recs[w].first_child + base_of.record),
                                                (uint32_t)(recs[w].first_tri + base_of.tri), rec, tri_off, &coarse))
                    return wide_error_text(code);
                coarse_here += coarse;
                for (int s = 0; s < 4; ++s)
                    if (slot_node[s] >= 0 && nodes[slot_node[s]].n_primitives > 0) emit_leaf(nodes[slot_node[s]], recs[w].first_tri + tri_off[s]);
                std::memcpy(&out->nodes[w * kWideNodeDwords], rec, sizeof(rec));
            }
            n_coarse += coarse_here;
            return nullptr;
        }))
        return why;
    if (n_coarse.load() * 8 > (long long)recs.size()) return wide_error_text(kWideErrCoarse);
    out->root_ref = base_of.record;
    out->n_records = (int)recs.size();
    {
        // a record with k children leaves at most k - 1 of them on the stack while the first is being walked
        std::vector<int> need(recs.size(), 0);
        for (size_t w = recs.size(); w-- > 0;) {
            const uint32_t* rec = &out->nodes[w * kWideNodeDwords];
            const uint32_t m[4] = {rec[0] & 0xffu, rec[1] & 0xffu, rec[2] & 0xffu, rec[3] >> 24};
            int k = 0, deepest = 0;
            for (int s = 0; s < 4; ++s) {
                if (m[s] == 0xffu) continue;
                ++k;
                if (m[s] & 0x80u) deepest = std::max(deepest, need[rec[10] - (uint32_t)base_of.record + (m[s] & 3u)]);
            }
            need[w] = k - 1 + deepest;
        }
        out->stack_need = need[0];
    }
    return nullptr;
}

}  // namespace pb

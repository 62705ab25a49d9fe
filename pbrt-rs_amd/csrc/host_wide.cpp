// host_wide.cpp — lays the 4-wide quantised records (wide_bvh.h) over a reference-order LinearBVHNode array.
//
// Host only. Two levels of the reference's tree (src/accelerators/bvh.rs:129-135, 774-811) become one record; the
// tree itself, its boxes and its visiting order are untouched (the 64-B child-pair records stay on the device beside
// these for the instrumented kernel and for rays the wide path does not take). Every rounding here is outward: a
// quantised box contains the float box it stands for in exact arithmetic, so the traversal kernel's filter can only
// pass more, never less (wide_bvh.h).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_wide.h"
#include "wide_bvh.h"

namespace pb {


namespace {

uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
float u2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// the largest float <= target whose low mantissa byte is `m` (the record keeps m[0..2] in the low bytes of base.xyz)
float base_with_byte(double target, uint32_t m) {
    float t = (float)target;
    if ((double)t > target) t = std::nextafter(t, -HUGE_VALF);
    uint32_t u = f2u(t);
    if (!(u & 0x80000000u)) {  // t >= +0: float order = integer order
        uint32_t cand = (u & ~0xffu) | m;
        if (cand <= u) return u2f(cand);
        if (u >= 0x100u) return u2f(cand - 0x100u);
        return u2f(0x80000000u | m);  // below the smallest positive step: a tiny negative value (or -0)
    }
    uint32_t mag = u & 0x7fffffffu, cand = (mag & ~0xffu) | m;  // negative: the magnitude has to be >= |t|
    if (cand < mag) cand += 0x100u;
    return u2f(0x80000000u | cand);
}
double next_up(double x) { return std::nextafter(x, HUGE_VAL); }
double next_down(double x) { return std::nextafter(x, -HUGE_VAL); }

}  // namespace

const char* build_wide_tree(const PbrtLinearBVHNode* nodes, int32_t n_nodes, const float* tris, int32_t n_slots,
                            WideTree* out, const WideBase& base_of) {
    if (n_nodes <= 0 || n_slots <= 0) return "empty tree";
    const bool opaque = tris == nullptr;
    // ---- the properties the exactness argument needs: checked, not assumed ----
    for (int32_t i = 0; i < n_nodes; ++i) {
        const PbrtLinearBVHNode& nd = nodes[i];
        for (int k = 0; k < 3; ++k) {
            if (!(std::fabs(nd.bounds_min[k]) <= kWideCoordLimit) || !(std::fabs(nd.bounds_max[k]) <= kWideCoordLimit))
                return "coordinates beyond 2^20";
            if (!(nd.bounds_min[k] <= nd.bounds_max[k])) return "inverted node box";
        }
        if (nd.n_primitives > 0) {
            if (nd.n_primitives > 4) return "leaf with more than 4 primitives";
            if (nd.n_primitives == 1 && !opaque) {
                // the single-triangle leaf box is recomputed from the vertices by the kernel: it has to BE the tight box
                const float* t = tris + 12 * (size_t)nd.offset;
                for (int k = 0; k < 3; ++k) {
                    float lo = std::fmin(t[k], std::fmin(t[3 + k], t[6 + k]));
                    float hi = std::fmax(t[k], std::fmax(t[3 + k], t[6 + k]));
                    if (lo != nd.bounds_min[k] || hi != nd.bounds_max[k]) return "single-triangle leaf box is not the triangle's bounds";
                }
            }
        } else {
            const PbrtLinearBVHNode* ch[2] = {&nodes[i + 1], &nodes[nd.offset]};
            for (int c = 0; c < 2; ++c)
                for (int k = 0; k < 3; ++k)
                    if (ch[c]->bounds_min[k] < nd.bounds_min[k] || ch[c]->bounds_max[k] > nd.bounds_max[k])
                        return "child box not inside its parent's";
        }
    }
    auto leaf_ref = [&](int64_t first_wide, int n) -> int32_t { return ~(int32_t)(((first_wide + base_of.tri) << 2) | (int64_t)(n - 1)); };
    if (!opaque) out->tris.assign((size_t)n_slots * 12, 0.0f);
    if (opaque) out->order.assign((size_t)n_slots, -1);
    out->leaf_boxes.assign((size_t)n_slots * 8, 0.0f);
    int64_t tri_cursor = 0;
    auto emit_leaf = [&](const PbrtLinearBVHNode& lf) -> int64_t {  // copies the leaf's triangles, returns their first wide position
        int64_t first = tri_cursor;
        for (int j = 0; j < lf.n_primitives; ++j) {
            if (opaque) {
                out->order[(size_t)tri_cursor] = lf.offset + j + base_of.slot;
                ++tri_cursor;
                continue;
            }
            const float* src = tris + 12 * (size_t)(lf.offset + j);
            float* dst = &out->tris[12 * (size_t)tri_cursor];
            std::memcpy(dst, src, 36);
            int32_t slot = lf.offset + j + base_of.slot;
            int32_t flags;
            std::memcpy(&flags, src + 11, 4);
            std::memcpy(dst + 9, &slot, 4);
            std::memcpy(dst + 10, &flags, 4);
            ++tri_cursor;
        }
        if (lf.n_primitives >= 2 || opaque) {
            float* b = &out->leaf_boxes[8 * (size_t)first];
            std::memcpy(b, lf.bounds_min, 12);
            std::memcpy(b + 4, lf.bounds_max, 12);
        }
        return first;
    };
    if (nodes[0].n_primitives > 0) {  // the whole tree is one leaf
        {
            const int64_t first = emit_leaf(nodes[0]);
            out->root_ref = leaf_ref(first, nodes[0].n_primitives);
        }
        out->nodes.assign(kWideNodeDwords, 0u);
        out->n_records = 0;
        return nullptr;
    }
    if ((int64_t)n_slots + base_of.tri >= (1ll << 29)) return "too many triangles for 30-bit wide references";
    // records in breadth-first order: the interior children of a record are consecutive
    // A record's index is fixed when its parent is laid out (the parent's interior children take the next free indices,
    // contiguously); the ORDER in which records are laid out decides which records and triangles end up near each other:
    // first-in-first-out = breadth-first (levels contiguous), last-in-first-out = depth-first (subtrees nearly contiguous).
    const char* order_env = std::getenv("PBRT_HIP_WIDE_ORDER");
    const bool depth_first = order_env && order_env[0] == 'd';
    std::vector<int32_t> roots;  // binary node of every record
    roots.push_back(0);
    std::vector<size_t> work{0};
    size_t work_head = 0;
    while (work_head < work.size()) {
        size_t w;
        if (depth_first) {
            w = work.back();
            work.pop_back();
        } else {
            w = work[work_head++];
        }
        const int32_t i = roots[w];
        const PbrtLinearBVHNode& nd = nodes[i];
        int32_t slot_node[4] = {-1, -1, -1, -1};
        const int32_t c[2] = {i + 1, nd.offset};
        int axis_c[2] = {0, 0};
        for (int j = 0; j < 2; ++j) {
            if (nodes[c[j]].n_primitives > 0) {
                slot_node[2 * j] = c[j];
            } else {
                slot_node[2 * j] = c[j] + 1;
                slot_node[2 * j + 1] = nodes[c[j]].offset;
                axis_c[j] = nodes[c[j]].axis;
            }
        }
        uint32_t rec[kWideNodeDwords] = {0};
        // children first: the bytes m[] are part of base.xyz
        uint32_t m[4] = {0xff, 0xff, 0xff, 0xff};
        int n_interior = 0, tri_off = 0;
        const int64_t first_child = (int64_t)roots.size();
        const int64_t first_tri = tri_cursor;
        for (int s = 0; s < 4; ++s) {
            if (slot_node[s] < 0) continue;
            const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
            if (ch.n_primitives > 0) {
                emit_leaf(ch);
                m[s] = (uint32_t)(tri_off << 2) | (uint32_t)(ch.n_primitives - 1);
                tri_off += ch.n_primitives;
            } else {
                m[s] = 0x80u | (uint32_t)n_interior;
                ++n_interior;
                roots.push_back(slot_node[s]);
                work.push_back(roots.size() - 1);
            }
        }
        float base[3];
        int e[3];
        for (int k = 0; k < 3; ++k) {
            // base: kWideSlack cells (and a little more) below the lower corner; cell 2^e: the smallest with 255 cells
            // reaching kWideSlack cells beyond the upper corner. The two depend on each other: settle in a few rounds.
            int ek = kExpMin;
            for (int round = 0; round < 8; ++round) {
                base[k] = base_with_byte((double)nd.bounds_min[k] - 2.0 * kWideSlack * std::ldexp(1.0, ek), k < 3 ? m[k] : 0);
                double extent = next_up((double)nd.bounds_max[k] - (double)base[k]);
                int need = kExpMin;
                while (need <= kExpMax && std::ldexp(255.0 - 2.0 * kWideSlack, need) < extent) ++need;
                if (need <= ek) break;
                ek = need;
            }
            if (ek > kExpMax) return "node extent beyond the exponent range";
            e[k] = ek;
        }
        uint32_t q[6] = {0, 0, 0, 0, 0, 0};
        for (int s = 0; s < 4; ++s) {
            uint32_t qlo[3] = {255, 255, 255}, qhi[3] = {0, 0, 0};  // empty slot: inverted (and masked out by m = 0xFF)
            if (slot_node[s] >= 0) {
                const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
                for (int k = 0; k < 3; ++k) {
                    double dlo = next_down((double)ch.bounds_min[k] - (double)base[k]);
                    double dhi = next_up((double)ch.bounds_max[k] - (double)base[k]);
                    double flo = std::floor(std::ldexp(dlo, -e[k]) - kWideSlack), fhi = std::ceil(std::ldexp(dhi, -e[k]) + kWideSlack);
                    if (flo < 0.0 || fhi > 255.0 || flo > fhi) return "quantisation out of range";  // excluded by the choice of base and e
                    qlo[k] = (uint32_t)flo;
                    qhi[k] = (uint32_t)fhi;
                }
            }
            for (int k = 0; k < 3; ++k) {
                q[2 * k] |= qlo[k] << (8 * s);
                q[2 * k + 1] |= qhi[k] << (8 * s);
            }
        }
        for (int k = 0; k < 3; ++k) rec[k] = f2u(base[k]);
        rec[3] = ((uint32_t)e[0] & 63u) | ((uint32_t)e[1] & 63u) << 6 | ((uint32_t)e[2] & 63u) << 12 |
                 (uint32_t)nd.axis << 18 | (uint32_t)axis_c[0] << 20 | (uint32_t)axis_c[1] << 22 | m[3] << 24;
        for (int k = 0; k < 6; ++k) rec[4 + k] = q[k];
        rec[10] = (uint32_t)(first_child + base_of.record);
        rec[11] = ~(uint32_t)((first_tri + base_of.tri) << 2);
        if (depth_first && n_interior > 1)  // walk the first interior child first, as BVHAccel's flattening does
            std::reverse(work.end() - n_interior, work.end());
        if (out->nodes.size() < roots.size() * kWideNodeDwords) out->nodes.resize(roots.size() * kWideNodeDwords, 0u);
        std::memcpy(&out->nodes[w * kWideNodeDwords], rec, sizeof(rec));
        if (roots.size() >= (1u << 31)) return "too many records";
    }
    out->root_ref = base_of.record;
    out->n_records = (int)roots.size();
    {
        // a record with k children leaves at most k - 1 of them on the stack while the first is being walked
        std::vector<int> need(roots.size(), 0);
        for (size_t w = roots.size(); w-- > 0;) {
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
    if (tri_cursor != n_slots) return "leaves do not cover the triangle list";
    return nullptr;
}

}  // namespace pb

// wide_check.cpp — host-side checker of the 4-wide quantised records (test infrastructure, built by tests/test_wide_host.py
// with g++; not part of the product library).
//
// Compiles the product's own builder (pbrt-rs_amd/csrc/host_wide.cpp) and the product's own filter arithmetic
// (pbrt-rs_amd/csrc/wide_bvh.h: wide_setup / wide_child_test, the functions the kernel calls) for the host and checks
// the three properties the exactness argument of wide_bvh.h rests on:
//   P1  conservative filter: whenever Bounds3f::intersect_p (src/core/geometry.rs:709-751) passes for a leaf box with
//       some t_max, the filter passes for every record child above that leaf, and its entry distance is <= the exact one;
//   P2  order: ranking the children of every record by the two dir_is_neg[axis] levels enumerates the leaves in the
//       order of BVHAccel::intersect's near-first walk (src/accelerators/bvh.rs:857-865), for all eight octants;
//   P3  structure: every leaf of the tree is referenced exactly once, with its triangles and (n >= 2) its exact box.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../pbrt-rs_amd/csrc/host_wide.cpp"

namespace {

constexpr float kMachEps = 5.9604644775390625e-08f;                       // f32::EPSILON / 2 (src/core/mod.rs:206)
constexpr float kGamma3 = 3.0f * kMachEps / (1.0f - 3.0f * kMachEps);   // src/core/pbrt.rs:88-91
constexpr float kSlabScale = 1.0f + 2.0f * kGamma3;

// Bounds3f::intersect_p(ray, inv_dir, dir_is_neg), geometry.rs:709-751 (D2 intended: the z far plane is widened too)
bool slab(const float lo[3], const float hi[3], const float o[3], const float id[3], float tmax_ray, float* entry) {
    const float* b[2] = {lo, hi};
    int neg[3] = {id[0] < 0.0f, id[1] < 0.0f, id[2] < 0.0f};
    float t_min = (b[neg[0]][0] - o[0]) * id[0];
    float t_max = (b[1 - neg[0]][0] - o[0]) * id[0];
    float ty_min = (b[neg[1]][1] - o[1]) * id[1];
    float ty_max = (b[1 - neg[1]][1] - o[1]) * id[1];
    t_max *= kSlabScale;
    ty_max *= kSlabScale;
    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    float tz_min = (b[neg[2]][2] - o[2]) * id[2];
    float tz_max = (b[1 - neg[2]][2] - o[2]) * id[2];
    tz_max *= kSlabScale;
    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    *entry = t_min;
    return (t_min < tmax_ray) && (t_max > 0.0f);
}

struct Ctx {
    const PbrtLinearBVHNode* nodes;
    const pb::WideTree* wt;
    const float* tris;
    // wide leaf position -> binary leaf node
    std::vector<int32_t> leaf_of_pos;
};

void decode(const uint32_t* rec, int s, bool* empty, bool* interior, int32_t* ref) {
    const uint32_t m = s < 3 ? (rec[s] & 0xffu) : (rec[3] >> 24);
    *empty = m == 0xffu;
    *interior = (m & 0x80u) != 0 && !*empty;
    *ref = *interior ? (int32_t)(rec[10] + (m & 3u)) : (int32_t)(rec[11] - m);
}

}  // namespace

extern "C" {

// slab test alone, for pinning this file's restatement against the oracle's from Python
int wide_check_slab(const float* lo, const float* hi, const float* o, const float* id, float tmax, float* entry) {
    return slab(lo, hi, o, id, tmax, entry) ? 1 : 0;
}

// Builds the records and checks P2 + P3. Returns 0 when they hold, a negative code otherwise; -100 = the builder
// declined (reason copied to `why`). stats: {records, stack_need, leaves, empty slots}.
int wide_check_structure(const PbrtLinearBVHNode* nodes, int32_t n_nodes, const float* tris, int32_t n_slots, int64_t* stats,
                         char* why, int why_len) {
    pb::WideTree wt;
    const char* e = pb::build_wide_tree(nodes, n_nodes, tris, n_slots, &wt);
    if (e) {
        std::snprintf(why, why_len, "%s", e);
        return -100;
    }
    stats[0] = wt.n_records;
    stats[1] = wt.stack_need;
    // the reference's leaf order per octant: near child first by dir_is_neg[axis]
    for (int oct = 0; oct < 8; ++oct) {
        std::vector<int32_t> ref_leaves;
        {
            std::vector<int32_t> st{0};
            while (!st.empty()) {
                int32_t i = st.back();
                st.pop_back();
                const PbrtLinearBVHNode& nd = nodes[i];
                if (nd.n_primitives > 0) {
                    ref_leaves.push_back(i);
                } else if ((oct >> nd.axis) & 1) {  // bvh.rs:857-865
                    st.push_back(i + 1);
                    st.push_back(nd.offset);
                } else {
                    st.push_back(nd.offset);
                    st.push_back(i + 1);
                }
            }
        }
        std::vector<int32_t> wide_leaves;  // as (first wide triangle, n) decoded back to the leaf's first slot
        std::vector<int64_t> seen_first;
        int64_t empties = 0;
        {
            std::vector<int32_t> st{wt.root_ref};
            while (!st.empty()) {
                int32_t ref = st.back();
                st.pop_back();
                if (ref < 0) {
                    int v = ~ref, n = (v & 3) + 1, first = v >> 2;
                    if (first < 0 || first + n > n_slots) return -2;
                    if (!tris) {  // opaque primitives: order[] maps the wide-order position back to the leaf slot
                        for (int j = 0; j < n; ++j)
                            if (wt.order[(size_t)first + j] != wt.order[(size_t)first] + j) return -3;
                        wide_leaves.push_back(wt.order[(size_t)first] * 8 + n);
                        continue;
                    }
                    int32_t slot0;
                    std::memcpy(&slot0, &wt.tris[12 * (size_t)first + 9], 4);
                    for (int j = 0; j < n; ++j) {  // P3: triangles copied in leaf order, slot ids consecutive
                        int32_t sl;
                        std::memcpy(&sl, &wt.tris[12 * (size_t)(first + j) + 9], 4);
                        if (sl != slot0 + j) return -3;
                        if (std::memcmp(&wt.tris[12 * (size_t)(first + j)], tris + 12 * (size_t)sl, 36) != 0) return -4;
                        if (std::memcmp(&wt.tris[12 * (size_t)(first + j) + 10], tris + 12 * (size_t)sl + 11, 4) != 0) return -5;
                    }
                    wide_leaves.push_back(slot0 * 8 + n);
                    continue;
                }
                if (ref >= wt.n_records) return -6;
                const uint32_t* rec = &wt.nodes[(size_t)ref * pb::kWideNodeDwords];
                const uint32_t dw3 = rec[3];
                const uint32_t f_root = (oct >> ((dw3 >> 18) & 3u)) & 1u, f_c0 = (oct >> ((dw3 >> 20) & 3u)) & 1u,
                               f_c1 = (oct >> ((dw3 >> 22) & 3u)) & 1u;
                const uint32_t x01 = (f_root << 1) | f_c0, x23 = (f_root << 1) | f_c1;
                const uint32_t rank[4] = {x01, x01 ^ 1u, x23 ^ 2u, x23 ^ 3u};
                int32_t by_rank[4] = {0, 0, 0, 0};
                bool have[4] = {false, false, false, false};
                for (int s = 0; s < 4; ++s) {
                    bool empty, interior;
                    int32_t r;
                    decode(rec, s, &empty, &interior, &r);
                    if (empty) {
                        ++empties;
                        continue;
                    }
                    if (have[rank[s]]) return -7;
                    have[rank[s]] = true;
                    by_rank[rank[s]] = r;
                }
                for (int k = 3; k >= 0; --k)
                    if (have[k]) st.push_back(by_rank[k]);
            }
        }
        if (wide_leaves.size() != ref_leaves.size()) return -8;
        for (size_t k = 0; k < ref_leaves.size(); ++k) {
            const PbrtLinearBVHNode& lf = nodes[ref_leaves[k]];
            if (wide_leaves[k] != lf.offset * 8 + lf.n_primitives) return -9;  // P2
        }
        stats[2] = (int64_t)ref_leaves.size();
        stats[3] = empties;
    }
    // P4: every quantised plane lies at least kWideSlack cells outside the float box of its child (checked in double:
    // base, q * cell and the coordinates are all dyadic and within 2^53 of each other here), inside [0, 255]
    {
        std::vector<int32_t> st{wt.root_ref};
        std::vector<int32_t> bin{0};  // binary node of each record on the stack
        while (!st.empty()) {
            int32_t ref = st.back(), bnode = bin.back();
            st.pop_back();
            bin.pop_back();
            if (ref < 0) continue;
            const uint32_t* rec = &wt.nodes[(size_t)ref * pb::kWideNodeDwords];
            const PbrtLinearBVHNode& nd = nodes[bnode];
            const int32_t c[2] = {bnode + 1, nd.offset};
            int32_t slot_node[4] = {-1, -1, -1, -1};
            for (int j = 0; j < 2; ++j) {
                if (nodes[c[j]].n_primitives > 0) {
                    slot_node[2 * j] = c[j];
                } else {
                    slot_node[2 * j] = c[j] + 1;
                    slot_node[2 * j + 1] = nodes[c[j]].offset;
                }
            }
            for (int s = 0; s < 4; ++s) {
                bool empty, interior;
                int32_t r;
                decode(rec, s, &empty, &interior, &r);
                if (empty != (slot_node[s] < 0)) return -12;
                if (empty) continue;
                const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
                if (interior != (ch.n_primitives == 0)) return -13;
                for (int k = 0; k < 3; ++k) {
                    float base;
                    std::memcpy(&base, &rec[k], 4);
                    int e = ((int)(rec[3] << (26 - 6 * k))) >> 26;
                    double cell = std::ldexp(1.0, e);
                    double lo = (double)base + (double)((rec[4 + 2 * k] >> (8 * s)) & 0xffu) * cell;
                    double hi = (double)base + (double)((rec[5 + 2 * k] >> (8 * s)) & 0xffu) * cell;
                    if (!(lo <= (double)ch.bounds_min[k] - 0.999 * pb::kWideSlack * cell)) return -14;
                    if (!(hi >= (double)ch.bounds_max[k] + 0.999 * pb::kWideSlack * cell)) return -15;
                }
                if (interior) {
                    st.push_back(r);
                    bin.push_back(slot_node[s]);
                }
            }
        }
    }
    // P3: exact boxes of the leaves with n >= 2
    {
        std::vector<int32_t> st{wt.root_ref};
        // map slot0 -> leaf node
        std::vector<int32_t> leaf_node(n_slots, -1);
        for (int32_t i = 0; i < n_nodes; ++i)
            if (nodes[i].n_primitives > 0) leaf_node[nodes[i].offset] = i;
        while (!st.empty()) {
            int32_t ref = st.back();
            st.pop_back();
            if (ref < 0) {
                int v = ~ref, n = (v & 3) + 1, first = v >> 2;
                int32_t slot0;
                if (tris) std::memcpy(&slot0, &wt.tris[12 * (size_t)first + 9], 4);
                else slot0 = wt.order[(size_t)first];
                const PbrtLinearBVHNode& lf = nodes[leaf_node[slot0]];
                if (lf.n_primitives != n) return -10;
                if ((n >= 2 || !tris) && (std::memcmp(&wt.leaf_boxes[8 * (size_t)first], lf.bounds_min, 12) != 0 ||
                               std::memcmp(&wt.leaf_boxes[8 * (size_t)first + 4], lf.bounds_max, 12) != 0))
                    return -11;
                continue;
            }
            const uint32_t* rec = &wt.nodes[(size_t)ref * pb::kWideNodeDwords];
            for (int s = 0; s < 4; ++s) {
                bool empty, interior;
                int32_t r;
                decode(rec, s, &empty, &interior, &r);
                if (!empty) st.push_back(r);
            }
        }
    }
    return 0;
}

}  // extern "C"

// P1 over every (ray, leaf) pair of the tree: rays = n x {o.xyz, d.xyz, t_max}. For each leaf whose exact box passes
// (with +inf, with the ray's t_max and with the smallest t_max that still lets it pass) every record child on the way
// down must pass the filter with the same t_max and report an entry distance <= the exact one. Returns the number of
// violations; counts: {covered rays, exact passes checked, filter passes at leaf children, exact passes at leaf children}.
namespace {
struct Anc {
    pb::WideSetup ws;
    uint32_t nq[3], fq[3];
    int s;
};
struct FilterWalk {
    const PbrtLinearBVHNode* nodes;
    const pb::WideTree* wt;
    const std::vector<int32_t>* leaf_node;
    float o[3], id[3], tmax;
    bool neg[3];
    std::vector<Anc> path;
    int64_t bad = 0;
    int64_t* counts;
    void visit(int32_t ref) {
        if (ref < 0) {
            int v = ~ref, first = v >> 2;
            int32_t slot0;
            if (wt->order.empty()) std::memcpy(&slot0, &wt->tris[12 * (size_t)first + 9], 4);
            else slot0 = wt->order[(size_t)first];  // opaque primitives
            const PbrtLinearBVHNode& lf = nodes[(*leaf_node)[slot0]];
            float entry_inf;
            bool pass_inf = slab(lf.bounds_min, lf.bounds_max, o, id, INFINITY, &entry_inf);
            if (!path.empty()) {  // how much looser the filter is at the leaf's own record child
                const Anc& a = path.back();
                float eq;
                if (pb::wide_child_test(a.ws, a.nq[0], a.nq[1], a.nq[2], a.fq[0], a.fq[1], a.fq[2], a.s, INFINITY, &eq)) counts[2] += 1;
                if (pass_inf) counts[3] += 1;
            }
            if (!pass_inf) return;
            const float tm[3] = {INFINITY, tmax, std::nextafter(entry_inf, INFINITY)};
            for (int k = 0; k < 3; ++k) {
                float entry;
                // a negative t_max is outside the claim: Triangle::intersect_test (triangle.rs:127-131) cannot return a hit
                // for it (t_scaled would have to be > 0 and <= t_max * det < 0), so which leaves are looked at is immaterial
                if (tm[k] < 0.0f) continue;
                if (!slab(lf.bounds_min, lf.bounds_max, o, id, tm[k], &entry)) continue;
                counts[1] += 1;
                for (const Anc& a : path) {
                    float eq;
                    bool p = pb::wide_child_test(a.ws, a.nq[0], a.nq[1], a.nq[2], a.fq[0], a.fq[1], a.fq[2], a.s, tm[k], &eq);
                    if (!p || !(eq <= entry) || !(eq < tm[k])) {
                        if (bad < 5 && std::getenv("WIDE_CHECK_VERBOSE"))
                            std::fprintf(stderr, "violation: pass %d eq %.9g exact entry %.9g tmax %.9g | o %.9g %.9g %.9g id %.9g %.9g %.9g | leaf [%.9g %.9g %.9g]-[%.9g %.9g %.9g] S %.9g %.9g %.9g An %.9g %.9g %.9g Af %.9g %.9g %.9g slot %d\n",
                                         (int)p, eq, entry, tm[k], o[0], o[1], o[2], id[0], id[1], id[2], lf.bounds_min[0], lf.bounds_min[1],
                                         lf.bounds_min[2], lf.bounds_max[0], lf.bounds_max[1], lf.bounds_max[2], a.ws.Sx, a.ws.Sy, a.ws.Sz,
                                         a.ws.Anx, a.ws.Any, a.ws.Anz, a.ws.Afx, a.ws.Afy, a.ws.Afz, a.s);
                        ++bad;
                    }
                }
            }
            return;
        }
        const uint32_t* rec = &wt->nodes[(size_t)ref * pb::kWideNodeDwords];
        Anc a;
        a.ws = pb::wide_setup(rec[0], rec[1], rec[2], rec[3], o[0], o[1], o[2], id[0], id[1], id[2]);
        for (int k = 0; k < 3; ++k) {
            a.nq[k] = neg[k] ? rec[5 + 2 * k] : rec[4 + 2 * k];
            a.fq[k] = neg[k] ? rec[4 + 2 * k] : rec[5 + 2 * k];
        }
        for (int s = 0; s < 4; ++s) {
            bool empty, interior;
            int32_t r;
            decode(rec, s, &empty, &interior, &r);
            if (empty) continue;
            a.s = s;
            path.push_back(a);
            visit(r);
            path.pop_back();
        }
    }
};
}  // namespace

extern "C" int64_t wide_check_filter(const PbrtLinearBVHNode* nodes, int32_t n_nodes, const float* tris, int32_t n_slots,
                                     const float* rays, int32_t n_rays, int64_t* counts) {
    pb::WideTree wt;
    if (pb::build_wide_tree(nodes, n_nodes, tris, n_slots, &wt)) return -100;
    std::vector<int32_t> leaf_node(n_slots, -1);
    for (int32_t i = 0; i < n_nodes; ++i)
        if (nodes[i].n_primitives > 0) leaf_node[nodes[i].offset] = i;
    FilterWalk w;
    w.nodes = nodes;
    w.wt = &wt;
    w.leaf_node = &leaf_node;
    w.counts = counts;
    for (int32_t ri = 0; ri < n_rays; ++ri) {
        const float* ry = rays + 7 * (size_t)ri;
        for (int k = 0; k < 3; ++k) {
            w.o[k] = ry[k];
            w.id[k] = 1.0f / ry[3 + k];
            w.neg[k] = w.id[k] < 0.0f;
        }
        w.tmax = ry[6];
        if (!pb::wide_ray_covered(w.o[0], w.o[1], w.o[2], w.id[0], w.id[1], w.id[2])) continue;
        counts[0] += 1;
        w.visit(wt.root_ref);
    }
    return w.bad;
}

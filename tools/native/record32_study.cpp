// record32_study.cpp — host costing of a 32-byte, two-load 4-wide record (VERDICT r4 "Next round" 3). Not product code.
//
// The shipped record (wide_bvh.h) is 48 B = three global_load_dwordx4 per lane and step: 16 B of FRAME (base.xyz, three cell
// exponents, split axes), 24 B of 8-bit planes, two child bases. The candidate drops the stored frame: a record's planes are
// quantised relative to the PARENT's quantised box of that child —
//     origin' = origin + q_lo * 2^e      (a multiple of 2^e: exact in float as long as |origin| / 2^e < 2^24)
//     e'      = e - k,  k = the largest k <= K_MAX with (q_hi - q_lo) * 2^k <= 255
// — so 24 B of planes + 4 B of descriptors / axes + ONE child base (records and triangles interleaved in one array of 16-B
// units) = 32 B = two loads. The frame then lives in the traversal state: registers for the record a lane is stepping, and on
// the STACK for the children it defers (modelled below: one 24-B frame entry per step that defers anything + one 8-B entry
// per deferred child, against one 8-B entry per deferred child today).
//
// This program builds both layouts over the same reference-order tree with the product's own builder arithmetic
// (host_wide.cpp, wide_build.h), walks them with the same rays the way trace_wide.h does (rank order = the reference's
// near-first order, first hit child next, the others deferred last-first with their filter entry distance and re-checked
// against the shrunk t_max at the pop, leaves confirmed with the exact slab test, triangles with a double-precision
// intersection that shrinks t_max) and counts, per ray class: record steps, filter passes that the exact child box would
// have failed (false positives), leaf candidates, triangle tests, deferrals, pops, and the stack's high-water mark in 8-B slots.
//
// usage: record32_study nodes.bin tris.bin rays.bin   (written by tools/record32_study.py)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../pbrt-rs_amd/csrc/host_wide.cpp"

namespace {

constexpr int kMaxShift = 3;  // K_MAX: two bits per axis in a deferred child's stack entry
constexpr float kMachEps = 5.9604644775390625e-08f;
constexpr float kGamma3 = 3.0f * kMachEps / (1.0f - 3.0f * kMachEps);
constexpr float kSlabScale = 1.0f + 2.0f * kGamma3;

template <class T>
std::vector<T> read_file(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) {
        std::fprintf(stderr, "cannot open %s\n", path);
        std::exit(2);
    }
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T));
    if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) std::exit(2);
    std::fclose(f);
    return v;
}

// Bounds3f::intersect_p (geometry.rs:709-751)
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

// ray / triangle in double (Moeller-Trumbore): the statistics need t_max to shrink as the reference's does, not its last bit
bool tri_hit(const float* v, const double o[3], const double d[3], double tmax, double* t_out) {
    double e1[3], e2[3], p[3], s[3], q[3];
    for (int k = 0; k < 3; ++k) {
        e1[k] = (double)v[3 + k] - v[k];
        e2[k] = (double)v[6 + k] - v[k];
    }
    p[0] = d[1] * e2[2] - d[2] * e2[1];
    p[1] = d[2] * e2[0] - d[0] * e2[2];
    p[2] = d[0] * e2[1] - d[1] * e2[0];
    double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det == 0.0) return false;
    double inv = 1.0 / det;
    for (int k = 0; k < 3; ++k) s[k] = o[k] - v[k];
    double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    if (u < 0.0 || u > 1.0) return false;
    q[0] = s[1] * e1[2] - s[2] * e1[1];
    q[1] = s[2] * e1[0] - s[0] * e1[2];
    q[2] = s[0] * e1[1] - s[1] * e1[0];
    double w = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
    if (w < 0.0 || u + w > 1.0) return false;
    double t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    if (!(t > 0.0) || !(t < tmax)) return false;
    *t_out = t;
    return true;
}

// ---- the candidate layout: per record the 24 plane bytes, descriptors, child references; frames are NOT stored ----
struct Rec32 {
    uint8_t qlo[3][4], qhi[3][4];
    int32_t child[4];  // >= 0 record, < 0 ~leaf node index, INT32_MIN empty
    int32_t bnode[4];  // binary node of the slot (for the false-positive count)
    uint8_t axis_root, axis_c0, axis_c1;
};
struct Frame {
    double o[3];
    int e[3];
};
struct Layout32 {
    std::vector<Rec32> recs;
    Frame root;
    int64_t clamped_shift = 0, coarse_children = 0, inexact_origins = 0;
};
constexpr int32_t kEmpty = INT32_MIN;

Frame child_frame(const Frame& f, const Rec32& r, int s, int* shifts = nullptr) {
    Frame c;
    for (int k = 0; k < 3; ++k) {
        int d = (int)r.qhi[k][s] - (int)r.qlo[k][s];
        if (d < 1) d = 1;
        int sh = 0;
        while (sh < kMaxShift && (d << (sh + 1)) <= 255) ++sh;
        c.o[k] = f.o[k] + (double)r.qlo[k][s] * std::ldexp(1.0, f.e[k]);
        c.e[k] = std::max(f.e[k] - sh, pb::kExpMin);
        if (shifts) shifts[k] = sh;
    }
    return c;
}

bool build32(const PbrtLinearBVHNode* nodes, Layout32* out) {
    // root frame: as the shipped builder's (cell = smallest power of two with 255 cells over the extent), origin on the cell grid
    for (int k = 0; k < 3; ++k) {
        int e = pb::wb_cell_exponent((double)nodes[0].bounds_max[k] - (double)nodes[0].bounds_min[k] + std::ldexp(4.0 * pb::kWideSlack, pb::kExpMin));
        for (;; ++e) {
            double cell = std::ldexp(1.0, e);
            double o = std::floor(((double)nodes[0].bounds_min[k] - 2.0 * pb::kWideSlack * cell) / cell) * cell;
            if (std::ceil(((double)nodes[0].bounds_max[k] - o) / cell + pb::kWideSlack) <= 255.0) {
                out->root.o[k] = o;
                out->root.e[k] = e;
                break;
            }
        }
    }
    struct Work {
        int32_t node, rec;
        Frame f;
    };
    std::vector<Work> work;
    out->recs.emplace_back();
    work.push_back({0, 0, out->root});
    while (!work.empty()) {
        Work w = work.back();
        work.pop_back();
        int32_t slot_node[4];
        int axis_c[2];
        pb::wide_slots_of(nodes, w.node, slot_node, axis_c);
        Rec32 r;
        std::memset(&r, 0, sizeof(r));
        r.axis_root = nodes[w.node].axis;
        r.axis_c0 = axis_c[0];
        r.axis_c1 = axis_c[1];
        for (int s = 0; s < 4; ++s) {
            r.child[s] = kEmpty;
            r.bnode[s] = slot_node[s];
            if (slot_node[s] < 0) continue;
            const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
            for (int k = 0; k < 3; ++k) {
                double cell = std::ldexp(1.0, w.f.e[k]);
                double flo = std::floor(((double)ch.bounds_min[k] - w.f.o[k]) / cell - pb::kWideSlack);
                double fhi = std::ceil(((double)ch.bounds_max[k] - w.f.o[k]) / cell + pb::kWideSlack);
                if (flo < 0.0 || fhi > 255.0 || flo > fhi) return false;
                r.qlo[k][s] = (uint8_t)flo;
                r.qhi[k][s] = (uint8_t)fhi;
            }
            r.child[s] = ch.n_primitives > 0 ? ~slot_node[s] : 0;  // record index patched below
        }
        for (int s = 0; s < 4; ++s) {
            if (slot_node[s] < 0 || nodes[slot_node[s]].n_primitives > 0) continue;
            int sh[3];
            Frame cf = child_frame(w.f, r, s, sh);
            for (int k = 0; k < 3; ++k) {
                int d = std::max(1, (int)r.qhi[k][s] - (int)r.qlo[k][s]);
                if ((d << (sh[k] + 1)) <= 255 && cf.e[k] > pb::kExpMin) out->clamped_shift += 1;  // K_MAX (not the range) held it back
                if (std::fabs(cf.o[k]) / std::ldexp(1.0, cf.e[k]) >= 16777216.0) out->inexact_origins += 1;
            }
            int32_t idx = (int32_t)out->recs.size();
            out->recs.emplace_back();
            r.child[s] = idx;
            work.push_back({slot_node[s], idx, cf});
        }
        out->recs[(size_t)w.rec] = r;
    }
    return true;
}

struct Counts {
    int64_t rays = 0, steps = 0, child_tests = 0, child_pass = 0, child_exact_pass = 0, leaf_cand = 0, leaf_pass = 0, tri_tests = 0, defers = 0,
            defer_steps = 0, pops = 0, pops_culled = 0, hits = 0;
    int64_t depth_hist[64] = {0};      // high-water mark of the stack per ray, in 8-B slots
    int64_t step_depth_over[4] = {0};  // record steps taken with more than {6, 8, 10, 12} slots in use
    void add_depth(int d) { depth_hist[std::min(d, 63)] += 1; }
};

struct Ray {
    float o[3], id[3];
    double od[3], dd[3];
    float tmax;
    bool any;
};

struct Walker {
    const PbrtLinearBVHNode* nodes;
    const float* tris;  // 12 floats per leaf slot
    // shipped layout
    const pb::WideTree* wt = nullptr;
    std::vector<int32_t> leaf_of_first;  // wide-order first triangle -> binary leaf node
    std::vector<int32_t> rec_bnode;      // shipped record -> the binary node it stands for
    // candidate layout
    const Layout32* l32 = nullptr;

    struct Entry {
        int32_t ref;
        float tn;
        Frame f;     // candidate only
        int slots;   // stack slots this entry accounts for when popped (candidate: 1, +3 for the last sibling of a frame entry)
    };

    bool leaf(int32_t leaf_node, Ray& r, Counts& c) {
        const PbrtLinearBVHNode& lf = nodes[leaf_node];
        c.leaf_cand += 1;
        float entry;
        if (!slab(lf.bounds_min, lf.bounds_max, r.o, r.id, r.tmax, &entry)) return false;
        c.leaf_pass += 1;
        for (int j = 0; j < lf.n_primitives; ++j) {
            c.tri_tests += 1;
            double t;
            if (tri_hit(tris + 12 * (size_t)(lf.offset + j), r.od, r.dd, (double)r.tmax, &t)) {
                r.tmax = (float)t;
                c.hits += 1;
                if (r.any) return true;
            }
        }
        return false;
    }

    // plane distances of one child in exact arithmetic (the kernel's float pads are 2^-19 relative: far below a cell)
    static bool child_test(const double o[3], const int e[3], const uint8_t qlo[3], const uint8_t qhi[3], const Ray& r, float* tn_out) {
        double t0 = 0.0, t1 = (double)r.tmax;
        double tn = -INFINITY;
        for (int k = 0; k < 3; ++k) {
            double cell = std::ldexp(1.0, e[k]);
            double lo = (o[k] + qlo[k] * cell - r.od[k]) * (double)r.id[k], hi = (o[k] + qhi[k] * cell - r.od[k]) * (double)r.id[k];
            if (lo > hi) std::swap(lo, hi);
            hi *= (double)kSlabScale;
            tn = std::max(tn, lo);
            t1 = std::min(t1, hi);
        }
        t0 = std::max(tn, 0.0);
        *tn_out = (float)tn;
        return t0 <= t1;
    }

    void walk(Ray r, Counts& c, bool candidate) {
        std::vector<Entry> st;
        int slots = 0, high = 0;
        c.rays += 1;
        Frame f = candidate ? l32->root : Frame();
        int32_t cur = candidate ? 0 : wt->root_ref;
        bool have = true;
        for (;;) {
            if (!have) {
                if (st.empty()) break;
                Entry e = st.back();
                st.pop_back();
                slots -= e.slots;
                c.pops += 1;
                if (!(e.tn < r.tmax)) {  // the reference's test at the pop: the far child's entry distance against the current t_max
                    c.pops_culled += 1;
                    continue;
                }
                cur = e.ref;
                f = e.f;
                have = true;
            }
            if (cur < 0) {  // leaf
                int32_t leaf_node = candidate ? ~cur : leaf_of_first[(size_t)((~cur) >> 2)];
                if (leaf(leaf_node, r, c) && r.any) break;
                have = false;
                continue;
            }
            c.steps += 1;
            for (int i = 0; i < 4; ++i)
                if (slots > 6 + 2 * i) c.step_depth_over[i] += 1;
            // children in slot order, then ranked as the kernel ranks them
            int32_t ref[4];
            float tn[4];
            bool hit[4];
            Frame cf[4];
            uint32_t ax_root, ax_c0, ax_c1;
            if (candidate) {
                const Rec32& rec = l32->recs[(size_t)cur];
                ax_root = rec.axis_root, ax_c0 = rec.axis_c0, ax_c1 = rec.axis_c1;
                for (int s = 0; s < 4; ++s) {
                    ref[s] = rec.child[s];
                    hit[s] = false;
                    if (ref[s] == kEmpty) continue;
                    uint8_t ql[3] = {rec.qlo[0][s], rec.qlo[1][s], rec.qlo[2][s]}, qh[3] = {rec.qhi[0][s], rec.qhi[1][s], rec.qhi[2][s]};
                    hit[s] = child_test(f.o, f.e, ql, qh, r, &tn[s]);
                    c.child_tests += 1;
                    if (hit[s]) {
                        c.child_pass += 1;
                        float en;
                        const PbrtLinearBVHNode& ch = nodes[rec.bnode[s]];
                        if (slab(ch.bounds_min, ch.bounds_max, r.o, r.id, r.tmax, &en)) c.child_exact_pass += 1;
                        if (ref[s] >= 0) cf[s] = child_frame(f, rec, s);
                    }
                }
            } else {
                const uint32_t* rec = &wt->nodes[(size_t)cur * pb::kWideNodeDwords];
                const uint32_t dw3 = rec[3];
                ax_root = (dw3 >> 18) & 3u, ax_c0 = (dw3 >> 20) & 3u, ax_c1 = (dw3 >> 22) & 3u;
                double o[3];
                int e[3];
                for (int k = 0; k < 3; ++k) {
                    float b;
                    std::memcpy(&b, &rec[k], 4);
                    o[k] = b;
                    e[k] = ((int)(dw3 << (26 - 6 * k))) >> 26;
                }
                // binary nodes of the slots, for the false-positive count
                int32_t slot_node[4];
                int axis_c[2];
                pb::wide_slots_of(nodes, rec_bnode[(size_t)cur], slot_node, axis_c);
                for (int s = 0; s < 4; ++s) {
                    const uint32_t m = s < 3 ? (rec[s] & 0xffu) : (rec[3] >> 24);
                    hit[s] = false;
                    ref[s] = kEmpty;
                    if (m == 0xffu) continue;
                    ref[s] = (m & 0x80u) ? (int32_t)(rec[10] + (m & 3u)) : (int32_t)(rec[11] - m);
                    uint8_t ql[3], qh[3];
                    for (int k = 0; k < 3; ++k) {
                        ql[k] = (rec[4 + 2 * k] >> (8 * s)) & 0xffu;
                        qh[k] = (rec[5 + 2 * k] >> (8 * s)) & 0xffu;
                    }
                    hit[s] = child_test(o, e, ql, qh, r, &tn[s]);
                    c.child_tests += 1;
                    if (hit[s]) {
                        c.child_pass += 1;
                        float en;
                        const PbrtLinearBVHNode& ch = nodes[slot_node[s]];
                        if (slab(ch.bounds_min, ch.bounds_max, r.o, r.id, r.tmax, &en)) c.child_exact_pass += 1;
                    }
                }
            }
            const uint32_t f_root = r.id[ax_root] < 0.0f, f_c0 = r.id[ax_c0] < 0.0f, f_c1 = r.id[ax_c1] < 0.0f;
            const uint32_t x01 = (f_root << 1) | f_c0, x23 = (f_root << 1) | f_c1;
            const uint32_t rank[4] = {x01, x01 ^ 1u, x23 ^ 2u, x23 ^ 3u};
            int by_rank[4] = {-1, -1, -1, -1};
            for (int s = 0; s < 4; ++s)
                if (ref[s] != kEmpty && hit[s]) by_rank[rank[s]] = s;
            int first = -1, n_defer = 0;
            for (int k = 0; k < 4; ++k)
                if (by_rank[k] >= 0) {
                    if (first < 0) first = by_rank[k];
                    else ++n_defer;
                }
            if (n_defer > 0) {
                c.defer_steps += 1;
                c.defers += n_defer;
                bool last = true;  // the deepest sibling releases the frame entry
                for (int k = 3; k >= 0; --k) {
                    int s = by_rank[k];
                    if (s < 0 || s == first) continue;
                    Entry e;
                    e.ref = ref[s];
                    e.tn = tn[s];
                    e.f = candidate ? cf[s] : Frame();
                    e.slots = 1 + ((candidate && last) ? 3 : 0);
                    last = false;
                    st.push_back(e);
                }
                slots += n_defer + (candidate ? 3 : 0);
                high = std::max(high, slots);
            }
            if (first < 0) {
                have = false;
                continue;
            }
            cur = ref[first];
            if (candidate && cur >= 0) f = cf[first];
        }
        c.add_depth(high);
    }
};

void report(const char* name, const Counts& a, const Counts& b) {
    auto per = [](int64_t v, int64_t n) { return n ? (double)v / (double)n : 0.0; };
    std::printf("%-28s %9lld rays | 48-B records: %6.2f steps/ray, filter passes %5.3f x exact (false positives %4.1f %% of passes), %5.2f leaf candidates, %5.2f triangle tests, "
                "%5.2f deferred, %5.2f pops (%4.1f %% culled at the pop)\n",
                name, (long long)a.rays, per(a.steps, a.rays), per(a.child_pass, a.child_exact_pass), 100.0 * (1.0 - per(a.child_exact_pass, a.child_pass)),
                per(a.leaf_cand, a.rays),
                per(a.tri_tests, a.rays), per(a.defers, a.rays), per(a.pops, a.rays), 100.0 * per(a.pops_culled, a.pops));
    std::printf("%-28s %9s      | 32-B records: %6.2f steps/ray (%+5.2f %%), filter passes %5.3f x exact (false positives %4.1f %% of passes), %5.2f leaf candidates, "
                "%5.2f triangle tests, %5.2f deferred in %5.2f steps, %5.2f pops (%4.1f %% culled)\n",
                "", "", per(b.steps, b.rays), 100.0 * (per(b.steps, b.rays) / per(a.steps, a.rays) - 1.0), per(b.child_pass, b.child_exact_pass),
                100.0 * (1.0 - per(b.child_exact_pass, b.child_pass)), per(b.leaf_cand, b.rays), per(b.tri_tests, b.rays), per(b.defers, b.rays),
                per(b.defer_steps, b.rays), per(b.pops, b.rays), 100.0 * per(b.pops_culled, b.pops));
    // wave-level memory instructions per ray: 3 (2) per record step, 3 per triangle, 2 per multi-triangle leaf box (counted as every leaf: upper bound)
    double vm48 = 3.0 * per(a.steps, a.rays) + 3.0 * per(a.tri_tests, a.rays) + 2.0 * per(a.leaf_cand, a.rays);
    double vm32 = 2.0 * per(b.steps, b.rays) + 3.0 * per(b.tri_tests, b.rays) + 2.0 * per(b.leaf_cand, b.rays);
    std::printf("%-28s %9s      | lane-level 16-B requests per ray: %6.1f -> %6.1f (%+5.1f %%)\n", "", "", vm48, vm32, 100.0 * (vm32 / vm48 - 1.0));
    auto depth_line = [&](const char* what, const Counts& c) {
        int64_t tot = 0, cum = 0;
        for (int i = 0; i < 64; ++i) tot += c.depth_hist[i];
        int p50 = -1, p90 = -1, p99 = -1, mx = 0;
        for (int i = 0; i < 64; ++i) {
            cum += c.depth_hist[i];
            if (c.depth_hist[i]) mx = i;
            if (p50 < 0 && cum * 2 >= tot) p50 = i;
            if (p90 < 0 && cum * 10 >= tot * 9) p90 = i;
            if (p99 < 0 && cum * 100 >= tot * 99) p99 = i;
        }
        std::printf("%-28s %9s      | stack high-water mark per ray, %s: median %d, 90 %% %d, 99 %% %d, max %d slots of 8 B; record steps taken above "
                    "6 / 8 / 10 / 12 slots: %4.1f / %4.1f / %4.1f / %4.1f %%\n",
                    "", "", what, p50, p90, p99, mx, 100.0 * per(c.step_depth_over[0], c.steps), 100.0 * per(c.step_depth_over[1], c.steps),
                    100.0 * per(c.step_depth_over[2], c.steps), 100.0 * per(c.step_depth_over[3], c.steps));
    };
    depth_line("48-B (8 B per deferred child)", a);
    depth_line("32-B (24-B frame entry per deferring step + 8 B per child)", b);
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) return 1;
    auto nodes = read_file<PbrtLinearBVHNode>(argv[1]);
    auto tris = read_file<float>(argv[2]);
    auto rays = read_file<float>(argv[3]);  // 8 floats per ray: o.xyz d.xyz t_max class (0 camera closest, 1 bounce closest, 2 shadow any, 3 MIS any)
    const int32_t n_nodes = (int32_t)nodes.size(), n_slots = (int32_t)(tris.size() / 12);
    pb::WideTree wt;
    if (const char* e = pb::build_wide_tree(nodes.data(), n_nodes, tris.data(), n_slots, &wt)) {
        std::printf("the shipped builder declines this tree: %s\n", e);
        return 1;
    }
    Layout32 l32;
    if (!build32(nodes.data(), &l32)) {
        std::printf("the 32-byte layout cannot be built over this tree (a plane left [0, 255])\n");
        return 1;
    }
    std::printf("# tree: %d binary nodes, %d triangles; 48-B records %d (%.1f MB); 32-B records %zu (%.1f MB); root cell exponents %d %d %d\n", n_nodes, n_slots,
                wt.n_records, wt.n_records * 48e-6, l32.recs.size(), l32.recs.size() * 32e-6, l32.root.e[0], l32.root.e[1], l32.root.e[2]);
    std::printf("# 32-B layout: shifts held back by K_MAX = %d on %lld child axes of %zu; frame origins that would not be exact floats: %lld\n", kMaxShift,
                (long long)l32.clamped_shift, 3 * l32.recs.size(), (long long)l32.inexact_origins);
    Walker w;
    w.nodes = nodes.data();
    w.tris = tris.data();
    w.wt = &wt;
    w.l32 = &l32;
    w.leaf_of_first.assign((size_t)n_slots, -1);
    {
        // wide-order first triangle of every leaf: walk the shipped records
        std::vector<int32_t> leaf_node((size_t)n_slots, -1);
        for (int32_t i = 0; i < n_nodes; ++i)
            if (nodes[i].n_primitives > 0) leaf_node[(size_t)nodes[i].offset] = i;
        for (int32_t p = 0; p < n_slots; ++p) {
            int32_t slot;
            std::memcpy(&slot, &wt.tris[12 * (size_t)p + 9], 4);
            if (leaf_node[(size_t)slot] >= 0) w.leaf_of_first[(size_t)p] = leaf_node[(size_t)slot];
        }
    }
    w.rec_bnode.assign((size_t)wt.n_records, -1);
    if (wt.n_records > 0) {
        std::vector<std::pair<int32_t, int32_t>> st{{wt.root_ref, 0}};
        while (!st.empty()) {
            auto [ref, bn] = st.back();
            st.pop_back();
            if (ref < 0) continue;
            w.rec_bnode[(size_t)ref] = bn;
            const uint32_t* rec = &wt.nodes[(size_t)ref * pb::kWideNodeDwords];
            int32_t slot_node[4];
            int axis_c[2];
            pb::wide_slots_of(nodes.data(), bn, slot_node, axis_c);
            for (int s = 0; s < 4; ++s) {
                const uint32_t m = s < 3 ? (rec[s] & 0xffu) : (rec[3] >> 24);
                if (m != 0xffu && (m & 0x80u)) st.push_back({(int32_t)(rec[10] + (m & 3u)), slot_node[s]});
            }
        }
    }
    const char* names[4] = {"camera rays, closest hit", "bounce rays, closest hit", "light-sample rays, any hit", "MIS rays, boolean any hit"};
    Counts a[5], b[5];
    const size_t n_rays = rays.size() / 8;
    for (size_t i = 0; i < n_rays; ++i) {
        const float* ry = &rays[8 * i];
        Ray r;
        for (int k = 0; k < 3; ++k) {
            r.o[k] = ry[k];
            r.id[k] = 1.0f / ry[3 + k];
            r.od[k] = ry[k];
            r.dd[k] = ry[3 + k];
        }
        r.tmax = ry[6];
        int cls = (int)ry[7];
        r.any = cls >= 2;
        if (!pb::wide_ray_covered(r.o[0], r.o[1], r.o[2], r.id[0], r.id[1], r.id[2])) continue;
        w.walk(r, a[cls], false);
        w.walk(r, b[cls], true);
    }
    auto sum = [](Counts& t, const Counts& c) {
        t.rays += c.rays, t.steps += c.steps, t.child_tests += c.child_tests, t.child_pass += c.child_pass, t.child_exact_pass += c.child_exact_pass;
        t.leaf_cand += c.leaf_cand, t.leaf_pass += c.leaf_pass, t.tri_tests += c.tri_tests, t.defers += c.defers, t.defer_steps += c.defer_steps;
        t.pops += c.pops, t.pops_culled += c.pops_culled, t.hits += c.hits;
        for (int i = 0; i < 64; ++i) t.depth_hist[i] += c.depth_hist[i];
        for (int i = 0; i < 4; ++i) t.step_depth_over[i] += c.step_depth_over[i];
    };
    for (int cls = 0; cls < 4; ++cls) {
        if (!a[cls].rays) continue;
        report(names[cls], a[cls], b[cls]);
        sum(a[4], a[cls]);
        sum(b[4], b[cls]);
    }
    report("all rays (the frame's mix)", a[4], b[4]);
    return 0;
}

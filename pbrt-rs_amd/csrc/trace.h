// trace.h — BVH traversal + watertight triangle test for gfx950, one ray per lane.
//
// Computes exactly what BVHAccel::intersect / intersect_p (src/accelerators/bvh.rs:828-932) compute
// with Bounds3f::intersect_p (src/core/geometry.rs:709-751) and Triangle::intersect_test
// (src/shapes/triangle.rs:74-158): same visiting order (near child first by dir_is_neg[axis]), same
// box test, same strictness in every comparison, same shrinking of ray.t_max, so the closest
// hit, its barycentrics and tie-breaks are bit-identical.
//
// Device layout (built by scene.h from the reference-order LinearBVHNode array):
//   * one 64-byte record per INTERIOR node holding BOTH children's boxes, child references and
//     the split axis. One coalesced 64-B fetch replaces the reference's two dependent 32-B fetches
//     (node, then child); the far child's box test result is kept on the stack as its entry
//     distance and re-checked against the current t_max when popped, which is the same decision
//     the reference makes when it tests the box at pop time.
//   * child reference: >= 0 interior record index; < 0 leaf: ~ref = (first_slot << count_bits) | (n - 1).
//   * triangles in leaf order, 48 B each: {v0.xyz v1.x | v1.yz v2.xy | v2.z prim_id material light}.
//   * traversal stack: STACK_LDS entries per lane in LDS ([entry][lane], conflict-free),
//     deeper entries spill to a per-lane global slab.
#pragma once
#include "dev_math.h"

namespace pb {

struct DevBVH {
    const float4* __restrict__ inodes;  // 4 x float4 per interior node
    const float4* __restrict__ tris;    // 3 x float4 per leaf slot
    float root_min[3], root_max[3];
    int root_ref;     // child-style reference of the root (leaf if the tree is a single leaf)
    int count_bits;   // bits of (n_primitives - 1) in a leaf reference
    int n_slots;
    uint2* __restrict__ spill;  // [entry][global lane] overflow stack
    int spill_stride;           // number of lanes the slab was sized for
};

#ifndef PB_STACK_LDS
#define PB_STACK_LDS 12
#endif
constexpr int kStackLds = PB_STACK_LDS;
constexpr int kStackSpill = 64 - PB_STACK_LDS;  // together: the reference's 64-entry stack (bvh.rs:839)
constexpr int kTraceBlock = 256;

struct TravRay {
    float ox, oy, oz, dx, dy, dz, tmax;
};
struct TravHit {
    float t, b0, b1, b2;
    int slot;  // leaf slot of the hit triangle, -1 = miss
};

// tri.z.w of the third float4 carries flags in the top bits of `light`: see scene.h
constexpr int kTriDegenerate = 1 << 30;  // Triangle::intersect returns false (triangle.rs:212-216)

PB_DEV bool slab_test(float bx0, float bx1, float by0, float by1, float bz0, float bz1, const TravRay& r, float idx,
                      float idy, float idz, float tmax_ray, float* entry) {
    // bx0 = bounds[dir_is_neg[0]].x, bx1 = bounds[1 - dir_is_neg[0]].x, ...
    float t_min = (bx0 - r.ox) * idx;
    float t_max = (bx1 - r.ox) * idx;
    float ty_min = (by0 - r.oy) * idy;
    float ty_max = (by1 - r.oy) * idy;
    t_max *= kSlabScale;
    ty_max *= kSlabScale;
    bool ok = !(t_min > ty_max || ty_min > t_max);
    t_min = (ty_min > t_min) ? ty_min : t_min;
    t_max = (ty_max < t_max) ? ty_max : t_max;
    float tz_min = (bz0 - r.oz) * idz;
    float tz_max = (bz1 - r.oz) * idz;
    tz_max *= kSlabScale;
    ok = ok && !(t_min > tz_max || tz_min > t_max);
    t_min = (tz_min > t_min) ? tz_min : t_min;
    t_max = (tz_max < t_max) ? tz_max : t_max;
    *entry = t_min;
    return ok && (t_min < tmax_ray) && (t_max > 0.0f);
}

// Per-ray constants of the watertight test (triangle.rs:84-101)
struct TriRayConst {
    int kz;
    float sx, sy, sz;
};
PB_DEV TriRayConst tri_ray_setup(const TravRay& r) {
    float ax = __builtin_fabsf(r.dx), ay = __builtin_fabsf(r.dy), az = __builtin_fabsf(r.dz);
    int kz = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
    // permute(kx, ky, kz): kz=0 -> (y,z,x); kz=1 -> (z,x,y); kz=2 -> (x,y,z)
    float px = kz == 0 ? r.dy : (kz == 1 ? r.dz : r.dx);
    float py = kz == 0 ? r.dz : (kz == 1 ? r.dx : r.dy);
    float pz = kz == 0 ? r.dx : (kz == 1 ? r.dy : r.dz);
    TriRayConst c;
    c.kz = kz;
    c.sx = -px / pz;
    c.sy = -py / pz;
    c.sz = 1.0f / pz;
    return c;
}
PB_DEV V3 permute_kz(V3 v, int kz) {
    return V3{kz == 0 ? v.y : (kz == 1 ? v.z : v.x), kz == 0 ? v.z : (kz == 1 ? v.x : v.y),
              kz == 0 ? v.x : (kz == 1 ? v.y : v.z)};
}

// triangle.rs:74-158. Returns true and (b0,b1,b2,t) when the ray hits within (0, tmax].
PB_DEV bool triangle_test(V3 p0, V3 p1, V3 p2, const TravRay& r, const TriRayConst& c, float tmax, float* b0o,
                          float* b1o, float* b2o, float* to) {
    V3 o = V3{r.ox, r.oy, r.oz};
    V3 p0t = permute_kz(p0 - o, c.kz);
    V3 p1t = permute_kz(p1 - o, c.kz);
    V3 p2t = permute_kz(p2 - o, c.kz);
    p0t.x += c.sx * p0t.z;
    p0t.y += c.sy * p0t.z;
    p1t.x += c.sx * p1t.z;
    p1t.y += c.sy * p1t.z;
    p2t.x += c.sx * p2t.z;
    p2t.y += c.sy * p2t.z;
    // edge functions in f64 (triangle.rs:109-111, D12)
    float e0 = (float)((double)p1t.x * (double)p2t.y - (double)p1t.y * (double)p2t.x);
    float e1 = (float)((double)p2t.x * (double)p0t.y - (double)p2t.y * (double)p0t.x);
    float e2 = (float)((double)p0t.x * (double)p1t.y - (double)p0t.y * (double)p1t.x);
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0t.z *= c.sz;
    p1t.z *= c.sz;
    p2t.z *= c.sz;
    float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < tmax * det)) return false;
    if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > tmax * det)) return false;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t = t_scaled * inv_det;
    float max_zt = max3(__builtin_fabsf(p0t.z), __builtin_fabsf(p1t.z), __builtin_fabsf(p2t.z));
    float delta_z = kGamma3 * max_zt;
    float max_xt = max3(__builtin_fabsf(p0t.x), __builtin_fabsf(p1t.x), __builtin_fabsf(p2t.x));
    float max_yt = max3(__builtin_fabsf(p0t.y), __builtin_fabsf(p1t.y), __builtin_fabsf(p2t.y));
    float delta_x = kGamma5 * (max_xt + max_zt);
    float delta_y = kGamma5 * (max_yt + max_zt);
    float delta_e = 2.0f * (kGamma2 * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = max3(__builtin_fabsf(e0), __builtin_fabsf(e1), __builtin_fabsf(e2));
    float delta_t = 3.0f * (kGamma3 * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * __builtin_fabsf(inv_det);
    if (t <= delta_t) return false;
    *b0o = b0;
    *b1o = b1;
    *b2o = b2;
    *to = t;
    return true;
}

PB_DEV void load_tri(const float4* __restrict__ tris, int slot, V3* p0, V3* p1, V3* p2, int* flags) {
    float4 a = tris[3 * slot], b = tris[3 * slot + 1], c = tris[3 * slot + 2];
    *p0 = V3{a.x, a.y, a.z};
    *p1 = V3{a.w, b.x, b.y};
    *p2 = V3{b.z, b.w, c.x};
    *flags = __float_as_int(c.w);
}

// One ray per lane. `lds_stack` points at this lane's column: entry e lives at lds_stack[e * kTraceBlock].
// ANY = true: BVHAccel::intersect_p (returns on the first triangle hit).
// COUNT = true: instrumented variant that also returns the number of box tests (bvh.rs:841-842)
// and triangle tests (triangle.rs:74) the REFERENCE's loop performs for this ray: the reference
// pushes the far child untested and tests it when popped, so here a far child whose box already
// failed is still pushed (entry distance +inf) and counted when popped; entries never popped
// (any-hit early exit) are not counted. These counts feed the algorithmic-byte roofline.
template <bool ANY, bool COUNT = false>
PB_DEV bool traverse(const DevBVH& bvh, const TravRay& r, TravHit* hit, uint2* lds_stack, int spill_lane,
                     uint32_t* n_node = nullptr, uint32_t* n_prim = nullptr) {
    float tmax = r.tmax;
    hit->t = tmax;
    hit->slot = -1;
    hit->b0 = hit->b1 = hit->b2 = 0.0f;
    const float idx = 1.0f / r.dx, idy = 1.0f / r.dy, idz = 1.0f / r.dz;  // bvh.rs:831
    const bool nx = idx < 0.0f, ny = idy < 0.0f, nz = idz < 0.0f;       // bvh.rs:832-836
    const TriRayConst trc = tri_ray_setup(r);
    float e;
    if (COUNT) *n_node += 1;
    if (!slab_test(nx ? bvh.root_max[0] : bvh.root_min[0], nx ? bvh.root_min[0] : bvh.root_max[0],
                   ny ? bvh.root_max[1] : bvh.root_min[1], ny ? bvh.root_min[1] : bvh.root_max[1],
                   nz ? bvh.root_max[2] : bvh.root_min[2], nz ? bvh.root_min[2] : bvh.root_max[2], r, idx, idy, idz,
                   tmax, &e))
        return false;
    const int count_mask = (1 << bvh.count_bits) - 1;
    int sp = 0;
    int cur = bvh.root_ref;
    bool found = false;
    bool running = true;
    while (running) {
        // ---- interior nodes ----
        while (running && cur >= 0) {
            const float4* nd = bvh.inodes + 4 * (size_t)cur;
            float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
            // child 0 box: min (q0.x q0.y q0.z) max (q0.w q1.x q1.y); child 1: min (q1.z q1.w q2.x) max (q2.y q2.z q2.w)
            float e0, e1;
            bool h0 = slab_test(nx ? q0.w : q0.x, nx ? q0.x : q0.w, ny ? q1.x : q0.y, ny ? q0.y : q1.x,
                                nz ? q1.y : q0.z, nz ? q0.z : q1.y, r, idx, idy, idz, tmax, &e0);
            bool h1 = slab_test(nx ? q2.y : q1.z, nx ? q1.z : q2.y, ny ? q2.z : q1.w, ny ? q1.w : q2.z,
                                nz ? q2.w : q2.x, nz ? q2.x : q2.w, r, idx, idy, idz, tmax, &e1);
            int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y), axis = __float_as_int(q3.z);
            // bvh.rs:857-865: if dir_is_neg[axis] visit the second child first
            bool neg = axis == 0 ? nx : (axis == 1 ? ny : nz);
            int near_c = neg ? c1 : c0, far_c = neg ? c0 : c1;
            bool near_h = neg ? h1 : h0, far_h = neg ? h0 : h1;
            float far_e = neg ? e0 : e1;
            if (COUNT) {
                *n_node += 1;                 // the near child is tested as soon as it is visited
                if (!near_h) *n_node += 1;    // near missed: the far child is popped and tested next
                if (!far_h) far_e = kInf;
            }
            if (near_h) {
                cur = near_c;
                if (far_h || COUNT) {
                    uint2 ent = make_uint2((uint32_t)far_c, __float_as_uint(far_e));
                    if (sp < kStackLds)
                        lds_stack[sp * kTraceBlock] = ent;
                    else
                        bvh.spill[(size_t)(sp - kStackLds) * bvh.spill_stride + spill_lane] = ent;
                    ++sp;
                }
            } else if (far_h) {
                cur = far_c;
            } else {
                // pop: skip entries whose entry distance no longer beats the shrunk t_max
                for (;;) {
                    if (sp == 0) {
                        running = false;
                        break;
                    }
                    --sp;
                    uint2 ent = (sp < kStackLds) ? lds_stack[sp * kTraceBlock]
                                                 : bvh.spill[(size_t)(sp - kStackLds) * bvh.spill_stride + spill_lane];
                    if (COUNT) *n_node += 1;
                    if (__uint_as_float(ent.y) < tmax) {
                        cur = (int)ent.x;
                        break;
                    }
                }
            }
        }
        if (!running) break;
        // ---- leaf ----
        {
            int ref = ~cur;
            int n = (ref & count_mask) + 1;
            int first = ref >> bvh.count_bits;
            for (int i = 0; i < n; ++i) {
                V3 p0, p1, p2;
                int flags;
                load_tri(bvh.tris, first + i, &p0, &p1, &p2, &flags);
                float b0, b1, b2, t;
                if (COUNT) *n_prim += 1;
                if (triangle_test(p0, p1, p2, r, trc, tmax, &b0, &b1, &b2, &t)) {
                    if (ANY) return true;
                    if (!(flags & kTriDegenerate)) {
                        tmax = t;  // primitive.rs:70
                        hit->t = t;
                        hit->b0 = b0;
                        hit->b1 = b1;
                        hit->b2 = b2;
                        hit->slot = first + i;
                        found = true;
                    }
                }
            }
            for (;;) {
                if (sp == 0) {
                    running = false;
                    break;
                }
                --sp;
                uint2 ent = (sp < kStackLds) ? lds_stack[sp * kTraceBlock]
                                             : bvh.spill[(size_t)(sp - kStackLds) * bvh.spill_stride + spill_lane];
                if (COUNT) *n_node += 1;
                if (__uint_as_float(ent.y) < tmax) {
                    cur = (int)ent.x;
                    break;
                }
            }
        }
    }
    return found;
}

// wave-reduce the instrumented counts, one atomic pair per wave
PB_DEV void count_flush(unsigned long long* counters, uint32_t n_node, uint32_t n_prim, uint32_t n_rays = 1) {
    // inactive lanes contribute 0 to __shfl_xor? No: they return their own stale register, so reduce
    // with ballot-guarded values instead: every active lane adds through LDS-free wave atomics.
    unsigned long long m = __ballot(1);
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t a = __shfl_xor(n_node, o, 64), b = __shfl_xor(n_prim, o, 64), c = __shfl_xor(n_rays, o, 64);
        bool peer_active = (m >> ((threadIdx.x & 63) ^ o)) & 1ull;
        n_node += peer_active ? a : 0u;
        n_prim += peer_active ? b : 0u;
        n_rays += peer_active ? c : 0u;
    }
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
        atomicAdd(&counters[0], (unsigned long long)n_node);
        atomicAdd(&counters[1], (unsigned long long)n_prim);
        atomicAdd(&counters[2], (unsigned long long)n_rays);
    }
}

}  // namespace pb

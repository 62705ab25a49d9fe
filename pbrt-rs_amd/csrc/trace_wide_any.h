// trace_wide_any.h — the rays of which only a BOOLEAN is wanted, over the 4-wide quantised records (wide_bvh.h), one-level scenes.
//
// BVHAccel::intersect_p (src/accelerators/bvh.rs:881-932) returns "some leaf whose box passes Bounds3f::intersect_p
// (src/core/geometry.rs:709-751) holds a triangle that passes Triangle::intersect_test (src/shapes/triangle.rs:74-158)", with a
// ray.t_max that never changes. The slab test is monotone in the box (wide_bvh.h: the rays for which it is not never reach this
// kernel), every ancestor's box contains the leaf's, so that boolean is a property of the SET of leaves and not of the order
// the reference happens to visit them in. Everything trace_wide.h does to reproduce that order is therefore left out here:
// no ranking of a record's four children by the direction signs (`sel` and seven v_perm_b32), no entry distances on the stack
// (4-byte entries: half the LDS), no re-check of a popped entry against a shrunken t_max, no hit record, no `raised` hand-over.
// What decides is unchanged — the reference's slab test on the EXACT leaf box, then its triangle test — so the boolean is the
// reference's, bit for bit. The same holds for `found` of BVHAccel::intersect (wf_state.h: RS_MIS_BOOL) with the triangles
// Triangle::intersect rejects left out (IO::strict).
//
// Who runs here: shadow rays and RS_MIS_BOOL rays of a wavefront (the sort key's top bit puts them at the end of the sorted
// queue; render.hip launches this kernel on that part), and pbrt_hip_intersect_p.
#pragma once
#include "trace_wide.h"

namespace pb {

#ifndef PB_ANY_WAVES
#define PB_ANY_WAVES 6
#endif
#ifndef PB_ANY_STACK_LDS
#define PB_ANY_STACK_LDS 12
#endif
#ifndef PB_ANY_INTERIOR_THRESH
#define PB_ANY_INTERIOR_THRESH 48
#endif
#ifndef PB_ANY_ORDERED
#define PB_ANY_ORDERED 0  // 1: walk the children in the reference's order all the same (measurement: what the order is worth to an any-hit ray)
#endif
constexpr int kAnyStackLds = PB_ANY_STACK_LDS;

template <class IO, bool COUNT = false>
PB_DEV void trace_wide_any(const WideTrees& wt, const IO& io, unsigned int* __restrict__ work_counter, uint32_t* lds_stack,
                           int spill_lane, unsigned long long* counters = nullptr) {
    const uint32_t n = io.n();
    const int lane = threadIdx.x & 63;
    TravRay r;
    float idx = 0.0f, idy = 0.0f, idz = 0.0f;
    int sp = 0;
    uint32_t index = 0;  // the ray's token (IO::token)
    bool nx = false, ny = false, nz = false;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.tmax = 0.0f;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    uint32_t c_rec = 0, c_cand = 0, c_tri = 0, c_special = 0;
    // lane codes as in trace_wide.h (a record index >= 0, a leaf reference < 0 above the codes)
    constexpr int kNeedPop = (int)0x80000001, kIdle = (int)0x80000002, kDoneHit = (int)0x80000003, kDoneMiss = (int)0x80000004,
                  kWait = (int)0x80000005;
    int cur = kIdle;
    int pend = 0;  // one postponed leaf (< 0) as in trace_wide.h; here it may be tested at any time: there is no order to keep
    auto is_idle = [&]() -> bool { return ((uint32_t)cur - (uint32_t)kIdle) <= 2u; };
    auto is_leaf_ref = [&]() -> bool { return cur < 0 && cur > kWait; };
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint32_t LdsEntry;
    typedef volatile __attribute__((address_space(3))) uint32_t LdsWord;
#else
    typedef uint32_t LdsEntry;
    typedef volatile uint32_t LdsWord;
#endif
    LdsEntry* const lds = (LdsEntry*)lds_stack;
    uint32_t* const spill = reinterpret_cast<uint32_t*>(wt.spill);  // the scene's slab, 4 of every 8 bytes used
    auto stack_write = [&](int pos, int ref) {
        if (pos < kAnyStackLds)
            lds[pos * kTraceBlock] = (uint32_t)ref;
        else
            spill[2 * ((size_t)(pos - kAnyStackLds) * wt.spill_stride + spill_lane)] = (uint32_t)ref;
    };
    auto stack_read = [&](int pos) -> int {
        uint32_t ent = *(LdsWord*)&lds[(pos < kAnyStackLds ? pos : kAnyStackLds - 1) * kTraceBlock];
        if (pos >= kAnyStackLds) ent = spill[2 * ((size_t)(pos - kAnyStackLds) * wt.spill_stride + spill_lane)];
        return (int)ent;
    };
    auto finish = [&](bool found) {
        cur = found ? kDoneHit : kDoneMiss;
        pend = 0;
    };
    auto flush_result = [&]() {
        if (cur == kDoneHit || cur == kDoneMiss) {
            io.store(index, true, cur == kDoneHit, r.tmax, 0.0f, 0.0f, 0.0f, 0, -1);
            cur = kIdle;
        }
    };
    // every entry on the stack passed the filter under the ray's one and only t_max: a pop is always live
    auto pop_one = [&]() {
        if (sp == 0) {
            if (pend < 0)
                cur = kWait;
            else
                finish(false);
        } else {
            --sp;
            cur = stack_read(sp);
        }
    };
    for (;;) {
        // ---------------- refill idle lanes (as trace_wide.h) ----------------
        unsigned long long idle_mask = __ballot(is_idle());
        int n_idle = popc64(idle_mask);
        if (!exhausted && (n_idle >= PB_WIDE_REFILL_THRESH)) {
            if (chunk_next >= chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned int)kChunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane(base);
                if (base < n) {
                    chunk_next = base;
                    chunk_end = (base + kChunk) < n ? (base + kChunk) : n;
                } else {
                    exhausted = true;
                }
            }
            uint32_t avail = chunk_end - chunk_next;
            uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0));
            flush_result();
            bool take = is_idle() && prefix < avail;
            uint32_t my = chunk_next + prefix;
            chunk_next += ((uint32_t)n_idle < avail) ? (uint32_t)n_idle : avail;
            bool special = false;
            if (take) {
                index = io.token(my);
                bool any_unused;
                bool real = io.load(index, &r, &any_unused);
                sp = 0;
                pend = 0;
                if (!real) {
                    finish(false);
                } else {
                    idx = 1.0f / r.dx;  // bvh.rs:884
                    idy = 1.0f / r.dy;
                    idz = 1.0f / r.dz;
                    nx = idx < 0.0f;  // bvh.rs:885-889
                    ny = idy < 0.0f;
                    nz = idz < 0.0f;
                    special = !wide_ray_covered(r.ox, r.oy, r.oz, idx, idy, idz);
                    if (COUNT && special) c_special += 1;
                    cur = special ? kIdle : wt.root_ref;  // uncovered rays: the binary kernel afterwards (wide_bvh.h)
                }
            }
            unsigned long long sm = __ballot(special);
            if (sm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(wt.special_count, (unsigned int)popc64(sm));
                base = (uint32_t)__builtin_amdgcn_readfirstlane(base);
                if (special)
                    wt.special_list[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0))] = index;
            }
        }
        if (!__any(!is_idle())) {
            if (exhausted) break;
            continue;
        }

        // ---------------- records ----------------
        for (;;) {
            if (pend >= 0 && is_leaf_ref()) {  // postpone the leaf, walk on
                pend = cur;
                cur = kNeedPop;
            }
            if (cur == kNeedPop) pop_one();
            bool interior = cur >= 0;
            int n_int = popc64(__ballot(cur >= 0) | __ballot(cur == kNeedPop));
            if (n_int == 0) break;
            if (n_int < PB_ANY_INTERIOR_THRESH) {
                bool leaf_pending = __any(is_leaf_ref() || cur == kWait);
                bool can_refill = !exhausted && (popc64(__ballot(is_idle())) >= PB_WIDE_REFILL_THRESH);
                if (leaf_pending || can_refill) break;
            }
            if (interior) {
                const uint4* nd = wt.nodes + 3 * (size_t)cur;
                uint4 q0 = nd[0], q1 = nd[1], q2 = nd[2];
                if (COUNT) c_rec += 1;
                const uint32_t dw3 = q0.w;
                const WideSetup ws = wide_setup(q0.x, q0.y, q0.z, dw3, r.ox, r.oy, r.oz, idx, idy, idz);
                // the near / far plane bytes of the four slots, and their descriptor bytes, in SLOT order
                uint32_t nqx = nx ? q1.y : q1.x, fqx = nx ? q1.x : q1.y;
                uint32_t nqy = ny ? q1.w : q1.z, fqy = ny ? q1.z : q1.w;
                uint32_t nqz = nz ? q2.y : q2.x, fqz = nz ? q2.x : q2.y;
                uint32_t mpack = __builtin_amdgcn_perm(q0.y, q0.x, 0x0c0c0400u) | __builtin_amdgcn_perm(dw3, q0.z, 0x07000c0cu);
                if (PB_ANY_ORDERED) {  // (measurement only: the reference's visiting order, as trace_wide.h)
                    const uint32_t negmask = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                    const uint32_t f_root = (negmask >> ((dw3 >> 18) & 3u)) & 1u;
                    const uint32_t f_c0 = (negmask >> ((dw3 >> 20) & 3u)) & 1u, f_c1 = (negmask >> ((dw3 >> 22) & 3u)) & 1u;
                    uint32_t sel = 0x03020100u ^ (f_c0 ? 0x00000101u : 0u) ^ (f_c1 ? 0x01010000u : 0u);
                    sel = __builtin_amdgcn_alignbit(sel, sel, f_root << 4);
                    nqx = __builtin_amdgcn_perm(0u, nqx, sel), fqx = __builtin_amdgcn_perm(0u, fqx, sel);
                    nqy = __builtin_amdgcn_perm(0u, nqy, sel), fqy = __builtin_amdgcn_perm(0u, fqy, sel);
                    nqz = __builtin_amdgcn_perm(0u, nqz, sel), fqz = __builtin_amdgcn_perm(0u, fqz, sel);
                    mpack = __builtin_amdgcn_perm(0u, mpack, sel);
                }
                const uint32_t child_base = q2.z, ntb = q2.w;
                bool h[4];
                int ref[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t m = (mpack >> (8 * k)) & 0xffu;
                    float tn_unused;
                    h[k] = wide_child_test(ws, nqx, nqy, nqz, fqx, fqy, fqz, k, r.tmax, &tn_unused) && (m != 0xffu);
                    ref[k] = (m & 0x80u) ? (int)(child_base + (m & 3u)) : (int)(ntb - m);
                }
                const bool p3 = h[3] && (h[0] || h[1] || h[2]), p2 = h[2] && (h[0] || h[1]), p1 = h[1] && h[0];
                if (!__any(sp > kAnyStackLds - 3)) {
                    // (as trace_wide.h: unconditional stores, a lane that does not push leaves its stack pointer where it was)
                    lds[sp * kTraceBlock] = (uint32_t)ref[3];
                    sp += p3 ? 1 : 0;
                    lds[sp * kTraceBlock] = (uint32_t)ref[2];
                    sp += p2 ? 1 : 0;
                    lds[sp * kTraceBlock] = (uint32_t)ref[1];
                    sp += p1 ? 1 : 0;
                } else {
                    if (p3) {
                        stack_write(sp, ref[3]);
                        ++sp;
                    }
                    if (p2) {
                        stack_write(sp, ref[2]);
                        ++sp;
                    }
                    if (p1) {
                        stack_write(sp, ref[1]);
                        ++sp;
                    }
                }
                if (h[0] || h[1] || h[2] || h[3]) {
                    cur = h[0] ? ref[0] : (h[1] ? ref[1] : (h[2] ? ref[2] : ref[3]));
                } else {
                    cur = kNeedPop;
                }
            }
        }

        // ---------------- leaves: the reference's box test on the exact leaf box, then its triangles ----------------
        if (pend < 0 || is_leaf_ref()) {
            const bool from_pend = pend < 0;
            const int v = ~(from_pend ? pend : cur);
            const int cnt = (v & 3) + 1;
            const int first = v >> 2;
            if (COUNT) c_cand += 1;
            if (COUNT) c_tri += 1;
            const float4* tp0 = wt.tris + 3 * (size_t)first;
            float4 ta = tp0[0], tb = tp0[1], tc = tp0[2];
            float lox, loy, loz, hix, hiy, hiz;
            if (cnt == 1) {
                // Triangle::world_bound (the union of the three vertices): exact, so it is the leaf node's box
                lox = wide_fmin(ta.x, wide_fmin(ta.w, tb.z));
                hix = wide_fmax(ta.x, wide_fmax(ta.w, tb.z));
                loy = wide_fmin(ta.y, wide_fmin(tb.x, tb.w));
                hiy = wide_fmax(ta.y, wide_fmax(tb.x, tb.w));
                loz = wide_fmin(ta.z, wide_fmin(tb.y, tc.x));
                hiz = wide_fmax(ta.z, wide_fmax(tb.y, tc.x));
            } else {
                const float4* bp = wt.leaf_boxes + 2 * (size_t)first;
                float4 b0 = bp[0], b1 = bp[1];
                lox = b0.x;
                loy = b0.y;
                loz = b0.z;
                hix = b1.x;
                hiy = b1.y;
                hiz = b1.z;
            }
            float entry;
            const bool pass = slab_test(nx ? hix : lox, nx ? lox : hix, ny ? hiy : loy, ny ? loy : hiy, nz ? hiz : loz, nz ? loz : hiz, r,
                                        idx, idy, idz, r.tmax, &entry);
            bool done = false;
            if (pass) {
                const TriRayConst trc = tri_ray_setup(r, idx, idy, idz);
                for (int i = 0; i < cnt; ++i) {
                    if (i > 0) {
                        if (COUNT) c_tri += 1;
                        const float4* tp = wt.tris + 3 * (size_t)(first + i);
                        ta = tp[0];
                        tb = tp[1];
                        tc = tp[2];
                    }
                    float b0, b1, b2, t;
                    if (triangle_test(V3{ta.x, ta.y, ta.z}, V3{ta.w, tb.x, tb.y}, V3{tb.z, tb.w, tc.x}, r, trc, r.tmax, &b0, &b1, &b2, &t)) {
                        if (io.strict(index) && (__float_as_int(tc.z) & kTriDegenerate)) continue;  // (IO::strict, trace_persistent.h)
                        done = true;
                        break;
                    }
                }
            }
            if (done)
                finish(true);
            else if (from_pend) {
                pend = 0;
                if (cur == kWait) cur = kNeedPop;  // (the pop finds the stack empty and finishes the ray)
            } else
                cur = kNeedPop;
        }
    }
    flush_result();
    if (COUNT) count_flush(counters + 4, c_rec, c_cand, c_tri, c_special);
}

}  // namespace pb

// trace_wide_pool.h — trace_wide.h's one-level walk with the LEAF work of a block's four waves pooled through LDS.
//
// trace_wide.h's waves alternate between the record loop (50 of 64 lanes) and the leaf phase — the reference's slab test on the
// exact leaf box and Triangle::intersect_test — which runs in most outer iterations with about TEN lanes (profiles/
// r02_wave_time_by_section.txt: 23 % of a wave's time). Every lane executes the same leaf code whoever's leaf it is, so here a lane
// that reaches a candidate leaf does not wait for ITS wave's next leaf phase: it posts the leaf — ray, t_max, leaf reference: 32
// bytes — in its slot of a block-wide table in LDS and walks on (exactly trace_wide.h's postponed leaf: what it walks meanwhile is a
// superset of what the reference walks, and it posts no second leaf before the first one's answer is in). Whichever wave of the
// block next finds enough posted leaves takes up to 64 of them, one per lane — FULL waves of leaf work —, and writes each answer
// (hit, t, barycentrics / any-hit found / t_max moved up) back into the slot, where the owner picks it up. Same tests on the same
// boxes and triangles with the same t_max in the same per-ray order, so the same hits bit for bit; only who executes them changes.
//
// Protocol (all in LDS, workgroup scope): slot state 0 free -> 1 posted (owner, after writing the item) -> 3 claimed (a serving
// lane's compare-and-swap) -> 2 answered (the server, after writing the answer) -> 0 (the owner, after reading it). `posted` counts
// the slots in state 1. No wave ever waits for a wave that is not already executing the work it waits for: every wave serves the
// table itself when it has nothing else to do, and a batch, once claimed, is finished by the wave that claimed it.
#pragma once
#include "trace_wide.h"

namespace pb {

#ifndef PB_POOL_WAVES
#define PB_POOL_WAVES 5
#endif
#ifndef PB_POOL_STACK_LDS
#define PB_POOL_STACK_LDS 11
#endif
#ifndef PB_POOL_SERVE_AT
#define PB_POOL_SERVE_AT 48  // a wave serves the table when this many leaves are posted (or when it cannot step records itself)
#endif
#ifndef PB_POOL_INTERIOR_THRESH
#define PB_POOL_INTERIOR_THRESH 40
#endif
constexpr int kPoolStackLds = PB_POOL_STACK_LDS;

struct PoolShared {
    uint2 stack[kPoolStackLds * kTraceBlock];
    float4 item_a[kTraceBlock];  // posted: o.xyz, d.x;                       answered: t, b0, b1, b2
    float4 item_b[kTraceBlock];  // posted: d.y, d.z, t_max, leaf word;       answered: .x = code (leaf slot >= 0 | -1 miss | -2 any-hit found | -3 t_max moved up)
    uint32_t state[kTraceBlock];
    uint32_t posted;
};

template <class IO>
PB_DEV void trace_wide_pool(const WideTrees& wt, const IO& io, unsigned int* __restrict__ work_counter, PoolShared* sh, int spill_lane) {
    constexpr int kLds = kPoolStackLds;
    const uint32_t n = io.n();
    const int lane = threadIdx.x & 63, me = threadIdx.x;
    TravRay r;
    float idx = 0.0f, idy = 0.0f, idz = 0.0f, tmax = 0.0f, hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f;
    int hit_slot = -1, sp = 0;
    uint32_t index = 0;
    bool nx = false, ny = false, nz = false, any = false;
    uint32_t negmask = 0;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.tmax = 0.0f;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    const int n_seg = io.segments();
    int seg = (int)(blockIdx.x % (unsigned)n_seg), seg_tries = 0;
    bool out = false;  // this lane has a leaf posted (its slot is in state 1, 3 or 2)
    constexpr int kNeedPop = (int)0x80000001, kIdle = (int)0x80000002, kDoneHit = (int)0x80000003, kDoneMiss = (int)0x80000004,
                  kWait = (int)0x80000005;  // kWait: nothing left to walk, the posted leaf's answer decides
    int cur = kIdle;
    auto is_idle = [&]() -> bool { return ((uint32_t)cur - (uint32_t)kIdle) <= 2u; };
    auto is_leaf_ref = [&]() -> bool { return cur < 0 && cur > kWait; };
#ifdef PB_LANE_STATS
    unsigned long long wstat[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned int wl_steps = 0, wl_posted = 0;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint2 LdsEntry;
    typedef volatile __attribute__((address_space(3))) unsigned long long LdsWord;
    typedef __attribute__((address_space(3))) float4 LdsVec;
    typedef volatile __attribute__((address_space(3))) uint32_t LdsU32;
#else
    typedef uint2 LdsEntry;
    typedef volatile unsigned long long LdsWord;
    typedef float4 LdsVec;
    typedef volatile uint32_t LdsU32;
#endif
    LdsEntry* const lds = (LdsEntry*)(sh->stack + me);
    LdsVec* const item_a = (LdsVec*)sh->item_a;
    LdsVec* const item_b = (LdsVec*)sh->item_b;
    LdsU32* const state = (LdsU32*)sh->state;
    LdsU32* const posted = (LdsU32*)&sh->posted;
    auto stack_write = [&](int pos, int ref, float entry) {
        uint2 ent = make_uint2((uint32_t)ref, __float_as_uint(entry));
        if (pos < kLds)
            lds[pos * kTraceBlock] = ent;
        else
            wt.spill[(size_t)(pos - kLds) * wt.spill_stride + spill_lane] = ent;
    };
    auto stack_read = [&](int pos) -> uint2 {
        unsigned long long raw = *(LdsWord*)&lds[(pos < kLds ? pos : kLds - 1) * kTraceBlock];
        uint2 ent = make_uint2((uint32_t)raw, (uint32_t)(raw >> 32));
        if (pos >= kLds) ent = wt.spill[(size_t)(pos - kLds) * wt.spill_stride + spill_lane];
        return ent;
    };
    auto finish = [&](bool found) { cur = found ? kDoneHit : kDoneMiss; };
    auto flush_result = [&]() {
        if (cur == kDoneHit || cur == kDoneMiss) {
            io.store(index, any, cur == kDoneHit, tmax, hb0, hb1, hb2, hit_slot, -1);
            cur = kIdle;
        }
    };
    auto pop_one = [&]() {
        if (sp == 0) {
            if (out)
                cur = kWait;
            else
                finish(hit_slot >= 0);
        } else {
            --sp;
            uint2 ent = stack_read(sp);
            if (__uint_as_float(ent.y) < tmax) cur = (int)ent.x;
        }
    };
    // the answer to this lane's posted leaf, if it is in
    auto take_answer = [&]() {
        if (out && state[me] == 2u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float4 a = item_a[me];
            const int code = __float_as_int(item_b[me].x);
            state[me] = 0u;
            out = false;
            if (code >= 0) {  // primitive.rs:70: the leaf's closest accepted hit
                tmax = a.x;
                hb0 = a.y;
                hb1 = a.z;
                hb2 = a.w;
                hit_slot = code;
                if (cur == kWait) cur = kNeedPop;
            } else if (code == -1) {
                if (cur == kWait) cur = kNeedPop;  // (the pop finds the stack empty and finishes the ray)
            } else if (code == -2) {
                finish(true);
            } else {  // t_max moved up (wide_bvh.h): the binary kernel traces the ray from scratch
                wt.special_list[atomicAdd(wt.special_count, 1u)] = index;
                cur = kIdle;
            }
        }
    };
    // Serve the table: every lane looks at four slots (its own number in each wave's quarter), claims the first posted one it
    // finds and does that leaf: the reference's box test on the exact leaf box, then its triangles (trace_wide.h's leaf phase).
    auto serve = [&]() {
        int slot = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int s = lane + 64 * j;
            if (slot < 0 && state[s] == 1u && atomicCAS((uint32_t*)&sh->state[s], 1u, 3u) == 1u) slot = s;
        }
        const unsigned long long got = __ballot(slot >= 0);
        PB_WSTAT(2, 1);
        PB_WSTAT(3, popc64(got));
        if (!got) return;
        if (lane == 0) atomicSub((uint32_t*)&sh->posted, (uint32_t)popc64(got));
        if (slot >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float4 a = item_a[slot], b = item_b[slot];
            TravRay q{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
            const uint32_t word = __float_as_uint(b.w);
            const bool q_any = (word >> 30) & 1u, q_strict = (word >> 31) != 0u;
            const int v = (int)(word & 0x3fffffffu);
            const int cnt = (v & 3) + 1, first = v >> 2;
            const float ix = 1.0f / q.dx, iy = 1.0f / q.dy, iz = 1.0f / q.dz;  // bvh.rs:831, as the owner computed them
            const bool bx = ix < 0.0f, by = iy < 0.0f, bz = iz < 0.0f;
            float qt = q.tmax;
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
                const float4 b0 = bp[0], b1 = bp[1];
                lox = b0.x;
                loy = b0.y;
                loz = b0.z;
                hix = b1.x;
                hiy = b1.y;
                hiz = b1.z;
            }
            float entry;
            const bool pass = slab_test(bx ? hix : lox, bx ? lox : hix, by ? hiy : loy, by ? loy : hiy, bz ? hiz : loz, bz ? loz : hiz, q, ix,
                                        iy, iz, qt, &entry);
            int code = -1;
            float rb0 = 0.0f, rb1 = 0.0f, rb2 = 0.0f;
            bool raised = false;
            if (pass) {
                const TriRayConst trc = tri_ray_setup(q, ix, iy, iz);
                for (int i = 0; i < cnt; ++i) {
                    if (i > 0) {
                        const float4* tp = wt.tris + 3 * (size_t)(first + i);
                        ta = tp[0];
                        tb = tp[1];
                        tc = tp[2];
                    }
                    float b0, b1, b2, t;
                    if (triangle_test(V3{ta.x, ta.y, ta.z}, V3{ta.w, tb.x, tb.y}, V3{tb.z, tb.w, tc.x}, q, trc, qt, &b0, &b1, &b2, &t)) {
                        if (q_any) {
                            if (q_strict && (__float_as_int(tc.z) & kTriDegenerate)) continue;  // (IO::strict, trace_persistent.h)
                            code = -2;
                            break;
                        }
                        if (!(__float_as_int(tc.z) & kTriDegenerate)) {
                            if (t > qt) raised = true;  // (see `raised` in trace_wide.h)
                            qt = t;                     // primitive.rs:70
                            rb0 = b0;
                            rb1 = b1;
                            rb2 = b2;
                            code = __float_as_int(tc.y);
                        }
                    }
                }
            }
            if (raised) code = -3;
            item_a[slot] = make_float4(qt, rb0, rb1, rb2);
            item_b[slot] = make_float4(__int_as_float(code), 0.0f, 0.0f, 0.0f);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            state[slot] = 2u;
        }
    };
    for (;;) {
        // ---------------- refill idle lanes (as trace_wide.h) ----------------
        PB_WCLOCK(t_refill0);
        unsigned long long idle_mask = __ballot(is_idle());
        int n_idle = popc64(idle_mask);
        if (!exhausted && (n_idle >= PB_WIDE_REFILL_THRESH)) {
            if (chunk_next >= chunk_end) {
                while (seg_tries < n_seg) {
                    uint32_t seg_begin = (uint32_t)(((unsigned long long)n * (unsigned)seg) / (unsigned)n_seg);
                    uint32_t seg_end = (uint32_t)(((unsigned long long)n * (unsigned)(seg + 1)) / (unsigned)n_seg);
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(work_counter + seg, (unsigned int)kChunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane(base) + seg_begin;
                    if (base < seg_end && base >= seg_begin) {
                        chunk_next = base;
                        chunk_end = (base + kChunk) < seg_end ? (base + kChunk) : seg_end;
                        break;
                    }
                    seg = (seg + 1 == n_seg) ? 0 : seg + 1;
                    seg_tries += 1;
                }
                if (chunk_next >= chunk_end) exhausted = true;
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
                bool real = io.load(index, &r, &any);
                tmax = r.tmax;
                hit_slot = -1;
                hb0 = hb1 = hb2 = 0.0f;
                sp = 0;
                if (!real) {
                    finish(false);
                } else {
                    idx = 1.0f / r.dx;  // bvh.rs:831
                    idy = 1.0f / r.dy;
                    idz = 1.0f / r.dz;
                    nx = idx < 0.0f;  // bvh.rs:832-836
                    ny = idy < 0.0f;
                    nz = idz < 0.0f;
                    negmask = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                    special = !wide_ray_covered(r.ox, r.oy, r.oz, idx, idy, idz);
                    cur = special ? kIdle : wt.root_ref;
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

        // ---------------- records (trace_wide.h's step; a candidate leaf is posted instead of postponed) ----------------
        PB_WCLOCK(t_rec0);
        PB_WSTAT(13, t_rec0 - t_refill0);
        PB_WSTAT(6, 1);
        PB_WSTAT(7, popc64(__ballot(!is_idle())));
        for (;;) {
            take_answer();
            {
                // post the candidate leaf and walk on (the count first: it may run ahead of the slots, never behind them)
                const bool posting = !out && is_leaf_ref();
                const unsigned long long pm = __ballot(posting);
                if (pm) {
                    if (lane == 0) atomicAdd((uint32_t*)&sh->posted, (uint32_t)popc64(pm));
                    if (posting) {
#ifdef PB_LANE_STATS
                        wl_posted += 1;
#endif
                        item_a[me] = make_float4(r.ox, r.oy, r.oz, r.dx);
                        item_b[me] = make_float4(r.dy, r.dz, tmax,
                                                 __uint_as_float((uint32_t)(~cur) | (any ? 1u << 30 : 0u) | ((any && io.strict(index)) ? 1u << 31 : 0u)));
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        state[me] = 1u;
                        out = true;
                        cur = kNeedPop;
                    }
                }
            }
            if (cur == kNeedPop) pop_one();
            const bool interior = cur >= 0;
            const int n_int = popc64(__ballot(cur >= 0) | __ballot(cur == kNeedPop));
            if (n_int == 0) break;
            if (n_int < PB_POOL_INTERIOR_THRESH) {
                const bool blocked = __any(is_leaf_ref() || cur == kWait);
                const bool can_refill = !exhausted && (popc64(__ballot(is_idle())) >= PB_WIDE_REFILL_THRESH);
                if (blocked || can_refill || *posted >= (uint32_t)PB_POOL_SERVE_AT) break;
            }
            PB_WSTAT(0, 1);
            PB_WSTAT(1, popc64(__ballot(interior)));
            if (interior) {
#ifdef PB_LANE_STATS
                wl_steps += 1;
#endif
                const uint4* nd = wt.nodes + 3 * (size_t)cur;
                const uint4 q0 = nd[0], q1 = nd[1], q2 = nd[2];
                const uint32_t dw3 = q0.w;
                const WideSetup ws = wide_setup(q0.x, q0.y, q0.z, dw3, r.ox, r.oy, r.oz, idx, idy, idz);
                const uint32_t f_root = (negmask >> ((dw3 >> 18) & 3u)) & 1u;
                const uint32_t f_c0 = (negmask >> ((dw3 >> 20) & 3u)) & 1u, f_c1 = (negmask >> ((dw3 >> 22) & 3u)) & 1u;
                uint32_t sel = 0x03020100u ^ (f_c0 ? 0x00000101u : 0u) ^ (f_c1 ? 0x01010000u : 0u);
                sel = __builtin_amdgcn_alignbit(sel, sel, f_root << 4);
                const uint32_t nqx = __builtin_amdgcn_perm(0u, nx ? q1.y : q1.x, sel), fqx = __builtin_amdgcn_perm(0u, nx ? q1.x : q1.y, sel);
                const uint32_t nqy = __builtin_amdgcn_perm(0u, ny ? q1.w : q1.z, sel), fqy = __builtin_amdgcn_perm(0u, ny ? q1.z : q1.w, sel);
                const uint32_t nqz = __builtin_amdgcn_perm(0u, nz ? q2.y : q2.x, sel), fqz = __builtin_amdgcn_perm(0u, nz ? q2.x : q2.y, sel);
                const uint32_t mslot = __builtin_amdgcn_perm(q0.y, q0.x, 0x0c0c0400u) | __builtin_amdgcn_perm(dw3, q0.z, 0x07000c0cu);
                const uint32_t mpack = __builtin_amdgcn_perm(0u, mslot, sel);
                const uint32_t child_base = q2.z, ntb = q2.w;
                float tn[4];
                bool h[4];
                int ref[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t m = (mpack >> (8 * k)) & 0xffu;
                    h[k] = wide_child_test(ws, nqx, nqy, nqz, fqx, fqy, fqz, k, tmax, &tn[k]) && (m != 0xffu);
                    ref[k] = (m & 0x80u) ? (int)(child_base + (m & 3u)) : (int)(ntb - m);
                }
                const bool p3 = h[3] && (h[0] || h[1] || h[2]), p2 = h[2] && (h[0] || h[1]), p1 = h[1] && h[0];
                if (!__any(sp > kLds - 3)) {
                    lds[sp * kTraceBlock] = make_uint2((uint32_t)ref[3], __float_as_uint(tn[3]));
                    sp += p3 ? 1 : 0;
                    lds[sp * kTraceBlock] = make_uint2((uint32_t)ref[2], __float_as_uint(tn[2]));
                    sp += p2 ? 1 : 0;
                    lds[sp * kTraceBlock] = make_uint2((uint32_t)ref[1], __float_as_uint(tn[1]));
                    sp += p1 ? 1 : 0;
                } else {
                    if (p3) {
                        stack_write(sp, ref[3], tn[3]);
                        ++sp;
                    }
                    if (p2) {
                        stack_write(sp, ref[2], tn[2]);
                        ++sp;
                    }
                    if (p1) {
                        stack_write(sp, ref[1], tn[1]);
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

        // ---------------- the leaves of the block ----------------
        PB_WCLOCK(t_leaf0);
        PB_WSTAT(14, t_leaf0 - t_rec0);
        const bool starved = popc64(__ballot(cur >= 0 || cur == kNeedPop)) < PB_POOL_INTERIOR_THRESH;
        const uint32_t np = *posted;
        if (np >= (uint32_t)PB_POOL_SERVE_AT || (np > 0u && starved)) {
            serve();
        } else if (starved && __any(out) && !(!exhausted && popc64(__ballot(is_idle())) >= PB_WIDE_REFILL_THRESH)) {
            __builtin_amdgcn_s_sleep(2);  // answers are on their way from a wave that is serving
            PB_WSTAT(9, 1);
        }
#ifdef PB_LANE_STATS
        wstat[15] += __builtin_readcyclecounter() - t_leaf0;
#endif
    }
    flush_result();
#ifdef PB_LANE_STATS
    {
        unsigned long long v[2] = {wl_steps, wl_posted};
        for (int k = 0; k < 2; ++k)
            for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        wstat[11] = v[0];
        wstat[12] = v[1];
        if (lane == 0)
            for (int i = 0; i < 32; ++i) atomicAdd(&g_wide_stats[i], wstat[i]);
    }
#endif
}

}  // namespace pb

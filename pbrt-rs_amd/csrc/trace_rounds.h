// trace_rounds.h — two-level scenes of ONE object aggregate (BASELINE config 5's shape) traced in ROUNDS: the top-level walk and
// the walks inside the instances in launches of their own (round 4; PBRT_TRAVERSAL_ROUNDS).
//
// TransformedPrimitive::intersect (src/core/primitive.rs:136-159) nests the object aggregate's BVHAccel::intersect inside the
// top-level one's leaf loop (src/accelerators/bvh.rs:844-850). trace_wide<.., INST> runs both in one persistent kernel, and a
// wave there is in one of three phases at a time (top-level or object records / instance entry and exit / triangles) with a
// third of its lanes. Here a ray still visits its instances strictly one after the other, in the
// reference's order, each with the ray.t_max the previous one left — but ACROSS launches:
//
//   k_rounds_top     walks the top-level records with the world ray. A candidate top-level leaf is confirmed by the reference's
//                    slab test on its exact box; its entries are tried in leaf order: world ray -> object space
//                    (geometry.rs:865-881, origin-error shift included), the object's root box. An attempt that misses costs
//                    nothing more. A ray that ENTERS is suspended: its object ray and everything the top-level walk needs to go
//                    on (world ray, t_max, the hit so far, the rest of the leaf, its short stack) go to an entry queue.
//   k_rounds_object  trace_wide's ONE-LEVEL kernel (INST = 0: 50 of 64 lanes per record iteration) over the object's records,
//                    one entry = one object-space ray; closest hit / any hit as the world ray asks; result beside the entry.
//   k_rounds_top     again, over the entries: t_max and hit updated as primitive.rs:140-143 says, the walk goes on from where
//                    it stopped; rays that enter another instance make the next round's entries.
//
// until a round leaves no entries. Same tests in the same order with the same t_max as the fused kernel, hence the same hits
// bit for bit (tests/test_gpu_rounds.py). Rays that fall out of what the wide records cover (an object-space ray beyond the
// filter's range, t_max moving up, a top-level stack deeper than kRoundsStackSave) are handed, whole, to the binary kernel as
// everywhere else (wide_bvh.h).
//
// State traffic per entry: 96 B + 8 B per stack entry written at the suspension, 32 B read + 20 B written by the object walk,
// 84 B + stack read at the resumption (SoA, consecutive entries = consecutive addresses).
#pragma once
#include "trace_wide.h"

namespace pb {

#ifndef PB_ROUNDS_TOP_WAVES
#define PB_ROUNDS_TOP_WAVES 5
#endif
#ifndef PB_ROUNDS_INTERIOR_THRESH
#define PB_ROUNDS_INTERIOR_THRESH 32
#endif
#ifndef PB_ROUNDS_REFILL_THRESH
#define PB_ROUNDS_REFILL_THRESH 16
#endif
constexpr int kRoundsStackLds = PB_WIDE_INST_STACK_LDS;  // the top-level walk's stack entries in LDS
constexpr int kRoundsStackSave = 20;                      // a suspended ray carries at most this many stack entries (deeper: binary kernel)
// Queue grabs and entry-slot reservations are same-address atomics, which saturate near 10^2 per microsecond chip-wide
// (DESIGN section 4.5): an activation here is a few record steps, not a whole ray, so a wave takes 512 rays / entries per
// grab and reserves entry slots 256 at a time (the first version reserved per refill: 38 M atomics per config-5 frame,
// 2.5x slower than the fused kernel for that alone, profiles/r04_rounds.txt).
constexpr uint32_t kRoundsChunk = 512, kRoundsSlotBlock = 256;

// One round's entries (SoA; capacity `cap`), written by k_rounds_top, read by k_rounds_object and by the next k_rounds_top
struct RoundEntries {
    float4* A0;  // object ray: o.xyz, d.x
    float4* A1;  // object ray: d.yz, t_max, kind (int bits: 0 closest hit, 1 any hit, 2 any hit strict — IO::strict)
    float4* A2;  // world ray: o.xyz, d.x
    float4* A3;  // world ray: d.yz, t_max, what is left of the top-level leaf (leaf_state, int bits)
    float4* A4;  // the hit so far: leaf slot (int bits, -1 none), b0, b1, b2
    uint4* A5;   // instance of that hit (top slot), the instance being entered (top slot), stack entries, the ray's token
    uint2* S;    // S[k * cap + e]: stack entry k of entry e
    float4* R0;  // written by the object walk: leaf slot of its hit (int bits; -1 none; -2 the ray left the wide path), b0, b1, b2
    float* R1;   // ... and its t
    unsigned int* count;  // slots handed out (whole blocks of kRoundsSlotBlock: a wave's last block ends in holes, kind = -1)
    unsigned int* real;   // entries among them
    uint32_t cap;
};

// IO policy of the object walks (trace_persistent.h): ray e of the launch is entry e
struct RoundObjectIO {
    RoundEntries en;
    PB_DEV uint32_t n() const { return *en.count; }
    PB_DEV int segments() const { return 1; }
    PB_DEV uint32_t token(uint32_t i) const { return i; }
    PB_DEV static constexpr uint32_t chunk() { return kRoundsChunk; }
    PB_DEV bool strict(uint32_t e) const { return __float_as_int(en.A1[e].w) == 2; }
    PB_DEV bool load(uint32_t e, TravRay* r, bool* any) const {
        const float4 a = en.A0[e], b = en.A1[e];
        *r = TravRay{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
        *any = __float_as_int(b.w) != 0;
        return __float_as_int(b.w) >= 0;  // a hole at the end of a wave's slot block: not a ray
    }
    PB_DEV void store(uint32_t e, bool any, bool found, float t, float b0, float b1, float b2, int slot, int inst) const {
        (void)any;
        (void)inst;
        en.R0[e] = make_float4(__int_as_float(found ? slot : -1), b0, b1, b2);
        en.R1[e] = t;
    }
};

// the entries whose object walk left the wide path (t_max moved up, wide_bvh.h): their rays go to the binary kernel, whole
__global__ void k_rounds_mark_abandoned(RoundEntries en, const uint32_t* __restrict__ list, const unsigned int* __restrict__ count) {
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        en.R0[list[i]] = make_float4(__int_as_float(-2), 0.0f, 0.0f, 0.0f);
}

// RESUME false: round 0, the rays of the wavefront (IO as trace_wide's); true: the entries of the previous round, each with
// the result of its object walk. `out`: the entries this launch makes.
template <class IO, bool RESUME>
PB_DEV void trace_rounds_top(const WideTrees& wt, const IO& io, const RoundEntries& in, const RoundEntries& out,
                             unsigned int* __restrict__ work_counter, uint2* lds_stack, int spill_lane, float* lds_world) {
    constexpr int kLds = kRoundsStackLds;
    const uint32_t n = RESUME ? *in.count : io.n();
    const int lane = threadIdx.x & 63;
    TravRay r;  // the world ray; the object ray from a successful entry attempt until the suspended state is written out
    float idx = 0.0f, idy = 0.0f, idz = 0.0f, tmax = 0.0f, tmax_world = 0.0f, hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f;
    int hit_slot = -1, hit_inst = -1, sp = 0, leaf_state = 0, cur_top_slot = -1;
    uint32_t index = 0;
    bool nx = false, ny = false, nz = false, any = false;
    uint32_t negmask = 0;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.tmax = 0.0f;
    uint32_t chunk_next = 0, chunk_end = 0;
    uint32_t slot_next = 0, slot_end = 0, n_suspended = 0;  // this wave's block of entry slots (wave-uniform)
    bool exhausted = false;
    // lane codes as in trace_wide.h; kResumed: back from an instance, the rest of its top-level leaf comes next;
    // kSuspended: entered an instance, the state waits in the registers for the wave's next refill
    constexpr int kResumed = (int)0x80000000, kNeedPop = (int)0x80000001, kIdle = (int)0x80000002, kDoneHit = (int)0x80000003,
                  kDoneMiss = (int)0x80000004, kSuspended = (int)0x80000005;
    int cur = kIdle;
    auto is_idle = [&]() -> bool { return ((uint32_t)cur - (uint32_t)kIdle) <= 3u; };  // idle, done (result pending), suspended (state pending)
    auto is_leaf_ref = [&]() -> bool { return cur < 0 && cur > kSuspended; };
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint2 LdsEntry;
    typedef volatile __attribute__((address_space(3))) unsigned long long LdsWord;
    typedef __attribute__((address_space(3))) float LdsFloat;
#else
    typedef uint2 LdsEntry;
    typedef volatile unsigned long long LdsWord;
    typedef float LdsFloat;
#endif
    LdsEntry* const lds = (LdsEntry*)lds_stack;
    LdsFloat* const w = (LdsFloat*)lds_world;  // the world ray, [component][lane]: o, d (read back when the lane is suspended)
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
    auto abandon = [&]() {  // the whole ray to the binary kernel (rare: one list append per such ray)
        wt.special_list[atomicAdd(wt.special_count, 1u)] = index;
        cur = kIdle;
    };
    auto set_ray_constants = [&]() {
        idx = 1.0f / r.dx;  // bvh.rs:831
        idy = 1.0f / r.dy;
        idz = 1.0f / r.dz;
        nx = idx < 0.0f;  // bvh.rs:832-836
        ny = idy < 0.0f;
        nz = idz < 0.0f;
        negmask = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    };
    auto pop_one = [&]() {
        if (sp == 0) {
            finish(hit_slot >= 0);
        } else {
            --sp;
            uint2 ent = stack_read(sp);
            if (__uint_as_float(ent.y) < tmax) cur = (int)ent.x;
        }
    };
    // results of finished rays and the state of suspended ones leave the registers together, at the wave's refill
    auto flush = [&]() {
        if (cur == kDoneHit || cur == kDoneMiss) {
            io.store(index, any, cur == kDoneHit, tmax, hb0, hb1, hb2, hit_slot, hit_inst);
            cur = kIdle;
        }
        const unsigned long long sm = __ballot(cur == kSuspended);
        if (sm) {
            const uint32_t cnt = (uint32_t)popc64(sm), rem = slot_end - slot_next;
            uint32_t fresh_block = 0;
            if (cnt > rem) {  // the wave's block of slots is used up: the first `rem` lanes take its end, the others start a new one
                if (lane == 0) fresh_block = atomicAdd(out.count, (unsigned int)kRoundsSlotBlock);
                fresh_block = (uint32_t)__builtin_amdgcn_readfirstlane(fresh_block);
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0));
            const uint32_t e = rank < rem ? slot_next + rank : fresh_block + (rank - rem);
            if (cnt > rem) {
                slot_next = fresh_block + (cnt - rem);
                slot_end = fresh_block + kRoundsSlotBlock;
            } else {
                slot_next += cnt;
            }
            n_suspended += cnt;
            if (cur == kSuspended && e >= out.cap) abandon();  // (never with the capacity render.hip allots)
            if (cur == kSuspended) {
                const int kind = any ? (io.strict(index) ? 2 : 1) : 0;
                out.A0[e] = make_float4(r.ox, r.oy, r.oz, r.dx);
                out.A1[e] = make_float4(r.dy, r.dz, tmax, __int_as_float(kind));
                out.A2[e] = make_float4(w[0 * kTraceBlock], w[1 * kTraceBlock], w[2 * kTraceBlock], w[3 * kTraceBlock]);
                out.A3[e] = make_float4(w[4 * kTraceBlock], w[5 * kTraceBlock], tmax_world, __int_as_float(leaf_state));
                out.A4[e] = make_float4(__int_as_float(hit_slot), hb0, hb1, hb2);
                out.A5[e] = make_uint4((uint32_t)hit_inst, (uint32_t)cur_top_slot, (uint32_t)sp, index);
                for (int k = 0; k < sp; ++k) out.S[(size_t)k * out.cap + e] = stack_read(k);
                cur = kIdle;
            }
        }
    };
    for (;;) {
        // ---------------- refill ----------------
        unsigned long long idle_mask = __ballot(is_idle());
        int n_idle = popc64(idle_mask);
        if (!exhausted && n_idle >= PB_ROUNDS_REFILL_THRESH) {
            if (chunk_next >= chunk_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned int)kRoundsChunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane(base);
                if (base < n) {
                    chunk_next = base;
                    chunk_end = (base + kRoundsChunk) < n ? (base + kRoundsChunk) : n;
                } else {
                    exhausted = true;
                }
            }
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0));
            flush();
            const bool take = is_idle() && prefix < avail;
            const uint32_t my = chunk_next + prefix;
            chunk_next += ((uint32_t)n_idle < avail) ? (uint32_t)n_idle : avail;
            if (take) {
                if (!RESUME) {
                    index = io.token(my);
                    const bool real = io.load(index, &r, &any);
                    tmax = r.tmax;
                    hit_slot = -1;
                    hit_inst = -1;
                    hb0 = hb1 = hb2 = 0.0f;
                    sp = 0;
                    leaf_state = 0;
                    if (!real) {
                        finish(false);  // placeholder of a path outside pixel_bounds
                    } else {
                        set_ray_constants();
                        cur = wide_ray_covered(r.ox, r.oy, r.oz, idx, idy, idz) ? wt.root_ref : kIdle;
                        if (cur == kIdle) abandon();
                    }
                } else {
                    const float4 a1 = in.A1[my], a2 = in.A2[my], a3 = in.A3[my], a4 = in.A4[my], r0 = in.R0[my];
                    const uint4 a5 = in.A5[my];
                    const float rt = in.R1[my];
                    const bool hole = __float_as_int(a1.w) < 0;
                    index = a5.w;
                    any = (index & 3u) >= 2u;  // wf_state.h: RS_SHADOW, RS_MIS_BOOL (batch IOs never come here)
                    r = TravRay{a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z};
                    tmax = a3.z;
                    leaf_state = __float_as_int(a3.w);
                    hit_slot = __float_as_int(a4.x);
                    hb0 = a4.y;
                    hb1 = a4.z;
                    hb2 = a4.w;
                    hit_inst = (int)a5.x;
                    sp = (int)a5.z;
                    if (hole) sp = 0;
                    for (int k = 0; k < sp; ++k) {
                        const uint2 ent = in.S[(size_t)k * in.cap + my];
                        stack_write(k, (int)ent.x, __uint_as_float(ent.y));
                    }
                    const int res = __float_as_int(r0.x);
                    set_ray_constants();
                    cur = kResumed;
                    if (hole) {
                        cur = kIdle;
                    } else if (res == -2) {
                        abandon();
                    } else if (res >= 0) {
                        if (any) {
                            finish(true);
                        } else {  // primitive.rs:140-143: the hit inside the instance becomes the ray's
                            hit_slot = res;
                            hb0 = r0.y;
                            hb1 = r0.z;
                            hb2 = r0.w;
                            hit_inst = (int)a5.y;
                            tmax = rt;
                            r.tmax = rt;
                        }
                    }
                }
                if (!is_idle()) {  // the world ray waits in LDS for the moment the lane is suspended
                    w[0 * kTraceBlock] = r.ox;
                    w[1 * kTraceBlock] = r.oy;
                    w[2 * kTraceBlock] = r.oz;
                    w[3 * kTraceBlock] = r.dx;
                    w[4 * kTraceBlock] = r.dy;
                    w[5 * kTraceBlock] = r.dz;
                }
            }
        }
        if (!__any(!is_idle())) {
            if (exhausted) break;
            continue;
        }

        // ---------------- top-level records (the record step of trace_wide.h) ----------------
        for (;;) {
            if (cur == kNeedPop) pop_one();
            const bool interior = cur >= 0;
            const int n_int = popc64(__ballot(cur >= 0) | __ballot(cur == kNeedPop));
            if (n_int == 0) break;
            if (n_int < PB_ROUNDS_INTERIOR_THRESH) {
                const bool leaf_pending = __any(is_leaf_ref() || cur == kResumed);
                const bool can_refill = !exhausted && (popc64(__ballot(is_idle())) >= PB_ROUNDS_REFILL_THRESH);
                if (leaf_pending || can_refill) break;
            }
            if (interior) {
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

        // ---------------- top-level leaves: the instance branch of trace_wide.h without the walk inside ----------------
        if (is_leaf_ref() || cur == kResumed) {
            const bool fresh = cur != kResumed;
            if (fresh) {
                const int v = ~cur;
                leaf_state = ((v >> 2) << 3) | ((v & 3) + 1);
            }
            bool walk = (leaf_state & 7) != 0;
            float4 e0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), e1 = e0, m0 = e0, m1 = e0, m2 = e0;
            if (walk) {
                const float4* ep = wt.top_slots + 5 * (size_t)(leaf_state >> 3);
                e0 = ep[0];
                e1 = ep[1];
                m0 = ep[2];
                m1 = ep[3];
                m2 = ep[4];
            }
            if (fresh) {
                // a candidate top-level leaf: the reference's test on its exact box with the world ray and the current t_max
                float e;
                walk = slab_test(nx ? e1.x : e0.x, nx ? e0.x : e1.x, ny ? e1.y : e0.y, ny ? e0.y : e1.y, nz ? e1.z : e0.z,
                                 nz ? e0.z : e1.z, r, idx, idy, idz, tmax, &e);
                if (!walk) leaf_state = 0;
            }
            bool entered = false, lost = false;
            while (walk) {
                leaf_state += 8 - 1;  // this entry is taken: next position, one entry less
                // TransformedPrimitive::intersect, first half (primitive.rs:136-139; enter_instance of trace_wide.h)
                const float x = r.ox, y = r.oy, z = r.oz, wdx = r.dx, wdy = r.dy, wdz = r.dz;
                float ox = m0.x * x + m0.y * y + m0.z * z + m0.w;
                float oy = m1.x * x + m1.y * y + m1.z * z + m1.w;
                float oz = m2.x * x + m2.y * y + m2.z * z + m2.w;
                const float xa = __builtin_fabsf(m0.x * x) + __builtin_fabsf(m0.y * y) + __builtin_fabsf(m0.z * z) + __builtin_fabsf(m0.w);
                const float ya = __builtin_fabsf(m1.x * x) + __builtin_fabsf(m1.y * y) + __builtin_fabsf(m1.z * z) + __builtin_fabsf(m1.w);
                const float za = __builtin_fabsf(m2.x * x) + __builtin_fabsf(m2.y * y) + __builtin_fabsf(m2.z * z) + __builtin_fabsf(m2.w);
                const float ex = xa * kGamma3, ey = ya * kGamma3, ez = za * kGamma3;
                const float dx = m0.x * wdx + m0.y * wdy + m0.z * wdz;
                const float dy = m1.x * wdx + m1.y * wdy + m1.z * wdz;
                const float dz = m2.x * wdx + m2.y * wdy + m2.z * wdz;
                const float l2 = dx * dx + dy * dy + dz * dz;
                float tm = tmax;
                if (l2 > 0.0f) {
                    const float dt = (__builtin_fabsf(dx) * ex + __builtin_fabsf(dy) * ey + __builtin_fabsf(dz) * ez) / l2;
                    ox = ox + dx * dt;
                    oy = oy + dy * dt;
                    oz = oz + dz * dt;
                    tm -= dt;
                }
                const TravRay ro{ox, oy, oz, dx, dy, dz, tm};
                const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // bvh.rs:831
                if (!wide_ray_covered(ox, oy, oz, ix, iy, iz)) {
                    lost = true;
                } else {
                    const bool bx = ix < 0.0f, by = iy < 0.0f, bz = iz < 0.0f;
                    float e;
                    // the object aggregate's own root box (bvh.rs:841-842)
                    if (slab_test(bx ? wt.obj0_max[0] : wt.obj0_min[0], bx ? wt.obj0_min[0] : wt.obj0_max[0], by ? wt.obj0_max[1] : wt.obj0_min[1],
                                  by ? wt.obj0_min[1] : wt.obj0_max[1], bz ? wt.obj0_max[2] : wt.obj0_min[2], bz ? wt.obj0_min[2] : wt.obj0_max[2],
                                  ro, ix, iy, iz, tm, &e)) {
                        entered = true;
                        cur_top_slot = __float_as_int(e0.w);
                        tmax_world = tmax;
                        r = ro;  // (the world ray is in LDS)
                        tmax = tm;
                    }
                }
                walk = (leaf_state & 7) != 0 && !entered && !lost;
                if (walk) {  // the leaf's next entry
                    const float4* ep = wt.top_slots + 5 * (size_t)(leaf_state >> 3);
                    e0 = ep[0];
                    e1 = ep[1];
                    m0 = ep[2];
                    m1 = ep[3];
                    m2 = ep[4];
                }
            }
            if (lost || (entered && sp > kRoundsStackSave)) {
                abandon();
            } else if (entered) {
                cur = kSuspended;
            } else {
                cur = kNeedPop;
            }
        }
    }
    flush();
    // what is left of the wave's slot block becomes holes; the wave's entries are added to the round's count
    for (uint32_t e = slot_next + (uint32_t)lane; e < slot_end; e += 64u)
        if (e < out.cap) out.A1[e] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(-1));
    if (lane == 0 && n_suspended) atomicAdd(out.real, n_suspended);
}

}  // namespace pb

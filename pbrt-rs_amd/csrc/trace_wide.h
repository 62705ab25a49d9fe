// trace_wide.h — closest-hit / any-hit traversal over the 4-wide quantised records (wide_bvh.h).
//
// Same results as trace_persistent.h, i.e. as BVHAccel::intersect / intersect_p (src/accelerators/bvh.rs:828-932)
// with Bounds3f::intersect_p (src/core/geometry.rs:709-751) and Triangle::intersect_test (src/shapes/triangle.rs:74-158):
// the records only decide which leaves are LOOKED AT, in the reference's near-first order; whether a leaf's
// triangles are tested is decided by the reference's own slab test on the leaf's exact box with the ray.t_max of that
// moment (the argument is in wide_bvh.h). Scheduling (persistent waves, per-lane ray replacement, postponed leaves)
// is trace_persistent.h's. Per step a lane issues three 16-B loads for four boxes instead of four for two.
//
// Rays the filter's error bound does not cover (a reciprocal direction component that is not finite or beyond
// 2^+-40, an origin beyond 2^24) are not traced here: their queue positions go to `special_list` and the binary
// kernel traces them in a second, usually empty, launch.
#pragma once
#include "trace_persistent.h"
#include "wide_bvh.h"

namespace pb {

// Waves per SIMD of the one-level kernel. 6 since round 5: the kernel needs 84 VGPRs, the compiler reaches the 80 that six waves
// allow without scratch, and six blocks' stacks (6 x 24.5 KB) still fit the CU's 160 KB of LDS: -1.6 % of the traversal time
// (profiles/r05_path_layout.txt item 6; in round 3, at 86 VGPRs, six waves spilled and gained nothing). Seven would need 72 VGPRs
// (24 bytes of scratch) and a 10-entry LDS stack. The two-level kernels stay at 5: their 31.7 KB of LDS per block allow no more.
#ifndef PB_WIDE_WAVES
#define PB_WIDE_WAVES 6
#endif
#ifndef PB_WIDE_INST_WAVES
#define PB_WIDE_INST_WAVES 5
#endif
#ifndef PB_WIDE_STACK_LDS
#define PB_WIDE_STACK_LDS 12
#endif
#ifndef PB_WIDE_INTERIOR_THRESH
#define PB_WIDE_INTERIOR_THRESH 48  // with postponed leaves (PB_WIDE_SPECULATE); 32 was the optimum without them, and is 3 % slower with
#endif
#ifndef PB_WIDE_INST_INTERIOR_THRESH
#define PB_WIDE_INST_INTERIOR_THRESH 24  // two-level scenes: more kinds of work wait behind the record loop (config 5: 24 +5 %, 40 -8 %)
#endif
#ifndef PB_WIDE_SPECULATE
#define PB_WIDE_SPECULATE 1  // one-level scenes: a lane keeps walking records with ONE candidate leaf postponed (see `pend`); two-level scenes never (measured slower, profiles/r03_two_level_ladder.txt)
#endif
#ifndef PB_WIDE_INST_STACK_LDS
#define PB_WIDE_INST_STACK_LDS 11  // two-level scenes: one stack entry less in LDS makes room for the world ray (below) at 5 blocks per CU
#endif
#ifndef PB_WIDE_INST_GATHER
#define PB_WIDE_INST_GATHER 1
#endif
#ifndef PB_WIDE_REFILL_THRESH
#define PB_WIDE_REFILL_THRESH 8
#endif
constexpr int kWideStackLds = PB_WIDE_STACK_LDS;
// Two-level scenes: the world ray of a lane inside an instance waits in LDS, [component][lane]: origin, direction and the
// reciprocal direction (so that leaving an instance costs nine LDS reads and no division). 9 KB per block of 256.
constexpr int kWideWorldFloats = 9;
constexpr int wide_stack_lds(int inst) { return inst ? PB_WIDE_INST_STACK_LDS : PB_WIDE_STACK_LDS; }
constexpr int wide_world_lds_bytes(int inst) { return inst ? kWideWorldFloats * 4 * 256 : 0; }

#ifdef PB_LANE_STATS
// development instrumentation (tools/lane_stats.py): wave-level iteration counts and lane sums of trace_wide
__device__ unsigned long long g_wide_stats[32];
#define PB_WSTAT(i, v) wstat[i] += (v)
#define PB_WCLOCK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define PB_WSTAT(i, v)
#define PB_WCLOCK(var)
#endif

// 64-bit population count as a 32-bit scalar (with __popcll the compare that follows is made in 64 bits, on the vector unit)
__device__ __forceinline__ int popc64(unsigned long long m) {
    return __builtin_popcount((uint32_t)m) + __builtin_popcount((uint32_t)(m >> 32));
}

// The rays trace_wide left to the binary kernel, as an IO policy of trace_persistent: ray i of this launch is the ray
// whose token (the IO's own: queue entry / batch position) is list[i]; their number is read from device memory (the wide
// launch counted them).
template <class Inner>
struct SpecialListIO {
    Inner inner;
    const uint32_t* __restrict__ list;
    const unsigned int* __restrict__ count;
    PB_DEV uint32_t n() const { return *count; }
    PB_DEV int segments() const { return 1; }
    PB_DEV uint32_t token(uint32_t i) const { return list[i]; }
    PB_DEV bool strict(uint32_t tok) const { return inner.strict(tok); }
    PB_DEV bool load(uint32_t tok, TravRay* r, bool* any) const { return inner.load(tok, r, any); }
    PB_DEV void store(uint32_t tok, bool any, bool found, float t, float b0, float b1, float b2, int slot, int inst) const {
        inner.store(tok, any, found, t, b0, b1, b2, slot, inst);
    }
};

// COUNT: also counts what this kernel itself fetches (records stepped, candidate leaves, triangles loaded, rays left to
// the binary kernel) into counters[4..7]: the inputs of bench.py's gather-rate roofline.
// INST: two-level scenes (primitive.rs:105-159), as trace_persistent's INST: the world ray walks the top-level records; a
// candidate top-level leaf is confirmed with the reference's slab test on its exact box, its entries are taken in leaf
// order: a TransformedPrimitive transforms the ray (geometry.rs:865-881) and walks its object's records with a stack
// floor, a plain triangle is tested in place. An object-space ray the filter's bound does not cover ends the wide
// traversal of that ray: it goes to the binary kernel like an uncovered world ray. INST == 1: instances of one object
// aggregate and nothing beside them (its root in the kernel arguments, no per-entry kind / object lookups); 2: general.
// The world ray is not kept in registers while a lane is inside an instance: it waits in LDS (lds_world, [component][lane],
// with its reciprocal direction), and an entry whose object root box the ray misses never replaces the lane's ray in the
// first place. The top-level entries are 80-byte records (wt.top_slots: the exact box of the leaf they belong to, two meta
// words, the world-to-object rows): whatever a lane's turn in the instance branch needs first arrives in ONE memory round
// trip (round 3; before, the leaf's box, then the entry's transform, then the object's root came one after the other and
// the branch cost 2.7 record iterations at 14 lanes).
template <class IO, bool COUNT = false, int INST = 0>
PB_DEV void trace_wide(const WideTrees& wt, const IO& io, unsigned int* __restrict__ work_counter, uint2* lds_stack,
                       int spill_lane, unsigned long long* counters = nullptr, float* lds_world = nullptr) {
    constexpr int kLds = wide_stack_lds(INST);  // stack entries per lane in LDS (the rest: wt.spill)
    // record / wide-order triangle index -> uint4 / float4 offset: 3 per element packed, 4 one 64-byte line each (wide_bvh.h:
    // vec_stride). Two-level scenes are always packed (their trees are small: a compile-time 3 keeps this kernel as it was);
    // one-level: 4 i - (i & m) with m = ~0 packed, 0 lines — a shift, an and, a subtract, no 64-bit multiply by a register
    const uint32_t packed_mask = INST ? ~0u : (wt.vec_stride == 3 ? ~0u : 0u);
    auto wide_vec_offset = [&](int i) -> size_t { return INST ? (size_t)3 * (size_t)i : (size_t)(((uint32_t)i << 2) - ((uint32_t)i & packed_mask)); };
    const uint32_t n = io.n();
    const int lane = threadIdx.x & 63;
    TravRay r;
    float idx = 0.0f, idy = 0.0f, idz = 0.0f, tmax = 0.0f, hb0 = 0.0f, hb1 = 0.0f, hb2 = 0.0f;
    int hit_slot = -1, sp = 0;
    uint32_t index = 0;  // the ray's token (IO::token): what load, store and the special list name it by
    bool nx = false, ny = false, nz = false, any = false;
    uint32_t negmask = 0;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.tmax = 0.0f;
    uint32_t chunk_next = 0, chunk_end = 0;
    bool exhausted = false;
    const int n_seg = io.segments();
    int seg = (int)(blockIdx.x % (unsigned)n_seg), seg_tries = 0;
    uint32_t c_rec = 0, c_cand = 0, c_tri = 0, c_special = 0;
    // two-level state (INST)
    float tmax_world = 0.0f;
    // base_sp >= 0: the lane is inside an instance whose stack floor is base_sp (-1 outside); the instance that holds the
    // current hit is hit_inst (a top-level slot is visited at most once per ray, so hit_inst == cur_top_slot says "hit
    // inside the instance being left"). No flags: see the lane codes above.
    // leaf_state: the top-level leaf being worked through: position of its next entry << 3 | entries left (0 = none)
    int leaf_state = 0, cur_top_slot = -1, hit_inst = -1, base_sp = -1;
    // What a lane is doing is all in `cur`: a record index (>= 0), a leaf reference (< 0, above the five codes below), or
    // one of the codes. Flags kept as separate booleans cost scalar mask bookkeeping in every iteration of the loops
    // that change them; one integer costs a vector compare where it is asked.
    constexpr int kLeaveInstance = (int)0x80000000;  // INST: the instance's stack floor was reached, leave it in the leaf phase
    constexpr int kNeedPop = (int)0x80000001;         // out of children: take a stack entry at the next record iteration
    constexpr int kIdle = (int)0x80000002;            // no ray
    constexpr int kDoneHit = (int)0x80000003;         // ray finished, its result is still in the registers ...
    constexpr int kDoneMiss = (int)0x80000004;        // ... (found / not found), stored at the next refill
    constexpr int kWait = (int)0x80000005;            // nothing left to walk, postponed leaves still to be tested (see `pend`)
    int cur = kIdle;
    auto is_idle = [&]() -> bool { return ((uint32_t)cur - (uint32_t)kIdle) <= 2u; };
    auto is_leaf_ref = [&]() -> bool { return cur < 0 && cur > kWait; };
#ifdef PB_LANE_STATS
    unsigned long long wstat[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // wave-level events (lane 0's copy is kept)
    unsigned int wl_steps = 0, wl_children = 0, wl_cand = 0, wl_pass = 0, wl_tris = 0;  // this lane's own events
    unsigned int wl_gate = 0, wl_gate_pass = 0, wl_try = 0, wl_entered = 0, wl_exit = 0;  // two-level: top-leaf box tests, entry attempts, exits
#endif

    // the LDS part of the stack through an LDS-typed pointer: with a generic one the compiler merges the two
    // branches below into flat stores
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) uint2 LdsEntry;
    typedef volatile __attribute__((address_space(3))) unsigned long long LdsWord;
#else  // the host pass only parses this function
    typedef uint2 LdsEntry;
    typedef volatile unsigned long long LdsWord;
#endif
    LdsEntry* const lds = (LdsEntry*)lds_stack;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) float LdsFloat;
#else
    typedef float LdsFloat;
#endif
    auto stack_write = [&](int pos, int ref, float entry) {
        uint2 ent = make_uint2((uint32_t)ref, __float_as_uint(entry));
        if (pos < kLds)
            lds[pos * kTraceBlock] = ent;
        else
            wt.spill[(size_t)(pos - kLds) * wt.spill_stride + spill_lane] = ent;
    };
    auto stack_read = [&](int pos) -> uint2 {
        // the LDS read is unconditional (a conditional one is merged with the spill read into a flat load)
        unsigned long long raw = *(LdsWord*)&lds[(pos < kLds ? pos : kLds - 1) * kTraceBlock];
        uint2 ent = make_uint2((uint32_t)raw, (uint32_t)(raw >> 32));
        if (pos >= kLds) ent = wt.spill[(size_t)(pos - kLds) * wt.spill_stride + spill_lane];
        return ent;
    };
    // A finished ray's result stays in the lane's registers (the lane is idle) until the wave next refills, and is
    // stored then, by all the lanes that finished meanwhile at once: with 64 rays per wave some lane finishes in most
    // record steps, and storing there cost every step the store sequence at one or two lanes.
    // Postponed leaf (one-level scenes): a lane that reaches a candidate leaf does not stop for the wave's next leaf phase
    // but keeps it in `pend` (< 0: a leaf reference, >= 0: none) and walks on — with the t_max it has, which the postponed
    // leaf may yet shorten: what it walks meanwhile is a superset of what the reference walks, and every leaf is still
    // confirmed with the t_max current when its turn comes. Turns are kept: the postponed leaf is tested before any leaf
    // found after it; a lane that finds a second one waits, as every lane did before. More lanes step records, and the
    // leaf phase finds more lanes with a leaf to test.
    constexpr bool SPEC = PB_WIDE_SPECULATE && INST == 0;
    int pend = 0;
    auto finish = [&](bool found) {
        cur = found ? kDoneHit : kDoneMiss;
        if (SPEC) pend = 0;
    };
    auto flush_result = [&]() {
        if (cur == kDoneHit || cur == kDoneMiss) {
            io.store(index, any, cur == kDoneHit, tmax, hb0, hb1, hb2, hit_slot, INST ? hit_inst : -1);
            cur = kIdle;
        }
    };
    // second half (primitive.rs:140-143): r.t_max = ray.t_max on a hit; back to the world ray, which waited in LDS with its
    // reciprocal direction (the same three quotients set_ray_constants computed when the ray was fetched)
    auto exit_instance = [&]() {
        if (hit_inst == cur_top_slot) tmax_world = tmax;
        const LdsFloat* w = (const LdsFloat*)lds_world;
        r.ox = w[0 * kTraceBlock];
        r.oy = w[1 * kTraceBlock];
        r.oz = w[2 * kTraceBlock];
        r.dx = w[3 * kTraceBlock];
        r.dy = w[4 * kTraceBlock];
        r.dz = w[5 * kTraceBlock];
        idx = w[6 * kTraceBlock];
        idy = w[7 * kTraceBlock];
        idz = w[8 * kTraceBlock];
        nx = idx < 0.0f;
        ny = idy < 0.0f;
        nz = idz < 0.0f;
        negmask = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
        r.tmax = tmax_world;
        tmax = tmax_world;
        base_sp = -1;
    };

    // A lane that ran out of children (need_pop) takes ONE stack entry at the top of the next record iteration, together
    // with the other such lanes: the next entry whose lower bound is still in front of the hit becomes `cur`; an entry
    // behind the hit costs the lane that iteration. Inside an instance the stack ends at the floor laid down on entry:
    // the lane then parks on kLeaveInstance and leaves the instance in the leaf phase (primitive.rs:140-143), where the
    // rest of the top-level leaf is taken up again. (The loop this replaces popped until it found a live entry, at one
    // or two lanes per iteration, with the exec-mask bookkeeping of a divergent loop in every record step.)
    auto pop_one = [&]() {
        if (INST && sp <= base_sp) {
            cur = (SPEC && pend < 0) ? kWait : kLeaveInstance;  // the postponed leaf belongs to this instance: test it first
        } else if (sp == 0) {
            if (SPEC && pend < 0) {  // nothing left to walk but the postponed leaves: wait for the leaf phase
                cur = kWait;
            } else {
                finish(hit_slot >= 0);
            }
        } else {
            --sp;
            uint2 ent = stack_read(sp);
            if (__uint_as_float(ent.y) < tmax) cur = (int)ent.x;
        }
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
    // TransformedPrimitive::intersect, first half (primitive.rs:136-139), on the entry's loaded record: the top slot it names,
    // its object word and the three world-to-object rows. 0 = the ray misses the object's root box, 1 = entered, 2 = the
    // object-space ray is outside what the filter's bound covers (the ray leaves the wide path)
    auto enter_instance = [&](const int top_slot, const int object, const float4 r0, const float4 r1, const float4 r2) -> int {
        const float x = r.ox, y = r.oy, z = r.oz, wdx = r.dx, wdy = r.dy, wdz = r.dz;  // the lane holds the world ray here
        float ox = r0.x * x + r0.y * y + r0.z * z + r0.w;
        float oy = r1.x * x + r1.y * y + r1.z * z + r1.w;
        float oz = r2.x * x + r2.y * y + r2.z * z + r2.w;
        float xa = __builtin_fabsf(r0.x * x) + __builtin_fabsf(r0.y * y) + __builtin_fabsf(r0.z * z) + __builtin_fabsf(r0.w);
        float ya = __builtin_fabsf(r1.x * x) + __builtin_fabsf(r1.y * y) + __builtin_fabsf(r1.z * z) + __builtin_fabsf(r1.w);
        float za = __builtin_fabsf(r2.x * x) + __builtin_fabsf(r2.y * y) + __builtin_fabsf(r2.z * z) + __builtin_fabsf(r2.w);
        float ex = xa * kGamma3, ey = ya * kGamma3, ez = za * kGamma3;
        float dx = r0.x * wdx + r0.y * wdy + r0.z * wdz;
        float dy = r1.x * wdx + r1.y * wdy + r1.z * wdz;
        float dz = r2.x * wdx + r2.y * wdy + r2.z * wdz;
        float l2 = dx * dx + dy * dy + dz * dz;
        float tm = tmax_world;
        if (l2 > 0.0f) {
            float dt = (__builtin_fabsf(dx) * ex + __builtin_fabsf(dy) * ey + __builtin_fabsf(dz) * ez) / l2;
            ox = ox + dx * dt;
            oy = oy + dy * dt;
            oz = oz + dz * dt;
            tm -= dt;
        }
        const TravRay ro{ox, oy, oz, dx, dy, dz, tm};
        const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;  // bvh.rs:831
        if (!wide_ray_covered(ox, oy, oz, ix, iy, iz)) return 2;
        const bool bx = ix < 0.0f, by = iy < 0.0f, bz = iz < 0.0f;
        float mnx, mny, mnz, mxx, mxy, mxz;
        int root;
        if (INST == 2) {
            const float4* ob = wt.objects + 2 * (size_t)object;
            const float4 o0 = ob[0], o1 = ob[1];
            mnx = o0.x, mny = o0.y, mnz = o0.z, mxx = o1.x, mxy = o1.y, mxz = o1.z;
            root = __float_as_int(o0.w);
        } else {
            mnx = wt.obj0_min[0], mny = wt.obj0_min[1], mnz = wt.obj0_min[2];
            mxx = wt.obj0_max[0], mxy = wt.obj0_max[1], mxz = wt.obj0_max[2];
            root = wt.obj0_root;
        }
        float e;
        // the object aggregate's own root box (bvh.rs:841-842): a leaf box passing implies it passes, so failing it ends
        // the visit, and the lane keeps the world ray it holds
        if (!slab_test(bx ? mxx : mnx, bx ? mnx : mxx, by ? mxy : mny, by ? mny : mxy, bz ? mxz : mnz, bz ? mnz : mxz, ro, ix, iy, iz, tm, &e))
            return 0;
        r = ro;
        tmax = tm;
        idx = ix;
        idy = iy;
        idz = iz;
        nx = bx;
        ny = by;
        nz = bz;
        negmask = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
        cur_top_slot = top_slot;
        base_sp = sp;
        cur = root;
        return 1;
    };
    for (;;) {
        // ---------------- refill idle lanes (as trace_persistent.h) ----------------
        PB_WCLOCK(t_refill0);
        unsigned long long idle_mask = __ballot(is_idle());
        int n_idle = popc64(idle_mask);
        if (!exhausted && (n_idle >= PB_WIDE_REFILL_THRESH)) {
            PB_WSTAT(9, 1);  // refills
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
            uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0));
            flush_result();
            bool take = is_idle() && prefix < avail;
            PB_WSTAT(10, popc64(__ballot(take)));  // rays fetched
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
                if (SPEC) pend = 0;
                if (!real) {
                    finish(false);  // placeholder of a path outside pixel_bounds: not a ray of the frame
                } else {
                    set_ray_constants();
                    if (INST) {
                        LdsFloat* w = (LdsFloat*)lds_world;
                        w[0 * kTraceBlock] = r.ox;
                        w[1 * kTraceBlock] = r.oy;
                        w[2 * kTraceBlock] = r.oz;
                        w[3 * kTraceBlock] = r.dx;
                        w[4 * kTraceBlock] = r.dy;
                        w[5 * kTraceBlock] = r.dz;
                        w[6 * kTraceBlock] = idx;
                        w[7 * kTraceBlock] = idy;
                        w[8 * kTraceBlock] = idz;
                        tmax_world = r.tmax;
                        base_sp = -1;
                        cur_top_slot = -2;
                        hit_inst = -1;
                        leaf_state = 0;
                    }
                    bool covered = wide_ray_covered(r.ox, r.oy, r.oz, idx, idy, idz);
                    special = !covered;
                    if (COUNT && special) c_special += 1;
                    cur = wt.root_ref;
                    if (special) cur = kIdle;  // traced by the binary kernel afterwards (results written there)
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
        PB_WCLOCK(t_rec0);
        PB_WSTAT(13, t_rec0 - t_refill0);
        for (;;) {
            if (SPEC && pend >= 0 && is_leaf_ref()) {  // postpone the leaf, walk on
                pend = cur;
                cur = kNeedPop;
            }
            if (cur == kNeedPop) pop_one();
            bool interior = cur >= 0;
            int n_int = popc64(__ballot(cur >= 0) | __ballot(cur == kNeedPop));  // two compare masks OR-ed on the scalar unit
            if (n_int == 0) break;
            if (n_int < (INST ? PB_WIDE_INST_INTERIOR_THRESH : PB_WIDE_INTERIOR_THRESH)) {
                // two-level scenes: lanes waiting to enter / leave an instance (the expensive, rarely taken branch of the
                // leaf phase) only count once PB_WIDE_INST_GATHER of them wait: the branch then runs for that many lanes
                bool leaf_pending = INST ? (__any((is_leaf_ref() && base_sp >= 0) || (SPEC && cur == kWait)) ||
                                            popc64(__ballot(cur == kLeaveInstance || (is_leaf_ref() && base_sp < 0))) >= PB_WIDE_INST_GATHER)
                                         : __any(is_leaf_ref() || (SPEC && cur == kWait));
                bool can_refill = !exhausted && (popc64(__ballot(is_idle())) >= PB_WIDE_REFILL_THRESH);
                if (leaf_pending || can_refill) break;
            }
            PB_WSTAT(0, 1);      // record iterations of this wave
            PB_WSTAT(1, popc64(__ballot(interior)));  // lanes stepping a record
            if (interior) {
                const uint4* nd = wt.nodes + wide_vec_offset(cur);
                uint4 q0 = nd[0], q1 = nd[1], q2 = nd[2];
                if (COUNT) c_rec += 1;
                const uint32_t dw3 = q0.w;
                const WideSetup ws = wide_setup(q0.x, q0.y, q0.z, dw3, r.ox, r.oy, r.oz, idx, idy, idz);
                // The reference's visiting order of the four slots is two levels of dir_is_neg[axis] (bvh.rs:857-865): the
                // first binary child's slots (0, 1) before the second's (2, 3) unless the root axis is negative, and the
                // like inside each pair. `sel` holds, per byte, the slot visited k-th; the plane bytes and the descriptor
                // bytes are put into that order once (v_perm_b32), so that everything below is indexed by rank.
                const uint32_t f_root = (negmask >> ((dw3 >> 18) & 3u)) & 1u;
                const uint32_t f_c0 = (negmask >> ((dw3 >> 20) & 3u)) & 1u, f_c1 = (negmask >> ((dw3 >> 22) & 3u)) & 1u;
                uint32_t sel = 0x03020100u ^ (f_c0 ? 0x00000101u : 0u) ^ (f_c1 ? 0x01010000u : 0u);
                sel = __builtin_amdgcn_alignbit(sel, sel, f_root << 4);
                const uint32_t nqx = __builtin_amdgcn_perm(0u, nx ? q1.y : q1.x, sel), fqx = __builtin_amdgcn_perm(0u, nx ? q1.x : q1.y, sel);
                const uint32_t nqy = __builtin_amdgcn_perm(0u, ny ? q1.w : q1.z, sel), fqy = __builtin_amdgcn_perm(0u, ny ? q1.z : q1.w, sel);
                const uint32_t nqz = __builtin_amdgcn_perm(0u, nz ? q2.y : q2.x, sel), fqz = __builtin_amdgcn_perm(0u, nz ? q2.x : q2.y, sel);
                // m[0..3]: the low bytes of base.xyz and the top byte of dw3, gathered into one dword, in visiting order
                const uint32_t mslot = __builtin_amdgcn_perm(q0.y, q0.x, 0x0c0c0400u) | __builtin_amdgcn_perm(dw3, q0.z, 0x07000c0cu);
                const uint32_t mpack = __builtin_amdgcn_perm(0u, mslot, sel);
                const uint32_t child_base = q2.z, ntb = q2.w;
                float tn[4];
                bool h[4];
                int ref[4];
                // (two children per v_pk_fma_f32 was tried: 12 fewer VALU instructions per step, 1.5 % slower)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t m = (mpack >> (8 * k)) & 0xffu;
                    h[k] = wide_child_test(ws, nqx, nqy, nqz, fqx, fqy, fqz, k, tmax, &tn[k]) && (m != 0xffu);
                    ref[k] = (m & 0x80u) ? (int)(child_base + (m & 3u)) : (int)(ntb - m);
                }
#ifdef PB_LANE_STATS
                wl_steps += 1;
                wl_children += (h[0] ? 1 : 0) + (h[1] ? 1 : 0) + (h[2] ? 1 : 0) + (h[3] ? 1 : 0);
#endif
                // the children the reference visits later go on the stack, last one deepest
                const bool p3 = h[3] && (h[0] || h[1] || h[2]), p2 = h[2] && (h[0] || h[1]), p1 = h[1] && h[0];
                if (!__any(sp > kLds - 3)) {
                    // all three fit the LDS part for every lane of the wave (nearly always): store unconditionally, a
                    // lane that does not push just leaves its stack pointer where it was (the entry above the top of a
                    // stack is never read). No exec-mask bookkeeping: the three predicated pushes below cost about as
                    // many scalar instructions as vector ones.
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

        // ---------------- leaves: the reference's box test on the exact leaf box, then its triangles ----------------
        PB_WCLOCK(t_leaf0);
        PB_WSTAT(14, t_leaf0 - t_rec0);
#ifdef PB_LANE_STATS
        {
            unsigned long long lm = __ballot(is_leaf_ref() || (INST && cur == kLeaveInstance));
            if (lm) {
                PB_WSTAT(2, 1);             // leaf sections
                PB_WSTAT(3, popc64(lm));  // lanes with a candidate leaf
            }
            PB_WSTAT(6, 1);  // outer iterations
            PB_WSTAT(7, popc64(__ballot(!is_idle())));
        }
#endif
        // (see the record loop: this branch waits for PB_WIDE_INST_GATHER lanes unless nothing else is left to do)
        bool inst_turn = false;
        if (INST) {
            const bool waits = cur == kLeaveInstance || (is_leaf_ref() && base_sp < 0);
            const bool others = __any(cur >= 0 || cur == kNeedPop || cur == kWait || (is_leaf_ref() && base_sp >= 0));
            inst_turn = waits && (!others || popc64(__ballot(waits)) >= PB_WIDE_INST_GATHER);
        }
        PB_WCLOCK(t_inst0);
        if (INST && inst_turn) {
            // Either the lane leaves an instance (and takes up what is left of the top-level leaf it was in) or it holds a
            // fresh top-level leaf. Both need the record of the leaf's next entry; a fresh leaf also needs the leaf's exact
            // box, which is part of every entry's record: ONE round trip to memory before anything is decided.
            const bool fresh = base_sp < 0;
            if (!fresh) {
#ifdef PB_LANE_STATS
                wl_exit += 1;
#endif
                exit_instance();
            } else {
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
#ifdef PB_LANE_STATS
                wl_gate += 1;
                wl_gate_pass += walk ? 1 : 0;
#endif
            }
#ifdef PB_LANE_STATS
            {
                const unsigned long long act = __ballot(true);
                if (lane == __builtin_ctzll(act)) {
                    atomicAdd(&g_wide_stats[21], 1ull);                               // instance sections
                    atomicAdd(&g_wide_stats[22], (unsigned long long)popc64(act));  // lanes in them
                }
            }
#endif
            bool entered = false, done = false, abandon = false;
            while (walk) {
                leaf_state += 8 - 1;  // this entry is taken: next position, one entry less
                const int top_slot = __float_as_int(e0.w), word = __float_as_int(e1.w);
                if (INST == 2 && (word & 0x40000000)) {
                    // a GeometricPrimitive beside the instances (primitive.rs:65-78): the world ray against its triangle
                    const int tslot = word & 0x3fffffff;
                    const float4* tp = wt.slot_tris + 3 * (size_t)tslot;
                    const float4 ta = tp[0], tb = tp[1], tc = tp[2];
                    float b0, b1, b2, t;
                    const TriRayConst c = tri_ray_setup(r, idx, idy, idz);
                    if (triangle_test(V3{ta.x, ta.y, ta.z}, V3{ta.w, tb.x, tb.y}, V3{tb.z, tb.w, tc.x}, r, c, tmax, &b0, &b1, &b2, &t)) {
                        if (any) {
                            done = !(io.strict(index) && (__float_as_int(tc.w) & kTriDegenerate));
                        } else if (!(__float_as_int(tc.w) & kTriDegenerate)) {
                            if (t > tmax) abandon = true;  // t_max would move UP: see `raised` in the triangle branch
                            tmax = t;
                            tmax_world = t;
                            hb0 = b0;
                            hb1 = b1;
                            hb2 = b2;
                            hit_slot = tslot;
                            hit_inst = -1;
                        }
                    }
                } else {
                    const int how = enter_instance(top_slot, word, m0, m1, m2);
#ifdef PB_LANE_STATS
                    wl_try += 1;
                    wl_entered += how == 1 ? 1 : 0;
                    if (lane == __builtin_ctzll(__ballot(true))) atomicAdd(&g_wide_stats[23], 1ull);  // entry-loop trips (wave level)
#endif
                    entered = how == 1;
                    abandon = how == 2;
                }
                walk = (leaf_state & 7) != 0 && !entered && !done && !abandon;
                if (walk) {  // the leaf's next entry (leaves of more than one primitive are rare in SAH trees over instances)
                    const float4* ep = wt.top_slots + 5 * (size_t)(leaf_state >> 3);
                    e0 = ep[0];
                    e1 = ep[1];
                    m0 = ep[2];
                    m1 = ep[3];
                    m2 = ep[4];
                }
            }
            if (abandon) {
                // left to the binary kernel, which traces the ray from scratch (rare: one list append per such ray)
                wt.special_list[atomicAdd(wt.special_count, 1u)] = index;
                cur = kIdle;
            } else if (done) {
                finish(true);
            } else if (!entered) {
                cur = kNeedPop;
            }
        }
        PB_WCLOCK(t_tri0);
        PB_WSTAT(24, t_tri0 - t_inst0);  // cycles in the instance branch
        if (!(INST && inst_turn) && ((SPEC && pend < 0) || (is_leaf_ref() && (!INST || base_sp >= 0)))) {
            const bool from_pend = SPEC && pend < 0;  // the postponed leaf first; a leaf in `cur` then waits for the next phase
            const int v = ~(from_pend ? pend : cur);
            const int cnt = (v & 3) + 1;
            const int first = v >> 2;
            if (COUNT) c_cand += 1;
            // The leaf's first triangle is fetched whatever the box test will say, and a leaf of several triangles fetches
            // its exact box in the same round trip (before round 3: the box first, then the triangles, one trip each).
            if (COUNT) c_tri += 1;
            const float4* tp0 = wt.tris + wide_vec_offset(first);
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
            bool pass = slab_test(nx ? hix : lox, nx ? lox : hix, ny ? hiy : loy, ny ? loy : hiy, nz ? hiz : loz, nz ? loz : hiz, r,
                                  idx, idy, idz, tmax, &entry);
            bool done = false;
            // Everything this kernel is (wide_bvh.h) rests on ray.t_max only ever SHRINKING. It nearly does: the range test of
            // triangle.rs:127-130 compares t_scaled with t_max * det, the quotient t = t_scaled / det is rounded once more and
            // can come out an ulp ABOVE the t_max it was accepted under (two hits within an ulp of each other: a ray through a
            // vertex several triangles share). The reference then carries on with the larger t_max, and a node it pruned
            // before may count again, or one it would have pruned may not. A ray to which that happens leaves this kernel: the
            // binary kernel, whose every box test is the reference's at the reference's moment, traces it from scratch.
            bool raised = false;
#ifdef PB_LANE_STATS
            wl_cand += 1;
            if (pass) {
                wl_pass += 1;
                wl_tris += cnt;
            }
#endif
#ifdef PB_LANE_STATS
            {
                const int trips = (__any(pass && cnt >= 1) ? 1 : 0) + (__any(pass && cnt >= 2) ? 1 : 0) + (__any(pass && cnt >= 3) ? 1 : 0) + (__any(pass && cnt >= 4) ? 1 : 0);
                const int nl = popc64(__ballot(true));
                if (lane == __builtin_ctzll(__ballot(true))) {
                    atomicAdd(&g_wide_stats[26], 1ull);                      // triangle sections
                    atomicAdd(&g_wide_stats[27], (unsigned long long)trips);  // triangle-loop trips (wave level)
                    atomicAdd(&g_wide_stats[28], (unsigned long long)nl);     // lanes in them
                }
            }
#endif
            if (pass) {
                const TriRayConst trc = tri_ray_setup(r, idx, idy, idz);
                for (int i = 0; i < cnt; ++i) {
                    if (i > 0) {
                        if (COUNT) c_tri += 1;
                        const float4* tp = wt.tris + wide_vec_offset(first + i);
                        ta = tp[0];
                        tb = tp[1];
                        tc = tp[2];
                    }
                    float b0, b1, b2, t;
                    bool hit = triangle_test(V3{ta.x, ta.y, ta.z}, V3{ta.w, tb.x, tb.y}, V3{tb.z, tb.w, tc.x}, r, trc, tmax, &b0, &b1,
                                             &b2, &t);
                    if (hit) {
                        if (any) {
                            if (io.strict(index) && (__float_as_int(tc.z) & kTriDegenerate)) continue;  // (IO::strict, trace_persistent.h)
                            done = true;
                            hit_slot = __float_as_int(tc.y);
                            break;
                        }
                        if (!(__float_as_int(tc.z) & kTriDegenerate)) {
                            if (t > tmax) raised = true;
                            tmax = t;  // primitive.rs:70
                            hb0 = b0;
                            hb1 = b1;
                            hb2 = b2;
                            hit_slot = __float_as_int(tc.y);
                            if (INST) {
                                hit_inst = cur_top_slot;
                            }
                        }
                    }
                }
            }
            if (raised) {
                wt.special_list[atomicAdd(wt.special_count, 1u)] = index;  // (one list append per such ray: they are rare)
                if (COUNT) c_special += 1;
                cur = kIdle;
                if (SPEC) pend = 0;
            } else if (done)
                finish(true);
            else if (from_pend) {
                pend = 0;
                if (cur == kWait) cur = kNeedPop;  // (the pop finds the stack empty and finishes the ray)
            } else
                cur = kNeedPop;
        }
#ifdef PB_LANE_STATS
        wstat[15] += __builtin_readcyclecounter() - t_leaf0;
        wstat[25] += __builtin_readcyclecounter() - t_tri0;  // cycles in the triangle branch
#endif
    }
    flush_result();
    if (COUNT) count_flush(counters + 4, c_rec, c_cand, c_tri, c_special);
#ifdef PB_LANE_STATS
    {
        unsigned long long v[10] = {wl_steps, wl_children, wl_cand, wl_pass, wl_tris, wl_gate, wl_gate_pass, wl_try, wl_entered, wl_exit};
        for (int k = 0; k < 10; ++k)
            for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
        wstat[8] = v[1];
        wstat[11] = v[0];
        wstat[12] = v[2];
        wstat[4] = v[3];
        wstat[5] = v[4];
        wstat[16] = v[5];
        wstat[17] = v[6];
        wstat[18] = v[7];
        wstat[19] = v[8];
        wstat[20] = v[9];
        if (lane == 0)
            for (int i = 0; i < 32; ++i) atomicAdd(&g_wide_stats[i], wstat[i]);
    }
#endif
}

}  // namespace pb

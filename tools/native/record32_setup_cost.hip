// instruction-count model of the per-step frame work: (A) the shipped record's setup from its stored frame, (B) the derived
// frame of a 32-byte record (descent) and (C) its reconstruction from a stack frame entry at a pop
#include <hip/hip_runtime.h>
#include "../../pbrt-rs_amd/csrc/wide_bvh.h"
using namespace pb;
struct Out { float v[9]; };
// (A) shipped: base.xyz + packed exponents -> S, An, Af
extern "C" __global__ void k_setup_stored(const uint4* recs, const float* ray, Out* out) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    uint4 r = recs[i];
    WideSetup w = wide_setup(r.x, r.y, r.z, r.w, ray[0], ray[1], ray[2], ray[3], ray[4], ray[5]);
    Out o = {{w.Sx, w.Sy, w.Sz, w.Anx, w.Afx, w.Any, w.Afy, w.Anz, w.Afz}};
    out[i] = o;
}
// (B) derived at a descent: parent's S.xyz, A.xyz (pads per ray) + the chosen child's q_lo, q_hi bytes (already in registers,
// byte 0 of six dwords after the rank permutation)
extern "C" __global__ void k_setup_derived(const float* st, const uint32_t* q, const float* pad, Out* out) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    float S[3] = {st[6 * i], st[6 * i + 1], st[6 * i + 2]}, A[3] = {st[6 * i + 3], st[6 * i + 4], st[6 * i + 5]};
    Out o;
    for (int k = 0; k < 3; ++k) {
        uint32_t lo = q[6 * i + 2 * k] & 0xffu, hi = q[6 * i + 2 * k + 1] & 0xffu;
        uint32_t d = hi - lo;
        d = d ? d : 1u;
        int sh = (int)__builtin_clz(d) - 24;           // (d << sh) in [128, 255]
        sh = sh > 3 ? 3 : sh;
        float A1 = __builtin_fmaf((float)lo, S[k], A[k]);
        float S1 = __builtin_ldexpf(S[k], -sh);
        o.v[k] = S1;
        o.v[3 + 2 * k] = A1 - pad[k];
        o.v[4 + 2 * k] = A1 + pad[k];
    }
    out[i] = o;
}
// (C) at a pop: frame entry (A.xyz of the PARENT + packed exponents) and the child's entry (q_lo.xyz, shifts) -> the child's frame
extern "C" __global__ void k_setup_pop(const float4* frames, const uint32_t* child, const float* ray, const float* pad, Out* out) {
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    float4 f = frames[i];
    uint32_t e = __float_as_uint(f.w), c = child[i];
    float A[3] = {f.x, f.y, f.z};
    Out o;
    for (int k = 0; k < 3; ++k) {
        float S = __builtin_ldexpf(ray[3 + k], ((int)(e << (26 - 6 * k))) >> 26);
        uint32_t lo = (c >> (8 * k)) & 0xffu;
        int sh = (c >> (24 + 2 * k)) & 3;
        float A1 = __builtin_fmaf((float)lo, S, A[k]);
        o.v[k] = __builtin_ldexpf(S, -sh);
        o.v[3 + 2 * k] = A1 - pad[k];
        o.v[4 + 2 * k] = A1 + pad[k];
    }
    out[i] = o;
}

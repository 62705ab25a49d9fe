// Microbenchmark behind DESIGN.md's "what bounds k_trace": 64-B records fetched at random from a 64 MB table
// (the shape of the interior-node fetch), three ways:
//   A  every lane fetches its own record with four 16-B loads (what k_trace does)
//   B  the four lanes of a quad fetch one record together, 16 B each, four records in turn (one cache-line access
//      serves four lanes); the data stay where they land (upper bound for the access path alone)
//   C  as B, then the pieces are handed to their owners through LDS (write b128 x4, read b128 x4)
// Build: hipcc -O3 --offload-arch=gfx950 -o probe_gather probe_gather.hip ; run: ./probe_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

constexpr int kIters = 256;

__device__ __forceinline__ uint32_t next_index(uint32_t x, uint32_t mask) { return (x * 1664525u + 1013904223u) & mask; }
__device__ __forceinline__ float sum4(float4 v) { return v.x + v.y + v.z + v.w; }

template <int MODE>
__global__ void __launch_bounds__(256, 6) k_gather(const float4* __restrict__ table, uint32_t mask, float* out) {
    __shared__ float4 stage[4 * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    float acc = 0.0f;
    float4* my_stage = stage + wave * 256;
    for (int it = 0; it < kIters; ++it) {
        idx = next_index(idx + (uint32_t)__float_as_uint(acc) * 0u, mask);
        if (MODE == 0) {
            const float4* r = table + 4 * (size_t)idx;
            float4 a = r[0], b = r[1], c = r[2], d = r[3];
            acc += sum4(a) + sum4(b) + sum4(c) + sum4(d);
        } else {
            float4 piece[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t n = (uint32_t)__shfl((int)idx, (lane & ~3) | k, 64);
                piece[k] = table[4 * (size_t)n + (lane & 3)];
            }
            if (MODE == 1) {
                acc += sum4(piece[0]) + sum4(piece[1]) + sum4(piece[2]) + sum4(piece[3]);
            } else {
                // piece[k] of lane 4q+j = part j of the record of lane 4q+k  ->  owner 4q+k reads parts 0..3
#pragma unroll
                for (int k = 0; k < 4; ++k) my_stage[((lane & ~3) | k) * 4 + (lane & 3)] = piece[k];
                float4 a = my_stage[lane * 4], b = my_stage[lane * 4 + 1], c = my_stage[lane * 4 + 2], d = my_stage[lane * 4 + 3];
                acc += sum4(a) + sum4(b) + sum4(c) + sum4(d);
                // make the next index depend on this record like a tree walk does
                idx += (uint32_t)(a.x > 2.0f);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  for (int lg : {8, 12, 15, 20}) {  // 16 KB (L1), 256 KB, 2 MB (L2), 64 MB (Infinity Cache)
    const uint32_t n_rec = 1u << lg;
    std::printf("table %u KB\n", n_rec * 64 / 1024);
    std::vector<float> h((size_t)n_rec * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    float4* table;
    float* out;
    const int blocks = 256 * 6 * 4;
    CK(hipMalloc(&table, h.size() * 4));
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* names[3] = {"A per-lane 4 x 16 B", "B quad-cooperative, no hand-over", "C quad-cooperative + LDS hand-over"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, table, n_rec - 1, out);
            if (mode == 1) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, table, n_rec - 1, out);
            if (mode == 2) hipLaunchKernelGGL(k_gather<2>, dim3(blocks), dim3(256), 0, 0, table, n_rec - 1, out);
            CK(hipGetLastError());
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double recs = (double)blocks * 256 * kIters;
        std::printf("%-36s %8.3f ms  %7.1f G records/s  %7.2f TB/s  %.2f records/cycle/CU at 2.4 GHz\n", names[mode], best,
                    recs / best * 1e-6, recs * 64 / best * 1e-9, recs / (best * 1e-3) / 256 / 2.4e9);
    }
    CK(hipFree(table));
    CK(hipFree(out));
  }
    return 0;
}

// probe.hip — pbrt_hip_probe_gather: the rate at which this GPU serves the traversal kernels' fetch pattern.
//
// Every lane walks its own chain of dependent record fetches (the next record index is computed from the bytes just
// loaded, as the next BVH record is), three 16-B loads for the 48-byte wide records (wide_bvh.h) or four for the 64-byte
// child-pair records (trace.h), from a table of a given size, at a given number of resident waves per SIMD and with
// nothing else to do. The result is a measured ceiling, for bench.py's roofline line (measurement only: no part of
// rendering calls this).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "scene.h"

namespace {

template <int LOADS, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_probe_gather(const uint4* __restrict__ table, uint32_t n_records, int iters, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_records;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint4* r = table + (size_t)LOADS * idx;
        uint4 a = r[0], b = r[1], c = r[2];
        uint32_t h = a.x ^ a.w ^ b.y ^ b.z ^ c.x ^ c.w;
        if (LOADS == 4) {
            uint4 d = r[3];
            h ^= d.y ^ d.z;
        }
        acc += h;
        idx = (h * 2654435761u + (uint32_t)it) % n_records;  // depends on every load of this step
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void k_probe_fill(uint4* table, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 747796405u + 2891336453u;
    auto step = [&]() {
        x ^= x >> 16;
        x *= 0x7feb352du;
        x ^= x >> 15;
        x *= 0x846ca68bu;
        x ^= x >> 16;
        return x;
    };
    table[i] = make_uint4(step(), step(), step(), step());
}

template <int LOADS>
void launch(int waves, dim3 grid, hipStream_t st, const uint4* table, uint32_t n_records, int iters, uint32_t* out) {
    switch (waves) {
        case 4: hipLaunchKernelGGL((k_probe_gather<LOADS, 4>), grid, dim3(256), 0, st, table, n_records, iters, out); break;
        case 5: hipLaunchKernelGGL((k_probe_gather<LOADS, 5>), grid, dim3(256), 0, st, table, n_records, iters, out); break;
        case 6: hipLaunchKernelGGL((k_probe_gather<LOADS, 6>), grid, dim3(256), 0, st, table, n_records, iters, out); break;
        default: hipLaunchKernelGGL((k_probe_gather<LOADS, 8>), grid, dim3(256), 0, st, table, n_records, iters, out); break;
    }
}

}  // namespace

extern "C" int pbrt_hip_probe_gather(PbrtHipContext* ctx, int64_t table_bytes, int32_t record_bytes, int32_t waves_per_simd,
                                     int32_t iters, double* records_per_second) {
    if (!ctx || !records_per_second || (record_bytes != 48 && record_bytes != 64) || table_bytes < record_bytes || iters <= 0 ||
        table_bytes > (64ll << 30))
        return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int waves = (waves_per_simd == 4 || waves_per_simd == 5 || waves_per_simd == 6) ? waves_per_simd : 8;
    const uint32_t n_records = (uint32_t)(table_bytes / record_bytes);
    const size_t n_vec = (size_t)n_records * (record_bytes / 16);
    const int blocks = ctx->n_cus * waves;  // 4 SIMDs x waves / 4 waves per block: every block resident, every SIMD at `waves`
    uint4* table = nullptr;
    uint32_t* out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&table, n_vec * sizeof(uint4)));
    if (!pb::hip_ok(ctx, hipMalloc((void**)&out, (size_t)blocks * 256 * sizeof(uint32_t)), "hipMalloc")) {
        (void)hipFree(table);
        return PBRT_HIP_ERR_DEVICE;
    }
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_probe_fill, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, st, table, n_vec);
    int rc = PBRT_HIP_OK;
    float best = 1e30f;
    for (int rep = 0; rep < 4 && rc == PBRT_HIP_OK; ++rep) {  // the first repetition warms the caches
        if (!pb::hip_ok(ctx, hipEventRecord(ctx->ev0, st), "hipEventRecord")) rc = PBRT_HIP_ERR_DEVICE;
        if (record_bytes == 48)
            launch<3>(waves, dim3(blocks), st, table, n_records, iters, out);
        else
            launch<4>(waves, dim3(blocks), st, table, n_records, iters, out);
        if (!pb::hip_ok(ctx, hipGetLastError(), "probe launch") || !pb::hip_ok(ctx, hipEventRecord(ctx->ev1, st), "hipEventRecord") ||
            !pb::hip_ok(ctx, hipEventSynchronize(ctx->ev1), "hipEventSynchronize"))
            rc = PBRT_HIP_ERR_DEVICE;
        float ms = 0.0f;
        if (rc == PBRT_HIP_OK && !pb::hip_ok(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1), "hipEventElapsedTime")) rc = PBRT_HIP_ERR_DEVICE;
        if (rc == PBRT_HIP_OK && rep > 0 && ms < best) best = ms;
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(table);
    (void)hipFree(out);
    if (rc == PBRT_HIP_OK) *records_per_second = (double)blocks * 256.0 * (double)iters / ((double)best * 1e-3);
    return rc;
}

// probe.hip — pbrt_hip_probe_gather: the rate at which this GPU serves the traversal kernels' fetch pattern.
//
// Every lane walks its own chain of dependent record fetches (the next record index is computed from the bytes just
// loaded, as the next BVH record is), three 16-B loads for the 48-byte wide records (wide_bvh.h) or four for the 64-byte
// child-pair records (trace.h), from a table of a given size, at a given number of resident waves per SIMD and with
// nothing else to do. The result is a measured ceiling, for bench.py's roofline line (measurement only: no part of
// rendering calls this).
#include <hip/hip_runtime.h>
#include "abi_guard.h"

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
                                     int32_t iters, double* records_per_second) try {
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
PB_ABI_CATCH

// ------------------------------------------------------------------------------------
// pbrt_hip_probe_state_stream: k_shade's access pattern with a known byte count, for calibrating rocprofv3's FETCH_SIZE /
// WRITE_SIZE on it (profiles/r04_fetch_size_calibration_shade.txt; the guide: FETCH_SIZE reads 1/2 for wide coalesced streams and
// is uncalibrated for other widths). A shade queue of ascending path numbers (`density` of all paths), and per queued path what
// k_shade touches in the SoA path state (wf_state.h): 16-B records (L, beta, hit records, the pending estimate), 32-B ray records
// at a 32-B stride in slot-major arrays, 8-B and 4-B scalars, one random 48-B triangle; stores of the same widths back, and
// 4-B queue entries at compacted positions. Measurement only.
// ------------------------------------------------------------------------------------
namespace {
constexpr int kProbeVec16 = 9, kProbeRaySlots = 2, kProbeWriteVec16 = 5, kProbeWriteRaySlots = 3, kProbeQueueWords = 7;
struct StateStreamArrays {
    const uint32_t* queue;
    float4* v16;       // [k][path]
    float4* ray;       // [slot][path][2]
    uint64_t* s8;      // [path]
    uint32_t* s4;      // [2][path]
    const float4* tris;  // 3 x float4 per record
    uint32_t n_tris;
    uint32_t* out_queue;  // [kProbeQueueWords][i]
    size_t n_paths;
};
__global__ void __launch_bounds__(256) k_probe_state_stream(StateStreamArrays a, uint32_t n, int parts) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = a.queue[i];
    float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    auto add = [&](const float4 v) {
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
    };
    if (parts & 1)
        for (int k = 0; k < kProbeVec16; ++k) add(a.v16[(size_t)k * a.n_paths + p]);
    if (parts & 2)
        for (int s = 0; s < kProbeRaySlots; ++s) {
            const float4* r = a.ray + ((size_t)s * a.n_paths + p) * 2;
            add(r[0]);
            add(r[1]);
        }
    uint64_t r8 = 0;
    uint32_t r4 = 0;
    if (parts & 4) {
        r8 = a.s8[p];
        r4 = a.s4[p] + a.s4[a.n_paths + p];
    }
    if (parts & 8) {
        const uint32_t t = (uint32_t)(((uint64_t)(p * 2654435761u + 12345u) * a.n_tris) >> 32);
        const float4* tp = a.tris + 3 * (size_t)t;
        add(tp[0]);
        add(tp[1]);
        add(tp[2]);
    }
    acc.x += (float)(r8 & 0xff) + (float)(r4 & 0xff);
    if (parts & 16) {
        for (int k = 0; k < kProbeWriteVec16; ++k) a.v16[(size_t)k * a.n_paths + p] = acc;
        for (int s = 0; s < kProbeWriteRaySlots; ++s) {
            float4* r = a.ray + ((size_t)s * a.n_paths + p) * 2;
            r[0] = acc;
            r[1] = acc;
        }
        a.s8[p] = r8 + 1;
        a.s4[p] = r4 + 1;
        a.s4[a.n_paths + p] = r4;
        for (int k = 0; k < kProbeQueueWords; ++k) a.out_queue[(size_t)k * n + i] = p + (uint32_t)k;
    } else if (acc.x == 12345.678f) {
        a.out_queue[i] = p;  // keeps the loads alive
    }
}
__global__ void k_probe_queue(uint32_t* queue, uint32_t keep_of_1024, uint32_t n_queue) {
    // queue entry i = the i-th kept path: paths are kept in a fixed pattern of keep_of_1024 per 1024 (ascending, as a shade queue is)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_queue) return;
    const uint32_t block = i / keep_of_1024, within = i % keep_of_1024;
    queue[i] = block * 1024u + (uint32_t)(((uint64_t)within * 1024u) / keep_of_1024);
}
}  // namespace

extern "C" int pbrt_hip_probe_state_stream(PbrtHipContext* ctx, int64_t n_paths, int32_t density_permille, int64_t gather_table_bytes,
                                           int32_t parts, int64_t* bytes_read, int64_t* bytes_written, double* ms_out) try {
    if (!ctx || n_paths < 1024 || n_paths > (1ll << 28) || density_permille < 1 || density_permille > 1000 || gather_table_bytes < 48 ||
        gather_table_bytes > (8ll << 30) || !bytes_read || !bytes_written)
        return PBRT_HIP_ERR_INVALID;
    PB_ENTER(ctx);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    n_paths &= ~1023ll;
    const uint32_t keep = (uint32_t)((1024ll * density_permille + 999) / 1000);
    const uint32_t n_queue = (uint32_t)(n_paths / 1024 * keep);
    StateStreamArrays a{};
    a.n_paths = (size_t)n_paths;
    a.n_tris = (uint32_t)(gather_table_bytes / 48);
    void* blocks[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t sizes[7] = {(size_t)n_queue * 4, (size_t)kProbeVec16 * n_paths * 16, (size_t)kProbeWriteRaySlots * n_paths * 32, (size_t)n_paths * 8,
                             (size_t)n_paths * 8, (size_t)a.n_tris * 48, (size_t)kProbeQueueWords * n_queue * 4};
    int rc = PBRT_HIP_OK;
    for (int k = 0; k < 7 && rc == PBRT_HIP_OK; ++k)
        if (!pb::hip_ok(ctx, hipMalloc(&blocks[k], sizes[k]), "hipMalloc (probe)")) rc = PBRT_HIP_ERR_OOM;
    hipStream_t st = ctx->stream;
    if (rc == PBRT_HIP_OK) {
        a.queue = (const uint32_t*)blocks[0];
        a.v16 = (float4*)blocks[1];
        a.ray = (float4*)blocks[2];
        a.s8 = (uint64_t*)blocks[3];
        a.s4 = (uint32_t*)blocks[4];
        a.tris = (const float4*)blocks[5];
        a.out_queue = (uint32_t*)blocks[6];
        for (int k = 1; k < 6; ++k) (void)hipMemsetAsync(blocks[k], 0, sizes[k], st);
        hipLaunchKernelGGL(k_probe_queue, dim3((n_queue + 255) / 256), dim3(256), 0, st, (uint32_t*)blocks[0], keep, n_queue);
        float best = 1e30f;
        for (int rep = 0; rep < 3 && rc == PBRT_HIP_OK; ++rep) {
            if (!pb::hip_ok(ctx, hipEventRecord(ctx->ev0, st), "hipEventRecord")) rc = PBRT_HIP_ERR_DEVICE;
            hipLaunchKernelGGL(k_probe_state_stream, dim3((n_queue + 255) / 256), dim3(256), 0, st, a, n_queue, parts);
            if (!pb::hip_ok(ctx, hipGetLastError(), "probe launch") || !pb::hip_ok(ctx, hipEventRecord(ctx->ev1, st), "hipEventRecord") ||
                !pb::hip_ok(ctx, hipEventSynchronize(ctx->ev1), "hipEventSynchronize"))
                rc = PBRT_HIP_ERR_DEVICE;
            float ms = 0.0f;
            if (rc == PBRT_HIP_OK && !pb::hip_ok(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1), "hipEventElapsedTime")) rc = PBRT_HIP_ERR_DEVICE;
            if (rc == PBRT_HIP_OK && ms < best) best = ms;
        }
        if (ms_out) *ms_out = best;
    }
    (void)hipStreamSynchronize(st);
    for (void* b : blocks)
        if (b) (void)hipFree(b);
    // bytes the lanes ask for, per launch
    int64_t rd = 4ll * n_queue, wr = 0;
    if (parts & 1) rd += 16ll * kProbeVec16 * n_queue;
    if (parts & 2) rd += 32ll * kProbeRaySlots * n_queue;
    if (parts & 4) rd += 16ll * n_queue;
    if (parts & 8) rd += 48ll * n_queue;
    if (parts & 16) wr += (16ll * kProbeWriteVec16 + 32ll * kProbeWriteRaySlots + 16ll + 4ll * kProbeQueueWords) * n_queue;
    *bytes_read = rd;
    *bytes_written = wr;
    return rc;
}
PB_ABI_CATCH

// Microbenchmark: does the ALIGNMENT of a 48-byte record matter to the fetch path? Every lane walks a chain of dependent
// record fetches (three 16-B loads per record, the next index computed from the bytes just loaded — the wide traversal's
// pattern) over the same NUMBER of records laid out (a) packed, 48-B stride: 23.4 MB for config 3's 487 623 records, one record
// in two straddling two 64-B lines; (b) one record per 64-B line, 64-B stride: 31.2 MB, never straddling. At 5 and 8 waves per
// SIMD, random start indices, and with a locality knob: with probability 1 - 1/2^k the next record lies within +-64 records
// of the current one (siblings and nearby subtrees, as sorted rays see them), else anywhere.
// Build: hipcc -O3 --offload-arch=gfx950 -o probe_stride probe_stride.hip ; run: ./probe_stride [n_records]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

constexpr int kIters = 512;

template <int STRIDE, int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_walk(const uint4* __restrict__ table, uint32_t n_records, int local_shift, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_records;
    uint32_t acc = 0;
    for (int it = 0; it < kIters; ++it) {
        const uint4* r = table + (size_t)STRIDE * idx;
        uint4 a = r[0], b = r[1], c = r[2];
        uint32_t h = a.x ^ a.w ^ b.y ^ b.z ^ c.x ^ c.w;
        acc += h;
        uint32_t g = h * 2654435761u + (uint32_t)it;
        const bool far = local_shift == 0 || (g >> (32 - local_shift)) == 0u;  // probability 2^-local_shift
        uint32_t near = idx + (g & 127u) + n_records - 64u;
        idx = (far ? (g >> 3) : near) % n_records;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void k_fill(uint4* table, size_t n) {
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

template <int STRIDE, int WAVES>
double run(const uint4* table, uint32_t n_records, int local_shift, uint32_t* out, int n_cus) {
    const int blocks = n_cus * WAVES;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_walk<STRIDE, WAVES>), dim3(blocks), dim3(256), 0, 0, table, n_records, local_shift, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    return (double)blocks * 256.0 * kIters / (best * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const uint32_t n_records = argc > 1 ? (uint32_t)std::atol(argv[1]) : 487623u;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    uint4 *packed, *aligned;
    uint32_t* out;
    CK(hipMalloc((void**)&packed, (size_t)n_records * 48));
    CK(hipMalloc((void**)&aligned, (size_t)n_records * 64));
    CK(hipMalloc((void**)&out, (size_t)n_cus * 8 * 256 * 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n_records * 3 + 255) / 256)), dim3(256), 0, 0, packed, (size_t)n_records * 3);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n_records * 4 + 255) / 256)), dim3(256), 0, 0, aligned, (size_t)n_records * 4);
    CK(hipDeviceSynchronize());
    std::printf("# %u records: packed %.1f MB (48-B stride), aligned %.1f MB (64-B stride); G records/s, three 16-B loads per record\n", n_records,
                n_records * 48e-6, n_records * 64e-6);
    for (int shift : {0, 2, 4}) {
        const double p5 = run<3, 5>(packed, n_records, shift, out, n_cus), a5 = run<4, 5>(aligned, n_records, shift, out, n_cus);
        const double p8 = run<3, 8>(packed, n_records, shift, out, n_cus), a8 = run<4, 8>(aligned, n_records, shift, out, n_cus);
        std::printf("far jumps %6.2f %% : 5 waves packed %7.2f aligned %7.2f (%+5.1f %%) | 8 waves packed %7.2f aligned %7.2f (%+5.1f %%)\n",
                    shift ? 100.0 / (1 << shift) : 100.0, p5, a5, 100.0 * (a5 / p5 - 1.0), p8, a8, 100.0 * (a8 / p8 - 1.0));
    }
    return 0;
}

// host_film.cpp — the host-only ends of the film path behind the C ABI (no device code, no HIP calls):
// Film::get_sample_bounds and the reconstruction-filter table (src/core/film.rs:52-81, src/filters/*.rs), the tile
// partition of SamplerIntegrator::render (src/core/integrator.rs:404-411), Film::write_image's XYZ -> RGB
// (src/core/film.rs:153-178) and the image writers the reference leaves as todo!() (src/core/imageio.rs:3-5).
#include <algorithm>
#include "abi_guard.h"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/pbrt_hip.h"

namespace {
constexpr int kTile = 16;  // integrator.rs:404 TILE_SIZE
constexpr float kPi = 3.14159265358979323846f;
}

// Film::get_sample_bounds (film.rs:76-81, D42 intended) of the whole film
extern "C" int pbrt_hip_sample_bounds(int32_t width, int32_t height, float rx, float ry, int32_t b[4]) try {
    if (!b || width <= 0 || height <= 0 || !(rx > 0.0f) || !(ry > 0.0f)) return PBRT_HIP_ERR_INVALID;
    b[0] = (int32_t)std::floor(0.0f + 0.5f - rx);
    b[1] = (int32_t)std::floor(0.0f + 0.5f - ry);
    b[2] = (int32_t)std::ceil((float)width - 0.5f + rx);
    b[3] = (int32_t)std::ceil((float)height - 0.5f + ry);
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

// Filter::evaluate of src/filters/*.rs tabulated as Film::new does (film.rs:52-63)
extern "C" int pbrt_hip_filter_table(int32_t type, float rx, float ry, float a, float b, float table[256]) try {
    if (!table || !(rx > 0.0f) || !(ry > 0.0f) || type < 0 || type > 4) return PBRT_HIP_ERR_INVALID;
    auto mitchell = [&](float x) {  // mitchell.rs:31-47
        x = std::fabs(2.0f * x);
        if (x > 1.0f)
            return ((-a - 6.0f * b) * x * x * x + (6.0f * a + 30.0f * b) * x * x + (-12.0f * a - 48.0f * b) * x +
                    (8.0f * a + 24.0f * b)) * (1.0f / 6.0f);
        return ((12.0f - 9.0f * a - 6.0f * b) * x * x * x + (-18.0f + 12.0f * a + 6.0f * b) * x * x + (6.0f - 2.0f * a)) *
               (1.0f / 6.0f);
    };
    auto sinc = [](float x) {  // sinc.rs:28-35
        x = std::fabs(x);
        return x < 1e-5f ? 1.0f : std::sin(kPi * x) / (kPi * x);
    };
    auto wsinc = [&](float x, float radius) {  // sinc.rs:37-45
        x = std::fabs(x);
        return x > radius ? 0.0f : sinc(x) * sinc(x / a);
    };
    int k = 0;
    for (int y = 0; y < 16; ++y)
        for (int x = 0; x < 16; ++x) {
            float px = ((float)x + 0.5f) * rx / 16.0f, py = ((float)y + 0.5f) * ry / 16.0f, v = 1.0f;
            switch (type) {
                case PBRT_FILTER_GAUSSIAN: {  // gaussian.rs:17-39
                    float ex = std::exp(-a * rx * rx), ey = std::exp(-a * ry * ry);
                    float gx = std::exp(-a * px * px) - ex, gy = std::exp(-a * py * py) - ey;
                    v = (gx > 0.0f ? gx : 0.0f) * (gy > 0.0f ? gy : 0.0f);
                    break;
                }
                case PBRT_FILTER_MITCHELL: v = mitchell(px * (1.0f / rx)) * mitchell(py * (1.0f / ry)); break;
                case PBRT_FILTER_LANCZOS: v = wsinc(px, rx) * wsinc(py, ry); break;
                case PBRT_FILTER_TRIANGLE: {  // triangle.rs:21-27
                    float tx = rx - std::fabs(px), ty = ry - std::fabs(py);
                    v = (tx > 0.0f ? tx : 0.0f) * (ty > 0.0f ? ty : 0.0f);
                    break;
                }
                default: v = 1.0f;  // boxf.rs:25-27
            }
            table[k++] = v;
        }
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_write_pfm(const char* path, const float* rgb, int32_t width, int32_t height) try {
    if (!path || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) return PBRT_HIP_ERR_INVALID;
    std::fprintf(f, "PF\n%d %d\n-1.0\n", width, height);  // negative scale = little endian
    for (int32_t y = height - 1; y >= 0; --y)               // PFM stores the bottom row first
        std::fwrite(rgb + (size_t)y * width * 3, sizeof(float), (size_t)width * 3, f);
    bool ok = std::fclose(f) == 0;
    return ok ? PBRT_HIP_OK : PBRT_HIP_ERR_INVALID;
}
PB_ABI_CATCH

// OpenEXR, the format pbrt-v3 writes by default: single-part scanline file, three 32-bit float channels (stored in
// alphabetical order B, G, R), no compression, one scanline per chunk. Linear values, nothing clamped.
namespace {
struct ExrBuf {
    std::vector<unsigned char> b;
    void bytes(const void* p, size_t n) { b.insert(b.end(), (const unsigned char*)p, (const unsigned char*)p + n); }
    void str(const char* s) { bytes(s, std::strlen(s) + 1); }
    void i32(int32_t v) { bytes(&v, 4); }  // little-endian host (x86-64)
    void f32(float v) { bytes(&v, 4); }
    void u8(unsigned char v) { b.push_back(v); }
    void attr(const char* name, const char* type, int32_t size) {
        str(name);
        str(type);
        i32(size);
    }
};
}  // namespace

extern "C" int pbrt_hip_write_exr(const char* path, const float* rgb, int32_t width, int32_t height) try {
    if (!path || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
    ExrBuf h;
    h.i32(20000630);  // magic 0x76 0x2f 0x31 0x01
    h.i32(2);         // version 2, no flags: single-part scanline
    h.attr("channels", "chlist", 3 * (2 + 4 + 4 + 4 + 4) + 1);
    for (const char* c : {"B", "G", "R"}) {
        h.str(c);
        h.i32(2);  // FLOAT
        h.u8(0);   // pLinear
        h.u8(0);
        h.u8(0);
        h.u8(0);
        h.i32(1);  // xSampling
        h.i32(1);  // ySampling
    }
    h.u8(0);
    h.attr("compression", "compression", 1);
    h.u8(0);  // NO_COMPRESSION
    for (const char* name : {"dataWindow", "displayWindow"}) {
        h.attr(name, "box2i", 16);
        h.i32(0);
        h.i32(0);
        h.i32(width - 1);
        h.i32(height - 1);
    }
    h.attr("lineOrder", "lineOrder", 1);
    h.u8(0);  // INCREASING_Y
    h.attr("pixelAspectRatio", "float", 4);
    h.f32(1.0f);
    h.attr("screenWindowCenter", "v2f", 8);
    h.f32(0.0f);
    h.f32(0.0f);
    h.attr("screenWindowWidth", "float", 4);
    h.f32(1.0f);
    h.u8(0);  // end of header
    FILE* f = std::fopen(path, "wb");
    if (!f) return PBRT_HIP_ERR_INVALID;
    const uint64_t row_bytes = (uint64_t)width * 3 * sizeof(float), chunk = 8 + row_bytes;
    const uint64_t first = h.b.size() + (uint64_t)height * 8;
    std::fwrite(h.b.data(), 1, h.b.size(), f);
    for (int32_t y = 0; y < height; ++y) {  // offset table
        uint64_t off = first + (uint64_t)y * chunk;
        std::fwrite(&off, 8, 1, f);
    }
    std::vector<float> line((size_t)width * 3);
    for (int32_t y = 0; y < height; ++y) {
        const float* src = rgb + (size_t)y * width * 3;
        for (int32_t x = 0; x < width; ++x) {
            line[x] = src[3 * x + 2];                      // B
            line[(size_t)width + x] = src[3 * x + 1];      // G
            line[2 * (size_t)width + x] = src[3 * x];      // R
        }
        int32_t head[2] = {y, (int32_t)row_bytes};
        std::fwrite(head, 4, 2, f);
        std::fwrite(line.data(), sizeof(float), line.size(), f);
    }
    bool ok = std::fclose(f) == 0;
    return ok ? PBRT_HIP_OK : PBRT_HIP_ERR_INVALID;
}
PB_ABI_CATCH

// 8-bit sRGB PNG (what pbrt-v3's WriteImage does for ".png": gamma_correct, 255 * v + 0.5 clamped to [0, 255]);
// the reference's own writer is todo!() (src/core/imageio.rs:3-5). zlib stream of stored (uncompressed) deflate
// blocks: no dependency, every PNG reader accepts it.
namespace {
struct Crc32 {
    uint32_t table[256];
    Crc32() {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            table[n] = c;
        }
    }
    uint32_t run(uint32_t crc, const unsigned char* p, size_t n) const {
        for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
        return crc;
    }
};
void png_chunk(FILE* f, const Crc32& crc, const char* type, const std::vector<unsigned char>& data) {
    unsigned char len[4] = {(unsigned char)(data.size() >> 24), (unsigned char)(data.size() >> 16),
                            (unsigned char)(data.size() >> 8), (unsigned char)data.size()};
    std::fwrite(len, 1, 4, f);
    std::fwrite(type, 1, 4, f);
    if (!data.empty()) std::fwrite(data.data(), 1, data.size(), f);
    uint32_t c = crc.run(0xffffffffu, (const unsigned char*)type, 4);
    if (!data.empty()) c = crc.run(c, data.data(), data.size());
    c ^= 0xffffffffu;
    unsigned char out[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c};
    std::fwrite(out, 1, 4, f);
}
float gamma_correct(float v) {  // pbrt.rs GammaCorrect: sRGB transfer curve
    if (v <= 0.0031308f) return 12.92f * v;
    return 1.055f * std::pow(v, 1.0f / 2.4f) - 0.055f;
}
}  // namespace

extern "C" int pbrt_hip_write_png(const char* path, const float* rgb, int32_t width, int32_t height) try {
    if (!path || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
    const size_t row = (size_t)width * 3 + 1;  // filter byte + pixels
    std::vector<unsigned char> raw(row * height);
    for (int32_t y = 0; y < height; ++y) {
        unsigned char* o = &raw[row * y];
        *o++ = 0;  // filter type None
        for (int32_t x = 0; x < width * 3; ++x) {
            float v = 255.0f * gamma_correct(rgb[(size_t)y * width * 3 + x]) + 0.5f;
            *o++ = (unsigned char)(v < 0.0f || v != v ? 0.0f : (v > 255.0f ? 255.0f : v));
        }
    }
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;  // Adler-32
    for (size_t pos = 0; pos < raw.size();) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back((unsigned char)(n & 0xff));
        z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff));
        z.push_back((unsigned char)((~n >> 8) & 0xff));
        for (size_t i = 0; i < n; ++i) {
            a = (a + raw[pos + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    }
    uint32_t adler = (b << 16) | a;
    for (int k = 3; k >= 0; --k) z.push_back((unsigned char)(adler >> (8 * k)));
    FILE* f = std::fopen(path, "wb");
    if (!f) return PBRT_HIP_ERR_INVALID;
    static const Crc32 crc;
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::fwrite(sig, 1, 8, f);
    std::vector<unsigned char> ihdr = {(unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8),
                                       (unsigned char)width, (unsigned char)(height >> 24), (unsigned char)(height >> 16),
                                       (unsigned char)(height >> 8), (unsigned char)height, 8, 2, 0, 0, 0};  // 8-bit RGB
    png_chunk(f, crc, "IHDR", ihdr);
    png_chunk(f, crc, "IDAT", z);
    png_chunk(f, crc, "IEND", {});
    bool ok = std::fclose(f) == 0;
    return ok ? PBRT_HIP_OK : PBRT_HIP_ERR_INVALID;
}
PB_ABI_CATCH

extern "C" void pbrt_hip_film_to_rgb(const float* film, int64_t n_pixels, float* rgb) {
    for (int64_t i = 0; i < n_pixels; ++i) {
        const float* p = film + 4 * i;
        float r = 3.240479f * p[0] - 1.537150f * p[1] - 0.498535f * p[2];
        float g = -0.969256f * p[0] + 1.875991f * p[1] + 0.041556f * p[2];
        float b = 0.055648f * p[0] - 0.204043f * p[1] + 1.057311f * p[2];
        if (p[3] != 0.0f) {
            float inv = 1.0f / p[3];
            r = std::max(r * inv, 0.0f);
            g = std::max(g * inv, 0.0f);
            b = std::max(b * inv, 0.0f);
        }
        rgb[3 * i] = r;
        rgb[3 * i + 1] = g;
        rgb[3 * i + 2] = b;
    }
}

// Morton (Z-order) code of a tile's grid position: x in the even bits, y in the odd ones.
static inline uint64_t tile_morton(uint32_t tx, uint32_t ty) {
    auto spread = [](uint64_t v) {
        v &= 0xffffffffull;
        v = (v | (v << 16)) & 0x0000ffff0000ffffull;
        v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
        v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
        v = (v | (v << 2)) & 0x3333333333333333ull;
        v = (v | (v << 1)) & 0x5555555555555555ull;
        return v;
    };
    return spread(tx) | (spread(ty) << 1);
}

extern "C" int pbrt_hip_tile_partition_order(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t rank, int32_t world,
                                             int32_t order, int32_t* origins_xy, int32_t capacity, int32_t* n_out) try {
    if (!n_out || x0 > x1 || y0 > y1 || world <= 0 || rank < 0 || rank >= world) return PBRT_HIP_ERR_INVALID;
    if (order != PBRT_TILE_ORDER_MORTON && order != PBRT_TILE_ORDER_ROW_MAJOR) return PBRT_HIP_ERR_INVALID;
    const int64_t ntx = ((int64_t)x1 - x0 + kTile - 1) / kTile, nty = ((int64_t)y1 - y0 + kTile - 1) / kTile;
    const int64_t total = ntx * nty;
    if (total > INT32_MAX) return PBRT_HIP_ERR_INVALID;
    // the k-th tile of the dealing order belongs to rank k % world (round-robin in Morton order)
    const int64_t n = total <= rank ? 0 : (total - rank + world - 1) / world;
    *n_out = (int32_t)n;
    if (!origins_xy) return PBRT_HIP_OK;
    if (n > capacity) return PBRT_HIP_ERR_INVALID;
    if (order == PBRT_TILE_ORDER_ROW_MAJOR) {
        int64_t k = 0;
        for (int64_t t = rank; t < total; t += world, ++k) {
            origins_xy[2 * k] = x0 + (int32_t)(t % ntx) * kTile;
            origins_xy[2 * k + 1] = y0 + (int32_t)(t / ntx) * kTile;
        }
        return PBRT_HIP_OK;
    }
    std::vector<std::pair<uint64_t, int32_t>> keyed((size_t)total);  // 16 bytes per tile of the frame while dealing (bad_alloc -> PB_ABI_CATCH)
    for (int64_t t = 0; t < total; ++t) keyed[(size_t)t] = {tile_morton((uint32_t)(t % ntx), (uint32_t)(t / ntx)), (int32_t)t};
    std::sort(keyed.begin(), keyed.end());
    // the rank's share of the deal, WALKED in row-major order: which tiles a rank owns is what balances the ranks; the order it
    // renders them in only decides the memory order of its camera-ray wavefront, and row-major measured 0.4 % faster on config 3
    // (profiles/r05_rank_of_world.txt) — with one rank the frame is then laid out exactly as under PBRT_TILE_ORDER_ROW_MAJOR
    std::vector<int32_t> mine;
    mine.reserve((size_t)n);
    for (int64_t i = rank; i < total; i += world) mine.push_back(keyed[(size_t)i].second);
    std::sort(mine.begin(), mine.end());
    for (int64_t k = 0; k < n; ++k) {
        const int64_t t = mine[(size_t)k];
        origins_xy[2 * k] = x0 + (int32_t)(t % ntx) * kTile;
        origins_xy[2 * k + 1] = y0 + (int32_t)(t / ntx) * kTile;
    }
    return PBRT_HIP_OK;
}
PB_ABI_CATCH

extern "C" int pbrt_hip_tile_partition(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t rank, int32_t world,
                                       int32_t* origins_xy, int32_t capacity, int32_t* n_out) try {
    return pbrt_hip_tile_partition_order(x0, y0, x1, y1, rank, world, PBRT_TILE_ORDER_MORTON, origins_xy, capacity, n_out);
}
PB_ABI_CATCH

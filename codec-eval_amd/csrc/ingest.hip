// Decoded-image ingest on the device: the per-pixel host passes the reference runs
// between a decoder and the metrics become one streaming kernel that writes the packed RGB8 slab slot.
//   RGBA8  -> RGB8 : ImageData::to_rgb8_vec, src/eval/session.rs:98-117 (alpha dropped)
//   RGB16 / RGBA16 holding 10-bit samples -> RGB8 : to_8bit, crates/codec-iter/src/avif_config.rs:122-125
//                                                   ((v * 255 + 512) / 1023).min(255), integer, bit-exact
#include "ce_internal.h"

namespace {

__device__ __forceinline__ uint8_t to_8bit(uint32_t v)
{
    const uint32_t q = (v * 255u + 512u) / 1023u;
    return (uint8_t)(q < 255u ? q : 255u);
}

// one pixel per thread; writes go out as bytes (3 B/px), reads are 4 or 8 B per pixel
template <int FORMAT>
__global__ __launch_bounds__(256) void k_ingest(const void *__restrict__ src, uint8_t *__restrict__ dst, size_t n_pixels)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += (size_t)gridDim.x * blockDim.x) {
        uint8_t r, g, b;
        if (FORMAT == CE_PIXEL_RGBA8) {
            const uchar4 p = reinterpret_cast<const uchar4 *>(src)[i];
            r = p.x, g = p.y, b = p.z;
        } else if (FORMAT == CE_PIXEL_RGB16_10BIT) {
            const uint16_t *p = reinterpret_cast<const uint16_t *>(src) + 3 * i;
            r = to_8bit(p[0]), g = to_8bit(p[1]), b = to_8bit(p[2]);
        } else {
            const ushort4 p = reinterpret_cast<const ushort4 *>(src)[i];
            r = to_8bit(p.x), g = to_8bit(p.y), b = to_8bit(p.z);
        }
        dst[3 * i] = r;
        dst[3 * i + 1] = g;
        dst[3 * i + 2] = b;
    }
}

// ---- ICC -> sRGB as a complete colour table (include/ce_metrics.h: ce_lut_*) ------------------------------------------
// the host's packed 3-byte table -> one dword per colour (r | g << 8 | b << 16), so a lookup is one aligned 4-byte gather
__global__ __launch_bounds__(256) void k_lut_expand(const uint8_t *__restrict__ packed, uint32_t *__restrict__ table, uint32_t n)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        table[i] = (uint32_t)packed[3 * (size_t)i] | ((uint32_t)packed[3 * (size_t)i + 1] << 8) | ((uint32_t)packed[3 * (size_t)i + 2] << 16);
}

// in place on one packed RGB8 image of the slab: pixel -> table[(r << 16) | (g << 8) | b]
__global__ __launch_bounds__(256) void k_lut_apply(uint8_t *__restrict__ rgb, const uint32_t *__restrict__ table, size_t n_pixels)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pixels; i += (size_t)gridDim.x * blockDim.x) {
        uint8_t *p = rgb + 3 * i;
        const uint32_t v = table[((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 8) | (uint32_t)p[2]];
        p[0] = (uint8_t)v;
        p[1] = (uint8_t)(v >> 8);
        p[2] = (uint8_t)(v >> 16);
    }
}

}  // namespace

int ce_launch_lut_expand(ce_ctx *ctx, hipStream_t stream, const uint8_t *d_packed, uint32_t *d_table)
{
    CE_LAUNCH_ON(ctx, stream, "lut_expand", k_lut_expand, dim3(8192), dim3(256), 0, d_packed, d_table, 1u << 24);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}

int ce_launch_lut_apply(ce_ctx *ctx, hipStream_t stream, uint8_t *d_rgb, const uint32_t *d_table, size_t n_pixels)
{
    if (n_pixels == 0) return CE_OK;
    const dim3 grid((uint32_t)std::min<size_t>((n_pixels + 255) / 256, 8192)), block(256);
    CE_LAUNCH_ON(ctx, stream, "lut_apply", k_lut_apply, grid, block, 0, d_rgb, d_table, n_pixels);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}

size_t ce_pixel_bytes(int format)
{
    switch (format) {
        case CE_PIXEL_RGB8: return 3;
        case CE_PIXEL_RGBA8: return 4;
        case CE_PIXEL_RGB16_10BIT: return 6;
        case CE_PIXEL_RGBA16_10BIT: return 8;
        default: return 0;
    }
}

int ce_launch_ingest(ce_ctx *ctx, hipStream_t stream, int format, const void *d_src, uint8_t *d_dst, size_t n_pixels)
{
    if (n_pixels == 0) return CE_OK;
    const dim3 grid((uint32_t)std::min<size_t>((n_pixels + 255) / 256, 8192)), block(256);
    switch (format) {
        case CE_PIXEL_RGBA8: CE_LAUNCH_ON(ctx, stream, "ingest_rgba8", k_ingest<CE_PIXEL_RGBA8>, grid, block, 0, d_src, d_dst, n_pixels); break;
        case CE_PIXEL_RGB16_10BIT:
            CE_LAUNCH_ON(ctx, stream, "ingest_rgb16", k_ingest<CE_PIXEL_RGB16_10BIT>, grid, block, 0, d_src, d_dst, n_pixels);
            break;
        case CE_PIXEL_RGBA16_10BIT:
            CE_LAUNCH_ON(ctx, stream, "ingest_rgba16", k_ingest<CE_PIXEL_RGBA16_10BIT>, grid, block, 0, d_src, d_dst, n_pixels);
            break;
        default: return CE_ERR_INVALID_ARG;
    }
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}

// Calibration kernels for the rocprofv3 traffic counters (profiles/make_traffic.py).
// MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read
// and other access widths are uncalibrated.  The kernels of this library read with 1, 4 and 16 bytes per lane, so
// each PMC pass also runs these known-byte-count streams and the correction factor of every width is MEASURED in
// the same pass (factor = bytes actually moved / bytes the counter reports) instead of assumed.
#include "ce_internal.h"

namespace {

template <int WIDTH>
struct lane_word;
template <>
struct lane_word<1> { using type = uint8_t; };
template <>
struct lane_word<4> { using type = uint32_t; };
template <>
struct lane_word<16> { using type = uint4; };

__device__ __forceinline__ uint32_t fold(uint8_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// every element of buf[0, n) is read exactly once, consecutive lanes reading consecutive words
template <int WIDTH>
__global__ __launch_bounds__(256) void k_calib_read(const typename lane_word<WIDTH>::type *__restrict__ buf, size_t n,
                                                    uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= fold(buf[i]);
    if (acc == sink[1]) sink[0] = acc;  // a run-time value the compiler cannot rule out (a constant out of a byte's range
                                        // let it delete the 1-byte loads): keeps every load alive
}

template <int WIDTH>
__global__ __launch_bounds__(256) void k_calib_write(typename lane_word<WIDTH>::type *__restrict__ buf, size_t n)
{
    typename lane_word<WIDTH>::type v{};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = v;
}

}  // namespace

int ce_calibrate_traffic(ce_ctx *ctx, size_t bytes)
{
    bytes &= ~(size_t)4095;
    if (bytes == 0) return CE_ERR_INVALID_ARG;
    uint8_t *buf = nullptr;
    uint32_t *sink = nullptr;
    CE_HIP(ctx, hipMalloc(&buf, bytes));
    CE_HIP(ctx, hipMalloc(&sink, 256));
    CE_HIP(ctx, hipMemsetAsync(buf, 1, bytes, ctx->stream));
    CE_HIP(ctx, hipMemsetAsync(sink, 0x5a, 256, ctx->stream));
    const dim3 grid(8192), block(256);
    CE_LAUNCH(ctx, "calib_read_b1", k_calib_read<1>, grid, block, 0, (const uint8_t *)buf, bytes, sink);
    CE_LAUNCH(ctx, "calib_read_b4", k_calib_read<4>, grid, block, 0, (const uint32_t *)buf, bytes / 4, sink);
    CE_LAUNCH(ctx, "calib_read_b16", k_calib_read<16>, grid, block, 0, (const uint4 *)buf, bytes / 16, sink);
    CE_LAUNCH(ctx, "calib_write_b4", k_calib_write<4>, grid, block, 0, (uint32_t *)buf, bytes / 4);
    CE_LAUNCH(ctx, "calib_write_b16", k_calib_write<16>, grid, block, 0, (uint4 *)buf, bytes / 16);
    CE_HIP(ctx, hipStreamSynchronize(ctx->stream));
    CE_HIP(ctx, hipFree(buf));
    CE_HIP(ctx, hipFree(sink));
    return CE_OK;
}

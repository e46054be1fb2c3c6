// PSNR on gfx950 — replaces calculate_psnr (/root/reference/src/metrics/mod.rs:312-331).
//
// The reference accumulates (r-t)^2 in f64; every partial sum is an integer below 2^53, so
// the sum is exact and order-independent.  The device therefore accumulates the same integer
// in u64 (wave reduction + one integer atomic per block, which is associative and hence
// deterministic) and the host finishes  10*log10(255^2 / (sse/n))  in f64 with the host libm:
// the result is bit-identical to the reference's.
//
// Inner loop: 16 bytes per lane per image; sum (a-b)^2 = sum a^2 + sum b^2 - 2 sum ab with
// three v_dot4_u32_u8 per dword pair (no per-byte unpacking).
#include <algorithm>

#include "ce_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kBlocksPerPair = 64;

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(kThreads) void k_psnr_sse(const uint8_t *__restrict__ refs,
                                                       const uint8_t *__restrict__ tests,
                                                       const uint32_t *__restrict__ pair_ref,
                                                       ce_dev_scores *__restrict__ scores, size_t img_bytes)
{
    const uint32_t p = blockIdx.y;
    const uint8_t *a = refs + (size_t)pair_ref[p] * img_bytes;
    const uint8_t *b = tests + (size_t)p * img_bytes;
    unsigned long long sse = 0;
    const size_t tid = (size_t)blockIdx.x * kThreads + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * kThreads;
    if ((img_bytes & 15) == 0) {  // slabs are 256-byte aligned, so every image starts 16-byte aligned
        const uint4 *a4 = reinterpret_cast<const uint4 *>(a);
        const uint4 *b4 = reinterpret_cast<const uint4 *>(b);
        const size_t n16 = img_bytes >> 4;
        for (size_t i = tid; i < n16; i += nthreads) {
            const uint4 va = a4[i], vb = b4[i];
            // per 16 bytes: each dot4 adds at most 4*255^2 < 2^18; twelve of them fit u32 easily
            uint32_t saa = 0, sbb = 0, sab = 0;
            saa = __builtin_amdgcn_udot4(va.x, va.x, saa, false);
            sbb = __builtin_amdgcn_udot4(vb.x, vb.x, sbb, false);
            sab = __builtin_amdgcn_udot4(va.x, vb.x, sab, false);
            saa = __builtin_amdgcn_udot4(va.y, va.y, saa, false);
            sbb = __builtin_amdgcn_udot4(vb.y, vb.y, sbb, false);
            sab = __builtin_amdgcn_udot4(va.y, vb.y, sab, false);
            saa = __builtin_amdgcn_udot4(va.z, va.z, saa, false);
            sbb = __builtin_amdgcn_udot4(vb.z, vb.z, sbb, false);
            sab = __builtin_amdgcn_udot4(va.z, vb.z, sab, false);
            saa = __builtin_amdgcn_udot4(va.w, va.w, saa, false);
            sbb = __builtin_amdgcn_udot4(vb.w, vb.w, sbb, false);
            sab = __builtin_amdgcn_udot4(va.w, vb.w, sab, false);
            sse += (unsigned long long)(saa + sbb) - 2ull * sab;
        }
    } else {
        for (size_t i = tid; i < img_bytes; i += nthreads) {
            const int d = (int)a[i] - (int)b[i];
            sse += (unsigned long long)(d * d);
        }
    }
    __shared__ unsigned long long s_part[kThreads / 64];
    sse = wave_sum_u64(sse);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sse;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
#pragma unroll
        for (int k = 0; k < kThreads / 64; k++) t += s_part[k];
        atomicAdd(&scores[p].sse, t);
    }
}

__global__ void k_psnr_clear(ce_dev_scores *scores, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scores[i].sse = 0;
}

}  // namespace

int ce_launch_psnr(ce_batch *b, const uint8_t *d_refs, uint32_t n_pairs)
{
    ce_ctx *ctx = b->ctx;
    CE_LAUNCH(ctx, "psnr_clear", k_psnr_clear, dim3((n_pairs + 255) / 256), dim3(256), 0, b->d_scores, n_pairs);
    size_t work = (b->img_bytes + 15) / 16;
    uint32_t blocks = (uint32_t)((work + kThreads - 1) / kThreads);
    // a few pairs alone cannot fill the chip with 64 blocks each and a block's loop is then pure latency (one 768x512 pair:
    // 48 us with 64 blocks): up to one 16-byte piece per thread while the launch stays under ~4096 blocks
    const uint32_t cap = std::max<uint32_t>(kBlocksPerPair, 4096u / n_pairs);
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    CE_LAUNCH(ctx, "psnr_sse", k_psnr_sse, dim3(blocks, n_pairs), dim3(kThreads), 0, d_refs, b->d_tests, b->d_pair_ref,
              b->d_scores, b->img_bytes);
    CE_HIP(ctx, hipGetLastError());
    return CE_OK;
}

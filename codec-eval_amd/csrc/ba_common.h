// Shared by butteraugli.hip and ba_stream.hip: the plane layout of the PsychoImage, plane geometry, the blur kernel table
// and the launcher of the streaming LF column pass (compiled in its own translation unit: no SLP packing, see the Makefile).
#pragma once
#include <cstddef>
#include <cstdint>

#include "ce_internal.h"

namespace ce_ba {

constexpr int PSY = 10;  // uhf0 uhf1 hf0 hf1 mf0 mf1 mf2 lf0 lf1 lf2
enum { UHF0 = 0, UHF1, HF0, HF1, MF0, MF1, MF2, LF0, LF1, LF2 };

struct geom {
    uint32_t w, h, pitch;
    size_t plane;
};

struct blur_kernel {
    int len;
    float k[40];
    // 1 / (sum of the valid weights) for an output d pixels from the low / high border of a line that is at least
    // `len` long, summed on the host in the order the pixel loop would sum them (low border: taps off-d .. len-1,
    // high border: taps 0 .. off+d)
    float lo[16], hi[16];
};

#if defined(__HIPCC__)
// image slot of launch index z: the references used come first, the distorted images follow the batch's max_refs slots
__device__ __forceinline__ uint32_t slot_of(uint32_t z, uint32_t n_refs_used, uint32_t max_refs)
{
    return z < n_refs_used ? z : max_refs + (z - n_refs_used);
}
#endif

// LF stage, column pass + split, streaming form (ba_stream.hip).  row_scale[y] = 1 / (sum of the 33-tap kernel's weights that
// fall inside [0, h) for output row y, summed in ascending tap order) - device memory, h floats.
int ce_ba_launch_v_lf_stream(ce_ctx *ctx, hipStream_t stream, const float *tmp, const float *xyb, float *psy, const geom &g,
                             const blur_kernel &bk, const float *row_scale, uint32_t n_refs_used, uint32_t max_refs, uint32_t z0,
                             float *aux_out, uint32_t nz);

}  // namespace ce_ba

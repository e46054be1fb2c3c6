// Shared by dssim.hip (tile kernels, launch sequence) and dssim_stream.hip (streaming kernels).  Not part of the ABI.
#pragma once

#include "ce_internal.h"

struct lvl_geom {
    uint32_t w, h, pitch;
    size_t plane;
};

#define CE_DSSIM_STRIP 60  // output columns of a streaming kernel's 64-lane strip (halo 2 on either side)

// Dssim::compare of level `level` for the first n_pairs pairs (dssim_stream.hip); *n_part = partial sums written per pair
int ce_dssim_compare_stream(ce_batch *b, int level, uint32_t n_pairs, uint32_t *n_part);

// The float32 full-band kernels without materialised spectra -- k_bands<float, 0, FULL, false>, 89 % of the headline step --
// live in their own translation unit (spart_bands_f32.hip) so that this one kernel family can be compiled with the
// instruction-scheduling strategy that measures fastest for it (build.py: TU_FLAGS) without imposing it on the float64
// column kernels, which it slows down, or on the store-bound materialising variants, which it does not help.
#pragma once

#include <hip/hip_runtime.h>

#include "spart_kernels.h"

namespace spart {
// launches k_bands<float, 0, full, false> (full = 1: one band sum per lane, 2: four) on `st`, grid = xcd_grid(nchunk)
// workgroups of TILE lanes; returns hipGetLastError()
hipError_t launch_bands_f32(int full, unsigned grid, hipStream_t st, const float* tab, const float* cst, int64_t Bp, int64_t B,
                            int chunk, float* bandsum);
}  // namespace spart

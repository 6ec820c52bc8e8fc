// Translation unit of the float32 full-band kernels k_bands<float, 0, FULL, false> (see spart_bands_f32.h).
#include "spart_bands_f32.h"

#include <cstring>

namespace spart {

hipError_t launch_bands_f32(int full, unsigned grid, hipStream_t st, const float* tab, const float* cst, int64_t Bp, int64_t B,
                            int chunk, float* bandsum) {
  MatPtrs<float> mp;                         // no materialised spectra in these variants
  std::memset(&mp, 0, sizeof(mp));
  if (full == 1) hipLaunchKernelGGL((k_bands<float, 0, 1, false>), dim3(grid), dim3(TILE), 0, st, tab, cst, Bp, B, chunk, mp, bandsum);
  else if (full == 2) hipLaunchKernelGGL((k_bands<float, 0, 2, false>), dim3(grid), dim3(TILE), 0, st, tab, cst, Bp, B, chunk, mp, bandsum);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace spart

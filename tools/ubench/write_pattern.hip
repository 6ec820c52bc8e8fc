// Micro-benchmark: HBM write bandwidth of the materialised-spectra store pattern of k_bands<.,MAT=true,.>
// (9 arrays, one 1 KB row segment per (sample, array, tile), row pitch 2162 floats) against a linear fill and
// against variants (padded pitch, dwordx4 stores, fewer concurrent arrays).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_linear(float4* __restrict__ out, size_t n4, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += st) out[i] = make_float4(v, v, v, v);
}

// pattern: grid = nchunk * 8 tiles (XCD-aware order as in spart_kernels.h), block 256; NARR arrays
template <int NARR>
__global__ __launch_bounds__(256) void k_pattern(float* __restrict__ base, size_t arr_stride, int pitch, int64_t B, int chunk, float v) {
  unsigned b = blockIdx.x;
  int tile = (int)((b >> 3) & 7u);
  int64_t ck = (int64_t)(b >> 6) * 8 + (b & 7u);
  int band = tile * 256 + threadIdx.x;
  if (ck * chunk >= B) return;
  int64_t s0 = ck * chunk, s1 = s0 + chunk < B ? s0 + chunk : B;
  bool active = band < 2162;
  for (int64_t s = s0; s < s1; ++s) {
    float x = v + (float)s;
    // some arithmetic between stores is irrelevant here: pure store stream
    if (active) {
#pragma unroll
      for (int a = 0; a < NARR; ++a) (base + a * arr_stride + s * pitch)[band] = x + a;
    }
  }
}

// same bytes, but each lane writes 4 consecutive floats of ONE array row segment (dwordx4), arrays round-robin over waves
template <int NARR>
__global__ __launch_bounds__(256) void k_pattern_x4(float* __restrict__ base, size_t arr_stride, int pitch, int64_t B, int chunk, float v) {
  unsigned b = blockIdx.x;
  int tile = (int)((b >> 3) & 7u);
  int64_t ck = (int64_t)(b >> 6) * 8 + (b & 7u);
  if (ck * chunk >= B) return;
  int64_t s0 = ck * chunk, s1 = s0 + chunk < B ? s0 + chunk : B;
  // 4 samples x 256 bands per array are gathered (as if through LDS) and written as 4 rows x 64 lanes x float4
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t s = s0; s + 3 < s1; s += 4) {
    float x = v + (float)s;
    int band = tile * 256 + lane * 4;
    if (band + 3 < 2162 && (pitch & 3) == 0) {
#pragma unroll
      for (int a = 0; a < NARR; ++a)
        *(float4*)(base + a * arr_stride + (s + wave) * pitch + band) = make_float4(x, x, x, x + a);
    }
  }
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); f(); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 3;
}

int main() {
  const int64_t B = 200000; const int NARR = 9;
  const int pitchA = 2162, pitchB = 2176;
  size_t stride = (size_t)B * pitchB;            // floats per array (room for the padded pitch)
  float* buf; CK(hipMalloc(&buf, stride * NARR * 4));
  size_t n4 = (size_t)B * pitchA * NARR / 4;
  float ms = timeit([&] { hipLaunchKernelGGL(k_linear, dim3(256 * 16), dim3(256), 0, 0, (float4*)buf, n4, 1.f); });
  printf("linear fill float4            %8.3f ms  %6.0f GB/s\n", ms, n4 * 16 / ms / 1e6);
  int chunk = (int)((B + 2047) / 2048); int64_t nchunk = (B + chunk - 1) / chunk; unsigned grid = (unsigned)(((nchunk + 7) / 8) * 64);
  double bytes = (double)B * 2162 * NARR * 4;
  ms = timeit([&] { hipLaunchKernelGGL((k_pattern<9>), dim3(grid), dim3(256), 0, 0, buf, stride, pitchA, B, chunk, 1.f); });
  printf("pattern 9 arrays pitch 2162   %8.3f ms  %6.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL((k_pattern<9>), dim3(grid), dim3(256), 0, 0, buf, stride, pitchB, B, chunk, 1.f); });
  printf("pattern 9 arrays pitch 2176   %8.3f ms  %6.0f GB/s\n", ms, bytes / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL((k_pattern<1>), dim3(grid), dim3(256), 0, 0, buf, stride, pitchA, B, chunk, 1.f); });
  printf("pattern 1 array  pitch 2162   %8.3f ms  %6.0f GB/s\n", ms, bytes / 9 / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL((k_pattern<4>), dim3(grid), dim3(256), 0, 0, buf, stride, pitchA, B, chunk, 1.f); });
  printf("pattern 4 arrays pitch 2162   %8.3f ms  %6.0f GB/s\n", ms, bytes * 4 / 9 / ms / 1e6);
  ms = timeit([&] { hipLaunchKernelGGL((k_pattern_x4<9>), dim3(grid), dim3(256), 0, 0, buf, stride, pitchB, B, chunk, 1.f); });
  printf("x4 stores 9 arrays pitch 2176 %8.3f ms  %6.0f GB/s (approx bytes)\n", ms, (double)B * 2160 * NARR * 4 / ms / 1e6);
  // occupancy sweep (dynamic LDS limits the resident workgroups per CU): does the store stream need many waves in flight?
  for (int kb : {20, 30, 40, 50, 64}) {
    ms = timeit([&] { hipLaunchKernelGGL((k_pattern<9>), dim3(grid), dim3(256), kb * 1024, 0, buf, stride, pitchB, B, chunk, 1.f); });
    printf("pattern 9 arrays pitch 2176, %2d KB LDS/block (%d blocks/CU) %8.3f ms  %6.0f GB/s\n", kb, 160 / kb, ms, bytes / ms / 1e6);
  }
  for (int ch : {16, 64, 256}) {
    int64_t nc = (B + ch - 1) / ch; unsigned g = (unsigned)(((nc + 7) / 8) * 64);
    ms = timeit([&] { hipLaunchKernelGGL((k_pattern<9>), dim3(g), dim3(256), 0, 0, buf, stride, pitchA, B, ch, 1.f); });
    printf("pattern 9 arrays chunk %4d   %8.3f ms  %6.0f GB/s\n", ch, ms, bytes / ms / 1e6);
  }
  return 0;
}

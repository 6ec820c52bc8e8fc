// variants of the exact-f32 MFMA rate probe: instruction shape, independent chains per wave, waves per SIMD, loop length
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
template <int NCH, int ITERS>
__global__ __launch_bounds__(256) void k32(float* out, float seed) {
  f16v acc[NCH];
  for (int c = 0; c < NCH; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = seed * (c + 1);
  float a = threadIdx.x * 1e-3f + seed, b = a * 0.5f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < NCH; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH, int ITERS>
__global__ __launch_bounds__(256) void k16(float* out, float seed) {
  f4v acc[NCH];
  for (int c = 0; c < NCH; ++c) for (int i = 0; i < 4; ++i) acc[c][i] = seed * (c + 1);
  float a = threadIdx.x * 1e-3f + seed, b = a * 0.5f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < NCH; ++c) for (int i = 0; i < 4; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
#define RUN32(NCH, ITERS, WG) { int blocks = 256 * WG; float ms = timeit([&] { hipLaunchKernelGGL((k32<NCH, ITERS>), dim3(blocks), dim3(256), 0, 0, o, 1.0f); }); \
  double n = (double)blocks * 4 * ITERS * NCH / 1024.0; printf("32x32x2 chains %d iters %6d waves/SIMD %d: %8.3f ms  %.2f ns/MFMA/SIMD  %.1f Tflop/s\n", NCH, ITERS, WG, ms, ms * 1e6 / n, n * 1024 * 4096 / (ms * 1e-3) / 1e12); }
#define RUN16(NCH, ITERS, WG) { int blocks = 256 * WG; float ms = timeit([&] { hipLaunchKernelGGL((k16<NCH, ITERS>), dim3(blocks), dim3(256), 0, 0, o, 1.0f); }); \
  double n = (double)blocks * 4 * ITERS * NCH / 1024.0; printf("16x16x4 chains %d iters %6d waves/SIMD %d: %8.3f ms  %.2f ns/MFMA/SIMD  %.1f Tflop/s\n", NCH, ITERS, WG, ms, ms * 1e6 / n, n * 1024 * 2048 / (ms * 1e-3) / 1e12); }
int main() {
  float* o; if (hipMalloc(&o, 256 * 8192 * 4) != hipSuccess) return 1;
  RUN32(4, 256, 1) RUN32(4, 4096, 1) RUN32(4, 32768, 1) RUN32(8, 4096, 1) RUN32(2, 4096, 2) RUN32(4, 4096, 2) RUN32(4, 32768, 2) RUN32(4, 4096, 4) RUN32(4, 32768, 4)
  RUN16(4, 4096, 1) RUN16(8, 4096, 1) RUN16(4, 4096, 2) RUN16(8, 4096, 2) RUN16(4, 4096, 4) RUN16(8, 32768, 2)
  return 0;
}

// Sustained rate of v_mfma_f32_32x32x2_f32 (the exact-f32 matrix instruction of the LUT scan) and of v_mfma_f64_16x16x4_f64:
// independent accumulator chains, nothing else in the loop.   hipcc -O3 --offload-arch=gfx950 -o mfma_f32_rate mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4v __attribute__((ext_vector_type(4)));
#define ITERS 4096

template <int NCH>
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
template <int NCH>
__global__ __launch_bounds__(256) void k64(double* out, double seed) {
  d4v acc[NCH];
  for (int c = 0; c < NCH; ++c) for (int i = 0; i < 4; ++i) acc[c][i] = seed * (c + 1);
  double a = threadIdx.x * 1e-3 + seed, b = a * 0.5;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < NCH; ++c) for (int i = 0; i < 4; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* o32; double* o64;
  CK(hipMalloc(&o32, 256 * 4096 * 4)); CK(hipMalloc(&o64, 256 * 4096 * 8));
  for (int wg_per_cu : {1, 2, 4}) {
    int blocks = 256 * wg_per_cu;
    float ms = timeit([&] { hipLaunchKernelGGL((k32<4>), dim3(blocks), dim3(256), 0, 0, o32, 1.0f); });
    double n = (double)blocks * 4 * ITERS * 4 / 1024.0;   // MFMAs per SIMD
    printf("v_mfma_f32_32x32x2_f32, 4 chains, %d wave(s)/SIMD: %.3f ms -> %.2f ns per MFMA per SIMD = %.1f Tflop/s (4096 flop each)\n", wg_per_cu, ms,
           ms * 1e6 / n, n * 1024 * 4096 / (ms * 1e-3) / 1e12);
    ms = timeit([&] { hipLaunchKernelGGL((k64<4>), dim3(blocks), dim3(256), 0, 0, o64, 1.0); });
    printf("v_mfma_f64_16x16x4_f64, 4 chains, %d wave(s)/SIMD: %.3f ms -> %.2f ns per MFMA per SIMD = %.1f Tflop/s (2048 flop each)\n", wg_per_cu, ms,
           ms * 1e6 / n, n * 1024 * 2048 / (ms * 1e-3) / 1e12);
  }
  return 0;
}

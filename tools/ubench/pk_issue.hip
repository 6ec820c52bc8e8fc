// Micro-benchmark: does v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (two fp32 per lane) issue at the rate of the
// plain v_fma_f32 on gfx950?  8 independent accumulators per lane, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float2v a[8], v0 = {0.999f + seed * 1e-9f, 0.998f}, v1 = {1e-3f, 2e-3f};
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = float2v{threadIdx.x * 1e-3f + i + seed, threadIdx.x * 2e-3f + i}; s[i] = a[i].x; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(v0), "v"(v1));
      if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(v0.x), "v"(v1.x));
      if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(v0));
      if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(v0.x));
      if (MODE == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(v1));
    }
  }
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, float* out, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double winst = (double)blocks * 4 * ITERS * 8;
  printf("%-16s %8.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / (winst / 1024.0));
}

int main() {
  float* out; int blocks = 256 * 8 * 4;
  if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
  run<1>("v_fma_f32", out, blocks);
  run<0>("v_pk_fma_f32", out, blocks);
  run<3>("v_mul_f32", out, blocks);
  run<2>("v_pk_mul_f32", out, blocks);
  run<4>("v_pk_add_f32", out, blocks);
  return 0;
}

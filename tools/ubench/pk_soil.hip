// Micro-benchmark: the soil model of the band kernel (soil_band, 55 VALU instructions, no data-dependent branch)
// for two samples per lane -- as two scalar evaluations and as one evaluation on packed float2 values
// (v_pk_mul/fma/add_f32) -- to price what "two samples per lane" would buy the band kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2048

template <typename V> struct S;   // scalar helpers for float and float2
template <> struct S<float> {
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
  static __device__ __forceinline__ float bc(float x) { return x; }
};
template <> struct S<f2> {
  static __device__ __forceinline__ f2 rcp(f2 x) { return f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
  static __device__ __forceinline__ f2 ex2(f2 x) { return f2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
  static __device__ __forceinline__ f2 bc(float x) { return f2{x, x}; }
};

// per-band table values are per-lane scalars (shared by the two samples); per-sample values are V
template <typename V>
__device__ __forceinline__ V soil(float cbac, float pw, float rw, float kw, V rdry, const V fm[7], V fmsum, V film2l) {
  const V one = S<V>::bc(1.0f);
  V rbac = one - (one - rdry) * (rdry * S<V>::bc(cbac) + one - rdry);
  V tw1 = S<V>::ex2(-film2l * S<V>::bc(kw));
  V x[6], d[6], pre[6];
  V xv = rbac;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    xv *= tw1; x[k] = xv; d[k] = one - S<V>::bc(pw) * xv; pre[k] = (k == 0) ? d[0] : pre[k - 1] * d[k];
  }
  V q = S<V>::rcp(pre[5]);
  V acc = S<V>::bc(0.0f);
#pragma unroll
  for (int k = 5; k >= 1; --k) { acc += fm[k + 1] * x[k] * (q * pre[k - 1]); q *= d[k]; }
  acc += fm[1] * x[0] * q;
  return rdry * fm[0] + S<V>::bc(rw) * fmsum + S<V>::bc((1.0f - rw) * (1.0f - pw)) * acc;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ cst, float seed) {
  const float cbac = 0.9f + threadIdx.x * 1e-4f, pw = 0.55f + threadIdx.x * 1e-4f, rw = 0.02f, kw = 0.01f * (threadIdx.x & 31);
  __shared__ float lds[64 * 12];
  for (int i = threadIdx.x; i < 64 * 12; i += 256) lds[i] = cst[i];
  __syncthreads();
  float r0 = 0.f, r1 = 0.f;
  for (int it = 0; it < ITERS; ++it) {
    const float* ca = lds + (it & 31) * 24;          // two samples' constants, wave-uniform (LDS broadcast)
    const float* cb = ca + 12;
    if (MODE == 0) {
      float fa[7], fb[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) { fa[i] = ca[i]; fb[i] = cb[i]; }
      r0 += soil<float>(cbac, pw, rw, kw, ca[7] + seed, fa, ca[8], ca[9]);
      r1 += soil<float>(cbac, pw, rw, kw, cb[7] + seed, fb, cb[8], cb[9]);
    } else {
      f2 f[7];
#pragma unroll
      for (int i = 0; i < 7; ++i) f[i] = f2{ca[i], cb[i]};
      f2 r = soil<f2>(cbac, pw, rw, kw, f2{ca[7] + seed, cb[7] + seed}, f, f2{ca[8], cb[8]}, f2{ca[9], cb[9]});
      r0 += r.x; r1 += r.y;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1;
}

template <int MODE> void run(const char* name, float* out, float* cst, int blocks) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cst, 0.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, cst, 0.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double pairs = (double)blocks * 4 * ITERS;          // (wave, sample pair) evaluations
  printf("%-28s %8.3f ms -> %.1f ns per (wave, sample) per SIMD\n", name, ms, ms * 1e6 / (pairs / 1024.0) / 2);
}

int main() {
  float *out, *cst; int blocks = 256 * 5 * 4;           // 5 workgroups per CU resident, as the band kernel
  if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess || hipMalloc(&cst, 64 * 12 * 4) != hipSuccess) return 1;
  float h[64 * 12];
  for (int i = 0; i < 64 * 12; ++i) h[i] = 0.05f + 0.9f * ((i * 37) % 101) / 101.0f * ((i % 12) < 7 ? 0.2f : 1.0f);
  if (hipMemcpy(cst, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 1;
  run<0>("soil, 2 samples, scalar", out, cst, blocks);
  run<1>("soil, 2 samples, packed", out, cst, blocks);
  return 0;
}

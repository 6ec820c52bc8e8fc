// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this library uses
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane)
// ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel moves exactly NBYTES = 1 GiB once (far past the 256 MiB Infinity Cache); run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calib      and      --pmc WRITE_SIZE
// and divide the counter (KB) by 1 048 576: tools/ingest_profiles.py does that and stores the factors next to the
// traffic numbers they correct.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t NBYTES = size_t(1) << 30;

template <typename T> __device__ __forceinline__ float as_f(T v);
template <> __device__ __forceinline__ float as_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float as_f<double>(double v) { return (float)v; }
template <> __device__ __forceinline__ float as_f<float4>(float4 v) { return v.x + v.y + v.z + v.w; }

// coalesced streaming read, sizeof(T) bytes per lane (the constants / G rows / atm rows of the structure-of-arrays
// workspace are read like this: 4 B and 8 B per lane)
template <typename T> __global__ __launch_bounds__(256) void read_coalesced(const T* __restrict__ in, size_t n, float* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i < n; i += st) acc += as_f<T>(in[i]);
  if (acc == 1.2345e-30f) *sink = acc;
}
template <typename T> __global__ __launch_bounds__(256) void write_coalesced(T* __restrict__ out, size_t n, T v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) out[i] = v;
}
// 16 B per lane at a 208 B stride (the (B, nslot, 4) G rows written by the band kernel: one float4 per sample and slot lane)
__global__ __launch_bounds__(256) void write_strided16(float4* __restrict__ out, size_t n, float4 v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  const size_t rows = n / 13;
  for (; i < n; i += st) {
    size_t slot = i / rows, row = i - slot * rows;     // lanes walk rows: addresses 208 B apart
    out[row * 13 + slot] = v;
  }
}
// 128-byte segments: 32 lanes x 4 B from each of 48 rows (stage_constants of the band kernels)
__global__ __launch_bounds__(256) void read_segments(const float* __restrict__ in, size_t pitch, size_t nblk, float* sink) {
  float acc = 0.f;
  for (size_t b = blockIdx.x; b < nblk; b += gridDim.x) {
    const int si = threadIdx.x & 31, i0 = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += in[(size_t)(i0 + 8 * k) * pitch + b * 32 + si];
  }
  if (acc == 1.2345e-30f) *sink = acc;
}

int main() {
  void* buf;
  float* sink;
  CK(hipMalloc(&buf, NBYTES));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, NBYTES));
  const int grid = 256 * 16;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_coalesced<float>, dim3(grid), dim3(256), 0, 0, (const float*)buf, NBYTES / 4, sink);
    hipLaunchKernelGGL(read_coalesced<double>, dim3(grid), dim3(256), 0, 0, (const double*)buf, NBYTES / 8, sink);
    hipLaunchKernelGGL(read_coalesced<float4>, dim3(grid), dim3(256), 0, 0, (const float4*)buf, NBYTES / 16, sink);
    hipLaunchKernelGGL(read_segments, dim3(grid), dim3(256), 0, 0, (const float*)buf, NBYTES / 4 / 48, NBYTES / 4 / 48 / 32, sink);
    hipLaunchKernelGGL(write_coalesced<float>, dim3(grid), dim3(256), 0, 0, (float*)buf, NBYTES / 4, 0.f);
    hipLaunchKernelGGL(write_coalesced<double>, dim3(grid), dim3(256), 0, 0, (double*)buf, NBYTES / 8, 0.0);
    hipLaunchKernelGGL(write_coalesced<float4>, dim3(grid), dim3(256), 0, 0, (float4*)buf, NBYTES / 16, make_float4(0, 0, 0, 0));
    hipLaunchKernelGGL(write_strided16, dim3(grid), dim3(256), 0, 0, (float4*)buf, (NBYTES / 16 / 13) * 13, make_float4(0, 0, 0, 0));
    CK(hipDeviceSynchronize());
  }
  printf("fetch_calib: every kernel moved %zu bytes\n", NBYTES);
  return 0;
}

// Accuracy of v_log_f32 / v_exp_f32 / v_rcp_f32 / v_sqrt_f32 on gfx950 against float64, in particular v_log_f32 near 1
// (relative error of log2(w) for w = 1 + x, x from 1e-7 to 0.5) -- decides whether log1p can be one v_log_f32 plus a
// rounding correction instead of a series with a reciprocal.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* lg, float* ex, float* rc, float* sq, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  lg[i] = __builtin_amdgcn_logf(1.0f + x[i]);
  ex[i] = __builtin_amdgcn_exp2f(-x[i] * 8.0f);
  rc[i] = __builtin_amdgcn_rcpf(1.0f + x[i] * 7.0f);
  sq[i] = __builtin_amdgcn_sqrtf(x[i] * 3.0f);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> hx(n), a(n), b(n), c(n), d(n);
  for (int i = 0; i < n; ++i) hx[i] = (float)std::pow(10.0, -7.0 + 6.7 * (double)i / n);   // 1e-7 .. 0.5
  float *x, *lg, *ex, *rc, *sq;
  if (hipMalloc(&x, n * 4) || hipMalloc(&lg, n * 4) || hipMalloc(&ex, n * 4) || hipMalloc(&rc, n * 4) || hipMalloc(&sq, n * 4)) return 1;
  if (hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice)) return 1;
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, lg, ex, rc, sq, n);
  if (hipMemcpy(a.data(), lg, n * 4, hipMemcpyDeviceToHost) || hipMemcpy(b.data(), ex, n * 4, hipMemcpyDeviceToHost) ||
      hipMemcpy(c.data(), rc, n * 4, hipMemcpyDeviceToHost) || hipMemcpy(d.data(), sq, n * 4, hipMemcpyDeviceToHost)) return 1;
  const double eps = 5.9604644775390625e-08;   // 2^-24
  double ml[7] = {0}, me = 0, mr = 0, ms = 0;
  for (int i = 0; i < n; ++i) {
    float w = 1.0f + hx[i];
    double ref = std::log2((double)w);
    if (ref != 0.0) {
      int dec = (int)std::floor(std::log10((double)hx[i])) + 7;   // decade of x: 0 = 1e-7..1e-6, ..., 6 = 0.1..0.5
      double e = std::fabs((double)a[i] - ref) / std::fabs(ref) / eps;
      if (dec >= 0 && dec < 7 && e > ml[dec]) ml[dec] = e;
    }
    double r2 = std::exp2((double)(-hx[i] * 8.0f));
    me = std::fmax(me, std::fabs((double)b[i] - r2) / r2 / eps);
    double r3 = 1.0 / (double)(1.0f + hx[i] * 7.0f);
    mr = std::fmax(mr, std::fabs((double)c[i] - r3) / r3 / eps);
    double r4 = std::sqrt((double)(hx[i] * 3.0f));
    ms = std::fmax(ms, std::fabs((double)d[i] - r4) / r4 / eps);
  }
  printf("v_log_f32(1 + x): max relative error in units of 2^-24, per decade of x (1e-7.., 1e-6.., ..., 0.1..0.5):\n ");
  for (int j = 0; j < 7; ++j) printf(" %.2f", ml[j]);
  printf("\nv_exp_f32 (args -4..0): %.2f   v_rcp_f32: %.2f   v_sqrt_f32: %.2f   (units of 2^-24 relative)\n", me, mr, ms);
  return 0;
}

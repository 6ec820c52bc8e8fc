// Probe: accuracy of v_rcp_f64 refined by ONE vs TWO Newton steps (the float64 band path's 1/x, spart_math.h Mx<double>::rcp)
// against IEEE division, over 2^24 random doubles in [2^-20, 2^20] and both signs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
  r1[i] = r;
  r = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 24;
  std::vector<double> x(n), a(n), b(n), c(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    double m = 1.0 + (double)rand() / RAND_MAX + (double)rand() / RAND_MAX * 1e-9;
    int e = rand() % 41 - 20;
    x[i] = std::ldexp(m, e) * ((rand() & 1) ? 1 : -1);
  }
  double *dx, *d0, *d1, *d2;
  CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8)); CK(hipMalloc(&d2, n * 8));
  CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  CK(hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost));
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) {
    double t = 1.0 / x[i];
    m0 = std::fmax(m0, std::fabs(a[i] - t) / std::fabs(t));
    m1 = std::fmax(m1, std::fabs(b[i] - t) / std::fabs(t));
    m2 = std::fmax(m2, std::fabs(c[i] - t) / std::fabs(t));
  }
  printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e (2^-53 = 1.11e-16)\n", m0, m1, m2);
  return 0;
}

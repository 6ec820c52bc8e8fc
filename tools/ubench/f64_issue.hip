// Micro-benchmark: sustained ISSUE cost of the float64 vector instructions the float64 band arithmetic is made of
// (spart_math.h Mx<double>), gfx950: 8 waves / SIMD, 8 independent accumulators per lane, no memory traffic.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench/f64_issue tools/ubench/f64_issue.hip && tools/ubench/f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)
#define ITERS 2048
#define ALL(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* __restrict__ sc, double seed) {
  double a0 = threadIdx.x * 1e-3 + seed, a1 = a0 + 1., a2 = a0 + 2., a3 = a0 + 3., a4 = a0 + 4., a5 = a0 + 5., a6 = a0 + 6., a7 = a0 + 7.;
  double v0 = a0 * 0.5, v1 = a1 * 0.25;
  const int le = (int)(threadIdx.x & 1u);
  const double s0 = sc[blockIdx.x & 1], s1 = sc[2 + (blockIdx.x & 1)];   // wave-uniform -> SGPR pair
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {
#define OP(x) x = __builtin_fma(x, v0, v1);
      ALL(OP)
#undef OP
    } else if (MODE == 1) {
#define OP(x) x = __builtin_fma(x, s0, v1);
      ALL(OP)
#undef OP
    } else if (MODE == 2) {
#define OP(x) x = __builtin_fma(x, v0, s1);
      ALL(OP)
#undef OP
    } else if (MODE == 3) {
#define OP(x) x = x * v0;
      ALL(OP)
#undef OP
    } else if (MODE == 4) {
#define OP(x) x = x + v0;
      ALL(OP)
#undef OP
    } else if (MODE == 5) {
#define OP(x) x = __builtin_amdgcn_rcp(x);
      ALL(OP)
#undef OP
    } else if (MODE == 6) {
#define OP(x) x = __builtin_amdgcn_rsq(x);
      ALL(OP)
#undef OP
    } else if (MODE == 7) {
#define OP(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
      ALL(OP)
#undef OP
    } else if (MODE == 8) {      // f64 -> f32 -> f64 round trip (two converts)
#define OP(x) x = (double)(float)x + 0.0;
      ALL(OP)
#undef OP
    } else if (MODE == 9) {
#define OP(x) x = __builtin_ldexp(x, le);
      ALL(OP)
#undef OP
    } else if (MODE == 10) {     // compare + 64-bit select
#define OP(x) x = (x > v0) ? v1 : x;
      ALL(OP)
#undef OP
    } else if (MODE == 11) {
#define OP(x) x = __builtin_fmax(x, v0);
      ALL(OP)
#undef OP
    } else if (MODE == 12) {     // f32 transcendental for reference
      float f0 = (float)a0, f1 = (float)a1;
#define OP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(f0));
      ALL(OP)
#undef OP
      a0 = f0; a1 = f1;
    } else if (MODE == 13) {     // 32-bit integer op
      int j0 = (int)a0;
#define OP(x) asm volatile("v_add_u32 %0, %0, 3" : "+v"(j0));
      ALL(OP)
#undef OP
      a0 = j0;
    } else if (MODE == 14) {     // frexp mantissa
#define OP(x) x = __builtin_amdgcn_frexp_mant(x);
      ALL(OP)
#undef OP
    } else if (MODE == 15) {     // fma with an inline constant addend
#define OP(x) x = __builtin_fma(x, v0, 0.5);
      ALL(OP)
#undef OP
    } else if (MODE == 16) {     // fma with a 64-bit literal (needs a v_mov pair or an SGPR pair)
#define OP(x) x = __builtin_fma(x, v0, 0.3183098861837907);
      ALL(OP)
#undef OP
    } else if (MODE == 17) {     // v_trig_preop / v_fract: skip; v_cvt_f64_i32 + v_cvt_i32_f64 round trip
#define OP(x) x = (double)(int)x;
      ALL(OP)
#undef OP
    } else if (MODE == 18) {     // f32 fma for reference
      float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7;
      float w0 = (float)v0, w1 = (float)v1;
#define OP(x) x = __builtin_fmaf(x, w0, w1);
      OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7)
#undef OP
      a0 = f0; a1 = f1; a2 = f2; a3 = f3; a4 = f4; a5 = f5; a6 = f6; a7 = f7;
    } else if (MODE == 19) {     // v_rcp_f32 for reference
      float f0 = (float)a0;
#define OP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(f0));
      ALL(OP)
#undef OP
      a0 = f0;
    } else if (MODE == 20) {     // v_log_f32
      float f0 = (float)a0;
#define OP(x) asm volatile("v_log_f32 %0, %0" : "+v"(f0));
      ALL(OP)
#undef OP
      a0 = f0;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char* name, double* out, double* sc, int blocks, int per = 8) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, sc, 1.0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, sc, 1.0);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double winst = (double)blocks * 4 * ITERS * per;
  printf("%-44s %8.3f ms  -> %.2f ns per wave-instruction (or group) per SIMD\n", name, ms, ms * 1e6 / (winst / 1024.0));
}

int main() {
  double *out, *sc;
  int blocks = 256 * 8 * 2;   // 8 waves / SIMD resident, 2 rounds
  if (hipMalloc(&out, (size_t)blocks * 256 * 8) != hipSuccess || hipMalloc(&sc, 32) != hipSuccess) return 1;
  double h[4] = {0.999, 0.998, 1e-3, 2e-3};
  if (hipMemcpy(sc, h, 32, hipMemcpyHostToDevice) != hipSuccess) return 1;
  run<18>("f32 fma vgpr (reference)", out, sc, blocks);
  run<0>("f64 fma vgpr,vgpr,vgpr", out, sc, blocks);
  run<1>("f64 fma vgpr,SGPR,vgpr", out, sc, blocks);
  run<2>("f64 fma vgpr,vgpr,SGPR", out, sc, blocks);
  run<15>("f64 fma vgpr,vgpr,inline 0.5", out, sc, blocks);
  run<16>("f64 fma vgpr,vgpr,64-bit literal", out, sc, blocks);
  run<3>("f64 mul", out, sc, blocks);
  run<4>("f64 add", out, sc, blocks);
  run<11>("f64 max", out, sc, blocks);
  run<5>("v_rcp_f64", out, sc, blocks);
  run<6>("v_rsq_f64", out, sc, blocks);
  run<7>("v_sqrt_f64", out, sc, blocks);
  run<14>("v_frexp_mant_f64", out, sc, blocks);
  run<9>("v_ldexp_f64", out, sc, blocks);
  run<8>("cvt f64->f32->f64 + add (3 instr)", out, sc, blocks);
  run<17>("cvt f64->i32->f64 (2 instr)", out, sc, blocks);
  run<10>("f64 cmp + 64-bit select (3 instr)", out, sc, blocks);
  run<12>("v_exp_f32 (reference)", out, sc, blocks);
  run<20>("v_log_f32 (reference)", out, sc, blocks);
  run<19>("v_rcp_f32 (reference)", out, sc, blocks);
  run<13>("v_add_u32 (reference)", out, sc, blocks);
  return 0;
}

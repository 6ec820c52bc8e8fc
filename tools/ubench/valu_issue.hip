// Micro-benchmark: sustained issue cost of fp32 VALU instructions on gfx950 with different operand kinds.
// Each kernel runs N dependent-free-ish FMA chains per lane; 256 threads x many blocks (8 waves / SIMD).
#include <hip/hip_runtime.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)
#include <cstdio>
#include <vector>

#define ITERS 4096

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ sc, float seed) {
  float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
        a6 = a0 + 6.f, a7 = a0 + 7.f;
  float v0 = a0 * 0.5f, v1 = a1 * 0.25f;
  const float s0 = sc[blockIdx.x & 1], s1 = sc[2 + (blockIdx.x & 1)];   // wave-uniform -> SGPR
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {        // VGPR operands only
#define OP(x) x = __builtin_fmaf(x, v0, v1);
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    } else if (MODE == 1) { // one SGPR operand
#define OP(x) x = __builtin_fmaf(x, s0, v1);
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    } else if (MODE == 2) { // literal constant (v_fmaak / v_fmamk)
#define OP(x) x = __builtin_fmaf(x, v0, 0.3183099f);
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    } else if (MODE == 3) { // v_mul with SGPR
#define OP(x) x = x * s0;
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    } else if (MODE == 4) { // rcp
#define OP(x) x = __builtin_amdgcn_rcpf(x);
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    } else if (MODE == 5) { // exp2
#define OP(x) x = __builtin_amdgcn_exp2f(x);
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    } else if (MODE == 6) { // dependent chain of 8 (single accumulator)
#define OP(x) a0 = __builtin_fmaf(a0, v0, v1);
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    } else if (MODE == 8) { // 8 literal FMAs interleaved with 2 rcp (do transcendentals overlap plain VALU?)
      a0 = __builtin_fmaf(a0, v0, 0.3183099f); a1 = __builtin_fmaf(a1, v0, 0.3183099f);
      a6 = __builtin_amdgcn_rcpf(a6);
      a2 = __builtin_fmaf(a2, v0, 0.3183099f); a3 = __builtin_fmaf(a3, v0, 0.3183099f);
      a4 = __builtin_fmaf(a4, v0, 0.3183099f); a5 = __builtin_fmaf(a5, v0, 0.3183099f);
      a7 = __builtin_amdgcn_rcpf(a7);
      a0 = __builtin_fmaf(a0, v1, 0.5f); a1 = __builtin_fmaf(a1, v1, 0.25f);
    } else if (MODE == 10) { // 8 FMAs, then the 2 rcp back to back
      a0 = __builtin_fmaf(a0, v0, 0.3183099f); a1 = __builtin_fmaf(a1, v0, 0.3183099f);
      a2 = __builtin_fmaf(a2, v0, 0.3183099f); a3 = __builtin_fmaf(a3, v0, 0.3183099f);
      a4 = __builtin_fmaf(a4, v0, 0.3183099f); a5 = __builtin_fmaf(a5, v0, 0.3183099f);
      a0 = __builtin_fmaf(a0, v1, 0.5f); a1 = __builtin_fmaf(a1, v1, 0.25f);
      asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(a6), "+v"(a7));
    } else if (MODE == 11) { // 32 FMAs then 8 rcp back to back (same ratio, bigger groups)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        a0 = __builtin_fmaf(a0, v0, 0.3183099f); a1 = __builtin_fmaf(a1, v0, 0.3183099f);
        a2 = __builtin_fmaf(a2, v0, 0.3183099f); a3 = __builtin_fmaf(a3, v0, 0.3183099f);
        a4 = __builtin_fmaf(a4, v0, 0.3183099f); a5 = __builtin_fmaf(a5, v0, 0.3183099f);
        a0 = __builtin_fmaf(a0, v1, 0.5f); a1 = __builtin_fmaf(a1, v1, 0.25f);
      }
      asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\t"
                   "v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1" : "+v"(a6), "+v"(a7));
    } else if (MODE == 9) { // the same 8 FMAs without the rcps
      a0 = __builtin_fmaf(a0, v0, 0.3183099f); a1 = __builtin_fmaf(a1, v0, 0.3183099f);
      a2 = __builtin_fmaf(a2, v0, 0.3183099f); a3 = __builtin_fmaf(a3, v0, 0.3183099f);
      a4 = __builtin_fmaf(a4, v0, 0.3183099f); a5 = __builtin_fmaf(a5, v0, 0.3183099f);
      a0 = __builtin_fmaf(a0, v1, 0.5f); a1 = __builtin_fmaf(a1, v1, 0.25f);
    } else if (MODE == 7) { // v_cndmask + v_cmp pairs
#define OP(x) x = (x > v0) ? v1 : x + s1;
      OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char* name, float* out, float* sc, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, sc, 1.0f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, sc, 1.0f);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double winst = (double)blocks * 4 * ITERS * 8;           // wave-instructions of the measured kind
  double per_simd = winst / 1024.0;
  printf("%-28s %8.3f ms  -> %.2f ns per wave-instr per SIMD (= cycles at 1 GHz; x clock GHz for cycles)\n", name, ms,
         ms * 1e6 / per_simd);
}

int main() {
  float *out, *sc;
  int blocks = 256 * 8 * 4;   // 8 waves / SIMD resident, 4 rounds
  if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess || hipMalloc(&sc, 16) != hipSuccess) return 1;
  float h[4] = {0.999f, 0.998f, 1e-3f, 2e-3f};
  if (hipMemcpy(sc, h, 16, hipMemcpyHostToDevice) != hipSuccess) return 1;
  run<0>("fma vgpr,vgpr,vgpr", out, sc, blocks);
  run<1>("fma vgpr,sgpr,vgpr", out, sc, blocks);
  run<2>("fma vgpr,vgpr,literal", out, sc, blocks);
  run<3>("mul vgpr,sgpr", out, sc, blocks);
  run<4>("rcp", out, sc, blocks);
  run<5>("exp2", out, sc, blocks);
  run<6>("fma dependent chain", out, sc, blocks);
  run<7>("cmp+cndmask+add", out, sc, blocks);
  run<9>("8 fma literal (per 8)", out, sc, blocks);
  run<8>("8 fma literal + 2 rcp (per 8)", out, sc, blocks);
  run<10>("8 fma, 2 rcp grouped (per 8)", out, sc, blocks);
  run<11>("32 fma, 8 rcp grouped (per 8; x4 work)", out, sc, blocks);
  return 0;
}

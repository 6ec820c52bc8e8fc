// Micro-benchmark: does the issue cost of a three-VGPR-source v_fma_f32 on gfx950 depend on WHICH registers it reads
// (register-file banks, register index mod 4)?  Hand-placed registers through inline assembly; 8 independent
// accumulators per lane, 8 waves / SIMD, like tools/ubench/valu_issue.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return; } } while (0)
#define ITERS 4096
#define CLOB "v8", "v12", "v16", "v20", "v24", "v28", "v32", "v36", "v9", "v13", "v17", "v21", "v25", "v29", "v33", "v37", \
             "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"
#define INIT asm volatile("v_mov_b32 v40, 0x3f7fbe77\n v_mov_b32 v44, 0x3a83126f\n v_mov_b32 v41, 0x3f7fbe77\n v_mov_b32 v45, 0x3a83126f\n" \
                          "v_mov_b32 v46, 0x3a83126f\n v_mov_b32 v42, 0x3f7fbe77\n v_mov_b32 v43, 0x3a83126f\n v_mov_b32 v47, 0x3f7fbe77\n" \
                          "v_mov_b32 v8, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v24, 1.0\n" \
                          "v_mov_b32 v28, 1.0\n v_mov_b32 v32, 1.0\n v_mov_b32 v36, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v13, 1.0\n" \
                          "v_mov_b32 v17, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v33, 1.0\n v_mov_b32 v37, 1.0" ::: CLOB)
// eight FMAs acc = acc * B + C on the bank-0 accumulators v8, v12, ..., v36
#define FMA8(B, C) "v_fma_f32 v8, v8, " B ", " C "\n v_fma_f32 v12, v12, " B ", " C "\n v_fma_f32 v16, v16, " B ", " C "\n" \
                   "v_fma_f32 v20, v20, " B ", " C "\n v_fma_f32 v24, v24, " B ", " C "\n v_fma_f32 v28, v28, " B ", " C "\n" \
                   "v_fma_f32 v32, v32, " B ", " C "\n v_fma_f32 v36, v36, " B ", " C "\n"
// the same with accumulators alternating between bank 0 and bank 1 (v8, v9, v12, v13, ...)
#define FMA8ALT(B, C) "v_fma_f32 v8, v8, " B ", " C "\n v_fma_f32 v9, v9, " B ", " C "\n v_fma_f32 v12, v12, " B ", " C "\n" \
                   "v_fma_f32 v13, v13, " B ", " C "\n v_fma_f32 v16, v16, " B ", " C "\n v_fma_f32 v17, v17, " B ", " C "\n" \
                   "v_fma_f32 v20, v20, " B ", " C "\n v_fma_f32 v21, v21, " B ", " C "\n"
#define FMAC8(B, C) "v_fmac_f32 v8, " B ", " C "\n v_fmac_f32 v12, " B ", " C "\n v_fmac_f32 v16, " B ", " C "\n" \
                   "v_fmac_f32 v20, " B ", " C "\n v_fmac_f32 v24, " B ", " C "\n v_fmac_f32 v28, " B ", " C "\n" \
                   "v_fmac_f32 v32, " B ", " C "\n v_fmac_f32 v36, " B ", " C "\n"
#define FMAAK8(B) "v_fmaak_f32 v8, v8, " B ", 0x3a83126f\n v_fmaak_f32 v12, v12, " B ", 0x3a83126f\n v_fmaak_f32 v16, v16, " B ", 0x3a83126f\n" \
                  "v_fmaak_f32 v20, v20, " B ", 0x3a83126f\n v_fmaak_f32 v24, v24, " B ", 0x3a83126f\n v_fmaak_f32 v28, v28, " B ", 0x3a83126f\n" \
                  "v_fmaak_f32 v32, v32, " B ", 0x3a83126f\n v_fmaak_f32 v36, v36, " B ", 0x3a83126f\n"
#define MUL8(B) "v_mul_f32 v8, v8, " B "\n v_mul_f32 v12, v12, " B "\n v_mul_f32 v16, v16, " B "\n v_mul_f32 v20, v20, " B "\n" \
                "v_mul_f32 v24, v24, " B "\n v_mul_f32 v28, v28, " B "\n v_mul_f32 v32, v32, " B "\n v_mul_f32 v36, v36, " B "\n"

template <int MODE> __global__ __launch_bounds__(256) void k(float* out) {
  INIT;
  asm volatile("s_mov_b32 s20, 0x3f7fbe77\n s_mov_b32 s21, 0x3a83126f" ::: "s20", "s21");
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) asm volatile(FMA8("v40", "v44") ::: CLOB);        // acc, B, C all in bank 0
    if (MODE == 1) asm volatile(FMA8("v41", "v46") ::: CLOB);        // banks 0, 1, 2
    if (MODE == 2) asm volatile(FMA8("v41", "v45") ::: CLOB);        // acc bank 0, B and C both bank 1
    if (MODE == 3) asm volatile(FMA8("v40", "v45") ::: CLOB);        // acc and B bank 0, C bank 1
    if (MODE == 4) asm volatile(FMA8("v41", "v41") ::: CLOB);        // B == C (two distinct registers read)
    if (MODE == 5) asm volatile(FMAC8("v41", "v46") ::: CLOB);       // v_fmac: acc += B * C, three banks
    if (MODE == 6) asm volatile(FMAC8("v40", "v44") ::: CLOB);       // v_fmac, one bank
    if (MODE == 7) asm volatile(MUL8("v41") ::: CLOB);               // two sources, banks 0, 1
    if (MODE == 8) asm volatile(MUL8("v40") ::: CLOB);               // two sources, one bank
    if (MODE == 9) asm volatile(FMA8ALT("v42", "v47") ::: CLOB);     // accumulators alternate bank 0 / 1, B bank 2, C bank 3
    if (MODE == 11) asm volatile(FMA8("s20", "v41") ::: CLOB, "s20", "s21");   // acc * SGPR + VGPR
    if (MODE == 12) asm volatile(FMA8("v41", "s20") ::: CLOB, "s20", "s21");   // acc * VGPR + SGPR
    if (MODE == 13) asm volatile(MUL8("s20") ::: CLOB, "s20", "s21");          // acc * SGPR
    if (MODE == 14) asm volatile(FMAC8("s20", "v41") ::: CLOB, "s20", "s21");  // v_fmac acc += SGPR * VGPR
    if (MODE == 15) asm volatile("v_cmp_gt_f32 vcc, v8, v41\n v_cndmask_b32 v8, v8, v42, vcc\n v_cmp_gt_f32 vcc, v12, v41\n v_cndmask_b32 v12, v12, v42, vcc\n"
                                 "v_cmp_gt_f32 vcc, v16, v41\n v_cndmask_b32 v16, v16, v42, vcc\n v_cmp_gt_f32 vcc, v20, v41\n v_cndmask_b32 v20, v20, v42, vcc\n" ::: CLOB, "vcc");
    if (MODE == 16) asm volatile("v_rcp_f32 v8, v8\n v_rcp_f32 v12, v12\n v_rcp_f32 v16, v16\n v_rcp_f32 v20, v20\n v_rcp_f32 v24, v24\n v_rcp_f32 v28, v28\n v_rcp_f32 v32, v32\n v_rcp_f32 v36, v36\n" ::: CLOB);
    if (MODE == 17) asm volatile("v_exp_f32 v8, v8\n v_exp_f32 v12, v12\n v_exp_f32 v16, v16\n v_exp_f32 v20, v20\n v_exp_f32 v24, v24\n v_exp_f32 v28, v28\n v_exp_f32 v32, v32\n v_exp_f32 v36, v36\n" ::: CLOB);
    if (MODE == 18) asm volatile("v_rcp_f32 v8, v8\n" FMA8("v41", "v46") "v_rcp_f32 v12, v12\n" FMA8("v41", "v46") ::: CLOB);   // 2 rcp among 16 fma (per 8: 9 instructions)
    if (MODE == 10) asm volatile(FMAAK8("v41") ::: CLOB);                // literal addend (v_fmaak), banks 0, 1
  }
  float r;
  asm volatile("v_add_f32 %0, v8, v12\n v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v20\n v_add_f32 %0, %0, v9" : "=v"(r) :: CLOB);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> void run(const char* name, float* out, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double per_simd = (double)blocks * 4 * ITERS * 8 / 1024.0;
  printf("%-58s %8.3f ms  -> %.2f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / per_simd);
}

int main() {
  float* out;
  int blocks = 256 * 8 * 4;
  if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return 1;
  run<0>("v_fma  acc b0, B b0, C b0 (one bank)", out, blocks);
  run<1>("v_fma  acc b0, B b1, C b2 (three banks)", out, blocks);
  run<2>("v_fma  acc b0, B b1, C b1", out, blocks);
  run<3>("v_fma  acc b0, B b0, C b1", out, blocks);
  run<4>("v_fma  acc b0, B = C b1 (two registers)", out, blocks);
  run<5>("v_fmac acc b0, B b1, C b2", out, blocks);
  run<6>("v_fmac acc b0, B b0, C b0", out, blocks);
  run<7>("v_mul  acc b0, B b1", out, blocks);
  run<8>("v_mul  acc b0, B b0", out, blocks);
  run<9>("v_fma  acc b0/b1 alternating, B b2, C b3", out, blocks);
  run<10>("v_fmaak acc b0, B b1, literal addend", out, blocks);
  run<11>("v_fma  acc * SGPR + VGPR", out, blocks);
  run<12>("v_fma  acc * VGPR + SGPR", out, blocks);
  run<13>("v_mul  acc * SGPR", out, blocks);
  run<14>("v_fmac acc += SGPR * VGPR", out, blocks);
  run<15>("4 x (v_cmp + v_cndmask) (per 8 instructions)", out, blocks);
  run<16>("v_rcp_f32", out, blocks);
  run<17>("v_exp_f32", out, blocks);
  run<18>("16 v_fma (three banks) + 2 v_rcp (per 8 = half of the group)", out, blocks);
  return 0;
}

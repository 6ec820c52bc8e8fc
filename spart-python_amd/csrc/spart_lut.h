// LUT inversion (no counterpart in the reference beyond its exact np.argmin nearest-index search,
// SPART.py:381-387): for every observed spectrum y_m find THE row b of the LUT that minimises
//
//     c_T(b, m) = sum_j w_j (x_bj - y_mj)^2   evaluated in the call's dtype T exactly like this:
//     c = 0;  for j = 0 .. nb-1:  d = x_bj - y_mj;  c = c + (w_j * d) * d      (every operation rounded to T, no FMA;
//                                                                                w = NULL: c = c + d * d)
//
// with ties going to the lowest row index and rows / observations with a non-finite cost never winning (-1 / +inf).
// That is what k_lut_reduce_exact / k_lut_fallback evaluate (under SPART_NO_CONTRACT; bands padded with zeros add exactly 0)
// and what the tests' brute force computes in numpy (tools/lut_brute_force.py).
//
// The search itself is a GEMM + argmin on the matrix cores, used as a FILTER with a proven error bound:
//   1. k_lut_centre   c_j = mean of a strided sample of LUT column j (any c is correct; a good one makes the bound small)
//   2. k_lut_prep     x' = x - c, n_b = sum w x'^2, laid out tile-major for the MFMA operand registers; rows with a
//                     non-finite entry become (0, .., 0, n = +inf): they can never come out as a minimum.
//                     Nmax = max_b sum |w| x'^2 over the finite rows (atomicMax on the bit pattern)
//   3. k_lut_scan_*   a~(b, m) = n_b - 2 sum_j w_j x'_bj y'_mj  on v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64
//                     (K = nb + 1: A[b][k] = x'_bk, n_b in column nb; Bq[k][m] = -2 w_k y'_mk, 1 in row nb).  Per
//                     (slice of the LUT, lane group) and observation the kernel keeps the smallest tile minimum, the tile
//                     it came from, and the SECOND smallest tile minimum (value only).
//   4. k_lut_reduce_exact   g = the smallest a~ over all partial results.  Every tile whose minimum is <= g + Delta is
//                     evaluated with the direct cost, row by row; if some partial result's SECOND minimum is also
//                     <= g + Delta there may be a candidate tile the scan did not remember: the observation goes to
//   5. k_lut_fallback / k_lut_fallback_merge   a plain vector-ALU brute force over the whole LUT with the direct cost
//                     (64 flagged observations on the lanes of a wave, LUT rows staged in LDS and read as broadcasts).
// So the result is the exact argmin whatever the data look like; the data only decide how much work steps 4-5 are.
//
// Delta.  u = unit roundoff of T (2^-24 / 2^-53), K = fused multiply-adds per accumulator chain (2 KS / 4 KS),
// N_b = sum |w| x'_b^2, Y = sum |w| y'^2, c(b) = the real-number cost.  With x' = (x - c)(1 + e), |e| <= u (same for y'):
//   (i)   | sum w (x - y)^2 - sum w (x' - y')^2 |          <= 4 u (N_b + Y)        (|d - (x'-y')| <= u (|x'| + |y'|))
//   (ii)  | n_b computed - sum w x'^2 |                    <= (nb + 2) u N_b
//   (iii) Bq = fl(-2 w y'): | 2 sum w x' y' e |            <= u (N_b + Y)          (2 |x' y'| <= x'^2 + y'^2)
//   (iv)  a chain of K fmas (any order):                   <= K u (|n_b| + 2 sum |w x' y'|) <= 2 K u (N_b + Y)
//   =>    | a~(b) + Y - c(b) | <= E = (nb + 2 K + 7) u (N_b + Y)
//   (v)   direct evaluation:  | c_T(b) - c(b) | <= F = (nb + 3) u sum |w| d^2 <= (2 nb + 6) u (N_b + Y)
// If b is the argmin of c_T and a the row with a~(a) = g:  a~(b) + Y <= c(b) + E_b <= c_T(b) + E_b + F_b <= c_T(a) + E_b + F_b
// <= c(a) + F_a + E_b + F_b <= g + Y + (E_a + F_a) + (E_b + F_b), i.e. a~(b) <= g + Delta with
//     Delta = (3 nb + 2 K + 13) u [(N_a + Y) + (N_b + Y)].
// N_a and N_b are not known to the reduce kernel, but bounded:
//   * always:  N_a, N_b <= Nmax  (the prep kernel's maximum over all finite rows);
//   * with non-negative weights (c is then a squared distance in the |w|-weighted norm), per observation:
//       N_a <= Na = max of n over the rows of the tile that produced g (read back from the operand image);
//       c(a) <= max(0, g + Y) + E_a =: cub;  c(b) <= c(a) (1 + 3 (nb + 3) u);  sqrt(N_b) <= sqrt(Y) + sqrt(c'(b)) with
//       c'(b) = sum w (x'_b - y')^2 <= c(b) + 4 u (Nmax + Y)   =>   N_b <= Nstar = (sqrt(Y) + sqrt(cub + 4 u (Nmax + Y)))^2.
//     For a LUT of spectra and an observation that resembles some of them Na and Nstar are ~Y, an order of magnitude below
//     Nmax: fewer candidate tiles, and far fewer observations for which the scan's top-1 + runner-up per slice is not enough.
// The code uses (3 nb + 2 K + 16) * 1.01 (second-order terms, the rounding of g + Delta and of Y itself, the square roots)
// plus 2^-100 / 1e-290 for products that underflow.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spart_math.h"      // SPART_NO_CONTRACT

namespace spart {

template <typename T> struct LutNum;
template <> struct LutNum<float> {
  static constexpr float u = 5.9604644775390625e-8f, tiny = 7.888609052210118e-31f;
  static __device__ __forceinline__ bool finite(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }
  static __device__ __forceinline__ unsigned long long bits(float v) { return (unsigned long long)__float_as_uint(v); }
  static __device__ __forceinline__ float from_bits(unsigned long long b) { return __uint_as_float((unsigned)b); }
};
template <> struct LutNum<double> {
  static constexpr double u = 1.1102230246251565e-16, tiny = 1e-290;
  static __device__ __forceinline__ bool finite(double v) {
    return ((unsigned long long)__double_as_longlong(v) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull;
  }
  static __device__ __forceinline__ unsigned long long bits(double v) { return (unsigned long long)__double_as_longlong(v); }
  static __device__ __forceinline__ double from_bits(unsigned long long b) { return __longlong_as_double((long long)b); }
};

// ctl[0] = bit pattern of Nmax (non-negative floats order like unsigned integers), ctl[1] = number of flagged observations
constexpr int LUT_CTL_WORDS = 2;
constexpr int LUT_CENTRE_ROWS = 8192;      // rows sampled for the column means
constexpr int LUT_FB_BLOCKS = 2048;        // workgroups (x 4 waves) of the brute-force fallback
// float32 scan: 32-observation blocks per wave (256 observations, 8 KS operand registers + 8 x 16 accumulators: 106 ... 226
// VGPRs for KS = 4 ... 16).  Eight independent accumulator chains per wave keep the matrix pipe fed from TWO or three resident
// waves per SIMD -- v_mfma_f32_32x32x2_f32 sustains 27.2 ns per instruction and SIMD (154 Tflop/s) with two issuing waves
// and 33.9 ns with four (tools/ubench/mfma_f32_rate2.hip) -- and every LUT tile loaded is used for twice the observations.
// Measured 1M x 65 536, nb = 13: 4 / 6 / 7 / 8 / 10 / 12 / 16 blocks: 15.25 / 15.24 / 15.11 / 14.45 / 15.46 / 15.71 / 14.97 ms
// (profiles/r4_lut_to_sweep2.txt); nb = 6: 9.25 -> 9.10 ms, nb = 21: unchanged.
#ifndef SPART_LUT_TO
#define SPART_LUT_TO 8
#endif
constexpr int LUT_TO = SPART_LUT_TO;

// column means of a strided sample (finite entries only) -> centre[nb]; also resets the control words
template <typename T>
__global__ __launch_bounds__(256) void k_lut_centre(const T* __restrict__ lut, int nb, int64_t B, T* __restrict__ centre,
                                                    unsigned long long* __restrict__ ctl) {
  __shared__ double ss[256];
  __shared__ int sn[256];
  const int j = blockIdx.x;
  const int64_t ns = B < LUT_CENTRE_ROWS ? B : LUT_CENTRE_ROWS;
  const int64_t stride = ns > 0 ? B / ns : 1;
  double s = 0.0;
  int n = 0;
  for (int64_t i = threadIdx.x; i < ns; i += 256) {
    const T v = lut[i * stride * nb + j];
    if (LutNum<T>::finite(v)) {
      s += (double)v;
      ++n;
    }
  }
  ss[threadIdx.x] = s;
  sn[threadIdx.x] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      ss[threadIdx.x] += ss[threadIdx.x + off];
      sn[threadIdx.x] += sn[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const T c = sn[0] > 0 ? (T)(ss[0] / sn[0]) : T(0);
    centre[j] = LutNum<T>::finite(c) ? c : T(0);
    if (j == 0) {
      ctl[0] = 0ull;
      ctl[1] = 0ull;
    }
  }
}

// One thread per LUT row (padding rows of the last tile included): centred row -> the operand-register image of its
// tile, [tile][kk][lane] with lane = (column % KPER) * ROWS + row % ROWS, column = KPER kk + lane / ROWS.
//   ROWS = 32, KPER = 2: v_mfma_f32_32x32x2_f32;  ROWS = 16, KPER = 4: v_mfma_f64_16x16x4_f64.
template <typename T, int KS, int ROWS>
__global__ __launch_bounds__(256) void k_lut_prep(const T* __restrict__ lut, const T* __restrict__ w, const T* __restrict__ centre,
                                                  int nb, int64_t B, int64_t ntile, T* __restrict__ tiles,
                                                  unsigned long long* __restrict__ ctl) {
  constexpr int KPER = 64 / ROWS;
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  T na = T(0);
  if (row < ntile * ROWS) {
    bool ok = row < B;
    T n = T(0);
    if (ok) {
      const T* x = lut + row * nb;
      for (int j = 0; j < nb; ++j) {
        const T v = x[j];
        ok = ok && LutNum<T>::finite(v);
        const T xc = v - centre[j];
        const T wj = w ? w[j] : T(1);
        n += wj * xc * xc;
        na += (wj < T(0) ? -wj : wj) * xc * xc;
      }
      ok = ok && LutNum<T>::finite(n) && LutNum<T>::finite(na);
    }
    if (!ok) na = T(0);
    T* dst = tiles + (row / ROWS) * (int64_t)(KS * 64) + (row % ROWS);
    for (int c = 0; c < KS * KPER; ++c) {
      T v = T(0);
      if (c < nb) v = ok ? lut[row * nb + c] - centre[c] : T(0);
      else if (c == nb) v = ok ? n : (T)INFINITY;
      dst[(c / KPER) * 64 + (c % KPER) * ROWS] = v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T o = __shfl_down(na, off, 64);
    na = o > na ? o : na;
  }
  if ((threadIdx.x & 63) == 0 && na > T(0)) atomicMax(&ctl[0], LutNum<T>::bits(na));
}

// float32 scan on the exact-f32 matrix cores.  A 32 x 32 x 2 MFMA takes ONE register of A (lane l: row l % 32,
// k = l / 32) and one of Bq (lane l: column l % 32, k = l / 32); K = 2 KS is covered by KS of them chained on one
// 16-register accumulator (lane l ends up with 16 LUT rows of observation l % 32).  Each wave keeps LUT_TO = 8 blocks of 32
// observations in registers (LUT_TO x KS operand registers) and streams the LUT tiles of its slice past them.
// The matrix pipe does the arithmetic (KS x 64 cycles per 1024 comparisons); the vector ALU only takes the minimum of
// the 16 accumulator values (v_min3) and keeps, per lane, the smallest and second smallest of those tile minima and
// WHICH TILE the smallest came from (v_cmp, v_med3, two v_cndmask per tile and block).
typedef float spart_f16v __attribute__((ext_vector_type(16)));

template <int KS>
__global__ __launch_bounds__(256, 2) void k_lut_scan_mfma(const float* __restrict__ tiles, const float* __restrict__ obs,
                                                          const float* __restrict__ w, const float* __restrict__ centre, int nb,
                                                          int64_t ntile, int64_t M, int nslice, float* __restrict__ part_cost,
                                                          float* __restrict__ part_sec, int* __restrict__ part_tile) {
  const int lane = threadIdx.x & 63;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (32 * LUT_TO);   // this wave's first observation
  if (m0 >= M) return;                                                                 // (whole wave; no barrier below)
  const int slice = blockIdx.y;
  const int j = lane & 31, half = lane >> 5;
  float bq[LUT_TO][KS];
#pragma unroll
  for (int blk = 0; blk < LUT_TO; ++blk) {
    const int64_t m = m0 + blk * 32 + j;
    const int64_t mc = m < M ? m : M - 1;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int col = 2 * kk + half;
      bq[blk][kk] = col < nb ? -2.0f * (w ? w[col] : 1.0f) * (obs[mc * nb + col] - centre[col]) : (col == nb ? 1.0f : 0.0f);
    }
  }
  const int64_t per = (ntile + nslice - 1) / nslice;
  const int64_t t0 = per * slice;
  const int64_t t1 = (t0 + per < ntile) ? t0 + per : ntile;
  float best[LUT_TO], sec[LUT_TO];
  int bt[LUT_TO];
#pragma unroll
  for (int blk = 0; blk < LUT_TO; ++blk) {
    best[blk] = INFINITY;
    sec[blk] = INFINITY;
    bt[blk] = -1;
  }
  if (t0 < t1) {
    const float* __restrict__ ap = tiles + t0 * (KS * 64) + lane;
    float a[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) a[kk] = ap[kk * 64];
    for (int64_t t = t0; t < t1; ++t) {
      if (t + 1 < t1) ap += KS * 64;                   // next tile's operands in flight during this tile's MFMAs
      float an[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) an[kk] = ap[kk * 64];
#pragma unroll
      for (int blk = 0; blk < LUT_TO; ++blk) {
        spart_f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], bq[blk][kk], acc, 0, 0, 0);
        // minimum of the 16 accumulator values: v_min3_f32 written out (fminf() makes the compiler canonicalise the MFMA
        // results first, two v_max per block; v_min / v_min3 return the non-NaN operand whatever its kind)
        float mn;
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(mn) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
#pragma unroll
        for (int r = 3; r < 15; r += 2) asm("v_min3_f32 %0, %1, %2, %3" : "=v"(mn) : "v"(mn), "v"(acc[r]), "v"(acc[r + 1]));
        asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(mn), "v"(acc[15]));
        // (best, sec) <- the two smallest of (best, sec, mn); a tile minimum EQUAL to the best so far becomes `sec`, so
        // an exact tie between tiles is seen by the reduce kernel
        const bool lt = mn < best[blk];
        sec[blk] = __builtin_amdgcn_fmed3f(best[blk], mn, sec[blk]);
        best[blk] = lt ? mn : best[blk];
        bt[blk] = lt ? (int)(t - t0) : bt[blk];
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) a[kk] = an[kk];
    }
  }
#pragma unroll
  for (int blk = 0; blk < LUT_TO; ++blk) {
    const int64_t m = m0 + blk * 32 + j;
    if (m < M) {
      const int64_t o = ((int64_t)slice * 2 + half) * M + m;
      part_cost[o] = best[blk];
      part_sec[o] = sec[blk];
      part_tile[o] = bt[blk] < 0 ? -1 : (int)(t0 + bt[blk]);
    }
  }
}

// float64: the same scan on v_mfma_f64_16x16x4_f64 (K steps of 4, 16 x 16 tiles, four accumulator values per lane: lane l
// holds four LUT rows of observation l % 16; the four lane groups l / 16 keep separate partial results).
typedef double spart_d4v __attribute__((ext_vector_type(4)));

template <int KS, int TO>
__global__ __launch_bounds__(256, 2) void k_lut_scan_mfma64(const double* __restrict__ tiles, const double* __restrict__ obs,
                                                            const double* __restrict__ w, const double* __restrict__ centre, int nb,
                                                            int64_t ntile, int64_t M, int nslice, double* __restrict__ part_cost,
                                                            double* __restrict__ part_sec, int* __restrict__ part_tile) {
  const int lane = threadIdx.x & 63;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (16 * TO);
  if (m0 >= M) return;
  const int slice = blockIdx.y;
  const int j = lane & 15, q = lane >> 4;
  double bq[TO][KS];
#pragma unroll
  for (int blk = 0; blk < TO; ++blk) {
    const int64_t m = m0 + blk * 16 + j;
    const int64_t mc = m < M ? m : M - 1;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      const int col = 4 * kk + q;
      bq[blk][kk] = col < nb ? -2.0 * (w ? w[col] : 1.0) * (obs[mc * nb + col] - centre[col]) : (col == nb ? 1.0 : 0.0);
    }
  }
  const int64_t per = (ntile + nslice - 1) / nslice;
  const int64_t t0 = per * slice;
  const int64_t t1 = (t0 + per < ntile) ? t0 + per : ntile;
  double best[TO], sec[TO];
  int bt[TO];
#pragma unroll
  for (int blk = 0; blk < TO; ++blk) {
    best[blk] = INFINITY;
    sec[blk] = INFINITY;
    bt[blk] = -1;
  }
  if (t0 < t1) {
    const double* __restrict__ ap = tiles + t0 * (KS * 64) + lane;
    double a[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) a[kk] = ap[kk * 64];
    for (int64_t t = t0; t < t1; ++t) {
      if (t + 1 < t1) ap += KS * 64;
      double an[KS];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) an[kk] = ap[kk * 64];
#pragma unroll
      for (int blk = 0; blk < TO; ++blk) {
        spart_d4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bq[blk][kk], acc, 0, 0, 0);
        const double mn = __builtin_fmin(__builtin_fmin(acc[0], acc[1]), __builtin_fmin(acc[2], acc[3]));
        // mn is never NaN for a finite observation (non-finite rows were replaced by n = +inf in k_lut_prep)
        const bool lt = mn < best[blk];
        sec[blk] = __builtin_fmin(sec[blk], __builtin_fmax(best[blk], mn));
        best[blk] = lt ? mn : best[blk];
        bt[blk] = lt ? (int)(t - t0) : bt[blk];
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) a[kk] = an[kk];
    }
  }
#pragma unroll
  for (int blk = 0; blk < TO; ++blk) {
    const int64_t m = m0 + blk * 16 + j;
    if (m < M) {
      const int64_t o = ((int64_t)slice * 4 + q) * M + m;
      part_cost[o] = best[blk];
      part_sec[o] = sec[blk];
      part_tile[o] = bt[blk] < 0 ? -1 : (int)(t0 + bt[blk]);
    }
  }
}

// One WAVE per observation (four per workgroup): lanes over the partial results (threshold), then the candidate tiles
// are evaluated with the direct cost by 64 / ROWS lane groups at a time, one LUT row per lane.
// coef = 2 (3 nb + 2 K + 16) * 1.01 * u (host: lut_delta_coef).
template <typename T>
__device__ __forceinline__ bool lut_better(T oc, int64_t oi, T bc, int64_t bi) {     // (cost, row) lexicographic; (inf, -1) is worst
  // a cost of -inf (negative weights whose products overflow) is NOT FINITE and never wins, like NaN and +inf
  return (oc < bc && oc > -(T)INFINITY) || (oc == bc && oi >= 0 && oi < bi);
}

template <typename T, int ROWS>
__global__ __launch_bounds__(256) void k_lut_reduce_exact(const T* __restrict__ part_cost, const T* __restrict__ part_sec,
                                                          const int* __restrict__ part_tile, const T* __restrict__ tiles, int ks,
                                                          const T* __restrict__ lut, const T* __restrict__ obs,
                                                          const T* __restrict__ w, const T* __restrict__ centre, int nb, int64_t B,
                                                          int64_t M, int npart, T coef_e, T coef_ef,
                                                          unsigned long long* __restrict__ ctl, int* __restrict__ flag_list,
                                                          int64_t* __restrict__ best_idx, T* __restrict__ best_cost) {
  constexpr int NG = 64 / ROWS;                        // candidate tiles evaluated side by side (= k columns per operand register)
  __shared__ T ysm[4][32], wsm[32];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t mraw = (int64_t)blockIdx.x * 4 + wv;
  const bool active = mraw < M;
  const int64_t m = active ? mraw : M - 1;
  const T yv = lane < nb ? obs[m * nb + lane] : T(0);
  if (lane < 32) ysm[wv][lane] = yv;                   // zero-padded to 32: a padded band adds (w * 0) * 0 = 0 to the cost
  if (threadIdx.x < 32) wsm[threadIdx.x] = (w && (int)threadIdx.x < nb) ? w[threadIdx.x] : T(0);
  __syncthreads();
  if (!active) return;
  const bool yfin = __all(LutNum<T>::finite(yv)) != 0;
  if (!yfin) {                                         // every direct cost is NaN or +inf: nothing wins
    if (lane == 0) {
      best_idx[m] = -1;
      best_cost[m] = (T)INFINITY;
    }
    return;
  }
  T ya = T(0);
  bool wneg = false;
  if (lane < nb) {
    const T yc = yv - centre[lane];
    const T wj = w ? w[lane] : T(1);
    wneg = wj < T(0);
    ya = (wneg ? -wj : wj) * yc * yc;
  }
  wneg = __any(wneg) != 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ya += __shfl_xor(ya, off, 64);
  T g = (T)INFINITY;
  int ta = -1;
  for (int p = lane; p < npart; p += 64) {
    const T c = part_cost[(int64_t)p * M + m];
    const int t = part_tile[(int64_t)p * M + m];
    if (t >= 0 && c < g) {
      g = c;
      ta = t;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T o = __shfl_xor(g, off, 64);
    const int ot = __shfl_xor(ta, off, 64);
    if (o < g || (o == g && ot > ta)) {                // (any tile that attains g will do; the tie rule only makes all lanes agree)
      g = o;
      ta = ot;
    }
  }
  const T nmax = LutNum<T>::from_bits(ctl[0]);
  T na = nmax, nstar = nmax;
  if (!wneg && ta >= 0) {
    // N_a <= the largest n among the rows of tile ta (n = sum w x'^2 = N for w >= 0; padding / non-finite rows hold +inf)
    T v = T(0);
    if (lane < ROWS) {
      const T n = tiles[((int64_t)ta * ks + nb / NG) * 64 + (nb % NG) * ROWS + lane];
      v = LutNum<T>::finite(n) ? n : T(0);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const T o = __shfl_xor(v, off, 64);
      v = o > v ? o : v;
    }
    const T gy = g + ya;
    const T cub = ((gy > T(0) ? gy : T(0)) + T(2) * coef_e * (v + ya) + T(4) * LutNum<T>::u * (nmax + ya)) * T(1.001);
    const T r = (T)__builtin_sqrt((double)ya) + (T)__builtin_sqrt((double)cub);
    const T ns = r * r * T(1.001);
    if (v < na) na = v;
    if (ns < nstar) nstar = ns;                        // (never above the global bound; NaN / inf leave it in place)
  }
  const T thr = g + (coef_ef * ((na + ya) + (nstar + ya)) + LutNum<T>::tiny);
  // no finite filter value (e.g. an all-NaN LUT) or no finite threshold (overflow): let the brute force decide
  bool over = !(g < (T)INFINITY) || !LutNum<T>::finite(thr);
  T bc = (T)INFINITY;
  int64_t bi = -1;
  const T* ys = ysm[wv];
  const int grp = lane / ROWS, rl = lane % ROWS;
  for (int p0 = 0; p0 < npart && !over; p0 += 64) {
    const int p = p0 + lane;
    int t = -1;
    bool cand = false, second = false;
    if (p < npart) {
      t = part_tile[(int64_t)p * M + m];
      cand = t >= 0 && part_cost[(int64_t)p * M + m] <= thr;
      second = cand && part_sec[(int64_t)p * M + m] <= thr;     // a second tile of this partial result may hold the minimum
    }
    if (__any(second)) {
      over = true;
      break;
    }
    unsigned long long mask = __ballot(cand);
    while (mask) {                                     // NG candidate tiles per round, one row per lane
      int myt = -1;
#pragma unroll
      for (int q = 0; q < NG; ++q) {
        if (mask) {
          const int L = __builtin_ctzll(mask);
          mask &= mask - 1;
          const int tq = __shfl(t, L, 64);
          if (grp == q) myt = tq;
        }
      }
      const int64_t r = (int64_t)myt * ROWS + rl;
      if (myt >= 0 && r < B) {
        SPART_NO_CONTRACT
        const T* x = lut + r * nb;
        T c = T(0);
        for (int j = 0; j < nb; j += 4) {              // four bands in flight; the padded ones contribute exactly 0
          const T x0 = x[j], x1 = j + 1 < nb ? x[j + 1] : T(0), x2 = j + 2 < nb ? x[j + 2] : T(0), x3 = j + 3 < nb ? x[j + 3] : T(0);
          const T d0 = x0 - ys[j], d1 = x1 - ys[j + 1], d2 = x2 - ys[j + 2], d3 = x3 - ys[j + 3];
          if (w) {
            c = c + (wsm[j] * d0) * d0;
            c = c + (wsm[j + 1] * d1) * d1;
            c = c + (wsm[j + 2] * d2) * d2;
            c = c + (wsm[j + 3] * d3) * d3;
          } else {
            c = c + d0 * d0;
            c = c + d1 * d1;
            c = c + d2 * d2;
            c = c + d3 * d3;
          }
        }
        if (lut_better(c, r, bc, bi)) {
          bc = c;
          bi = r;
        }
      }
    }
  }
  if (over) {
    if (lane == 0) {
      const unsigned long long pos = atomicAdd(&ctl[1], 1ull);
      flag_list[pos] = (int)m;
    }
    return;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T oc = __shfl_xor(bc, off, 64);
    const int64_t oi = __shfl_xor(bi, off, 64);
    if (lut_better(oc, oi, bc, bi)) {
      bc = oc;
      bi = oi;
    }
  }
  if (lane == 0) {
    best_idx[m] = bi;
    best_cost[m] = bc;
  }
}

// Work split of the fallback: `count` flagged observations in groups of 64 (one per lane), each group's scan of the
// LUT cut into `spg` slices of `per` rows so that the W waves of the launch all have work.
__host__ __device__ inline void lut_fb_partition(unsigned count, int64_t B, int W, int& ngroups, int& spg, int64_t& per) {
  ngroups = (int)((count + 63u) / 64u);
  int64_t cap = (B + 255) / 256;
  if (cap < 1) cap = 1;
  int64_t s = ngroups > 0 ? W / ngroups : 1;
  if (s < 1) s = 1;
  if (s > cap) s = cap;
  spg = (int)s;
  per = (B + s - 1) / s;
}

// Brute force with the direct cost for the flagged observations.  Lane = observation (its bands and the weights in
// registers, NBC = nb rounded up to a multiple of 4, zero-padded: a padded band adds exactly 0); the wave stages LUT_FB_ROWS
// LUT rows at a time in its own LDS block (zero-padded to NBC) and reads them back as wave-uniform broadcasts.
// Dynamic LDS: 4 waves x LUT_FB_ROWS x NBC x sizeof(T).
constexpr int LUT_FB_ROWS = 32;

template <typename T, int NBC>
__global__ __launch_bounds__(256) void k_lut_fallback(const T* __restrict__ lut, const T* __restrict__ obs, const T* __restrict__ w,
                                                      int nb, int64_t B, const unsigned long long* __restrict__ ctl,
                                                      const int* __restrict__ flag_list, T* __restrict__ fb_cost,
                                                      int64_t* __restrict__ fb_idx) {
  SPART_NO_CONTRACT
  extern __shared__ char lut_smem[];
  const unsigned count = (unsigned)ctl[1];
  if (count == 0u) return;
  const int W = (int)gridDim.x * 4;
  int ngroups, spg;
  int64_t per;
  lut_fb_partition(count, B, W, ngroups, spg, per);
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  T* xs = reinterpret_cast<T*>(lut_smem) + (size_t)wv * LUT_FB_ROWS * NBC;
  T wr[NBC];
#pragma unroll
  for (int j = 0; j < NBC; ++j) wr[j] = (w && j < nb) ? w[j] : T(0);
  const int64_t nitem = (int64_t)ngroups * spg;
  for (int64_t item = (int64_t)blockIdx.x * 4 + wv; item < nitem; item += W) {
    const int group = (int)(item / spg), s = (int)(item % spg);
    const unsigned f = (unsigned)group * 64u + (unsigned)lane;
    const int64_t m = flag_list[f < count ? f : (unsigned)group * 64u];     // (lanes past the end repeat the group's first)
    T y[NBC];
#pragma unroll
    for (int j = 0; j < NBC; ++j) y[j] = j < nb ? obs[m * nb + j] : T(0);
    const int64_t r0 = (int64_t)s * per, r1 = (r0 + per < B) ? r0 + per : B;
    T bc = (T)INFINITY;
    int64_t bi = -1;
    for (int64_t rb = r0; rb < r1; rb += LUT_FB_ROWS) {
      const int nrow = (int)(r1 - rb < LUT_FB_ROWS ? r1 - rb : LUT_FB_ROWS);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                 // (the previous block has been read)
      __builtin_amdgcn_wave_barrier();
      for (int e = lane; e < LUT_FB_ROWS * NBC; e += 64) {
        const int rr = e / NBC, j = e % NBC;
        xs[e] = (rr < nrow && j < nb) ? lut[(rb + rr) * nb + j] : T(0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      for (int rr = 0; rr < nrow; ++rr) {
        const T* x = xs + rr * NBC;
        T c = T(0);
        if (w) {
#pragma unroll
          for (int j = 0; j < NBC; ++j) {
            const T d = x[j] - y[j];
            c = c + (wr[j] * d) * d;
          }
        } else {
#pragma unroll
          for (int j = 0; j < NBC; ++j) {
            const T d = x[j] - y[j];
            c = c + d * d;
          }
        }
        if (c < bc && c > -(T)INFINITY) {              // ascending rows + strict '<': ties to the lowest row; -inf is not finite
          bc = c;
          bi = rb + rr;
        }
      }
    }
    fb_cost[item * 64 + lane] = bc;
    fb_idx[item * 64 + lane] = bi;
  }
}

// one wave per flagged observation, lanes over the slices of its brute-force scan
template <typename T>
__global__ __launch_bounds__(256) void k_lut_fallback_merge(int64_t B, int W, const unsigned long long* __restrict__ ctl,
                                                            const int* __restrict__ flag_list, const T* __restrict__ fb_cost,
                                                            const int64_t* __restrict__ fb_idx, int64_t* __restrict__ best_idx,
                                                            T* __restrict__ best_cost) {
  const unsigned count = (unsigned)ctl[1];
  const int lane = threadIdx.x & 63;
  const unsigned nwave = gridDim.x * 4u;
  int ngroups, spg;
  int64_t per;
  lut_fb_partition(count, B, W, ngroups, spg, per);
  for (unsigned f = blockIdx.x * 4u + (threadIdx.x >> 6); f < count; f += nwave) {
    const int64_t base = (int64_t)(f / 64u) * spg;
    const int l = (int)(f % 64u);
    T bc = (T)INFINITY;
    int64_t bi = -1;
    for (int s = lane; s < spg; s += 64) {
      const T c = fb_cost[(base + s) * 64 + l];
      const int64_t i = fb_idx[(base + s) * 64 + l];
      if (lut_better(c, i, bc, bi)) {
        bc = c;
        bi = i;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const T oc = __shfl_xor(bc, off, 64);
      const int64_t oi = __shfl_xor(bi, off, 64);
      if (lut_better(oc, oi, bc, bi)) {
        bc = oc;
        bi = oi;
      }
    }
    if (lane == 0) {
      const int m = flag_list[f];
      best_idx[m] = bi;
      best_cost[m] = bc;
    }
  }
}

}  // namespace spart

// SPART hot path: per-band and per-sample arithmetic, shared by every kernel.
//
// Everything here is a plain templated inline function of registers (no memory traffic, no
// thread indexing), so the same code is compiled by hipcc for gfx950 (the product) and by
// g++ for the CPU-side arithmetic checks in tests/hostmath (test infrastructure only).
//
// The formulas are algebraic rearrangements of the reference's (file:line cited at each
// function) chosen so that float32 keeps ~1e-6 relative accuracy where the reference's
// literal float64 forms would cancel catastrophically:
//   * leaf plate transmittance  tau = (1-K)e^-K + K^2 E1(K) == 2 E3(K)  evaluated from
//     fitted forms that also return u = 1 - tau without cancellation,
//   * every "1 - r - t" absorptance of the plate model is carried analytically,
//   * Stokes N-layer terms are divided through by b^(2(N-1)) (no overflow),
//   * SAIL: m^2 = a^2 - sigb^2 == (1 - rho - tau)(a + sigb),  rinf == sigb / (a + m),
//     J1/J2 through phi(d) = (1 - e^-d)/d.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SPART_HD __host__ __device__ __forceinline__
#else
#define SPART_HD inline
#endif

#if defined(__clang__)
// sample-level double arithmetic must round exactly like the reference's (e.g. the hot-spot
// test dso == 0, sailh.py:78,120): no FMA contraction there
#define SPART_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define SPART_NO_CONTRACT
#endif

// True when the condition holds in ANY lane of the wave (device) / for this value (host): lets a rare per-lane
// branch be skipped with a scalar branch instead of being issued under an all-zero exec mask (short masked
// blocks get no s_cbranch_execz from the compiler and cost their full issue slots)
// SPART_KEEP_BRANCH(x), placed inside such a block, stops the compiler from if-converting it back into
// unconditional arithmetic + select (an empty, non-speculatable asm that "touches" x).
#if defined(__HIP_DEVICE_COMPILE__)
#define SPART_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#define SPART_KEEP_BRANCH(x) asm volatile("" : "+v"(x))
#else
#define SPART_WAVE_ANY(c) (c)
#define SPART_KEEP_BRANCH(x) ((void)0)
#endif

#include "spart_e3_coeffs.h"
#include "spart_f64_tables.h"

namespace spart {

constexpr int NWL = 2001;    // 400..2400 nm, 1 nm          (SPART.py:303)
constexpr int NWLT = 161;    // thermal pad                  (SPART.py:307-309)
constexpr int NWLS = 2162;   // NWL + NWLT                   (SPART.py:310)
constexpr int NLAYER = 60;   // canopy layers                (sailh.py:345)
constexpr int NLINCL = 13;   // leaf inclination classes     (sailh.py:346)
constexpr int NEVAL = 2002;  // band evaluations per spectrum: 2001 optical + ONE thermal (all 161 are identical)
constexpr int NPARAM = 27;
constexpr int NCOEF = 48;    // SMAC coefficient rows used   (smac.py:44-92)

constexpr double PI = 3.14159265358979323846;

// ------------------------------------------------------------------------------------------
// scalar math wrappers
template <typename T> struct Mx;

template <> struct Mx<float> {
  static SPART_HD float exp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
    return __expf(x);
#else
    return ::expf(x);
#endif
  }
  static SPART_HD float exp2(float x) {   // arguments here are <= 0 and far from the denormal range of the result
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
    return __builtin_amdgcn_exp2f(x);     // v_exp_f32
#else
    return ::exp2f(x);
#endif
  }
  static SPART_HD float log(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
    return __builtin_amdgcn_logf(x) * 0.693147181f;   // v_log_f32 (log2), arguments here are normal floats
#else
    return ::logf(x);
#endif
  }
  static SPART_HD float sqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
    return __builtin_amdgcn_sqrtf(x);
#else
    return ::sqrtf(x);
#endif
  }
  static SPART_HD float rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
  }
  // log1p for x >= 0 (the only use: ln b, b >= 1): 2 atanh(x/(2+x)) series below 0.5, log(1+x) above
  static SPART_HD float log1p(float x) {
#if defined(SPART_FAST_MATH)
    float s = x * rcp(2.0f + x);
    float s2 = s * s;
    float p = 2.0f * s * (1.0f + s2 * (0.333333333f + s2 * (0.2f + s2 * (0.142857143f + s2 * 0.111111111f))));
    return (x < 0.5f) ? p : log(1.0f + x);
#else
    return ::log1pf(x);
#endif
  }
  // 1 - e^-z for z >= 0 without cancellation: Taylor below 0.25 (z^8/8! = 3.8e-10), direct above
  static SPART_HD float one_minus_exp_neg(float z) {
#if defined(SPART_FAST_MATH)
    float p = z * (1.0f + z * (-0.5f + z * (0.166666667f + z * (-0.0416666667f + z * (8.33333333e-3f +
              z * (-1.38888889e-3f + z * 1.98412698e-4f))))));
    return (z < 0.25f) ? p : 1.0f - exp(-z);
#else
    return -::expm1f(-z);
#endif
  }
  // the same with ez = e^-z already at hand: the Taylor side is only issued when some lane needs it
  static SPART_HD float one_minus_exp_neg(float z, float ez) {
#if defined(SPART_FAST_MATH)
    float v = 1.0f - ez;
    if (SPART_WAVE_ANY(z < 0.25f)) {
      float p = z * (1.0f + z * (-0.5f + z * (0.166666667f + z * (-0.0416666667f + z * (8.33333333e-3f +
                z * (-1.38888889e-3f + z * 1.98412698e-4f))))));
      v = (z < 0.25f) ? p : v;
    }
    return v;
#else
    (void)ez;
    return -::expm1f(-z);
#endif
  }
  static SPART_HD float fabs(float x) { return ::fabsf(x); }
  static SPART_HD float fmax(float a, float b) { return ::fmaxf(a, b); }
  static SPART_HD float tiny() { return 1e-30f; }
};

#if defined(__HIPCC__)
// (Device tables in this header are `static`: the library is built from more than one translation unit -- the float32
// full-band kernels are compiled on their own, spart_bands_f32.hip -- and each unit carries its own copy.)
// Coefficient tables in constant memory are read through a pointer the optimiser cannot prove loop-invariant
// (SPART_FRESH): left alone, hipcc hoists all 48 float64 coefficients of the band path (96 SGPRs) out of the sample
// loop, runs out of SGPRs and parks them in VGPR lanes -- 32 v_readlane / v_writelane per band and sample in
// k_prospect<double>, 16 % of its VALU instructions.  Reloaded per use they cost a few wide scalar loads (scalar
// cache) and no VALU slot.
#ifndef SPART_FRESH_COEF
#define SPART_FRESH_COEF 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
typedef const double __attribute__((address_space(4))) * spart_cdp;   // constant address space: uniform loads stay scalar
#else
typedef const double* spart_cdp;
#endif
__device__ __forceinline__ spart_cdp spart_fresh(const double* p) {
  spart_cdp q = (spart_cdp)p;
#if SPART_FRESH_COEF && defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(q));
#endif
  return q;
}
// One Horner step p * x + c with the coefficient c taken from an SGPR pair as the ADDEND of a three-address
// v_fma_f64.  Written as C++, hipcc selects the two-address v_fmac_f64 (which overwrites its addend) and first copies
// every coefficient into a VGPR pair: two v_mov_b32 per step, 138 of 467 VALU instructions in k_prospect<double>.
__device__ __forceinline__ double spart_horner(double p, double x, double c_uniform) {
#if defined(__HIP_DEVICE_COMPILE__) && SPART_FRESH_COEF
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(x), "s"(c_uniform));
  return d;
#else
  return __builtin_fma(p, x, c_uniform);
#endif
}
// exp / log for the float64 arithmetic (device, SPART_FAST_MATH).
//  * exp_poly: range reduction to |r| <= ln2/2 + Taylor to r^13, coefficients in constant memory (scalar loads -> SGPR
//    operands).  Needs no table: the sample-level prelude (one lane per sample, no workgroup staging) uses it.
//  * exp / exp2 / log of the BAND kernels: table-driven.  e^x = 2^(k >> 8) T[k & 255] e^r with T[j] = 2^(j/256) and
//    |r| <= ln2/512 (Taylor to r^4, remainder 4e-17): 17 instructions, 3 of them 32-bit integer, instead of 22 float64
//    ones; ln x = e ln2 + L[i] + log1p(z) with i = the top 8 mantissa bits, z = m R[i] - 1, |z| <= 1/512 (Taylor to z^6,
//    remainder 8e-18 relative): 15 instructions (5 of them 32-bit) instead of 27 -- and no reciprocal.  Both tables
//    (csrc/spart_f64_tables.h, mpmath) sit in LDS: 6 KB per workgroup, staged by stage_f64_tables() at kernel entry;
//    the lookups are per-lane ds_reads (neighbouring bands mostly hit the same entry: broadcast).  EVERY kernel that
//    evaluates band arithmetic in float64 must call stage_f64_tables() first.
//    exp(x) is 0 below x = -745 (including -inf); log is only called with positive finite normal arguments.  Neither
//    propagates a NaN argument by itself: in leaf_band / soil_band / canopy_core every result of exp / log is multiplied
//    with terms that are NaN when the argument is (test_nan_and_nonphysical_inputs_do_not_crash).  Relative error <= 3e-16 (exp), absolute <= 3e-16 / relative <= 2e-15 away from
//    x = 1 (log).
static __device__ __constant__ double c_EXP_F64[12] = {1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320,
                                                1.0 / 362880, 1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double g_F64_EXP_TAB[F64_EXP_TAB] = SPART_F64_EXP_TABLE;
static __device__ const double g_F64_LOG_TAB[2 * F64_LOG_TAB] = SPART_F64_LOG_TABLE;
__shared__ __attribute__((aligned(16))) double s_f64_exp_tab[F64_EXP_TAB];
__shared__ __attribute__((aligned(16))) double s_f64_log_tab[2 * F64_LOG_TAB];
// all threads of the workgroup; ends with a barrier
__device__ __forceinline__ void stage_f64_tables() {
  for (int i = threadIdx.x; i < F64_EXP_TAB; i += blockDim.x) s_f64_exp_tab[i] = g_F64_EXP_TAB[i];
  for (int i = threadIdx.x; i < 2 * F64_LOG_TAB; i += blockDim.x) s_f64_log_tab[i] = g_F64_LOG_TAB[i];
  __syncthreads();
}
#else
__device__ __forceinline__ void stage_f64_tables() {}     // (host pass of hipcc: declaration only)
#endif
#endif

template <> struct Mx<double> {
  // p x + c with a CONSTANT c (device: SGPR addend of a three-address v_fma_f64, see spart_horner)
  static SPART_HD double horner_c(double p, double x, double c) {
#if defined(__HIPCC__)
    return spart_horner(p, x, c);
#else
    return p * x + c;
#endif
  }
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
  static SPART_HD double exp_poly(double x) {
    // 2^k * e^r with k = rint(x / ln 2), |r| <= ln 2 / 2: Taylor to r^13 (r^14/14! < 4e-18)
    x = (x < -800.0) ? -800.0 : x;                          // e^-800 underflows to 0 below; a NaN stays a NaN
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    spart_cdp ce = spart_fresh(c_EXP_F64);
    double p = ce[11];
#pragma unroll
    for (int i = 10; i >= 0; --i) p = spart_horner(p, r, ce[i]);
    p = __builtin_fma(p * r, r, r);                       // r + r^2 (1/2 + ...)
    return __builtin_ldexp(1.0 + p, (int)k);
  }
  // 2^(k/256) e^r, k integral (as a double), |r| <= ln2/512
  static SPART_HD double exp_finish(double k, double r) {
    const int ki = (int)k;                                 // (NaN -> 0: r is NaN then, and so is the result)
    const double t = s_f64_exp_tab[ki & (F64_EXP_TAB - 1)];
    double p = spart_horner(1.0 / 24.0, r, 1.0 / 6.0);      // (three-address FMA, constant addend from an SGPR pair: no copies)
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r * r, r);                        // e^r - 1
    return __builtin_ldexp(__builtin_fma(t, p, t), ki >> 8);
  }
  static SPART_HD double exp(double x) {
    x = __builtin_fmax(x, -800.0);      // e^-800 underflows to 0 in ldexp.  (One v_max instead of a compare + two selects; a NaN
                                        //  argument gives 0 here -- every caller multiplies the result with NaN terms anyway)
    const double k = __builtin_rint(x * 369.32993046757463);             // 256 / ln 2
    double r = __builtin_fma(-k, 0.0027076061737716373, x);              // ln2/256, high part (33 bits: k * hi is exact)
    r = __builtin_fma(-k, 2.9064910585985925e-13, r);                    // low part
    return exp_finish(k, r);
  }
  // the same for callers that rely on exp(NaN) = NaN (smac_band: a NaN gas column reaches the result only through exp);
  // a compare + select instead of the one v_max
  static SPART_HD double exp_keepnan(double x) {
    x = (x < -800.0) ? -800.0 : x;
    const double k = __builtin_rint(x * 369.32993046757463);
    double r = __builtin_fma(-k, 0.0027076061737716373, x);
    r = __builtin_fma(-k, 2.9064910585985925e-13, r);
    return exp_finish(k, r);
  }
  static SPART_HD double exp2(double x) {                  // 2^x = 2^(k/256) e^((256 x - k) ln2 / 256)
    const double t = x * 256.0;
    const double k = __builtin_rint(t);
    return exp_finish(k, (t - k) * 0.0027076061740622863);               // ln2 / 256
  }
  static SPART_HD double log(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned hi = (unsigned)(u >> 32);
    const int e = (int)(hi >> 20) - 1023;
    const unsigned i = (hi >> 12) & (unsigned)(F64_LOG_TAB - 1);
    const double m = __longlong_as_double((long long)((u & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));   // [1, 2)
    const double R = s_f64_log_tab[2 * i], L = s_f64_log_tab[2 * i + 1];  // one ds_read_b128
    const double z = __builtin_fma(m, R, -1.0);             // (a NaN / inf argument gives a finite value: see the callers)
    double q = spart_horner(-1.0 / 6.0, z, 0.2);
    q = spart_horner(q, z, -0.25);
    q = spart_horner(q, z, 1.0 / 3.0);
    q = __builtin_fma(q, z, -0.5);
    const double p = __builtin_fma(z * z, q, z);           // log1p(z)
    return __builtin_fma((double)e, 0.6931471805599453, L + p);
  }
#else
  static SPART_HD double exp_poly(double x) { return ::exp(x); }
  static SPART_HD double exp(double x) { return ::exp(x); }
  static SPART_HD double exp_keepnan(double x) { return ::exp(x); }
  static SPART_HD double exp2(double x) { return ::exp2(x); }
  static SPART_HD double log(double x) { return ::log(x); }
#endif
  // square root: v_rsq_f64 (4.6e-8 relative) refined by ONE Newton step, sqrt x = g (1 + e/2 + O(e^2)), g = x y0,
  // e = 1 - g y0: relative error 3 e^2 / 8 < 1e-15 in 5 instructions (the library's sequence: ~14).  The argument of the
  // rsq is kept away from 0 by ADDING 1e-300 (x = 0 -> 0 * rsq(1e-300) = 0 instead of 0 * inf; x + 1e-300 == x for every
  // normal x): a negative x <= -1e-300 gives NaN like np.sqrt (a max() here returned finite garbage), a NaN x
  // stays NaN.  Outside that: -1e-300 < x < 0 returns a tiny negative finite value (|result| < 1e-150) instead of NaN, and a
  // positive x below 1e-300 returns x * 1e150 (too small); no quantity of the model comes near either range.
  static SPART_HD double sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
    const double y0 = __builtin_amdgcn_rsq(x + 1e-300);
    const double g = x * y0;
    const double e = __builtin_fma(-g, y0, 1.0);
    return __builtin_fma(g * 0.5, e, g);
#else
    return ::sqrt(x);
#endif
  }
  // reciprocal: v_rcp_f64 (4.6e-8 relative, tools/ubench/rcp64_probe) refined by ONE Newton step r (2 - x r):
  // relative error e^2 = 2e-15 in 3 instructions (the IEEE division sequence: ~12); the host build divides
  static SPART_HD double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SPART_FAST_MATH)
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#else
    return 1.0 / x;
#endif
  }
  // ln(1+x), x >= 0: 2 atanh(x/(2+x)) series below 0.1 (s <= 0.048, s^15/15 < 1e-21), log(1+x) above
  // (whose rounding of 1+x costs at most 1e-16/0.1 relative)
  static SPART_HD double log1p(double x) {
#if defined(SPART_FAST_MATH)
    if (x < 0.1) {
      double s = x * rcp(2.0 + x), s2 = s * s;
      double p = horner_c(1.0 / 13.0, s2, 1.0 / 11.0);
      p = horner_c(p, s2, 1.0 / 9.0);
      p = horner_c(p, s2, 1.0 / 7.0);
      p = horner_c(p, s2, 0.2);
      p = horner_c(p, s2, 1.0 / 3.0);
      p = p * s2 + 1.0;
      return 2.0 * s * p;
    }
    return log(1.0 + x);
#else
    return ::log1p(x);
#endif
  }
  // 1 - e^-z: Taylor for |z| below 0.02 (z^9/9! < 2e-21), direct otherwise (cancellation <= 1e-16/0.02 relative)
  static SPART_HD double omen_taylor(double z) {
    double p = -1.0 / 40320.0;
    p = p * z + 1.0 / 5040.0;
    p = p * z - 1.0 / 720.0;
    p = p * z + 1.0 / 120.0;
    p = p * z - 1.0 / 24.0;
    p = p * z + 1.0 / 6.0;
    p = p * z - 0.5;
    p = p * z + 1.0;
    return z * p;
  }
  static SPART_HD double one_minus_exp_neg(double z) {
#if defined(SPART_FAST_MATH)
    if (::fabs(z) < 0.02) return omen_taylor(z);   // (|z|: a negative z -- N < 1, LAI < 0 -- must not take the series)
    return 1.0 - exp(-z);
#else
    return -::expm1(-z);
#endif
  }
  // (the prelude's form: no table, see exp_poly)
  static SPART_HD double one_minus_exp_neg_poly(double z) {
#if defined(SPART_FAST_MATH)
    return (::fabs(z) < 0.02) ? omen_taylor(z) : 1.0 - exp_poly(-z);
#else
    return -::expm1(-z);
#endif
  }
  static SPART_HD double one_minus_exp_neg(double z, double ez) {
#if defined(SPART_FAST_MATH)
    return (::fabs(z) < 0.02) ? omen_taylor(z) : 1.0 - ez;
#else
    (void)ez;
    return -::expm1(-z);
#endif
  }
  static SPART_HD double fabs(double x) { return ::fabs(x); }
  static SPART_HD double fmax(double a, double b) { return ::fmax(a, b); }
  static SPART_HD double tiny() { return 1e-300; }
};

template <typename T> SPART_HD T divx(T a, T b) { return a * Mx<T>::rcp(b); }

// SAIL J-functions (sailh.py:154-183) for x = -1 / x = 0:
//   J1 = (e^-mL - e^-kL)/(k - m) = L e^-kL phi((m-k)L),   J2 = (1 - e^-kL e^-mL)/(k + m) = L phi((k+m)L),
//   phi(d) = (1 - e^-d)/d  (-> 1 as d -> 0)
// e1 = exp(-m L) is shared between the four of them; below |d| < THRESH the Taylor polynomial of phi replaces
// the difference of nearly equal exponentials (float: 0.06, next term d^5/720 = 1e-9).
template <typename T> struct SailJ;
template <> struct SailJ<float> {
  static constexpr float THRESH = 0.06f;
  static SPART_HD float poly(float d) {
    return 1.0f + d * (-0.5f + d * (0.166666667f + d * (-0.0416666667f + d * 8.33333333e-3f)));
  }
};
template <> struct SailJ<double> {
  // |d| < 2e-3: Taylor polynomial of phi to d^5 (next term d^6/5040 < 2e-20); outside, the difference of
  // exponentials loses at most 1e-16/2e-3 relative
  static constexpr double THRESH = 2e-3;
  static SPART_HD double poly(double d) {
    return 1.0 + d * (-0.5 + d * (1.0 / 6.0 + d * (-1.0 / 24.0 + d * (1.0 / 120.0 - d * (1.0 / 720.0)))));
  }
};
// J1 = L (tk - e1)/d with tk = e^-kL, e1 = e^-mL, d = (m - k) L, id = 1/d (unused on the Taylor side)
template <typename T> SPART_HD T sail_j1_d(T L, T tk, T e1, T d, T id) {
  T v = L * (tk - e1) * id;              // (id is a finite dummy where |d| is small)
  const bool small = Mx<T>::fabs(d) < SailJ<T>::THRESH;
  if (SPART_WAVE_ANY(small)) {           // m within 0.06 / LAI of k or K: rare, issued only when some lane needs it
    SPART_KEEP_BRANCH(d);
    v = small ? L * tk * SailJ<T>::poly(d) : v;
  }
  return v;
}
// J2 = (1 - tk e1)/(k + m), kpm = k + m > 0, ikpm = 1/kpm; (k + m) L < THRESH selects L phi((k + m) L) -- rare, and
// the Taylor side is only issued when some lane of the wave needs it
template <typename T> SPART_HD T sail_j2_d(T L, T tk, T e1, T kpm, T ikpm) {
  T d = kpm * L;
  T v = (T(1) - tk * e1) * ikpm;
  if (SPART_WAVE_ANY(d < SailJ<T>::THRESH)) {
    SPART_KEEP_BRANCH(d);
    v = (d < SailJ<T>::THRESH) ? L * SailJ<T>::poly(d) : v;
  }
  return v;
}

// ------------------------------------------------------------------------------------------
// tau(K) = (1-K) exp(-K) + K^2 E1(K) = 2 E3(K)           (prospect_5d.py:183-196)
// returns tau and u = 1 - tau; K <= 0 -> tau = 1 (prospect_5d.py:195)
template <typename T> struct E3c;
template <> struct E3c<float> {
  static constexpr int GD = E3_G_DEG_F32, WD = E3_W_DEG_F32;
  static constexpr float PSCALE = 2.0f;      // pt() holds P; the factor 2 is folded into the literals by the compiler
  static SPART_HD const float* gt() { return E3_G_F32; }     // (instruction literals after unrolling)
  static SPART_HD const float* pt() { return E3_P_F32; }
  static SPART_HD const float* qt() { return E3_Q_F32; }
};
#if defined(__HIPCC__)
// The 27 float64 coefficients of the plate transmittance live in constant memory on the device: they arrive through
// scalar loads and enter the FMAs as SGPR operands.  As literals they were parked in 54 VGPRs for the whole sample
// loop and copied before every Horner step (v_fmac_f64 overwrites its addend); without them the float64 band kernel
// fits three waves per SIMD with 38 spilled values instead of two with 7 (38.4 -> 36.2 ms per 1M spectra).  Doing
// the same to the smaller polynomials (log1p, 1 - e^-z, phi) measured slower: their scalar loads sit in rarely
// taken branches.  (float32 coefficients are instruction literals and cost nothing.)
static_assert(E3_G_DEG_F64 == 12 && E3_W_DEG_F64 == 6, "tables below list the coefficients one by one");
#define SPART_L13(a) {a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]}
#define SPART_L7(a) {a[0], a[1], a[2], a[3], a[4], a[5], a[6]}
static __device__ __constant__ double c_E3_G_F64[13] = SPART_L13(E3_G_F64);
#define SPART_L7X2(a) {2 * a[0], 2 * a[1], 2 * a[2], 2 * a[3], 2 * a[4], 2 * a[5], 2 * a[6]}
static __device__ __constant__ double c_E3_P_F64[7] = SPART_L7X2(E3_P_F64);   // 2 P: the factor 2 of tau = 2 E3 folded in (exact)
static __device__ __constant__ double c_E3_Q_F64[7] = SPART_L7(E3_Q_F64);
#endif
template <> struct E3c<double> {
  static constexpr int GD = E3_G_DEG_F64, WD = E3_W_DEG_F64;
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr double PSCALE = 1.0;      // the device table already holds 2 P (seven multiplications per band saved)
#else
  static constexpr double PSCALE = 2.0;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
  static SPART_HD spart_cdp gt() { return spart_fresh(c_E3_G_F64); }
  static SPART_HD spart_cdp pt() { return spart_fresh(c_E3_P_F64); }
  static SPART_HD spart_cdp qt() { return spart_fresh(c_E3_Q_F64); }
#else
  static SPART_HD const double* gt() { return E3_G_F64; }
  static SPART_HD const double* pt() { return E3_P_F64; }
  static SPART_HD const double* qt() { return E3_Q_F64; }
#endif
};

#if defined(__HIPCC__)
// horner_uniform(p, x, c) = p x + c where c MUST be wave-uniform (a constant): on the device the float64 overload passes it
// through an SGPR operand (spart_horner) -- a lane-varying c would silently become lane 0's value.
SPART_HD double horner_uniform(double p, double x, double c) { return spart_horner(p, x, c); }
#else
SPART_HD double horner_uniform(double p, double x, double c) { return p * x + c; }
#endif
SPART_HD float horner_uniform(float p, float x, float c) { return p * x + c; }

template <typename T> SPART_HD void plate_tau(T K, T& tau, T& u) {
  using C = E3c<T>;
  // K <= 0 (only Kall > 0 entries are replaced in the reference, prospect_5d.py:182,195) is clamped to a
  // tiny positive value: tau -> 1, u -> 0+, and the Stokes terms below reach the reference's
  // zero-absorption limit (prospect_5d.py:233-235) continuously.  NaN propagates.
  T x = (K < Mx<T>::tiny()) ? Mx<T>::tiny() : K;   // (a select, not fmax: a NaN K must stay NaN)
  const bool small = x < T(1);
  if (sizeof(T) == 8) {   // float64: tau and u formed in each branch (a select of a double is two instructions)
    T tl, ul;             // (locals, not the reference arguments: those made the compiler route the values through scratch)
    if (small) {
      const auto cg = C::gt();
      T g = cg[C::GD];
#pragma unroll
      for (int i = C::GD - 1; i >= 0; --i) g = horner_uniform(g, x, cg[i]);
      ul = x * (g + x * Mx<T>::log(x));
      tl = T(1) - ul;
    } else {
      const auto cp = C::pt(), cq = C::qt();
      T pn = C::PSCALE * cp[0], qn = cq[0];
#pragma unroll
      for (int i = 1; i <= C::WD; ++i) {
        pn = horner_uniform(pn, x, C::PSCALE * cp[i]);
        qn = horner_uniform(qn, x, cq[i]);
      }
      tl = Mx<T>::exp(-x) * pn * Mx<T>::rcp((x + T(3)) * qn);
      ul = T(1) - tl;
    }
    tau = tl;
    u = ul;
    return;
  }
  T v;  // u on the small branch, tau on the large one (one value, so nothing is spilled to select them)
  if (small) {
    const auto cg = C::gt();
    T g = cg[C::GD];
#pragma unroll
    for (int i = C::GD - 1; i >= 0; --i) g = horner_uniform(g, x, cg[i]);
    v = x * (g + x * Mx<T>::log(x));
  } else {
    // P(t)/Q(t) with t = 1/x, written in x (coefficients reversed) so that a single reciprocal is needed:
    // tau = e^-x * 2 Pr(x) / ((x + 3) Qr(x)),  Pr(x) = x^n P(1/x)
    const auto cp = C::pt(), cq = C::qt();
    T pn = C::PSCALE * cp[0], qn = cq[0];  // (the factor 2 is folded into P's constants: exact)
#pragma unroll
    for (int i = 1; i <= C::WD; ++i) {
      pn = horner_uniform(pn, x, C::PSCALE * cp[i]);
      qn = horner_uniform(qn, x, cq[i]);
    }
    v = Mx<T>::exp(-x) * pn * Mx<T>::rcp((x + T(3)) * qn);
  }
  T w = T(1) - v;
  u = small ? v : w;
  tau = small ? w : v;
}

// ------------------------------------------------------------------------------------------
// calculate_tav (prospect_5d.py:249-311), double only: used once per context for the
// table-only interface transmissivities
inline double calculate_tav(double alpha, double nr) {
  const double rd = PI / 180.0;
  double n2 = nr * nr, n_p = n2 + 1, nm = n2 - 1;
  double a = (nr + 1) * (nr + 1) / 2;
  double k = -(n2 - 1) * (n2 - 1) / 4;
  double sa = std::sin(alpha * rd);
  double b1 = 0;
  if (alpha != 90) b1 = std::sqrt((sa * sa - n_p / 2) * (sa * sa - n_p / 2) + k);
  double b2 = sa * sa - n_p / 2;
  double b = b1 - b2;
  double b3 = b * b * b, a3 = a * a * a;
  double ts = (k * k / (6 * b3) + k / b - b / 2) - (k * k / (6 * a3) + k / a - a / 2);
  double tp1 = -2 * n2 * (b - a) / (n_p * n_p);
  double tp2 = -2 * n2 * n_p * std::log(b / a) / (nm * nm);
  double tp3 = n2 * (1 / b - 1 / a) / 2;
  double tp4 = 16 * n2 * n2 * (n2 * n2 + 1) * std::log((2 * n_p * b - nm * nm) / (2 * n_p * a - nm * nm)) /
               (n_p * n_p * n_p * nm * nm);
  double tp5 = 16 * n2 * n2 * n2 * (1 / (2 * n_p * b - nm * nm) - 1 / (2 * n_p * a - nm * nm)) / (n_p * n_p * n_p);
  double tp = tp1 + tp2 + tp3 + tp4 + tp5;
  return (ts + tp) / (2 * sa * sa);
}

// ------------------------------------------------------------------------------------------
// per-band table slice (lives in registers for the whole sample loop)
// row order of the device table block ctx->tab[NTAB][NWL]
enum TabRow {
  TAB_KAB = 0, TAB_KCA, TAB_KDM, TAB_KW, TAB_KS, TAB_KANT, TAB_CBC, TAB_PROT,  // prospect_5d.py:158-167
  TAB_TALF, TAB_T12, TAB_T21,   // tav(40,nr), tav(90,nr), tav(90,nr)/nr^2   (prospect_5d.py:200-205)
  TAB_GSV0, TAB_GSV1, TAB_GSV2,  // bsm.py:45
  TAB_CBAC, TAB_PW, TAB_RW,      // tav(90,2/nw)/tav(90,2), 1-tav(90,nw)/nw^2, 1-tav(40,nw) (bsm.py:110-119)
  NTAB
};

template <typename T> struct BandTab {
  T kab, kca, kdm, kw, ks, kant, kcbc, kprot, talf, t12, t21, g0, g1, g2, cbac, pw, rw;
};

// ------------------------------------------------------------------------------------------
// per-sample constants consumed by the band kernels (written by the prelude kernel,
// read through wave-uniform scalar loads).  Layout = array of NCONST values of T.
enum ConstIdx {
  // leaf: concentrations already divided by N (prospect_5d.py:170-179)
  C_CAB = 0, C_CCA, C_CDM, C_CW, C_CS, C_CANT, C_CBC, C_PROT, C_NM1, C_RHO_TH, C_TAU_TH,
  // soil (bsm.py:49-52, 101, 121-122): GSV factors, wet flag, Poisson weights and their sum, 2 film log2(e)
  C_F1, C_F2, C_F3, C_WET, C_FM0, C_FM1, C_FM2, C_FM3, C_FM4, C_FM5, C_FM6, C_FMSUM, C_FILM2L,
  // canopy (sailh.py:93-97, 200-203, 216, 219); the six geometric factors of sailh.py:100-105 are (k +- bf)/2,
  // (1 +- bf)/2, (K +- bf)/2 and are formed from ks, ko, bf/2 where they are used (canopy_core)
  C_SOB, C_SOF, C_HBF, C_KS, C_KO, C_LAI, C_LAI2, C_TSS, C_TOO, C_Z, C_HOT, C_PSO2W,
  C_RSV0, C_RSV1, C_RSV2, C_RSV3,
  NCONST  // 40: 36 used, padded to a multiple of 8 (stage_constants)
};
constexpr int NCONST_USED = C_RSV0;   // rows the prelude writes
static_assert(NCONST == 40, "constant block is 40 values");

// per-sample atmosphere scalars (double), read by the sensor-band kernel
enum AtmIdx {
  A_US = 0, A_UV, A_M, A_PEQ, A_PA, A_AOT, A_CKSI, A_KSID, A_LAF,
  A_LOGPEQ, A_LOGM, A_LOGO3M, A_LOGH2OM,   // ln Peq, ln m, ln(uo3 m), ln(uh2o m): x^n is evaluated as exp(n ln x)
  A_RSV,                                   // (uo3 and uh2o themselves are not kept: smac_band reads them through the two logarithms only)
  NATM = 16
};
constexpr int NATM_USED = A_RSV;      // rows the prelude writes

// ------------------------------------------------------------------------------------------
// PROSPECT-5D / PRO, one band                                        (prospect_5d.py:170-241)
// cN[] = concentrations / N.  Returns refl, tran, absorptance = 1 - refl - tran, K = Kall.
template <typename T>
SPART_HD void leaf_band(const BandTab<T>& tb, T cab, T cca, T cdm, T cw, T cs, T cant, T cbc, T prot, T nm1,
                        T& refl, T& tran, T& absb, T& Kall) {
  T K = cab * tb.kab + cca * tb.kca + cdm * tb.kdm + cw * tb.kw + cs * tb.ks + cant * tb.kant + cbc * tb.kcbc +
        prot * tb.kprot;  // :170-179
  Kall = K;
  T tau, u;
  plate_tau(K, tau, u);  // :182-196
  T r21 = T(1) - tb.t21, r12 = T(1) - tb.t12, ralf = T(1) - tb.talf;  // :201-205
  T x = r21 * tau;
  T inv = Mx<T>::rcp(T(1) - x * x);  // :208
  T c = tau * tb.t21 * inv;
  T Ta = tb.talf * c;   // :209
  T Ra = ralf + x * Ta;  // :210
  T t = tb.t12 * c;      // :213
  T r = r12 + x * t;     // :214
  // 1 - r - t = t12 (1-tau)/(1 - r21 tau);  1 - Ra - Ta = talf (1-tau)/(1 - r21 tau);  1/(1-x) = (1+x)/(1-x^2)
  T gq = u * (T(1) + x) * inv;
  T a1 = tb.t12 * gq;
  T atop = tb.talf * gq;
  // Stokes system for the N-1 lower layers (:219-230), written in a-1, b-1 and b^-(N-1)
  T tt = Mx<T>::fmax(t, Mx<T>::tiny());
  T D = Mx<T>::sqrt((T(1) + r + tt) * (T(1) + r - tt) * (T(1) - r + tt) * a1);  // :219
  T i2rt = Mx<T>::rcp(T(2) * r * tt);                          // 1/(2r) = tt i2rt, 1/(2tt) = r i2rt
  T am1 = (a1 * (T(1) - r + tt) + D) * (tt * i2rt);            // a - 1, a from :222
  T bm1 = (a1 * (T(1) - tt + r) + D) * (r * i2rt);             // b - 1, b from :223
  T a = T(1) + am1;
  T z = nm1 * Mx<T>::log1p(bm1);  // (N-1) ln b
  T sq = Mx<T>::exp(-z);          // b^-(N-1)
  T omsq = Mx<T>::one_minus_exp_neg(z, sq);  // 1 - b^-(N-1)
  T omq = omsq * (T(1) + sq);     // 1 - b^-2(N-1)
  T A2 = am1 * (a + T(1));        // a^2 - 1
  // Rsub = a omq / den, Tsub = sq A2 / den, den = A2 + omq (:229-230 divided by b^2(N-1)); combined with the top
  // layer (:239-241) everything shares the ONE denominator E = den (1 - Rsub r) = den - a omq r:
  T aomq = a * omq;
  T iE = Mx<T>::rcp(A2 + omq - aomq * r);
  // zero absorption (r + t >= 1, :233-235) is the a1 -> 0+ limit of these expressions:
  // Tsub -> t/(t + (1-t)(N-1)), Rsub -> 1 - Tsub; plate_tau keeps a1 > 0 so no branch is needed.
  T TaE = Ta * iE;
  tran = TaE * sq * A2;                                   // Ta Tsub / (1 - Rsub r)          (:240)
  refl = Ra + TaE * aomq * t;                             // Ra + Ta Rsub t / (1 - Rsub r)   (:241)
  absb = atop + TaE * (am1 * omsq * (a - sq) + aomq * a1);  // 1 - refl - tran, every term >= 0
}

// ------------------------------------------------------------------------------------------
// BSM + soilwat, one band                                        (bsm.py:49-52, 99-124)
// the water film's transmittance for one layer (:122 with k = 1; film2l = 2 film log2(e)): a function of the band and of
// the film thickness only, so a kernel may evaluate it once for a run of samples that share the film thickness
template <typename T> SPART_HD T soil_tw1(const BandTab<T>& tb, T film2l) { return Mx<T>::exp2(-film2l * tb.kw); }

template <typename T>
SPART_HD void soil_band_tw(const BandTab<T>& tb, T rdry, T wet, const T fm[7], T fmsum16, T tw1, T& rwet) {
  T rbac = T(1) - (T(1) - rdry) * (rdry * tb.cbac + T(1) - rdry);  // :110-112
  // rwet = rdry f0 + sum_k f_k [Rw + (1-Rw)(1-p) x_k/(1 - p x_k)],  x_k = tw1^k rbac   (:123-124)
  // The six reciprocals 1/d_k, d_k = 1 - p x_k, come from ONE reciprocal of their product (prefix products
  // forward, peel-off backward): a v_rcp costs about five plain VALU ops in this instruction mix
  // (profiles/r1_ubench_valu_issue.txt), the 15 extra multiplies three.
  T x[6], d[6], pre[6];
  T xv = rbac;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    xv *= tw1;
    x[k] = xv;
    d[k] = T(1) - tb.pw * xv;
    pre[k] = (k == 0) ? d[0] : pre[k - 1] * d[k];
  }
  T q = Mx<T>::rcp(pre[5]);
  T acc = T(0);
#pragma unroll
  for (int k = 5; k >= 1; --k) {
    acc += fm[k + 1] * x[k] * (q * pre[k - 1]);    // 1/d_k = (prod_{j<k} d_j) / (prod_{j<=k} d_j)
    q *= d[k];
  }
  acc += fm[1] * x[0] * q;
  T v = rdry * fm[0] + tb.rw * fmsum16 + (T(1) - tb.rw) * (T(1) - tb.pw) * acc;
  rwet = (wet > T(0)) ? v : rdry;                                  // :102-103
}
template <typename T>
SPART_HD void soil_band(const BandTab<T>& tb, T rdry, T wet, const T fm[7], T fmsum16, T film2l, T& rwet) {
  soil_band_tw<T>(tb, rdry, wet, fm, fmsum16, soil_tw1<T>(tb, film2l), rwet);
}

template <typename T> SPART_HD T soil_dry(const BandTab<T>& tb, T f1, T f2, T f3) {
  return f1 * tb.g0 + f2 * tb.g1 + f3 * tb.g2;                     // bsm.py:52
}

// ------------------------------------------------------------------------------------------
// SAILH, one band                                                  (sailh.py:142-233)
// (the six geometric factors sdb..dof of sailh.py:100-105 are (k +- bf)/2, (1 +- bf)/2, (K +- bf)/2; canopy_band
// works from ks, ko, bf directly -- the prelude still writes them to the constant block for inspection)
template <typename T> struct CanopyPar {
  T sob, sof, hbf, ks, ko, lai, lai2, tss, too, Z, hot, pso2w;   // hbf = bf/2, lai2 = LAI log2(e)
};

// The solve in two parts: everything that depends on the leaf only (canopy_core), and the coupling with the soil
// background (canopy_soil, :222-233).  canopy_band = both.
template <typename T> struct CanopyCore {
  T rho_so, rho_dd, tau_dd, tau_sd, tau_do, rho_sd, rho_do;
};

template <typename T>
SPART_HD CanopyCore<T> canopy_core(const CanopyPar<T>& c, T rho, T tau, T absb) {
  // scattering coefficients (:142-148).  With sdb/sdf = (k +- bf)/2, ddb/ddf = (1 +- bf)/2, dob/dof = (K +- bf)/2
  // (:100-105) they are P, k P, K P plus/minus Mn, where P = (rho + tau)/2 and Mn = bf (rho - tau)/2:
  //   sigb = P + Mn, sigf = P - Mn, sb/sf = k P +- Mn, vb/vf = K P +- Mn
  T P = T(0.5) * (rho + tau);
  T Mn = c.hbf * (rho - tau);
  T sigb = P + Mn;                     // diffuse backscatter
  T w = c.sob * rho + c.sof * tau;     // bidirectional
  T a = T(1) - P + Mn;                 // 1 - sigf  (:149)
  // m^2 = a^2 - sigb^2 = (a - sigb)(a + sigb), a - sigb = 1 - rho - tau, a + sigb = 1 + 2 Mn   (:150)
  T m = Mx<T>::sqrt(absb * (T(1) + T(2) * Mn));
  // three reciprocals from one: 1/(a+m), 1/(ks+m), 1/(ko+m)
  T apm = a + m, ksm = c.ks + m, kom = c.ko + m;
  T kk = ksm * kom;
  T i3 = Mx<T>::rcp(apm * kk);
  T iam = i3 * kk;
  T ikk = i3 * apm;
  T iks = ikk * kom, iko = ikk * ksm;
  T rinf = sigb * iam;                 // == (a - m)/sigb   (:151)
  T rinf2 = rinf * rinf;
  // 1 - rinf = (a - sigb + m)/(a + m) = (absorptance + m)/(a + m): no cancellation for nearly
  // conservative leaves (rinf -> 1)
  T omr = (absb + m) * iam;            // 1 - rinf
  T opr = T(1) + rinf;
  T omr2 = omr * opr;                  // 1 - rinf^2
  T L = c.lai;
  // J1(-1) = (e^-mL - e^-kL)/(k - m), J2(0) = (1 - e^-kL e^-mL)/(k + m)   (:154-183)
  T e1 = Mx<T>::exp2(-m * c.lai2);     // e^-mL, :185-189
  const T thr = SailJ<T>::THRESH;
  T d1 = (m - c.ks) * L, d2 = (m - c.ko) * L;
  T d1s = (Mx<T>::fabs(d1) < thr) ? T(1) : d1, d2s = (Mx<T>::fabs(d2) < thr) ? T(1) : d2;   // (the Taylor side needs no 1/d)
  T idd = Mx<T>::rcp(d1s * d2s);
  T J1k = sail_j1_d<T>(L, c.tss, e1, d1, idd * d2s);
  T J1K = sail_j1_d<T>(L, c.too, e1, d2, idd * d1s);
  T J2k = sail_j2_d<T>(L, c.tss, e1, ksm, iks);
  T J2K = sail_j2_d<T>(L, c.too, e1, kom, iko);
  T ome2 = T(1) - e1 * e1;
  T re = rinf * e1;
  T i1 = Mx<T>::rcp(omr2 * (T(1) + rinf2));  // sic: 1/(1 - rinf2**2) (:189)
  T i2 = (T(1) + rinf2) * i1;                 // 1/(1 - rinf2)  (:214)
  // s1 = sf + rinf sb, s2 = sf rinf + sb, v1 = vf + rinf vb, v2 = vf rinf + vb (:191-198) in terms of
  // U = P (1 + rinf), V = Mn (1 - rinf):
  T U = P * opr, V = Mn * omr;
  T s1 = c.ks * U - V;
  T s2 = c.ks * U + V;
  T v1 = c.ko * U - V;
  T v2 = c.ko * U + V;
  T Pss = s1 * J1k, Qss = s2 * J2k, Poo = v1 * J1K, Qoo = v2 * J2K;
  T tau_dd = omr2 * e1 * i1;             // :205-210
  T rho_dd = rinf * ome2 * i1;
  T tau_sd = (Pss - re * Qss) * i1;
  T tau_do = (Poo - re * Qoo) * i1;
  T rho_sd = (Qss - re * Pss) * i1;
  T rho_do = (Qoo - re * Poo) * i1;
  T T1 = v2 * s1 * (c.Z - J1k * c.too) * iko + v1 * s2 * (c.Z - J1K * c.tss) * iks;  // :212
  T T2 = -(Qoo * rho_sd + Poo * tau_sd) * rinf;                                      // :213
  T rho_sod = (T1 + T2) * i2;                                                        // :214
  T rho_so = rho_sod + w * c.hot;                                 // :216-217
  return CanopyCore<T>{rho_so, rho_dd, tau_dd, tau_sd, tau_do, rho_sd, rho_do};
}

template <typename T>
SPART_HD void canopy_soil(const CanopyPar<T>& c, const CanopyCore<T>& k, T rs, T& rso, T& rdo, T& rsd, T& rdd) {
  const T rho_so = k.rho_so, rho_dd = k.rho_dd, tau_dd = k.tau_dd, tau_sd = k.tau_sd, tau_do = k.tau_do, rho_sd = k.rho_sd,
          rho_do = k.rho_do;
  T g = rs * Mx<T>::rcp(T(1) - rs * rho_dd);                      // rs / (1 - rs rho_dd)   (:222)
  T h = g * tau_dd;
  T tst = c.tss + tau_sd;
  rso = rho_so + rs * c.pso2w + ((tau_sd + c.tss * rs * rho_dd) * c.too + tst * tau_do) * g;  // :224-230
  rdo = rho_do + (c.too + tau_do) * h;                            // :231
  rsd = rho_sd + tst * h;                                         // :232
  rdd = rho_dd + tau_dd * h;                                      // :233
}

template <typename T>
SPART_HD void canopy_band(const CanopyPar<T>& c, T rho, T tau, T absb, T rs, T& rso, T& rdo, T& rsd, T& rdd) {
  const CanopyCore<T> k = canopy_core<T>(c, rho, tau, absb);
  canopy_soil<T>(c, k, rs, rso, rdo, rsd, rdd);
}

// ------------------------------------------------------------------------------------------
// sample-level (band independent) arithmetic: always double
// ------------------------------------------------------------------------------------------

// calculate_leafangles.dcum (sailh.py:368-384), literal fixed-point iteration with the reference's
// stopping rule; y = a sin x + b/2 sin 2x is evaluated as sin x (a + b cos x) (one sincos per step).
SPART_HD double lidf_dcum(double a, double b, double theta_deg) {
  const double rd = PI / 180.0;
  if (a > 1.0) return 1.0 - ::cos(theta_deg * rd);  // :371-372
  double x = 2.0 * rd * theta_deg;
  const double theta2 = x;
  double y = 0.0, dx;
  int it = 0;
  do {  // :378-382; the cap only guards against non-convergence for non-physical |a|+|b| >> 1
    double sn, cs;
    ::sincos(x, &sn, &cs);
    y = sn * (a + b * cs);
    dx = 0.5 * (y - x + theta2);
    x += dx;
  } while (::fabs(dx) > 1e-8 && ++it < 100000);
  return (2.0 * y + theta2) / PI;
}

// The same cumulative distribution from the exact root of x = theta2 + a sin x + b/2 sin 2x (Newton,
// quadratic convergence).  The reference stops its linear iteration at |dx| <= 1e-8, i.e. its F carries
// an error of up to ~5e-8; the root differs from it by that much.  Used by the float32 path only
// (tolerance 1e-4); |a| + |b| >= 0.95 (derivative may vanish) falls back to the literal iteration.
SPART_HD double lidf_theta(int i);
SPART_HD double lidf_sin_2theta(int i);
SPART_HD double lidf_cos_2theta(int i);
SPART_HD void lidf_sincos(double u, double s2, double c2, double& sn, double& cs);
SPART_HD double lidf_dcum_newton(double a, double b, int i) {
  const double theta_deg = lidf_theta(i);
  if (!(::fabs(a) + ::fabs(b) < 0.95)) return lidf_dcum(a, b, theta_deg);
  const double rd = PI / 180.0;
  const double theta2 = 2.0 * rd * theta_deg;
  // Unknown y = x - theta2 (|y| <= |a| + |b|/2 < 1): sin x, cos x come from the angle-addition formulas with
  // (sin, cos)(theta2) -- tabulated constants --
  // and the Taylor polynomials of sin y, cos y (to y^19 / y^18: 3e-17 at |y| = 1), instead of a library sincos of
  // x in every Newton step.  Start: one fixed-point step y0 = sin theta2 (a + b cos theta2) (error <~ 0.05), then
  // 0.05 -> 1e-3 -> 1e-6 -> 1e-12; the loop leaves as soon as a step is below 1e-9 (the next iterate is then exact
  // to ~1e-17, far inside the ~1e-8 of the reference's own stopping rule).
  const double s2 = lidf_sin_2theta(i), c2 = lidf_cos_2theta(i);
  double y = s2 * (a + b * c2);
  for (int it = 0; it < 12; ++it) {
    double sn, cs;
    lidf_sincos(y, s2, c2, sn, cs);                                // sin x, cos x
    const double f = y - sn * (a + b * cs);
    const double fp = 1.0 - a * cs - b * (2.0 * cs * cs - 1.0);
    const double d = f / fp;
    y -= d;
    if (::fabs(d) < 1e-9) break;
  }
  return (2.0 * y + theta2) / PI;
}

// F(theta_i) nodes: 10..80 step 10, 82..88 step 2, then F = 1   (sailh.py:388-394)
SPART_HD double lidf_theta(int i) { return (i < 8) ? 10.0 * (i + 1) : 80.0 + 2.0 * (i - 7); }
// litab: 5,15,...,75,81,83,...,89                                  (sailh.py:49)
SPART_HD double lidf_litab(int i) { return (i < 8) ? 5.0 + 10.0 * i : 81.0 + 2.0 * (i - 8); }

// sin x, cos x for x = 2 theta + u from the angle-addition formulas with the tabulated (sin, cos)(2 theta) = (s2, c2)
// and the Taylor polynomials of sin u, cos u (to u^19 / u^18; |u| <= 1: 1/21! = 2e-20): ~25 multiply-adds instead of a
// library sincos with its argument reduction
SPART_HD void lidf_sincos(double u, double s2, double c2, double& sn, double& cs) {
  const double u2 = u * u;
  double ps = -1.0 / 121645100408832000.0;            // sin u / u:  ... - u^18/19!
  ps = ps * u2 + 1.0 / 355687428096000.0;
  ps = ps * u2 - 1.0 / 1307674368000.0;
  ps = ps * u2 + 1.0 / 6227020800.0;
  ps = ps * u2 - 1.0 / 39916800.0;
  ps = ps * u2 + 1.0 / 362880.0;
  ps = ps * u2 - 1.0 / 5040.0;
  ps = ps * u2 + 1.0 / 120.0;
  ps = ps * u2 - 1.0 / 6.0;
  ps = ps * u2 + 1.0;
  double pc = 1.0 / 6402373705728000.0;                // (1 - cos u) / u^2:  1/2 - u^2/24 + ... + u^16/18!
  pc = pc * u2 - 1.0 / 20922789888000.0;
  pc = pc * u2 + 1.0 / 87178291200.0;
  pc = pc * u2 - 1.0 / 479001600.0;
  pc = pc * u2 + 1.0 / 3628800.0;
  pc = pc * u2 - 1.0 / 40320.0;
  pc = pc * u2 + 1.0 / 720.0;
  pc = pc * u2 - 1.0 / 24.0;
  pc = pc * u2 + 0.5;
  const double su = u * ps, cu = 1.0 - u2 * pc;
  sn = s2 * cu + c2 * su;
  cs = c2 * cu - s2 * su;
}
// (sn, cs) = (sin, cos)(x) -> (sin, cos)(x + d) for a step |d| <= 1/4: the angle-addition formulas with the Taylor
// polynomials of sin d (to d^11) and 1 - cos d (to d^12) -- truncation 1e-17 relative / 4e-20 --, 19 multiply-adds instead of the
// 26 of lidf_sincos.  The fixed-point iteration moves by ever smaller steps (in the usual parameter ranges every step after the
// second is below 1/4), so nearly every pass -- and every Newton step of the jump -- can take this form; each use adds ~1e-16 of
// rounding to (sn, cs), so the callers re-anchor with lidf_sincos every eighth pass.
SPART_HD void lidf_rotate(double d, double& sn, double& cs) {
  const double d2 = d * d;
  double ps = -1.0 / 39916800.0;
  ps = ps * d2 + 1.0 / 362880.0;
  ps = ps * d2 - 1.0 / 5040.0;
  ps = ps * d2 + 1.0 / 120.0;
  ps = ps * d2 - 1.0 / 6.0;
  const double sd = d + d * (d2 * ps);                 // sin d
  double pc = -1.0 / 479001600.0;
  pc = pc * d2 + 1.0 / 3628800.0;
  pc = pc * d2 - 1.0 / 40320.0;
  pc = pc * d2 + 1.0 / 720.0;
  pc = pc * d2 - 1.0 / 24.0;
  pc = pc * d2 + 0.5;
  const double cm = d2 * pc;                           // 1 - cos d
  const double s0 = sn, c0 = cs;
  sn = s0 + (c0 * sd - s0 * cm);
  cs = c0 - (s0 * sd + c0 * cm);
}
#ifndef SPART_LIDF_ROTATE
#define SPART_LIDF_ROTATE 1
#endif
constexpr double LIDF_ROT_MAX = 0.25;
// (sn, cs) at the iterate u_new = u_old + d: by rotation when `rot` (this lane's own decision), by lidf_sincos otherwise.
// Each form is issued only when some lane of the wave takes it; a lane's result never depends on its neighbours' choices.
SPART_HD void lidf_sincos_step(bool rot, double d, double u_new, double s2, double c2, double& sn, double& cs) {
  double sr = sn, cr = cs;
  if (SPART_WAVE_ANY(rot)) lidf_rotate(d, sr, cr);
  if (SPART_WAVE_ANY(!rot)) {
    double sf, cf;
    lidf_sincos(u_new, s2, c2, sf, cf);
    sr = rot ? sr : sf;
    cr = rot ? cr : cf;
  }
  sn = sr;
  cs = cr;
}

// The literal iteration (the reference's stopping rule and iterates, sailh.py:378-382) in the small unknown
// u = x - 2 theta:   x <- x + 1/2 (y - x + 2 theta)   ==   u <- u + g(u),  g(u) = 1/2 (y(u) - u),
// y(u) = sin x (a + b cos x).  The iterates differ from the x-form's by rounding (1e-16) only.
// |a| + |b|/2 > 1 (non-physical) and a > 1 keep the library form (lidf_dcum).
//
// JUMP: the iteration converges linearly, e_{n+1} = phi(e_n) = r e_n + c2 e_n^2 + ..., e = u - u*, with
// r = (1 + y'(u*))/2 up to 0.9 in the usual parameter ranges, i.e. 30 ... 170 steps down to |g| <= 1e-8 -- the
// largest single cost of the float64 prelude.  Once an iterate is close to the fixed point the rest of the
// trajectory is known in closed form: with the Koenigs function h(e) = e + a2 e^2 + a3 e^3 + a4 e^4,
// h(phi(e)) = r h(e), every later iterate is e_{m+k} = h^-1(r^k h(e_m)).  So: iterate literally until the expansion
// parameter c2max |e| / (r (1 - r)) is below 5e-3 (truncation below 1e-9 relative), find u* by Newton, evaluate
// r, c2, c3, c4 from the derivatives of y there, find the FIRST k with |g(u_{m+k})| <= 1e-8 -- the reference's
// stopping index -- and return the reference's F = (2 y(u_{m+k}) + 2 theta)/pi at that iterate.  The result differs
// from the literally iterated one by ~1e-15 (the CPU-side arithmetic tests compare the two over the reference's whole LIDF
// grid); lanes with r outside [0.3, 0.98], or whose Newton step does not settle, simply finish the literal loop.
// The jump itself costs about nine literal steps (three Newton steps, five reciprocals, r^k by binary descent over
// r^(2^j) -- no logarithm, no exponential).
template <bool JUMP> SPART_HD double lidf_dcum_lit_impl(double a, double b, int i, int* jumped = nullptr) {
  if (!(::fabs(a) + 0.5 * ::fabs(b) <= 1.0)) return lidf_dcum(a, b, lidf_theta(i));
  const double rd = PI / 180.0;
  const double theta2 = 2.0 * rd * lidf_theta(i);
  const double s2 = lidf_sin_2theta(i), c2 = lidf_cos_2theta(i);
#ifndef SPART_LIDF_KJUMP
#define SPART_LIDF_KJUMP 2e-2   // (5e-3 until round 5: same iterates -- 0 mismatches in 5.4M solves against the literal loop -- two passes fewer)
#endif
  const double kjump = SPART_LIDF_KJUMP / (0.25 * (::fabs(a) + 2.0 * ::fabs(b)) + 0.02);   // 5e-3 / (bound of |y''| / 4, + margin)
  double u = 0.0, y, dx = 0.0, dprev;
  double sn = s2, cs = c2;                 // (sin, cos)(2 theta + u), carried from pass to pass
  bool more, ready = false;
  int it = 0;
  // literal phase: a lane leaves when it has converged (the reference's test) or is close enough to jump; the lanes of
  // a wave reconverge behind the loop, so the jump below runs once per wave
  do {
    dprev = dx;
    u += dx;                               // (the step found in the previous pass; 0 in the first)
    // sin / cos at the new iterate: by rotating the previous pass's pair through the step when THIS lane's step is small, from
    // scratch otherwise and every eighth pass.  The choice is the lane's own (a sample's result must not depend on the samples
    // it shares a wave with); the wave only skips a form that none of its lanes takes.
    // (pass 0 has u = 0 and dx = 0: the rotation through 0 returns the tabulated pair unchanged)
    lidf_sincos_step(SPART_LIDF_ROTATE && (it & 7) != 7 && ::fabs(dx) < LIDF_ROT_MAX, dx, u, s2, c2, sn, cs);
    y = sn * (a + b * cs);
    dx = 0.5 * (y - u);
    more = ::fabs(dx) > 1e-8;              // sailh.py:382 -- y belongs to the iterate BEFORE the update
#ifndef SPART_LIDF_GATE
#define SPART_LIDF_GATE 1e-2    // (4e-3 until round 5)
#endif
    if (JUMP && ::fabs(dx) < SPART_LIDF_GATE) {
      // rho = dx / dprev estimates r; ready when 0.3 < rho < 0.98 and |dx| < 5e-3 rho (1 - rho)^2 / c2max, written
      // without the division: |dprev|^3 < K (|dprev| - |dx|)^2
      const double ad = ::fabs(dx), ap = ::fabs(dprev), df = ap - ad;
      ready = dx * dprev > 0.0 && ad > 0.3 * ap && ad < 0.98 * ap && ap * ap * ap < kjump * (df * df);
    }
  } while (more && !ready && ++it < 100000);
  if (JUMP && more && ready) {
    // state: iterate u_m = u with g(u_m) = dx (not yet applied), previous step dprev
    using Md = Mx<double>;
    const double d0 = dx * dprev * Md::rcp(dprev - dx);
    double us = u + d0;                    // u + dx / (1 - rho): geometric extrapolation of the fixed point
    double nstep = 1.0;
    // (sn, cs) belong to the iterate u: the first Newton point is a small step away (|d0| <~ 0.06 by the readiness test), every
    // later one a tiny one
    lidf_sincos_step(SPART_LIDF_ROTATE && ::fabs(d0) < LIDF_ROT_MAX, d0, us, s2, c2, sn, cs);
    for (int k = 0; k < 3; ++k) {          // Newton on g(u) = 0: 1e-5 -> 1e-9 -> 1e-17
      const double f = sn * (a + b * cs) - us;                            // 2 g
      const double fp = a * cs + b * (2.0 * cs * cs - 1.0) - 1.0;         // 2 g'
      nstep = f * Md::rcp(fp);
      us -= nstep;
      lidf_sincos_step(SPART_LIDF_ROTATE && ::fabs(nstep) < LIDF_ROT_MAX, -nstep, us, s2, c2, sn, cs);
    }
    const double s2x = 2.0 * sn * cs, c2x = 2.0 * cs * cs - 1.0;          // sin 2x, cos 2x at the fixed point
    const double y1 = a * cs + b * c2x, y2 = -a * sn - 2.0 * b * s2x, y3 = -a * cs - 4.0 * b * c2x, y4 = a * sn + 8.0 * b * s2x;
    const double r = 0.5 * (1.0 + y1), q2 = 0.25 * y2, q3 = y3 * (1.0 / 12.0), q4 = y4 * (1.0 / 48.0);
    const double r2 = r * r;
    const double i1 = Md::rcp(r - r2);                                    // 1 / (r (1 - r))
    const double a2 = q2 * i1;
    const double a3 = (q3 + 2.0 * a2 * r * q2) * i1 * Md::rcp(1.0 + r);             // / (r - r^3)
    const double a4 = (q4 + a2 * (q2 * q2 + 2.0 * r * q3) + 3.0 * a3 * r2 * q2) * i1 * Md::rcp(1.0 + r + r2);   // / (r - r^4)
    const double em = u - us;
    const double hm = em * (1.0 + em * (a2 + em * (a3 + em * a4)));
    const bool ok = r > 0.25 && r < 0.985 && ::fabs(a2 * em) < 0.02 && ::fabs(us) <= 1.0 && ::fabs(nstep) < 1e-12;
    if (ok) {
      // largest k with (1 - r) |h_m| r^k > 1e-8 (leading order of |g(u_{m+k})|) by binary descent over r^(2^j): the
      // stopping index is k + 1 up to the higher-order terms, which the three candidates below settle
      double pw[9];
      pw[0] = r;
      for (int j = 1; j < 9; ++j) pw[j] = pw[j - 1] * pw[j - 1];
      double G = (1.0 - r) * ::fabs(hm), rk = 1.0;
      for (int j = 8; j >= 0; --j) {
        const bool take = G * pw[j] > 1e-8;
        G = take ? G * pw[j] : G;
        rk = take ? rk * pw[j] : rk;
      }
      const double b3 = 2.0 * a2 * a2 - a3;
      double H = hm * rk, ek = 0.0;        // k: still above the threshold at leading order; then k + 1, k + 2
      bool found = false;
      for (int c = 0; c < 3; ++c) {
        const double e = H * (1.0 - H * (a2 - H * b3));                  // h^-1(H)
        const double g = e * ((r - 1.0) + e * (q2 + e * q3));              // g(u* + e)
        const bool hit = !found && !(::fabs(g) > 1e-8);
        ek = hit ? e : ek;
        found = found || hit;
        H *= r;
      }
      if (found) {
        y = us + ek * ((2.0 * r - 1.0) + ek * (0.5 * y2 + ek * (y3 * (1.0 / 6.0))));   // y(u* + e), y(u*) = u*
        more = false;
        if (jumped) *jumped = 1;
      }
    }
  }
  while (more && ++it < 100000) {           // lanes that did not jump: the rest of the literal iteration
    u += dx;
    lidf_sincos(u, s2, c2, sn, cs);
    y = sn * (a + b * cs);
    dx = 0.5 * (y - u);
    more = ::fabs(dx) > 1e-8;
  }
  return (2.0 * y + theta2) / PI;
}
#ifndef SPART_LIDF_JUMP
#define SPART_LIDF_JUMP 1
#endif
SPART_HD double lidf_dcum_lit(double a, double b, int i) { return lidf_dcum_lit_impl<(SPART_LIDF_JUMP != 0)>(a, b, i); }

// sin / cos of the 13 class-centre inclinations litab(i) and of twice the 12 class boundaries theta(i): constants of
// the model (sailh.py:49, 388-394), tabulated (math.sin / math.cos of the same double arguments) so that no sample
// pays for 38 library calls on them
SPART_HD double lidf_sin_litab(int i) {
  static constexpr double t[NLINCL] = {0.08715574274765817, 0.25881904510252074, 0.42261826174069944, 0.573576436351046, 0.7071067811865475, 0.8191520442889918, 0.9063077870366499, 0.9659258262890683, 0.9876883405951378, 0.992546151641322, 0.9961946980917455, 0.9986295347545738, 0.9998476951563913};
  return t[i];
}
SPART_HD double lidf_cos_litab(int i) {
  static constexpr double t[NLINCL] = {0.9961946980917455, 0.9659258262890683, 0.9063077870366499, 0.8191520442889918, 0.7071067811865476, 0.5735764363510462, 0.42261826174069944, 0.25881904510252074, 0.15643446504023092, 0.12186934340514749, 0.08715574274765814, 0.052335956242943966, 0.0174524064372836};
  return t[i];
}
SPART_HD double lidf_sin_2theta(int i) {
  static constexpr double t[NLINCL - 1] = {0.3420201433256687, 0.6427876096865393, 0.8660254037844386, 0.984807753012208, 0.984807753012208, 0.8660254037844387, 0.6427876096865395, 0.3420201433256689, 0.2756373558169992, 0.20791169081775931, 0.13917310096006533, 0.06975647374412552};
  return t[i];
}
SPART_HD double lidf_cos_2theta(int i) {
  static constexpr double t[NLINCL - 1] = {0.9396926207859084, 0.766044443118978, 0.5000000000000001, 0.17364817766693041, -0.1736481776669303, -0.4999999999999998, -0.7660444431189779, -0.9396926207859083, -0.9612616959383189, -0.9781476007338057, -0.9902680687415704, -0.9975640502598242};
  return t[i];
}

// lidf[13] = diff of the cumulative distribution (sailh.py:386-398)
SPART_HD void leaf_angles(double a, double b, double lidf[NLINCL]) {
  double prev = 0.0;
  for (int i = 0; i < NLINCL; ++i) {
    double F = (i < NLINCL - 1) ? lidf_dcum_lit(a, b, i) : 1.0;
    lidf[i] = F - prev;
    prev = F;
  }
}

// _volscatt for one leaf inclination (sailh.py:401-446).  Only the two arccosines are evaluated as such: the
// sines / cosines the reference takes of bts, bto, bt1, bt2, bt3 follow algebraically, because
// cos bts = -Cs/As (sin >= 0 on [0, pi]), delta1 = |bts - bto|, delta2 = pi - |bts + bto - pi|, and
// (bt1, bt2, bt3) is (min, median, max) of {psi, delta1, delta2} (delta1 <= delta2 always).
SPART_HD void volscatt1(double sin_tts, double cos_tts, double sin_tto, double cos_tto, double psi_rad,
                        double sin_psi, double cos_psi, double sin_l, double cos_l, double& chi_s, double& chi_o,
                        double& frho, double& ftau) {
  double Cs = cos_l * cos_tts, Ss = sin_l * sin_tts;
  double Co = cos_l * cos_tto, So = sin_l * sin_tto;
  double As = ::fmax(Ss, Cs), Ao = ::fmax(So, Co);
  double cbs = -Cs / As, cbo = -Co / Ao;                         // cos bts, cos bto
  double bts = ::acos(cbs), bto = ::acos(cbo);
  using Md = Mx<double>;           // (device: v_rsq_f64 + one Newton step, 1e-15; the two divisions above stay IEEE: -Cs / As must be
                                   //  EXACTLY -1 when As = Cs, acos magnifies anything less near +-1)
  double sbs = Md::sqrt(::fmax(0.0, 1.0 - cbs * cbs)), sbo = Md::sqrt(::fmax(0.0, 1.0 - cbo * cbo));
  chi_o = 2.0 / PI * ((bto - PI / 2) * Co + sbo * So);
  chi_s = 2.0 / PI * ((bts - PI / 2) * Cs + sbs * Ss);
  double delta1 = ::fabs(bts - bto);
  double delta2 = PI - ::fabs(bts + bto - PI);
  double cd1 = cbs * cbo + sbs * sbo, sd1 = ::fabs(sbs * cbo - cbs * sbo);   // cos / sin delta1
  double cd2 = cbs * cbo - sbs * sbo, sd2 = ::fabs(sbs * cbo + cbs * sbo);   // cos / sin delta2
  // bt1 = min(psi, delta1), bt3 = max(psi, delta2), bt2 = the remaining one (:429-431)
  bool p_lt_d1 = psi_rad < delta1, p_gt_d2 = psi_rad > delta2;
  double cbt1 = p_lt_d1 ? cos_psi : cd1;
  double cbt3 = p_gt_d2 ? cos_psi : cd2;
  double bt2 = p_lt_d1 ? delta1 : (p_gt_d2 ? delta2 : psi_rad);
  double sbt2 = p_lt_d1 ? sd1 : (p_gt_d2 ? sd2 : sin_psi);
  double T1 = 2.0 * Cs * Co + Ss * So * cos_psi;
  double T2 = sbt2 * (2.0 * As * Ao + Ss * So * cbt1 * cbt3);
  double Jmin = bt2 * T1 - T2;
  double Jplus = (PI - bt2) * T1 + T2;
  constexpr double I2PI2 = 1.0 / (2.0 * PI * PI);                 // (a multiplication: <= 1 ulp from the reference's division)
  frho = ::fmax(0.0, Jplus * I2PI2);
  ftau = ::fmax(0.0, -Jmin * I2PI2);
}

// Hot-spot integrals (sailh.py:115-135, 216, 219).  The reference integrates Psofunction over
// each of the 61 layers [xl_j - dx, xl_j] with QUADPACK and uses only
//   sum_{j<60} Pso_j * dx = int_{-1}^{0} f,   Pso_60 = (1/dx) int_{-1-dx}^{-1} f   (sic: below the canopy).
// Both are evaluated here with Gauss-Legendre panels (10-point; 8-point in the FAST prelude) that halve towards x = 0, where
// f varies on the scale 1/max(alpha, (K+k)LAI).
struct PsoFn {
  double A, C, alpha;
  bool hot;  // dso == 0
  SPART_HD double operator()(double x) const {
    if (hot) return Mx<double>::exp_poly(A * x);  // A holds (K+k-sqrt(Kk)) LAI in this branch (:127)
    return Mx<double>::exp_poly(A * x + C * Mx<double>::one_minus_exp_neg_poly(-alpha * x));  // :121-125 (x <= 0)
  }
};

// 10-point rule on [-1,1] (positive half): the default prelude, panels halved until rate * h <= 2.  Against the 16-point
// rule with rate * h <= 4 it replaces: <= 4e-15 relative on both integrals over 20 000 random geometries (q 0.001 ... 0.5,
// LAI 0.005 ... 10, exact hot spot included), 6e-16 against 30-digit quadrature, with 20-30 percent fewer integrand
// evaluations
static constexpr double GL10_X[5] = {0.148874338981631210884826, 0.4333953941292471907992659, 0.6794095682990244062343274,
                                     0.8650633666889845107320967, 0.973906528517171720077964};
static constexpr double GL10_W[5] = {0.295524224714752870173893, 0.2692667193099963550912269, 0.2190863625159820439955349,
                                     0.1494513491505805931457763, 0.06667134430868813759356881};

// 8-point rule (the FAST prelude: panels are halved until rate * h <= 2, error <= 2e-12)
static constexpr double GL8_X[4] = {0.1834346424956498049394761, 0.5255324099163289858177390,
                                    0.7966664774136267395915539, 0.9602898564975362316835609};
static constexpr double GL8_W[4] = {0.3626837833783619829651504, 0.3137066458778872873379622,
                                    0.2223810344533744705443560, 0.1012285362903762591525314};

template <bool FAST> SPART_HD double gl_panel(const PsoFn& f, double a, double b) {
  double h = 0.5 * (b - a), c = 0.5 * (b + a), s = 0.0;
  if (FAST) {
    for (int i = 0; i < 4; ++i) s += GL8_W[i] * (f(c + h * GL8_X[i]) + f(c - h * GL8_X[i]));
  } else {
    for (int i = 0; i < 5; ++i) s += GL10_W[i] * (f(c + h * GL10_X[i]) + f(c - h * GL10_X[i]));
  }
  return s * h;
}

// The same two integrals in closed form (round 4).  With t = e^(alpha x) the integrand exp(A x + C (1 - e^(alpha x))) dx
// becomes t^(a-1) e^(C (1 - t)) dt / alpha, a = A / alpha, i.e. incomplete gamma functions, and Kummer's all-positive series
//     e^C int_0^T t^(a-1) e^(-C t) dt = e^(C (1 - T)) T^a g(a, C T),   g(a, x) = sum_n x^n / (a (a+1) ... (a+n)),
// has no cancellation.  The n = 0 terms of the two limits are taken out (they nearly cancel when a is small) and combined
// into 1 - e^(-z) forms:  g = 1/a + x g1,  g1(a, x) = sum_{n>=1} x^(n-1) / (a ... (a+n)),  and for a stretch of length L,
// with tL = e^(-alpha L), zL = A L - C (1 - tL) >= A L / 2, EL = e^-zL:
//     J(A, C, alpha, L) = int_0^L exp(-A u + C (1 - e^(-alpha u))) du = (1 - EL) / A + (C / alpha) [g1(a, C) - EL tL g1(a, C tL)]
//     int_{-1}^{0} f = J(A, C, alpha, 1),     int_{-1-dx}^{-1} f = f(-1) J(A, C e^-alpha, alpha, dx)
// (below the canopy the integrand is f(-1) times the same form with C e^-alpha in place of C).
// Against 40-digit quadrature over random geometries with C <= 8: <= 1.5e-14 relative (the reference's QUADPACK: ~1e-13).
// Used when 0 < C <= 8 (97 % of the benchmark's samples have C <= 2, all but a few in 10^4 C <= 8): a wave of 64 samples
// then needs ~20-40 series terms instead of the 8.3 panels x 10 points its slowest lane used to cost; the few lanes with
// C > 8 (a small alpha, i.e. a SMOOTH integrand) keep the panels, and need few of them.
SPART_HD double hotspot_g1(double a, double x) {
  using Md = Mx<double>;
  double t = Md::rcp(a * (a + 1.0)), s = t, d = a + 1.0;
  for (int n = 2; n < 200; n += 2) {      // two terms per pass: both divisions from ONE reciprocal of the product
    const double d1 = d + 1.0, d2 = d + 2.0;
    const double r = Md::rcp(d1 * d2);
    const double t1 = (t * x) * (d2 * r);   // t x / d1
    const double t2 = (t1 * x) * (d1 * r);  // t1 x / d2
    s += t1;
    s += t2;
    t = t2;
    d = d2;
    if (!(t2 > 1e-17 * s)) break;         // (also ends on NaN; at most one term beyond the old stopping point)
  }
  return s;
}

// J(A, C, alpha, L) = int_0^L exp(-A u + C (1 - e^(-alpha u))) du  (tL = e^(-alpha L) supplied by the caller)
SPART_HD double hotspot_J(double A, double C, double alpha, double a, double iA, double ialpha, double L, double tL) {
  using Md = Mx<double>;
  const double zL = A * L - C * Md::one_minus_exp_neg(alpha * L, tL);         // >= A L / 2
  const double EL = Md::exp_poly(-zL);
  return Md::one_minus_exp_neg(zL, EL) * iA + C * ialpha * (hotspot_g1(a, C) - EL * tL * hotspot_g1(a, C * tL));
}

// nl = canopy.nlayers (sailh.py:48): it enters the model ONLY through the width dx = 1 / nl of the stretch below the canopy
// (Pso[nl], :131-135, 219) -- sum(Pso[0:nl]) iLAI = LAI int_{-1}^{0} f whatever nl is (:216).  The default (the constant
// NLAYER, folded at compile time) is the SAIL assumption of CanopyStructure (:345).
SPART_HD bool hotspot_series(double A, double C, double alpha, double& int_canopy, double& pso2w, int nl = NLAYER) {
  using Md = Mx<double>;
  // C <= 8: Kummer's series needs <= ~45 terms; A + alpha >= 2: otherwise the integrand is nearly flat, two panels do, and
  // the bracket of J would lose digits (its relative gap is ~ alpha L)
  if (!(C > 0.0 && C <= 8.0 && A + alpha >= 2.0 && A > 1e-3 && alpha > 0.0 && alpha < 1e300)) return false;
  const double dx = 1.0 / nl;
  const double ialpha = Md::rcp(alpha), iA = Md::rcp(A), a = A * ialpha;
  const double t1 = Md::exp_poly(-alpha), td = Md::exp_poly(-alpha * dx);
  // int_{-1}^{0} f = J(A, C, alpha, 1);  f(-1 - u) = f(-1) exp(-A u + C t1 (1 - e^(-alpha u))):
  // int_{-1-dx}^{-1} f = f(-1) J(A, C t1, alpha, dx),  f(-1) = exp(-(A - C (1 - t1)))
  int_canopy = hotspot_J(A, C, alpha, a, iA, ialpha, 1.0, t1);
  const double f1 = Md::exp_poly(-(A - C * Md::one_minus_exp_neg(alpha, t1)));
  pso2w = f1 * hotspot_J(A, C * t1, alpha, a, iA, ialpha, dx, td) * (double)nl;
  return true;
}

template <bool FAST>
SPART_HD void hotspot_integrals(double K, double k, double LAI, double q, double dso, double& int_canopy,
                                double& pso2w, int nl = NLAYER) {
  const double dx = 1.0 / nl;
  PsoFn f;
  double rate;
  if (dso != 0.0) {
    f.hot = false;
    f.alpha = (dso / q) * 2.0 / (k + K);
    f.A = (K + k) * LAI;
    f.C = ::sqrt(K * k) * LAI / f.alpha;
    rate = ::fmax(f.alpha, f.A + ::sqrt(K * k) * LAI);
#ifndef SPART_HOTSPOT_SERIES
#define SPART_HOTSPOT_SERIES 1
#endif
    if (SPART_HOTSPOT_SERIES && hotspot_series(f.A, f.C, f.alpha, int_canopy, pso2w, nl)) return;
  } else {
    f.hot = true;
    f.alpha = 0.0;
    f.C = 0.0;
    f.A = (K + k) * LAI - ::sqrt(K * k) * LAI;
    rate = f.A;
  }
  // number of halvings so that rate * 2^-m <= 2; NaN / inf rates fall through with m = 0 / 40
  const double lim = 2.0;
  int m = 0;
  double hw = 1.0;
  while (rate * hw > lim && m < 40) {
    hw *= 0.5;
    ++m;
  }
  double tot = 0.0, lo = -1.0;
  for (int i = 0; i < m; ++i) {
    tot += gl_panel<FAST>(f, lo, 0.5 * lo);
    lo *= 0.5;
  }
  tot += gl_panel<FAST>(f, lo, 0.0);
  int_canopy = tot;
  if (nl >= NLAYER) {
    pso2w = gl_panel<FAST>(f, -1.0 - dx, -1.0) / dx;   // smooth there: rate * dx/2 <= 2 unless rate > 240
  } else {
    // fewer, thicker layers (a user's canopy.nlayers): the stretch [-1 - dx, -1] in panels no wider than the default's 1/60
    const int npan = (NLAYER + nl - 1) / nl;
    const double h = dx / npan;
    double acc = 0.0, hi = -1.0;
    for (int i = 0; i < npan; ++i) {
      acc += gl_panel<FAST>(f, hi - h, hi);
      hi -= h;
    }
    pso2w = acc / dx;
  }
}

// ------------------------------------------------------------------------------------------
// The prelude: 27 parameters -> band-kernel constants (T) + atmosphere scalars (double)
enum PreludeMask { PRE_LEAF = 1, PRE_SOIL = 2, PRE_CANOPY = 4, PRE_ATM = 8, PRE_ALL = 15 };

// FAST (used for T = float, tolerance 1e-4): Newton LIDF and the 8-point hot-spot rule, both ~1e-7 from
// the literal forms; the default keeps the reference's iteration (lidf_dcum_lit) and uses the 10-point rule.
// Results leave through `out` as soon as they exist -- out.c(ConstIdx, v), out.a(AtmIdx, v), out.l(i, lidf_i) -- so
// that a kernel can store them straight to memory instead of holding 64 float64 values in registers across the
// LIDF iteration and the hot-spot quadrature (k_prelude: 262 -> fewer VGPRs, two waves per SIMD instead of one).
// Groups excluded by `mask` are not written at all.
// USER (the canopy state the reference's SAILH reads from the object at call time, sailh.py:48, 51): lidf_row = the sample's
// canopy.lidf[13] (nullptr: derived from LIDFa / LIDFb as below) and nl_user = canopy.nlayers.  Without USER both are
// compile-time facts (no lidf pointer, NLAYER) and the code is the one every earlier round measured.
template <bool FAST, bool USER = false, typename In, typename Out>
SPART_HD void sample_prelude_to(const In& in /* in(i) = parameter i of 27, read when first needed */, double rho_th,
                                double tau_th, int mask, Out& out, const double* lidf_row = nullptr, int nl_user = NLAYER) {
  SPART_NO_CONTRACT
  const double d2r = PI / 180.0;
  const int nl = USER ? nl_user : NLAYER;
  const bool given = USER && lidf_row != nullptr;
  if (mask & PRE_LEAF) {
  // ---- leaf (prospect_5d.py:135-155, 170-179)
  double Cab = in(0), Cdm = in(1), Cw = in(2), Cs = in(3), Cca = in(4), Cant = in(5), N = in(6), PROT = in(7), CBC = in(8);
  if ((PROT > 0.0 || CBC > 0.0) && Cdm > 0.0) Cdm = 0.0;  // PROSPECT-PRO rule (:148-155)
  double iN = 1.0 / N;
  out.c(C_CAB, Cab * iN);
  out.c(C_CCA, Cca * iN);
  out.c(C_CDM, Cdm * iN);
  out.c(C_CW, Cw * iN);
  out.c(C_CS, Cs * iN);
  out.c(C_CANT, Cant * iN);
  out.c(C_CBC, CBC * iN);
  out.c(C_PROT, PROT * iN);
  out.c(C_NM1, N - 1.0);
  out.c(C_RHO_TH, rho_th);
  out.c(C_TAU_TH, tau_th);
  }
  if (mask & PRE_SOIL) {
  // ---- soil (bsm.py:49-52, 99-103, 121)
  double Bs = in(9), lat = in(10), lon = in(11), SMp = in(12), SMC = in(13), film = in(14);
  out.c(C_F1, Bs * ::sin(lat * d2r));
  out.c(C_F2, Bs * ::cos(lat * d2r) * ::sin(lon * d2r));
  out.c(C_F3, Bs * ::cos(lat * d2r) * ::cos(lon * d2r));
  double mu = (SMp - 5.0) / SMC;
  bool wet = mu > 0.0;
  out.c(C_WET, wet ? 1.0 : 0.0);
  {
    double e = wet ? ::exp(-mu) : 1.0, pw = 1.0, fact = 1.0, fsum = 0.0;
    for (int k = 0; k < 7; ++k) {  // poisson.pmf(k, mu) = e^-mu mu^k / k!
      if (k > 0) {
        pw *= mu;
        fact *= k;
      }
      double f = wet ? e * pw / fact : (k == 0 ? 1.0 : 0.0);
      out.c(C_FM0 + k, f);
      if (k > 0) fsum += f;
    }
    out.c(C_FMSUM, fsum);
  }
  out.c(C_FILM2L, 2.0 * film * 1.4426950408889634);
  }
  double tts = in(19), tto = in(20), rel = in(21);
  if (mask & PRE_CANOPY) {
  // ---- canopy geometry (sailh.py:46-105)
  double LAI = in(15), LIDFa = given ? 0.0 : in(16), LIDFb = given ? 0.0 : in(17), q = in(18);
  double psi = ::fabs(rel - 360.0 * ::rint(rel / 360.0));  // :65 (Python round = half-to-even = rint)
  double psi_rad = psi * d2r;
  double sin_tts = ::sin(tts * d2r), cos_tts = ::cos(tts * d2r), tan_tts = ::tan(tts * d2r);
  double sin_tto = ::sin(tto * d2r), cos_tto = ::cos(tto * d2r), tan_tto = ::tan(tto * d2r);
  double sin_psi, cos_psi;
  ::sincos(psi_rad, &sin_psi, &cos_psi);
  double dso = ::sqrt(tan_tts * tan_tts + tan_tto * tan_tto - 2.0 * tan_tts * tan_tto * cos_psi);  // :78
  double ks = 0, ko = 0, bf = 0, sob = 0, sof = 0, Fprev = 0;
  for (int i = 0; i < NLINCL; ++i) {
    double li;
    if (given) {
      li = lidf_row[i];                                           // canopy.lidf as handed in (sailh.py:51)
    } else {
      double F = (i < NLINCL - 1) ? (FAST ? lidf_dcum_newton(LIDFa, LIDFb, i)
                                          : lidf_dcum_lit(LIDFa, LIDFb, i))
                                  : 1.0;
      li = F - Fprev;
      Fprev = F;
    }
    out.l(i, li);
    double sl = lidf_sin_litab(i), cl = lidf_cos_litab(i);   // sin / cos of litab(i), sailh.py:81
    double chi_s, chi_o, frho, ftau;
    volscatt1(sin_tts, cos_tts, sin_tto, cos_tto, psi_rad, sin_psi, cos_psi, sl, cl, chi_s, chi_o, frho, ftau);  // :81-83
    ks += chi_s / cos_tts * li;                                                               // :85, 93
    ko += chi_o / cos_tto * li;                                                               // :86, 94
    bf += cl * cl * li;                                                                       // :90, 95
    sob += frho * PI / (cos_tts * cos_tto) * li;                                              // :88, 96
    sof += ftau * PI / (cos_tts * cos_tto) * li;                                              // :89, 97
  }
  out.c(C_SOB, sob);
  out.c(C_SOF, sof);
  out.c(C_HBF, 0.5 * bf);
  out.c(C_KS, ks);
  out.c(C_KO, ko);
  out.c(C_LAI, LAI);
  out.c(C_LAI2, LAI * 1.4426950408889634);
  double tss = ::exp(-ks * LAI), too = ::exp(-ko * LAI);  // :200-201
  out.c(C_TSS, tss);
  out.c(C_TOO, too);
  out.c(C_Z, (1.0 - tss * too) / (ko + ks));             // :203
  double ic, p2w;
  hotspot_integrals<FAST>(ko, ks, LAI, q, dso, ic, p2w, nl);
  out.c(C_HOT, ic * LAI);   // sum(Pso[0:60]) * iLAI  (:216)
  out.c(C_PSO2W, p2w);      // Pso[60]               (:219)
  }
  if (mask & PRE_ATM) {
  // ---- atmosphere scalars (smac.py:94-102, 128-138) + ET factor (SPART.py:345-353)
  double psi_s = in(21);  // SMAC uses rel_angle unfolded (smac.py:38)
  double Pa = in(25);
  const double cdr = PI / 180.0, crd = 180.0 / PI;
  double us = ::cos(tts * cdr), uv = ::cos(tto * cdr);
  double cksi = -((us * uv) + (::sqrt(1.0 - us * us) * ::sqrt(1.0 - uv * uv) * ::cos(psi_s * crd)));  // sic (:130)
  if (cksi < -1.0) cksi = -1.0;                                                                        // :134-135
  const double am = 1.0 / us + 1.0 / uv, peq = Pa / 1013.25;
  out.a(A_US, us);
  out.a(A_UV, uv);
  out.a(A_M, am);
  out.a(A_PEQ, peq);
  out.a(A_PA, Pa);
  out.a(A_AOT, in(22));
  out.a(A_CKSI, cksi);
  out.a(A_KSID, crd * ::acos(cksi));
  out.a(A_LOGPEQ, ::log(peq));
  out.a(A_LOGM, ::log(am));
  out.a(A_LOGO3M, ::log(in(23) * am));
  out.a(A_LOGH2OM, ::log(in(24) * am));
  double b = 2.0 * PI * in(26) / 365.0;
  double sb, cb;
  ::sincos(b, &sb, &cb);                                    // cos 2b, sin 2b by the double-angle formulas; cos(tts) = us:
  double corr = 1.00011 + 0.034221 * cb + 0.00128 * sb + 0.000719 * (cb * cb - sb * sb) +   // three library calls less,
                0.000077 * (2.0 * sb * cb);                                                 // 1e-16 from the literal form
  out.a(A_LAF, corr * us / PI);                             // SPART.py:345-353 (cos(tts pi/180) there)
  }
}

// the same into plain arrays (tests/hostmath): unwritten entries are zero
template <typename T> struct PreludeArrays {
  T* cst;
  double* atm;
  double* lidf;
  SPART_HD void c(int i, double v) { cst[i] = T(v); }
  SPART_HD void a(int i, double v) { atm[i] = v; }
  SPART_HD void l(int i, double v) { lidf[i] = v; }
};
template <typename T, bool FAST = false>
SPART_HD void sample_prelude(const double* p /*[27]*/, double rho_th, double tau_th, int mask, T* cst /*[NCONST]*/,
                             double* atm /*[NATM]*/, double* lidf_out /*[13]*/, const double* lidf_in = nullptr, int nl = 0) {
  for (int i = 0; i < NCONST; ++i) cst[i] = T(0);
  for (int i = 0; i < NATM; ++i) atm[i] = 0.0;
  for (int i = 0; i < NLINCL; ++i) lidf_out[i] = 0.0;
  PreludeArrays<T> out{cst, atm, lidf_out};
  if (lidf_in || nl > 0) sample_prelude_to<FAST, true>([p](int i) { return p[i]; }, rho_th, tau_th, mask, out, lidf_in, nl > 0 ? nl : NLAYER);
  else sample_prelude_to<FAST>([p](int i) { return p[i]; }, rho_th, tau_th, mask, out);
}

// ------------------------------------------------------------------------------------------
// SMAC for one (sample, sensor band)                                     (smac.py:94-211)
// coef points at this band's 48 coefficients with stride `cs` between rows.
struct SmacOut {
  double Ta_s, Ta_o, Tg, Ra_dd, Ra_so, Ta_ss, Ta_sd, Ta_oo, Ta_do;
};

enum CoefRow {
  K_AH2O = 0, K_NH2O, K_AO3, K_NO3, K_AO2, K_NO2, K_PO2, K_ACO2, K_NCO2, K_PCO2, K_ACH4, K_NCH4, K_PCH4,
  K_ANO2, K_NNO2, K_PNO2, K_ACO, K_NCO, K_PCO, K_A0S, K_A1S, K_A2S, K_A3S, K_A0T, K_A1T, K_A2T, K_A3T,
  K_TAUR, K_A0TAUP, K_A1TAUP, K_WO, K_GC, K_A0P, K_A1P, K_A2P, K_A3P, K_A4P, K_REST1, K_REST2, K_REST3,
  K_REST4, K_RESR1, K_RESR2, K_RESR3, K_RESA1, K_RESA2, K_RESA3, K_RESA4
};

// atm: anything indexable by AtmIdx (a plain array of NATM doubles, or a sample's column of the LDS copy in k_columns)
// C(r): coefficient row r (CoefRow) of this band
template <typename A, typename CF> SPART_HD SmacOut smac_band_c(const A& atm, const CF& C) {
  using Md = Mx<double>;
  // exp / sqrt: the library's (default), or with SPART_SMAC_LIBM=0 the float64 band arithmetic's own table-driven exp (LDS
  // tables staged by the calling kernel: stage_f64_tables) and Newton-refined rsq: 17 % fewer instructions, more registers
#ifndef SPART_SMAC_LIBM
#define SPART_SMAC_LIBM 1     // measured: the table exp needs 168 VGPRs here (three waves per SIMD: 0.51 ms per 1M spectra, spilling at
#endif                        // 128: 0.95 ms) against 0.47 ms with the library's exp at 128 (profiles/r5_ab_smac_exp.txt)
#if SPART_SMAC_LIBM
  auto EXP = [](double x) { return ::exp(x); };
  auto SQRT = [](double x) { return ::sqrt(x); };
#else
  auto EXP = [](double x) { return Md::exp_keepnan(x); };
  auto SQRT = [](double x) { return Md::sqrt(x); };
#endif
  double us = atm[A_US], uv = atm[A_UV], m = atm[A_M], Peq = atm[A_PEQ], Pa = atm[A_PA];
  double taup550 = atm[A_AOT], cksi = atm[A_CKSI], ksiD = atm[A_KSID];
  double lpeq = atm[A_LOGPEQ], lm = atm[A_LOGM];
  double ius = Md::rcp(us), iuv = Md::rcp(uv);
  double taup = C(K_A0TAUP) + C(K_A1TAUP) * taup550;  // :103
  // gaseous transmittances t = exp(a (u m)^n), u = Peq^p (:105-119), with x^n evaluated as exp(n ln x) from the
  // per-sample logarithms (arguments are positive; ln 0 = -inf gives 0^n = 0 as numpy does).  A gas whose
  // coefficient a is zero in this band has t = exp(0) = 1 exactly (most of CO, CH4, NO2, O2, CO2 in most
  // bands): it is skipped, which leaves the product below bit-identical.
  auto gas = [&](int ka, int kn, double lum) -> double {
    double av = C(ka);
    return (av != 0.0) ? EXP(av * EXP(C(kn) * lum)) : 1.0;
  };
  auto pgas = [&](int ka, int kn, int kp) -> double {
    double av = C(ka);
    return (av != 0.0) ? EXP(av * EXP(C(kn) * (C(kp) * lpeq + lm))) : 1.0;
  };
  double to3 = gas(K_AO3, K_NO3, atm[A_LOGO3M]);
  double th2o = gas(K_AH2O, K_NH2O, atm[A_LOGH2OM]);
  double to2 = pgas(K_AO2, K_NO2, K_PO2);
  double tco2 = pgas(K_ACO2, K_NCO2, K_PCO2);
  double tch4 = pgas(K_ACH4, K_NCH4, K_PCH4);
  double tno2 = pgas(K_ANO2, K_NNO2, K_PNO2);
  double tco = pgas(K_ACO, K_NCO, K_PCO);
  SmacOut o;
  o.Tg = th2o * to3 * to2 * tco2 * tch4 * tco * tno2;  // :119
  o.Ra_dd = C(K_A0S) * Peq + C(K_A3S) + C(K_A1S) * taup550 + C(K_A2S) * taup550 * taup550;  // :122
  double tP = C(K_A2T) * Peq + C(K_A3T);
  o.Ta_s = C(K_A0T) + C(K_A1T) * taup550 * ius + tP * Md::rcp(1.0 + us);  // :125
  o.Ta_o = C(K_A0T) + C(K_A1T) * taup550 * iuv + tP * Md::rcp(1.0 + uv);  // :126
  double taur = C(K_TAUR);
  double iusuv = ius * iuv;
  double ray_phase = 0.7190443 * (1.0 + (cksi * cksi)) + 0.0412742;  // :141
  double ray_ref = (taur * ray_phase) * (0.25 * iusuv);              // :142
  ray_ref = ray_ref * Pa * (1.0 / 1013.25);                          // :143 (a multiplication: <= 1 ulp from the division)
  double taurz = taur * Peq;                                         // :144
  double aer_phase = C(K_A0P) + C(K_A1P) * ksiD + C(K_A2P) * ksiD * ksiD + C(K_A3P) * ksiD * ksiD * ksiD +
                     C(K_A4P) * (ksiD * ksiD) * (ksiD * ksiD);  // :146-148
  double wo = C(K_WO), gc = C(K_GC);
  double g3 = 3.0 - wo * 3.0 * gc;
  double ig3 = Md::rcp(g3);
  double ak2 = (1.0 - wo) * g3;  // :149-150
  double ak = SQRT(ak2);
  double idus = Md::rcp(1.0 - ak2 * us * us);
  double e = -3.0 * us * us * wo * 0.25 * idus;  // :153-157
  double f = -(1.0 - wo) * 3.0 * gc * us * us * wo * 0.25 * idus;
  double dp = e * ius * (1.0 / 3.0) + us * f;
  double d = e + f;
  double b = 2.0 * ak * ig3;
  double eak = EXP(ak * taup), emak = Md::rcp(eak);
  double delta = eak * (1.0 + b) * (1.0 + b) - emak * (1.0 - b) * (1.0 - b);  // :158
  double ww = wo * 0.25;
  double ss = us * idus;
  double q1 = 2.0 + 3.0 * us + (1.0 - wo) * 3.0 * gc * us * (1.0 + 2.0 * us);
  double q2 = 2.0 - 3.0 * us - (1.0 - wo) * 3.0 * gc * us * (1.0 - 2.0 * us);
  const double e_us = EXP(-taup * ius);      // e^(-taup / us)
  double q3 = q2 * e_us;
  double wsd = ww * ss * Md::rcp(delta);
  double c1 = wsd * (q1 * eak * (1.0 + b) + q3 * (1.0 - b));    // :164
  double c2 = -wsd * (q1 * emak * (1.0 - b) + q3 * (1.0 + b));  // :165
  double cp1 = c1 * ak * ig3;
  double cp2 = -c2 * ak * ig3;
  double z = d - wo * 3.0 * gc * uv * dp + wo * aer_phase * 0.25;  // :168-173
  double x = c1 - wo * 3.0 * gc * uv * cp1;
  double y = c2 - wo * 3.0 * gc * uv * cp2;
  // aa1 = uv/(1 + ak uv), aa2 = uv/(1 - ak uv), aa3 = us uv/(us + uv); taup/aa_i needs no division
  double n1 = 1.0 + ak * uv, n2 = 1.0 - ak * uv, n3 = us + uv;
  double aa1 = uv * Md::rcp(n1), aa2 = uv * Md::rcp(n2), aa3 = us * uv * Md::rcp(n3);
  // taup n1 / uv = taup / uv + ak taup, taup n2 / uv = taup / uv - ak taup, taup n3 / (us uv) = taup / uv + taup / us: the three
  // exponentials of :175-179 are products of e^(-taup / uv) with e^(-+ ak taup) and e^(-taup / us), which :158 and :163 need
  // anyway -- one exp instead of three, each product within 2 ulp of the direct exponential
#if defined(SPART_SMAC_SHARE_EXP) && !SPART_SMAC_SHARE_EXP
  double aer_ref1 = x * aa1 * (1.0 - EXP(-taup * n1 * iuv));  // :175-179
  double aer_ref2 = y * aa2 * (1.0 - EXP(-taup * n2 * iuv));
  double aer_ref3 = z * aa3 * (1.0 - EXP(-taup * n3 * iusuv));
#else
  const double e_uv = EXP(-taup * iuv);
  double aer_ref1 = x * aa1 * (1.0 - e_uv * emak);  // :175-179
  double aer_ref2 = y * aa2 * (1.0 - e_uv * eak);
  double aer_ref3 = z * aa3 * (1.0 - e_uv * e_us);
#endif
  double aer_ref = (aer_ref1 + aer_ref2 + aer_ref3) * iusuv;
  double rr = taur * ray_phase * iusuv;
  double Res_ray = C(K_RESR1) + C(K_RESR2) * rr + C(K_RESR3) * (rr * rr);  // :182-186
  double ta = taup * m * cksi;
  double Res_aer = (C(K_RESA1) + C(K_RESA2) * ta + C(K_RESA3) * (ta * ta)) + C(K_RESA4) * (ta * ta * ta);  // :189-191
  double tautot = taup + taurz;  // :194
  double tt = tautot * m * cksi;
  double Res_6s = (C(K_REST1) + C(K_REST2) * tt + C(K_REST3) * (tt * tt)) + C(K_REST4) * (tt * tt * tt);  // :196-198
  o.Ra_so = ray_ref - Res_ray + aer_ref - Res_aer + Res_6s;  // :201
  o.Ta_ss = EXP(-tautot * ius);                            // :204-207
  o.Ta_oo = EXP(-tautot * iuv);
  o.Ta_sd = o.Ta_s - o.Ta_ss;
  o.Ta_do = o.Ta_o - o.Ta_oo;
  return o;
}
// the same with the coefficients in a (48, cs) row-major block: coef points at this band's column
template <typename A> SPART_HD SmacOut smac_band(const A& atm, const double* coef, int cs) {
  return smac_band_c(atm, [coef, cs](int r) { return coef[(size_t)r * cs]; });
}

// TOC -> TOA (SPART.py:243-252)
SPART_HD void toc_to_toa(const SmacOut& a, double rv_so, double rv_do, double rv_dd, double rv_sd, double La,
                         double& R_TOC, double& R_TOA, double& L_TOA) {
  using Md = Mx<double>;
  const double imr = Md::rcp(1.0 - rv_dd * a.Ra_dd);          // (device: v_rcp_f64 + one Newton step, 2e-15 relative)
  double rtoa0 = a.Ra_so + a.Ta_ss * rv_so * a.Ta_oo;
  double rtoa1 = (a.Ta_sd * rv_do + a.Ta_ss * rv_sd * a.Ra_dd * rv_do) * a.Ta_oo * imr;
  double rtoa2 = (a.Ta_ss * rv_sd + a.Ta_sd * rv_dd) * a.Ta_do * imr;
  R_TOC = (a.Ta_ss * rv_so + a.Ta_sd * rv_do) * Md::rcp(a.Ta_ss + a.Ta_sd);
  R_TOA = a.Tg * (rtoa0 + rtoa1 + rtoa2);
  L_TOA = La * R_TOA;
}

}  // namespace spart

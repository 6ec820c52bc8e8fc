// HIP kernels of the SPART hot path for gfx950 (MI355X).
//
// Layout: the band axis is mapped onto lanes.  A workgroup of 256 lanes owns
// 256 consecutive bands and walks a chunk of samples; each lane keeps ITS band's 17 table
// values in VGPRs for the whole walk (the tables are read from memory once per workgroup,
// coalesced along the band axis), while the per-sample constants are wave-uniform and are
// staged through LDS from the prelude's workspace (broadcast reads into VGPRs).  2001 optical bands + one
// thermal evaluation (the 161 thermal bands are identical, SPART.py:427-470) fill
// 2002 of the 2048 lanes of eight workgroups.
#pragma once

#include "spart_math.h"

namespace spart {

// (Kernels that are not templates are `static`: this header is included by two translation units.)
constexpr int TILE = 256;                 // lanes (= bands) per workgroup
constexpr int NTILE = 8;                  // 8 * 256 = 2048 >= NEVAL
constexpr int NTILE_FULL = 9;             // 9 * 256 >= 2162 (standalone SAILH: arbitrary thermal inputs)
constexpr int MAX_NB = 64;

// XCD-aware blockIdx -> (chunk, tile) map (cdna guide T1).  Workgroups are dealt round-robin over the 8
// XCDs, so blocks b and b + 8 share an XCD (a speed assumption only).  Within each group of 64 consecutive
// blocks the 8 tiles of one chunk get ids {x, x+8, ..., x+56}: they land on ONE XCD and read the chunk's
// per-sample constants through that XCD's L2 once, instead of once per XCD (8x the HBM fetches).
// Grids are padded to a multiple of 64 blocks; padded chunks (ck >= nchunk) exit.
__device__ __forceinline__ void xcd_map(unsigned b, int& tile, int64_t& ck) {
  tile = (int)((b >> 3) & 7u);
  ck = (int64_t)(b >> 6) * 8 + (b & 7u);
}
inline unsigned xcd_grid(int64_t nchunk) { return (unsigned)(((nchunk + 7) / 8) * 64); }

struct ParamPtrs {
  const double* p[NPARAM];
  const double* rho_th;
  const double* tau_th;
  const double* lidf;   // optional (B, 13) row-major canopy.lidf as the caller set it (sailh.py:51); read by k_prelude<., true> only
  int nlayers;          // canopy.nlayers (sailh.py:48); read by k_prelude<., true> only
};

// ------------------------------------------------------------------------------------------
// Workspace layout of the per-sample quantities: STRUCTURE OF ARRAYS with row pitch Bp (= B rounded up to 64):
//   cst[i][s]  (NCONST = 40 rows, 36 used, float and / or double), atm[i][s] (16 rows, 13 used, double).
// Every kernel that has the sample on its lanes (prelude, column kernel) then reads and writes
// them coalesced; the band kernels (band on lanes) stage 32 samples x 40 constants per workgroup copy.
inline int64_t row_pitch_of(int64_t B) { return (B + 63) & ~int64_t(63); }

// A sample's constants seen from a band kernel: either contiguous in LDS (plain pointer) or column s of the
// structure-of-arrays block (wave-uniform address -> scalar loads).
template <typename T> struct ConstCol {
  const T* p;
  int64_t stride;
  __device__ __forceinline__ T operator[](int i) const { return p[(int64_t)i * stride]; }
};

// One workgroup copy of 32 samples x NCONST (40) constants from the structure-of-arrays block (blk = &cst[0][first sample]) into
// LDS as [sample][NCONST]: lanes 32 r .. 32 r + 31 read 128 contiguous bytes of row r + 8 k.  The row base is wave-uniform
// (SGPRs) and the lane carries ONE loop-invariant 32-bit byte offset (host: 8 Bp sizeof(T) < 4 GB), so nothing but
// that register stays live across the sample loop.
template <typename T>
__device__ __forceinline__ void stage_constants(T* lds_c, const T* __restrict__ blk, int64_t Bp, int nsub) {
  static_assert(TILE == 256 && NCONST % 8 == 0, "8 rows of 32 samples per pass");
  const int si = threadIdx.x & 31, i0 = threadIdx.x >> 5;
  const unsigned lane_off = ((unsigned)i0 * (unsigned)Bp + (unsigned)si) * (unsigned)sizeof(T);
  if (si < nsub) {
#pragma unroll
    for (int k = 0; k < NCONST / 8; ++k) {
      const T* row = blk + (int64_t)(8 * k) * Bp;
      lds_c[si * NCONST + i0 + 8 * k] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + lane_off);
    }
  }
}

// The same copy in two halves -- global loads into registers, registers into LDS -- for the double-buffered band kernel: the
// loads of the NEXT 32 samples are in flight while the current 32 are evaluated.
// (SUB samples per copy: TILE / SUB rows of the block per pass, NCONST * SUB / TILE values per lane.)
template <typename T, int SUB>
__device__ __forceinline__ void stage_fetch(T (&r)[NCONST * SUB / TILE], const T* __restrict__ blk, int64_t Bp, int nsub) {
  constexpr int RP = TILE / SUB;
  static_assert(TILE % SUB == 0 && NCONST % RP == 0, "whole passes");
  const int si = threadIdx.x % SUB, i0 = threadIdx.x / SUB;
  const unsigned lane_off = ((unsigned)i0 * (unsigned)Bp + (unsigned)si) * (unsigned)sizeof(T);
  if (si < nsub) {
#pragma unroll
    for (int k = 0; k < NCONST / RP; ++k) {
      const T* row = blk + (int64_t)(RP * k) * Bp;
      r[k] = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(row) + lane_off);
    }
  }
}
template <typename T, int SUB>
__device__ __forceinline__ void stage_put(T* lds_c, const T (&r)[NCONST * SUB / TILE], int nsub) {
  constexpr int RP = TILE / SUB;
  const int si = threadIdx.x % SUB, i0 = threadIdx.x / SUB;
  if (si < nsub) {
#pragma unroll
    for (int k = 0; k < NCONST / RP; ++k) lds_c[si * NCONST + i0 + RP * k] = r[k];
  }
}

// K1: one lane per sample, float64: parameters -> band constants + atmosphere scalars.
// FAST: Newton LIDF + 8-point hot-spot panels (~1e-7 from the literal forms; only the legacy float32-columns mode),
// otherwise the reference's own LIDF iteration and 16-point panels.  The constants are computed in float64 and
// written in float32 (cstF, for the float32 band kernels) and / or float64 (cstD: float64 band kernels and the
// float64 column kernel of the default float32 mode).
struct PreludeStore {            // sample_prelude_to's sink: straight to the structure-of-arrays workspace
  float* cstF;
  double* cstD;
  double* atm;
  int64_t Bp, s;
  __device__ __forceinline__ void c(int i, double v) const {
    if (cstF) cstF[i * Bp + s] = (float)v;
    if (cstD) cstD[i * Bp + s] = v;
  }
  __device__ __forceinline__ void a(int i, double v) const {
    if (atm) atm[i * Bp + s] = v;
  }
  __device__ __forceinline__ void l(int, double) const {}
};

#ifndef SPART_PRELUDE_WAVES
#define SPART_PRELUDE_WAVES 3     // waves per SIMD the prelude is compiled for (168 VGPRs, a handful of spilled values)
#endif
// Which lane takes which sample of the workgroup's 256: the literal LIDF iteration (sailh.py:378-382) runs 12 fixed-point
// solves per sample whose pass counts differ from sample to sample -- 6 passes on average, 15 for the slowest of 64 lanes,
// and a wave pays its slowest lane.  The pass count grows with |LIDFa| + |LIDFb| (the contraction rate of the iteration), so
// the workgroup's samples are dealt to its four waves IN THE ORDER OF THAT KEY (a 32-bucket counting sort in LDS, ~60
// instructions): the slowest lane of a wave then has 10.9 passes on average (host-side emulation of the wave, 65 536 rows of
// the benchmark's distribution; 1024-sample groups would give 9.9).  A sample's arithmetic does not depend on the lane that
// runs it: results are bit-identical with and without the permutation.  Rows stay coalesced at the cache-line level (a
// workgroup still reads / writes one 2 KB segment of every row).
#ifndef SPART_PRELUDE_SORT
#define SPART_PRELUDE_SORT 1
#endif
// USER: the call carries canopy state of its own -- pp.lidf (the 12 fixed-point solves are then skipped: no sort either)
// and / or pp.nlayers; the default instantiation does not look at either.
template <bool FAST, bool USER = false>
__global__ __launch_bounds__(256, SPART_PRELUDE_WAVES) void k_prelude(ParamPtrs pp, int mask, int64_t B, int64_t Bp, float* __restrict__ cstF,
                                                 double* __restrict__ cstD, double* __restrict__ atm) {
  const int64_t base = (int64_t)blockIdx.x * blockDim.x;
  int64_t s = base + threadIdx.x;
  if (SPART_PRELUDE_SORT && !FAST && (mask & PRE_CANOPY) && !(USER && pp.lidf)) {      // (block-uniform condition)
    constexpr int NB = 32;
    __shared__ int cnt[NB], start[NB];
    __shared__ unsigned char perm[256];
    const int t = threadIdx.x;
    int bucket = NB - 1;                                       // lanes past the end of the batch: last, so that whole waves idle
    if (s < B) {
      const double key = ::fabs(pp.p[16][s]) + ::fabs(pp.p[17][s]);     // |LIDFa| + |LIDFb| (valid inputs: <= ~1.5)
      const double q = key * (NB / 1.0);
      bucket = !(q < (double)(NB - 1)) ? NB - 1 : (q > 0.0 ? (int)q : 0);   // (NaN -> last bucket)
    }
    if (t < NB) cnt[t] = 0;
    __syncthreads();
    const int pos = atomicAdd(&cnt[bucket], 1);
    __syncthreads();
    if (t == 0) {
      int acc = 0;
      for (int i = 0; i < NB; ++i) { start[i] = acc; acc += cnt[i]; }
    }
    __syncthreads();
    perm[start[bucket] + pos] = (unsigned char)t;
    __syncthreads();
    s = base + perm[t];
  }
  if (s >= B) return;
  double rho_th = pp.rho_th ? pp.rho_th[s] : 0.01;  // LeafBiology defaults (prospect_5d.py:82-83)
  double tau_th = pp.tau_th ? pp.tau_th[s] : 0.01;
  PreludeStore out{cstF, cstD, atm, Bp, s};
  if (USER)
    sample_prelude_to<FAST, true>([&pp, s](int i) { return pp.p[i] ? pp.p[i][s] : 0.0; }, rho_th, tau_th, mask, out,
                                  pp.lidf ? pp.lidf + s * NLINCL : nullptr, pp.nlayers);
  else
    sample_prelude_to<FAST>([&pp, s](int i) { return pp.p[i] ? pp.p[i][s] : 0.0; }, rho_th, tau_th, mask, out);
}

// leaf-angle distribution only (CanopyStructure.lidf, sailh.py:348)
static __global__ __launch_bounds__(256) void k_lidf(const double* __restrict__ a, const double* __restrict__ b, int64_t B,
                                              double* __restrict__ lidf) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= B) return;
  double li[NLINCL];
  leaf_angles(a[s], b[s], li);
#pragma unroll
  for (int i = 0; i < NLINCL; ++i) lidf[s * NLINCL + i] = li[i];
}

// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ BandTab<T> load_tab(const T* __restrict__ tab, int i) {
  BandTab<T> t;
  t.kab = tab[TAB_KAB * NWL + i];
  t.kca = tab[TAB_KCA * NWL + i];
  t.kdm = tab[TAB_KDM * NWL + i];
  t.kw = tab[TAB_KW * NWL + i];
  t.ks = tab[TAB_KS * NWL + i];
  t.kant = tab[TAB_KANT * NWL + i];
  t.kcbc = tab[TAB_CBC * NWL + i];
  t.kprot = tab[TAB_PROT * NWL + i];
  t.talf = tab[TAB_TALF * NWL + i];
  t.t12 = tab[TAB_T12 * NWL + i];
  t.t21 = tab[TAB_T21 * NWL + i];
  t.g0 = tab[TAB_GSV0 * NWL + i];
  t.g1 = tab[TAB_GSV1 * NWL + i];
  t.g2 = tab[TAB_GSV2 * NWL + i];
  t.cbac = tab[TAB_CBAC * NWL + i];
  t.pw = tab[TAB_PW * NWL + i];
  t.rw = tab[TAB_RW * NWL + i];
  return t;
}

template <typename T, typename C> __device__ __forceinline__ CanopyPar<T> load_canopy(const C& c) {
  CanopyPar<T> cp;
  cp.sob = c[C_SOB]; cp.sof = c[C_SOF];
  cp.hbf = c[C_HBF]; cp.ks = c[C_KS]; cp.ko = c[C_KO]; cp.lai = c[C_LAI]; cp.lai2 = c[C_LAI2];
  cp.tss = c[C_TSS]; cp.too = c[C_TOO]; cp.Z = c[C_Z]; cp.hot = c[C_HOT]; cp.pso2w = c[C_PSO2W];
  return cp;
}

// Store v at (row + byte_off) where `row` is wave-uniform and loop-invariant: SGPR base + 32-bit VGPR byte offset,
// the "saddr" form of global_store.  Keeping the per-array bases in SGPRs (instead of one 64-bit per-lane pointer
// per output array, 22 VGPRs for 11 arrays) is what lets the materialising kernel run at the occupancy of the
// columns-only one.
// Spectrum rows are written once and never read back by the kernel that writes them: non-temporal stores (the `nt` bit)
// keep them from displacing the constants / tables in the caches -- WHEN every 256-byte wave store covers whole 128-byte
// lines, i.e. when the row pitch keeps rows on the line grid (spart_ctx_set_row_pitch; the Python engine's default).
// Interleaved A/B on one box (tools/mat_ab.py, tools/prospect_bench.py): materialised band kernel 3.15 -> 3.05 ms,
// k_prospect<double> 10k 0.182 -> 0.177 ms.  With DENSE rows (8648-byte pitch: the stores straddle lines) the same bit
// defeats the write combining in L2 and costs 20 % (3.9 -> 4.8 ms), so NT is a template parameter of the storing kernels
// that the host picks from the pitch (nt_ok below; as a run-time flag its branches cost what the bit gains); the thermal
// pad (unaligned tails) always uses plain stores.
template <bool NT = false, typename T> __device__ __forceinline__ void store_row(T* row, unsigned byte_off, T v) {
  T* p = reinterpret_cast<T*>(reinterpret_cast<char*>(row) + byte_off);
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
inline bool nt_ok(int pitch_elems, size_t elem_size) { return ((size_t)pitch_elems * elem_size) % 128 == 0; }

template <typename T> struct MatPtrs {
  T *leaf_refl, *leaf_tran, *leaf_kchl, *soil_refl, *soil_dry, *rso, *rdo, *rsd, *rdd;
  const T* rdry_in; // optional (B, 2001) user dry-soil spectra (SoilParametersFromFile, bsm.py:42-43)
  int pf, po;       // row pitch (elements) of the 2162-wide / 2001-wide spectrum arrays (spart_ctx_set_row_pitch)
};

// ------------------------------------------------------------------------------------------
// K2: the fused band kernel: PROSPECT + BSM + SAILH for every band of every sample.
// grid.x = nchunk * NTILE (tile fastest), block = 256.
//
// FULL >= 1 (the default mode of spart_run_batch): every one of the 2002 band evaluations of every sample is
// observable -- each lane accumulates its band's canopy reflectances over the chunk and stores the sum: FULL = 1 one
// value per lane (rso + rdo + rsd + rdd: bandsum[chunk][2048], 8 KB per chunk), FULL = 2 the four sums separately
// (bandsum[chunk][2048][4], reduced to batch-mean spectra by k_bandmean; chosen when band_mean is requested).
// Without an observable result per band the compiler is free to drop the arithmetic (an early version sank most of the
// soil / canopy model under a conditional store and skipped it for 23 of 32 waves).  FULL = 0 exists only together with
// MAT >= 1 (prune_unused_bands + materialised spectra: the stored spectra are the observable result).
// MAT: 0 = sensor columns only; 1 = also store the requested full spectra; 2 = 1 + per-sample dry-soil spectra are
// READ (rdry_in).  The read is its own variant because a global load inside the sample loop makes the compiler wait
// for vmcnt(0) -- i.e. for every outstanding store of the previous sample -- once per sample, which serialises the
// store stream with the arithmetic.
// fp64: left alone hipcc takes 262 VGPRs = ONE wave per SIMD; asking for two workgroups per CU (256 VGPRs, 7 spilled)
// made the fp64 mode 24 % faster (50.7 -> 38.6 ms per 1M spectra); with the plate-model coefficients in constant
// memory (spart_math.h, E3c<double>) three fit with 38 spilled values: 36.2 ms (with the coefficients as literals
// three workgroups meant 96 spilled values and 104 ms).
// The sensor columns never come from this kernel: in every mode R_TOC / R_TOA / L_TOA (and the debug rsoil column) are
// produced by k_columns from the <= 2 nb bands they depend on, so that they are the SAME numbers -- by
// construction, not by compiler luck -- whether the full spectra are evaluated in float64, in float32, or not at all.
template <typename T, int MAT, int FULL, bool NT = false>
__global__ __launch_bounds__(TILE, (sizeof(T) == 8 ? 3 : 1))
void k_bands(const T* __restrict__ tab, const T* __restrict__ cst, int64_t Bp, int64_t B, int chunk, MatPtrs<T> mat,
             T* __restrict__ bandsum) {
  int tile;
  int64_t ck;
  xcd_map(blockIdx.x, tile, ck);
  if (ck * chunk >= B) return;                         // (block-uniform)
  if (sizeof(T) == 8) stage_f64_tables();              // exp / log tables of the float64 band arithmetic -> LDS
  const int band = tile * TILE + threadIdx.x;          // 0..2047
  const bool active = band < NEVAL;
  const bool thermal = band == NWL;                    // the single thermal evaluation
  const int ti = band < NWL ? band : NWL - 1;          // thermal soil = soil at 2400 nm (SPART.py:440)
  const BandTab<T> tb = load_tab(tab, ti);
  const int64_t s0 = ck * chunk;
  const int64_t s1 = (s0 + chunk < B) ? s0 + chunk : B;
  T sum_so = T(0), sum_do = T(0), sum_sd = T(0), sum_dd = T(0);
  // materialised rows: array bases of this chunk's first sample stay in SGPRs for the whole chunk; the lane carries
  // (band + row * pitch) as a 32-bit byte offset that advances by one pitch per sample (host: chunk * pitch < 2 GB)
  unsigned off_f = (unsigned)band * (unsigned)sizeof(T), off_o = off_f;
  if (MAT) {
    const int64_t b0f = s0 * mat.pf, b0o = s0 * mat.po;
    if (mat.leaf_refl) mat.leaf_refl += b0f;
    if (mat.leaf_tran) mat.leaf_tran += b0f;
    if (mat.soil_refl) mat.soil_refl += b0f;
    if (mat.rso) mat.rso += b0f;
    if (mat.rdo) mat.rdo += b0f;
    if (mat.rsd) mat.rsd += b0f;
    if (mat.rdd) mat.rdd += b0f;
    if (mat.leaf_kchl) mat.leaf_kchl += b0o;
    if (mat.soil_dry) mat.soil_dry += b0o;
  }
  // Per-sample constants are staged through LDS, 32 samples (5 KB fp32) at a time: one coalesced copy by the
  // workgroup, then every wave reads sample s's 40 values with wave-uniform ds_read_b128 (a broadcast).  They
  // land in VGPRs: on gfx950 a VALU op with an SGPR source issues ~1.6x slower than with VGPR / literal sources
  // (profiles/r1_ubench_valu_issue.txt), and ~60 ops per band use these constants -- the same kernel with
  // scalar loads (s_load_dwordx8 -> SGPR operands) is 6 % slower.
#ifndef SPART_BANDS_SUB
#define SPART_BANDS_SUB 32
#endif
  constexpr int SUB = SPART_BANDS_SUB;                 // (64 needs PINGPONG: the single-buffer copy is written for 32)
#ifndef SPART_HOIST_FILM
#define SPART_HOIST_FILM (sizeof(T) == 8)
#endif
  constexpr bool HOIST_FILM = SPART_HOIST_FILM;
  // Double-buffered staging (PINGPONG): the global loads of the next 32 samples' constants are issued BEFORE the sample loop
  // and land in LDS after it, so a workgroup meets ONE barrier per 32 samples and never waits for memory on its critical
  // path (single buffer: barrier, load, wait, barrier).  Five more VGPRs per lane (float32) while the loop runs.
#ifndef SPART_BANDS_PINGPONG
#define SPART_BANDS_PINGPONG (sizeof(T) == 4 && MAT == 0)
#endif
  constexpr bool PINGPONG = SPART_BANDS_PINGPONG;
  __shared__ __attribute__((aligned(16))) T lds_all[(PINGPONG ? 2 : 1) * SUB * NCONST];
  static_assert(PINGPONG || SUB == 32, "stage_constants copies 32 samples");
  T stg[NCONST * SUB / TILE];
  int cur = 0;
  if (PINGPONG) {
    const int n0 = (int)((s1 - s0 < SUB) ? (s1 - s0) : SUB);
    stage_fetch<T, SUB>(stg, cst + s0, Bp, n0);
    stage_put<T, SUB>(lds_all, stg, n0);
    __syncthreads();
  }
  for (int64_t sb = s0; sb < s1; sb += SUB) {
  const int nsub = (int)((s1 - sb < SUB) ? (s1 - sb) : SUB);
  T* const lds_c = lds_all + cur * (SUB * NCONST);
  const int nnext = (int)((s1 - sb - SUB < SUB) ? (s1 - sb - SUB) : SUB);      // (<= 0: this is the last sub-chunk)
  if (PINGPONG) {
    if (nnext > 0) stage_fetch<T, SUB>(stg, cst + sb + SUB, Bp, nnext);
  } else {
    __syncthreads();                                   // the previous sub-chunk has been consumed by every wave
    stage_constants<T>(lds_c, cst + sb, Bp, nsub);
    __syncthreads();
  }
  // The water film's single-layer transmittance exp2(-film2l kw) depends on the band and the film thickness only.  LUTs
  // are usually generated with ONE film thickness (BASELINE configs 3-5 fix it), so when the samples staged here all share
  // it the lane evaluates it once per 32 samples instead of once per sample (same value: the same function of the same
  // arguments); any other input takes the per-sample path below.
  bool film_same = false;
  T tw1s = T(0);
  if (HOIST_FILM) {
    const int l = threadIdx.x & 63;
    const T f0 = lds_c[C_FILM2L];
    const T fl = lds_c[(l < nsub ? l : 0) * NCONST + C_FILM2L];
    film_same = __all(fl == f0) != 0;                  // (wave-uniform; a NaN thickness compares unequal: per-sample path)
    if (film_same) tw1s = soil_tw1<T>(tb, f0);
  }
  // the nine leaf constants are read one sample ahead (they are the first thing an iteration needs: without the
  // prefetch every iteration starts by waiting for its own LDS reads).  Not in the materialising variants: there the
  // nine registers cost a wave of occupancy (109 VGPRs).
  constexpr bool AHEAD = MAT == 0 && sizeof(T) == 4;   // (float64: the 18 registers are missed elsewhere, +0.4 %)
  T lc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) lc[i] = lds_c[i];
  for (int si = 0; si < nsub; ++si) {
    const int64_t s = sb + si;
    const T* c = lds_c + si * NCONST;                  // uniform LDS address -> broadcast ds_read into VGPRs
    T refl, tran, absb, K;
    static_assert(C_CAB == 0 && C_NM1 == 8, "lc[] = constants 0..8 in leaf_band's argument order");
    if (!AHEAD) {
#pragma unroll
      for (int i = 0; i < 9; ++i) lc[i] = c[i];
    }
    leaf_band<T>(tb, lc[0], lc[1], lc[2], lc[3], lc[4], lc[5], lc[6], lc[7], lc[8], refl, tran, absb, K);
    if (AHEAD) {
      const T* cn = lds_c + ((si + 1 < nsub) ? si + 1 : si) * NCONST;   // next sample's leaf constants, in flight early
#pragma unroll
      for (int i = 0; i < 9; ++i) lc[i] = cn[i];
    }
    T rho = refl, tau = tran, ab = absb;
    if (tile == NTILE - 1) {                           // block-uniform: only the last tile holds the thermal evaluation
      rho = thermal ? c[C_RHO_TH] : refl;              // SPART.py:463-466
      tau = thermal ? c[C_TAU_TH] : tran;
      ab = thermal ? (T(1) - c[C_RHO_TH] - c[C_TAU_TH]) : absb;
    }
    // materialised spectra are stored as soon as they exist (leaf, then soil, then canopy) so that the store stream
    // is spread over the iteration instead of arriving as one burst of 9-11 stores at its end
    if (MAT && active) {
      if (mat.leaf_refl) store_row<NT>(mat.leaf_refl, off_f, rho);
      if (mat.leaf_tran) store_row<NT>(mat.leaf_tran, off_f, tau);
      if (!thermal && mat.leaf_kchl) store_row<NT>(mat.leaf_kchl, off_o, (K > T(0)) ? divx(c[C_CAB] * tb.kab, K) : T(0));  // prospect_5d.py:197-198
    }
    // order: leaf -> canopy solve for the leaf alone -> soil -> coupling with the soil background.  The soil model
    // sits between the two canopy parts because that schedule measured fastest (interleaved A/B of four orders).
    const CanopyPar<T> cp = load_canopy<T>(c);
    const CanopyCore<T> core = canopy_core<T>(cp, rho, tau, ab);
    T rdry = (MAT == 2) ? mat.rdry_in[s * mat.po + ti] : soil_dry<T>(tb, c[C_F1], c[C_F2], c[C_F3]);
    T fm[7] = {c[C_FM0], c[C_FM1], c[C_FM2], c[C_FM3], c[C_FM4], c[C_FM5], c[C_FM6]};
    T rwet;
    if (HOIST_FILM) {
      const T tw1 = film_same ? tw1s : soil_tw1<T>(tb, c[C_FILM2L]);      // (wave-uniform choice)
      soil_band_tw<T>(tb, rdry, c[C_WET], fm, c[C_FMSUM], tw1, rwet);
    } else {
      soil_band<T>(tb, rdry, c[C_WET], fm, c[C_FMSUM], c[C_FILM2L], rwet);
    }
    if (MAT && active) {
      if (mat.soil_refl) store_row<NT>(mat.soil_refl, off_f, rwet);
      if (!thermal && mat.soil_dry) store_row<NT>(mat.soil_dry, off_o, rdry);
    }
    T rso, rdo, rsd, rdd;
    canopy_soil<T>(cp, core, rwet, rso, rdo, rsd, rdd);
    if (FULL == 2) {
      sum_so += rso; sum_do += rdo; sum_sd += rsd; sum_dd += rdd;
    } else if (FULL == 1) {
      sum_so += (rso + rdo) + (rsd + rdd);
    }
    if (MAT) {
      if (active) {
        if (mat.rso) store_row<NT>(mat.rso, off_f, rso);
        if (mat.rdo) store_row<NT>(mat.rdo, off_f, rdo);
        if (mat.rsd) store_row<NT>(mat.rsd, off_f, rsd);
        if (mat.rdd) store_row<NT>(mat.rdd, off_f, rdd);
      }
      // thermal padding (SPART.py:427-470): the wave that holds the thermal evaluation (band 2001) copies it over
      // bands 2002..2161 of the padded spectra -- 160 values per array, three coalesced stores per lane
      constexpr int TH_WAVE = (NWL % TILE) / 64, TH_LANE = (NWL % TILE) % 64;
      if (tile == NTILE - 1 && (int)(threadIdx.x >> 6) == TH_WAVE) {
        // this lane's band is 1792 + 192 + l; the pad starts at band 2002 = this wave's row offset + (TH_LANE + 1 + l)
        const unsigned pad = off_f + (unsigned)((TH_LANE + 1) * (int)sizeof(T));
        T* arrs[7] = {mat.leaf_refl, mat.leaf_tran, mat.soil_refl, mat.rso, mat.rdo, mat.rsd, mat.rdd};
        const T vals[7] = {rho, tau, rwet, rso, rdo, rsd, rdd};
        const int l = threadIdx.x & 63;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const T v = __shfl(vals[q], TH_LANE, 64);
          if (arrs[q]) {
            store_row(arrs[q], pad, v);
            store_row(arrs[q], pad + 64u * (unsigned)sizeof(T), v);
            if (l < NWLT - 1 - 128) store_row(arrs[q], pad + 128u * (unsigned)sizeof(T), v);
          }
        }
      }
      off_f += (unsigned)mat.pf * (unsigned)sizeof(T);
      off_o += (unsigned)mat.po * (unsigned)sizeof(T);
    }
  }
  if (PINGPONG) {
    if (nnext > 0) stage_put<T, SUB>(lds_all + (cur ^ 1) * (SUB * NCONST), stg, nnext);
    __syncthreads();            // everyone is done with this buffer (it is refilled one sub-chunk from now) and sees the other
    cur ^= 1;
  }
  }
  if (FULL == 2) {
    T* bs = bandsum + (ck * (NTILE * TILE) + band) * 4;
    bs[0] = sum_so; bs[1] = sum_do; bs[2] = sum_sd; bs[3] = sum_dd;
  } else if (FULL == 1) {
    bandsum[ck * (NTILE * TILE) + band] = sum_so;
  }
}

// ------------------------------------------------------------------------------------------
// K3: THE COLUMN KERNEL.  R_TOC / R_TOA / L_TOA (and the debug rsoil column, La) of every sample, from the <= 2 nb
// spectral bands they depend on: for each sensor band the canopy model (PROSPECT + BSM + SAILH, the arithmetic of
// leaf_band / soil_band / canopy_core / canopy_soil that k_bands runs for all 2162 bands) is evaluated at the np.interp
// support points of the band centre (SPART.py:220-223: one grid point for the integer centres of Sentinel-2 / Landsat 4,
// 5, 8, two for the fractional centres of MODIS / OLCI / Landsat 7), interpolated, and taken through SMAC (smac.py) and
// TOC -> TOA (SPART.py:243-252) IN THE SAME WAVE: the four canopy reflectances never leave registers.  (Rounds 1-4 ran
// this as two kernels -- a sensor-slot pass writing a (nslot, 4, B) float64 array, a sensor kernel reading it back:
// 0.83 GB of the pruned step's 2.16 GB per 1M spectra and a launch.)
// In EVERY mode of spart_run_batch the columns come from this kernel, so they are the SAME numbers -- by construction --
// whether the full spectra are evaluated beside it in float64, in float32, or not at all (prune_unused_bands).
//   TG: arithmetic of the canopy model (double, except spart_materialize.f32_columns); SMAC and TOC -> TOA are float64
//   TO: dtype of the (B, nb) outputs;  TR: element type of the optional user dry-soil spectra (the call's dtype).
// Mapping: a workgroup owns 64 consecutive samples (lane = sample); its 36 band constants and 13 atmosphere scalars
// are copied into LDS once ([row][64]: lane l reads word l of a row, conflict-free) and wave w walks the sensor bands
// j = w, w + 4, ...: the band index -- and with it the 17 table values, the 48 SMAC coefficients, the interpolation
// support and the "is this gas absent in this band" tests -- is wave-uniform (scalar loads, scalar branches).
template <typename T> struct LdsCol {
  const T* p;
  __device__ __forceinline__ T operator[](int i) const { return p[i * 64]; }
};

// band0 / band1: (nb) evaluation index (0..2000 = 400..2400 nm, 2001 = the thermal evaluation) of the grid point at / below
// the band centre, and of the next grid point (== band0 where the centre sits on a grid point, frac == 0).  The sensor
// tables are separate __restrict__ arguments (a pointer inside a by-value struct cannot be declared noalias).

constexpr int COL_WAVES = 4;   // (more waves per 64-sample workgroup measured slower for both former kernels)
#ifndef SPART_COLUMNS_WAVES
#define SPART_COLUMNS_WAVES 4  // waves per SIMD the column kernel is compiled for (128 VGPRs)
#endif
// Results leave straight from the lane that computed them (stride nb between lanes: the (64 x nb) block of a workgroup is
// completed by its four waves within a few hundred cycles and merges in L2) instead of through an LDS transpose: the 10 KB
// of staging were what limited the kernel to three workgroups per CU.
#ifndef SPART_COLUMNS_DIRECT
#define SPART_COLUMNS_DIRECT 1
#endif

template <typename TG, typename TO, typename TR>
__global__ __launch_bounds__(64 * COL_WAVES, SPART_COLUMNS_WAVES) void k_columns(
    const TG* __restrict__ tab, const TG* __restrict__ cst, const double* __restrict__ atm, int64_t Bp,
    const int* __restrict__ band0, const int* __restrict__ band1, const double* __restrict__ frac,
    const double* __restrict__ coef, const double* __restrict__ econv, int nb, const TR* __restrict__ rdry_in, int po, int64_t B,
    TO* __restrict__ R_TOC, TO* __restrict__ R_TOA, TO* __restrict__ L_TOA, TO* __restrict__ o_rsoil, TO* __restrict__ o_La) {
  constexpr bool DIRECT = SPART_COLUMNS_DIRECT != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  TO* stage = reinterpret_cast<TO*>(smem_raw);          // [narr][64 * nb]  (not used with DIRECT)
  __shared__ TG lds_c[NCONST_USED * 64];
  __shared__ double lds_a[NATM_USED * 64];
  __shared__ double lds_k[COL_WAVES * 64];              // per wave: the 48 SMAC coefficients of the band it is working on
  if (sizeof(TG) == 8 || !SPART_SMAC_LIBM) stage_f64_tables();   // exp / log tables of the float64 arithmetic
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  const int64_t s = s0 + lane;
  const bool ok = s < B;
  const int64_t sc = ok ? s : B - 1;                    // (lanes past the end repeat the last sample and store nothing)
  for (int i = wave; i < NCONST_USED; i += COL_WAVES) lds_c[i * 64 + lane] = cst[(int64_t)i * Bp + sc];
  for (int i = wave; i < NATM_USED; i += COL_WAVES) lds_a[i * 64 + lane] = atm[(int64_t)i * Bp + sc];
  __syncthreads();
  const LdsCol<TG> c{lds_c + lane};
  const LdsCol<double> a{lds_a + lane};
  const int tile = 64 * nb;
  const bool want_soil = o_rsoil != nullptr;
  // (starting wave w at band (w + workgroup) mod 4, so that the four-band wave of a 13-band sensor moves from SIMD to SIMD,
  //  changed nothing: 0.464 / 0.466 ms, profiles/r5_ab_columns.txt)
  for (int j = wave; j < nb; j += COL_WAVES) {
    const int b0 = band0[j], b1 = band1[j];
    const double f = frac[j];
    // this band's 48 SMAC coefficients: ONE vector load (lane r fetches row r), issued before the canopy model and parked in
    // the wave's LDS row behind it; smac_band then reads each with a wave-uniform (broadcast) ds_read.  Left to the
    // compiler they were ~50 dependent global loads per band, each followed by its own s_waitcnt vmcnt(0).
    const double kv = coef[(size_t)(lane < NCOEF ? lane : NCOEF - 1) * nb + j];
    double y0[5], y1[5];                                // rso, rdo, rsd, rdd, wet soil at the two support points
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {
      if (e == 1 && b1 == b0) break;                    // (wave-uniform)
      const int band = e ? b1 : b0;
      const bool thermal = band == NWL;
      const int ti = band < NWL ? band : NWL - 1;       // thermal soil = soil at 2400 nm (SPART.py:440)
      const BandTab<TG> tb = load_tab(tab, ti);
      TG refl, tran, absb, K;
      leaf_band<TG>(tb, c[C_CAB], c[C_CCA], c[C_CDM], c[C_CW], c[C_CS], c[C_CANT], c[C_CBC], c[C_PROT], c[C_NM1], refl,
                    tran, absb, K);
      const TG rho = thermal ? c[C_RHO_TH] : refl;      // SPART.py:463-466
      const TG tau = thermal ? c[C_TAU_TH] : tran;
      const TG ab = thermal ? (TG(1) - c[C_RHO_TH] - c[C_TAU_TH]) : absb;
      const CanopyPar<TG> cp = load_canopy<TG>(c);      // (same order of the parts as in k_bands)
      const CanopyCore<TG> core = canopy_core<TG>(cp, rho, tau, ab);
      const TG rdry = rdry_in ? (TG)rdry_in[sc * po + ti] : soil_dry<TG>(tb, c[C_F1], c[C_F2], c[C_F3]);
      TG fm[7] = {c[C_FM0], c[C_FM1], c[C_FM2], c[C_FM3], c[C_FM4], c[C_FM5], c[C_FM6]};
      TG rwet;
      soil_band<TG>(tb, rdry, c[C_WET], fm, c[C_FMSUM], c[C_FILM2L], rwet);
      TG rso, rdo, rsd, rdd;
      canopy_soil<TG>(cp, core, rwet, rso, rdo, rsd, rdd);
      const double v[5] = {(double)rso, (double)rdo, (double)rsd, (double)rdd, (double)rwet};
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        if (e == 0) y0[q] = v[q];
        y1[q] = v[q];                                   // (one support point: y1 = y0, as np.interp sees it)
      }
    }
    double v[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) v[q] = y0[q] + (y1[q] - y0[q]) * f;    // np.interp (SPART.py:220-223)
    double* kw = lds_k + wave * 64;
    kw[lane] = kv;
    // the reads below are other lanes' words: a wavefront-scope release fence + wave barrier make the store visible to the
    // wave under the HIP memory model (neither emits an instruction: LDS serves one wave's accesses in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const SmacOut so = smac_band_c(a, [kw](int r) { return kw[r]; });
    const double La = a[A_LAF] * econv[j];               // SPART.py:353, 394
    double rtoc, rtoa, ltoa;
    toc_to_toa(so, v[0], v[1], v[3], v[2], La, rtoc, rtoa, ltoa);
    if (DIRECT) {
      if (ok) {
        const int64_t o = s * nb + j;
        R_TOC[o] = (TO)rtoc;
        R_TOA[o] = (TO)rtoa;
        L_TOA[o] = (TO)ltoa;
        if (want_soil) o_rsoil[o] = (TO)v[4];           // SPART.py:262-267
        if (o_La) o_La[o] = (TO)La;
      }
    } else {
      const int o = lane * nb + j;
      stage[o] = (TO)rtoc;
      stage[tile + o] = (TO)rtoa;
      stage[2 * tile + o] = (TO)ltoa;
      int na = 3;
      if (want_soil) {
        stage[na * tile + o] = (TO)v[4];
        ++na;
      }
      if (o_La) stage[na * tile + o] = (TO)La;
    }
  }
  if (DIRECT) return;
  __syncthreads();
  const int64_t rem = B - s0;
  const int n = (int)((rem < 64 ? rem : 64) * nb);
  const int64_t base = s0 * nb;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    R_TOC[base + i] = stage[i];
    R_TOA[base + i] = stage[tile + i];
    L_TOA[base + i] = stage[2 * tile + i];
    int na = 3;
    if (want_soil) {
      o_rsoil[base + i] = stage[na * tile + i];
      ++na;
    }
    if (o_La) o_La[base + i] = stage[na * tile + i];
  }
}

// batch-mean canopy spectra from the per-chunk band sums: out (4, 2162) = mean over samples of rso, rdo, rsd, rdd
template <typename T>
__global__ __launch_bounds__(256) void k_bandmean(const T* __restrict__ bandsum, int64_t nchunk, int64_t B,
                                                  T* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;   // 0 .. 4*2162-1
  if (i >= 4 * NWLS) return;
  int q = i / NWLS, band = i % NWLS;
  int ev = band < NWL ? band : NWL;                // thermal bands share evaluation 2001
  double acc = 0.0;
  for (int64_t c = 0; c < nchunk; ++c) acc += (double)bandsum[(c * (NTILE * TILE) + ev) * 4 + q];
  out[i] = (T)(acc / (double)B);
}

// ------------------------------------------------------------------------------------------
// Walk samples s0..s1 with their 40 constants staged through LDS, 32 samples per coalesced workgroup copy (the
// stage-level kernels below; k_bands has the same loop written out).  body(s, c): c points at sample s's constants
// in LDS (wave-uniform address -> broadcast reads).  Without the staging every sample starts with a scalar load
// from global memory whose latency nothing hides: k_bsm, 55 VALU instructions per band, was bound by exactly that.
// STAGE = false is the plain loop (constants through scalar loads): measured per kernel and type, staging wins for
// k_prospect<float> (7.1 -> 5.4 ms per 1M leaves) and k_bsm (8.2 -> 4.6 ms fp32, 14.2 -> 10.2 fp64), and loses 6-12 %
// for k_prospect<double> (float64 ops take SGPR operands at no extra cost, the extra VGPRs are not free) and k_sailh
// (its per-band input loads already overlap the scalar loads).
template <typename T, bool STAGE, typename F>
__device__ __forceinline__ void for_samples_staged(const T* __restrict__ cst, int64_t Bp, int64_t s0, int64_t s1, F&& body) {
  if (!STAGE) {
    for (int64_t s = s0; s < s1; ++s) body(s, ConstCol<T>{cst + s, Bp});
    return;
  }
  constexpr int SUB = 32;
  __shared__ __attribute__((aligned(16))) T lds_c[SUB * NCONST];
  for (int64_t sb = s0; sb < s1; sb += SUB) {
    const int nsub = (int)((s1 - sb < SUB) ? (s1 - sb) : SUB);
    __syncthreads();                                   // the previous sub-chunk has been consumed by every wave
    stage_constants<T>(lds_c, cst + sb, Bp, nsub);
    __syncthreads();
    for (int si = 0; si < nsub; ++si) body(sb + si, (const T*)(lds_c + si * NCONST));
  }
}

// ------------------------------------------------------------------------------------------
// standalone PROSPECT-5D / PRO: (B,2001) spectra out
template <typename T, bool NT>
__global__ __launch_bounds__(TILE) void k_prospect(const T* __restrict__ tab, const T* __restrict__ cst, int64_t Bp, int64_t B,
                                                   int chunk, int po, T* __restrict__ o_refl,
                                                   T* __restrict__ o_tran, T* __restrict__ o_kchl) {
  int tile;
  int64_t ck;
  xcd_map(blockIdx.x, tile, ck);
  if (ck * chunk >= B) return;
  if (sizeof(T) == 8) stage_f64_tables();
  const int band = tile * TILE + threadIdx.x;
  const bool active = band < NWL;
  const BandTab<T> tb = load_tab(tab, active ? band : NWL - 1);
  const int64_t s0 = ck * chunk;
  const int64_t s1 = (s0 + chunk < B) ? s0 + chunk : B;
  // SGPR row bases of the chunk + one 32-bit per-lane offset that advances by a pitch per sample (see k_bands)
  unsigned off = (unsigned)band * (unsigned)sizeof(T);
  if (o_refl) o_refl += s0 * po;
  if (o_tran) o_tran += s0 * po;
  if (o_kchl) o_kchl += s0 * po;
  for_samples_staged<T, sizeof(T) == 4>(cst, Bp, s0, s1, [&](int64_t, auto c) {
    T refl, tran, absb, K;
#ifndef SPART_EXPERIMENT
    leaf_band<T>(tb, c[C_CAB], c[C_CCA], c[C_CDM], c[C_CW], c[C_CS], c[C_CANT], c[C_CBC], c[C_PROT], c[C_NM1], refl,
                 tran, absb, K);
    if (active) {
#else
    // MEASUREMENT VARIANTS, never part of a product build (tools/prospect_split.sh builds them with
    // build.py's `extra` flags, which are hashed into spart_build_id; nothing else defines the macro):
    // SPART_EXPERIMENT = 1 arithmetic only (the stores sit behind a test no value passes), = 2 stores only (placeholder
    // arithmetic -- NOT leaf spectra).  profiles/r4_prospect_split.txt is what they measured.
#if SPART_EXPERIMENT == 2
    refl = c[C_CAB] * tb.kab; tran = c[C_CW] * tb.kw; K = c[C_CDM] + tb.kdm; absb = 0;
#else
    leaf_band<T>(tb, c[C_CAB], c[C_CCA], c[C_CDM], c[C_CW], c[C_CS], c[C_CANT], c[C_CBC], c[C_PROT], c[C_NM1], refl,
                 tran, absb, K);
#endif
    if (active && (SPART_EXPERIMENT != 1 || refl + tran + K == T(-12345.678))) {
#endif
      if (o_refl) store_row<NT>(o_refl, off, refl);
      if (o_tran) store_row<NT>(o_tran, off, tran);
      if (o_kchl) store_row<NT>(o_kchl, off, (K > T(0)) ? divx(c[C_CAB] * tb.kab, K) : T(0));
    }
    off += (unsigned)po * (unsigned)sizeof(T);
  });
}

// standalone BSM: (B,2001) wet and dry soil spectra; optional user dry spectra (bsm.py:42-43)
template <typename T, bool NT>
__global__ __launch_bounds__(TILE) void k_bsm(const T* __restrict__ tab, const T* __restrict__ cst, int64_t Bp, int64_t B,
                                              int chunk, int po, const T* __restrict__ rdry_in,
                                              T* __restrict__ o_refl, T* __restrict__ o_dry) {
  int tile;
  int64_t ck;
  xcd_map(blockIdx.x, tile, ck);
  if (ck * chunk >= B) return;
  if (sizeof(T) == 8) stage_f64_tables();
  const int band = tile * TILE + threadIdx.x;
  const bool active = band < NWL;
  const BandTab<T> tb = load_tab(tab, active ? band : NWL - 1);
  const int64_t s0 = ck * chunk;
  const int64_t s1 = (s0 + chunk < B) ? s0 + chunk : B;
  unsigned off = (unsigned)band * (unsigned)sizeof(T);
  if (o_refl) o_refl += s0 * po;
  if (o_dry) o_dry += s0 * po;
  for_samples_staged<T, true>(cst, Bp, s0, s1, [&](int64_t s, auto c) {
    T rdry = rdry_in ? (active ? rdry_in[s * po + band] : T(0)) : soil_dry<T>(tb, c[C_F1], c[C_F2], c[C_F3]);
    T fm[7] = {c[C_FM0], c[C_FM1], c[C_FM2], c[C_FM3], c[C_FM4], c[C_FM5], c[C_FM6]};
    T rwet;
    soil_band<T>(tb, rdry, c[C_WET], fm, c[C_FMSUM], c[C_FILM2L], rwet);
    if (active) {
      if (o_refl) store_row<NT>(o_refl, off, rwet);
      if (o_dry) store_row<NT>(o_dry, off, rdry);
    }
    off += (unsigned)po * (unsigned)sizeof(T);
  });
}

// standalone SAILH: leaf / soil spectra in, four canopy reflectance spectra out, all (B,2162)
template <typename T, bool NT>
__global__ __launch_bounds__(TILE) void k_sailh(const T* __restrict__ cst, int64_t Bp, int64_t B, int chunk, int pf,
                                                const T* __restrict__ i_rho, const T* __restrict__ i_tau,
                                                const T* __restrict__ i_rs, T* __restrict__ o_rso,
                                                T* __restrict__ o_rdo, T* __restrict__ o_rsd, T* __restrict__ o_rdd) {
  const int tile = blockIdx.x % NTILE_FULL;
  const int64_t ck = blockIdx.x / NTILE_FULL;
  if (sizeof(T) == 8) stage_f64_tables();
  const int band = tile * TILE + threadIdx.x;
  const bool active = band < NWLS;
  const int64_t s0 = ck * chunk;
  const int64_t s1 = (s0 + chunk < B) ? s0 + chunk : B;
  // SGPR row bases of the chunk + one 32-bit per-lane offset (loads and stores alike)
  unsigned off = (unsigned)(active ? band : 0) * (unsigned)sizeof(T);
  i_rho += s0 * pf; i_tau += s0 * pf; i_rs += s0 * pf;
  o_rso += s0 * pf; o_rdo += s0 * pf; o_rsd += s0 * pf; o_rdd += s0 * pf;
  for_samples_staged<T, false>(cst, Bp, s0, s1, [&](int64_t, auto c) {
    const T rho = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(i_rho) + off);
    const T tau = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(i_tau) + off);
    const T rs = *reinterpret_cast<const T*>(reinterpret_cast<const char*>(i_rs) + off);
    const CanopyPar<T> cp = load_canopy<T>(c);
    T rso, rdo, rsd, rdd;
    canopy_band<T>(cp, rho, tau, T(1) - rho - tau, rs, rso, rdo, rsd, rdd);
    if (active) {
      store_row<NT>(o_rso, off, rso); store_row<NT>(o_rdo, off, rdo); store_row<NT>(o_rsd, off, rsd); store_row<NT>(o_rdd, off, rdd);
    }
    off += (unsigned)pf * (unsigned)sizeof(T);
  });
}

// ------------------------------------------------------------------------------------------
// standalone SMAC: nine (B,nb) float64 outputs
struct Out9 {
  double* o[9];
};
static __global__ __launch_bounds__(256) void k_smac(const double* __restrict__ coef, int nb, const double* __restrict__ atm,
                                              int64_t Bp, int64_t B, Out9 out) {
  if (!SPART_SMAC_LIBM) stage_f64_tables();             // smac_band's table exp
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nb) return;
  int64_t s = i / nb;
  int j = (int)(i - s * nb);
  double a[NATM];
#pragma unroll
  for (int q = 0; q < NATM_USED; ++q) a[q] = atm[q * Bp + s];      // (the rows the prelude writes)
  SmacOut so = smac_band(a, coef + j, nb);
  out.o[0][i] = so.Ta_s; out.o[1][i] = so.Ta_o; out.o[2][i] = so.Tg; out.o[3][i] = so.Ra_dd; out.o[4][i] = so.Ra_so;
  out.o[5][i] = so.Ta_ss; out.o[6][i] = so.Ta_sd; out.o[7][i] = so.Ta_oo; out.o[8][i] = so.Ta_do;
}

// ------------------------------------------------------------------------------------------
// Context-creation kernel: SRF convolution of the extraterrestrial irradiance
// (calculate_spectral_convolution, SPART.py:358-396).  One wave per sensor band; lanes stride
// over the SRF samples, then a 64-lane shuffle reduction of sum(Ea[idx] p) and sum(p).
static __global__ __launch_bounds__(64) void k_econv(const double* __restrict__ Ea, const double* __restrict__ wl_srf,
                                              const double* __restrict__ p_srf, int nsrf, int nb,
                                              double* __restrict__ econv) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  double num = 0.0, den = 0.0;
  for (int i = lane; i < nsrf; i += 64) {
    double w = wl_srf[(size_t)i * nb + b];
    double p = p_srf[(size_t)i * nb + b];
    // nearest grid wavelength, ties to the lower one; NaN -> index 0 (numpy argmin over NaNs), SPART.py:381-387
    int idx = 0;
    if (w == w) {
      double r = ::ceil(w - 0.5) - 400.0;
      r = r < 0.0 ? 0.0 : (r > (double)(NWL - 1) ? (double)(NWL - 1) : r);
      idx = (int)r;
    }
    num += Ea[idx] * p;
    den += p;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    num += __shfl_down(num, off, 64);
    den += __shfl_down(den, off, 64);
  }
  if (lane == 0) econv[b] = num / den;
}

}  // namespace spart

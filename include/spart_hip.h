/* libspart_hip.so -- C ABI of the MI355X-native batched SPART evaluator.
 *
 * The reference (wirrell/SPART-python) has no FFI layer: its hot path is the five plain
 * Python callables
 *     BSM(soilpar, optical_params)                  src/SPART/bsm.py:17
 *     PROSPECT_5D(leafbio, optical_params)          src/SPART/prospect_5d.py:117
 *     SAILH(soil, leafopt, canopy, angles)          src/SPART/sailh.py:14
 *     SMAC(angles, atm, coefs)                      src/SPART/smac.py:14
 *     SPART(...).run()                              src/SPART/SPART.py:162
 * Each entry point below replaces the arithmetic of one of them for a BATCH of B samples;
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every array pointer is DEVICE memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()); the library allocates only its context and never frees caller memory;
 *   - per-sample inputs are structure-of-arrays, float64, length B (sample-level geometry is
 *     always computed in float64);
 *   - spectra and result columns are row-major (B, nbands), band-contiguous, in `dtype`; the row pitch of the
 *     2162- / 2001-wide spectrum arrays is the row width unless spart_ctx_set_row_pitch says otherwise;
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on it (spart_run_batch may run some
 *     of its kernels on a side stream the context owns -- one per caller stream, for the first 32 caller streams a context
 *     sees; calls on further streams run every kernel on `stream` itself: same results, no overlap -- and joins them back
 *     into `stream` before it returns, so the caller sees plain stream semantics, HIP-graph capture included);
 *   - a context is THREAD-SAFE: its tables are immutable after creation and the little per-call state it has (side
 *     streams, events) is guarded, so any number of host threads may call into one context on any streams.  Calls
 *     that run concurrently on the GPU (different streams) need different workspaces; if two streams do pass the
 *     same workspace, the later call is ordered after the earlier one (hipStreamWaitEvent on its completion) -- slow,
 *     never a race, for any number of (workspace, stream) pairs: the context remembers every use until it has seen
 *     it complete -- or fails with SPART_ERR_INVALID when that order cannot be expressed (e.g. across a stream
 *     capture).  HIP-graph REPLAYS are outside the library's view: a captured call's workspace belongs to its graph.  The
 *     first call on a stream (and the first use of a workspace on it) creates that stream's side stream / events: issue
 *     one ordinary call on the stream before capturing it, so that the capture itself creates nothing;
 *   - return value 0 = ok, <0 = error (spart_last_error gives the text).  Numerical trouble
 *     propagates as NaN/inf exactly like the reference (no clamping).
 */
#ifndef SPART_HIP_H
#define SPART_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPART_NWL 2001      /* 400..2400 nm            (SPART.py:303)      */
#define SPART_NWLS 2162     /* + 161 thermal bands      (SPART.py:307-310)  */
#define SPART_NLINCL 13     /* leaf inclination classes (sailh.py:346)      */
#define SPART_NPARAM 27     /* parameter columns of spart_run_batch         */
#define SPART_NCOEF 48      /* SMAC coefficient rows    (smac.py:44-92)     */

#define SPART_NLAYERS 60    /* canopy layers, CanopyStructure's default (sailh.py:345) */
#define SPART_MAX_NLAYERS 1000000

/* Version of THIS interface (struct layouts, argument lists, stage count): spart_abi_version() of a loaded library must
 * equal the header's a binding was written against (the Python loader checks, so that a library built from another
 * round's header -- e.g. through SPART_HIP_LIB -- is refused instead of being called with shifted arguments).
 *   6: spart_materialize.lidf_in / .nlayers, spart_sailh_batch(lidf_in, nlayers), spart_abi_version itself */
#define SPART_ABI_VERSION 6

#define SPART_F32 0
#define SPART_F64 1

#define SPART_OK 0
#define SPART_ERR_INVALID (-1)    /* bad argument                          */
#define SPART_ERR_HIP (-2)        /* HIP runtime error                     */
#define SPART_ERR_WORKSPACE (-3)  /* workspace missing or too small        */
#define SPART_ERR_NOSENSOR (-4)   /* context was created without a sensor  */

typedef struct spart_ctx spart_ctx;

/* Static tables, HOST pointers, float64; copied to the device by spart_ctx_create.
 * Spectral tables are the arrays of the reference's optical_params.pkl / ET_irradiance.pkl
 * (SPART.py:399-416), length 2001; the sensor block is sensor_information/<sensor>.pkl
 * (SPART.py:419-424): keys wl_smac, SMAC_coef, wl_srf_smac, p_srf_smac.  nb = 0 builds a
 * context without sensor (leaf / soil / canopy entry points only). */
typedef struct spart_tables {
  const double *nr, *Kab, *Kca, *Kdm, *Kw, *Ks, *Kant, *cbc, *prot; /* prospect_5d.py:158-167 */
  const double *GSV;                                                /* (2001,3) row-major, bsm.py:45 */
  const double *nw;                                                 /* bsm.py:55 */
  const double *Ea;                                                 /* SPART.py:181 */
  int32_t nb;                                                       /* sensor bands */
  const double *wl_smac;                                            /* (nb,) band centres, nm */
  const double *coef;                                               /* (48, nb) row-major, rows in smac.py:44-92 order
                                                                       (ah2o nh2o ao3 no3 ao2 no2 po2 aco2 nco2 pco2 ach4 nch4 pch4
                                                                       ano2 nno2 pno2 aco nco pco a0s a1s a2s a3s a0T a1T a2T a3T taur
                                                                       a0taup a1taup wo gc a0P a1P a2P a3P a4P Rest1..4 Resr1..3 Resa1..4) */
  int32_t nsrf;                                                     /* SRF samples per band */
  const double *wl_srf;                                             /* (nsrf, nb) row-major, NaN padded */
  const double *p_srf;                                              /* (nsrf, nb) row-major */
} spart_tables;

/* Optional full-spectrum outputs of spart_run_batch (the reference object's soilopt /
 * leafopt / canopyopt attributes, SPART.py:66-81) and evaluation options.  NULL members are skipped;
 * passing opt = NULL means: columns only, all bands evaluated. */
typedef struct spart_materialize {
  void *leaf_refl, *leaf_tran; /* (B,2162) thermal-padded, SPART.py:445-470 */
  void *leaf_kchl;             /* (B,2001) kChlrel, prospect_5d.py:197-198   */
  void *soil_refl;             /* (B,2162) wet soil, padded SPART.py:427-442 */
  void *soil_refl_dry;         /* (B,2001)                                    */
  void *rso, *rdo, *rsd, *rdd; /* (B,2162) sailh.py:224-233                  */
  void *rsoil;                 /* (B,nb) debug column, SPART.py:262-267      */
  void *La;                    /* (B,nb) convolved ET radiance, SPART.py:183 */
  const void *rdry_in;         /* INPUT, optional: (B,2001) user dry-soil spectra in `dtype` (SoilParametersFromFile,
                                  bsm.py:42-43, 155-199); params[9..11] (B, lat, lon) may then be NULL */
  void *band_mean;             /* (4,2162) batch means of rso, rdo, rsd, rdd (LUT summary; no reference counterpart) */
  int32_t prune_unused_bands;  /* R_TOC / R_TOA / L_TOA (and rsoil) ALWAYS come from the <= 2 nb spectral bands they depend on
                                  (np.interp support points, SPART.py:220-223): prelude -> column kernel.
                                  0 (default): beside that, every one of the 2162 bands of every sample is evaluated by the
                                  fused full-band kernel (band sums / band_mean / the spectra requested above); 1: that
                                  kernel only runs for requested spectra -- identical columns (the same kernels produce
                                  them), the work is not "full spectra" */
  int32_t f32_columns;         /* dtype SPART_F32 only.  0 (default): the column path (prelude constants, the canopy model at
                                  the sensor bands, SMAC, TOC->TOA) is float64 whatever the dtype, so R_TOC / R_TOA / L_TOA are the float64
                                  mode's values rounded once to float32 (the 1e-4 contract then also holds for nearly
                                  conservative PROSPECT-PRO leaves, where the reference's canopy formulas cancel,
                                  sailh.py:185-214); materialised spectra and band_mean are float32 arithmetic.
                                  1: the canopy model at the sensor bands is evaluated in float32 as well (fast prelude) */
  int32_t f32_bands;           /* dtype SPART_F64 only.  1: R_TOC / R_TOA / L_TOA (and rsoil, La) are float64 and IDENTICAL
                                  to the float64 mode's -- they come from the same float64 column path -- while the
                                  evaluation of all 2162 bands of every sample (the band sums) runs in float32.
                                  Materialised spectra, band_mean and rdry_in cannot be combined with it (SPART_ERR_INVALID). */
  int32_t fast_prelude;        /* 0 (default): the reference's own LIDF fixed-point iteration with its |dx| <= 1e-8 stopping rule
                                  (sailh.py:378-382) and 10-point hot-spot panels -- the columns agree with the reference to
                                  ~1e-11.  1: the ROOT of the same equation by Newton (the reference stops up to ~5e-8 short
                                  of it) and 8-point panels; the per-sample prelude kernel takes 0.41 instead of 0.69 ms
                                  per 1M spectra (round 5; 0.45 / 0.93 when this option was introduced).  It is a different function at the 1e-8 level.  Measured over all 2 x 13M
                                  float64 column entries of BASELINE configs 4 and 5 (1M rows each; bench.py
                                  configs.fast_prelude.float64_columns_vs_default and test_fast_prelude_at_size repeat the
                                  measurement): on SURVEY 8(d)'s metric |d| / max(|ref|, 1e-6) the median is 2.5e-10, 99.999 % of
                                  the entries of R_TOC / R_TOA / L_TOA are within 1.5e-7, and 15 + 27 of the 2 x 13M R_TOC
                                  entries exceed 1e-6 (maximum 1.7e-5) -- every one of them an entry whose own magnitude is
                                  below 1e-3 (strongly absorbing bands, |d| < 1e-9).  So: inside the float32 contract (1e-4)
                                  everywhere, inside the float64 contract (1e-6) wherever the value is at least 1e-3.
                                  Implied by f32_columns */
  const double *lidf_in;       /* INPUT, optional: (B,13) float64 row-major, the leaf inclination distribution the reference's
                                  SAILH reads from canopy.lidf at call time (sailh.py:51 -> k, K, bf, sob, sof at :93-97).  NULL
                                  (default): derived from params[16..17] = LIDFa, LIDFb exactly as CanopyStructure's constructor
                                  does (sailh.py:348, 351-398).  Given: used as it is (no normalisation, like the reference's
                                  dot products), LIDFa / LIDFb are not read and may be NULL */
  int32_t nlayers;             /* canopy.nlayers (sailh.py:48): 0 = the default 60.  It enters the model only as dx = 1 / nlayers,
                                  the width of the stretch below the canopy that Pso[nlayers] averages (:131-135, 219); one value
                                  per call.  1 ... SPART_MAX_NLAYERS */
} spart_materialize;

int spart_ctx_create(spart_ctx **out, int device, const spart_tables *tables);
int spart_ctx_destroy(spart_ctx *ctx);
const char *spart_last_error(const spart_ctx *ctx); /* text of the last error raised ON THE CALLING THREAD (ctx may be NULL) */

/* Identity of this binary: 12 hex digits over the kernel / ABI sources, compiler flags and math variant it was built
 * from (spart-python_amd/build.py: source_id).  The Python loader refuses a library whose id is not that of the
 * sources next to it, and bench.py prints it, so a stale .so cannot be measured silently.  No reference counterpart. */
const char *spart_build_id(void);
int spart_abi_version(void);                              /* SPART_ABI_VERSION the library was compiled with */

int spart_ctx_nb(const spart_ctx *ctx);                    /* sensor bands of the context */
int spart_ctx_econv(const spart_ctx *ctx, double *host_out); /* (nb,) SRF-convolved ET irradiance (SPART.py:389-394), copied to HOST */

/* Row pitch, in elements, of EVERY (B,2162) and (B,2001) spectrum array this context reads or writes (the outputs /
 * inputs of spart_prospect_batch, spart_bsm_batch, spart_sailh_batch and the spart_materialize members of
 * spart_run_batch; (B,nb) columns and band_mean stay dense).  Default = dense (2162 / 2001); 0 restores it.
 * A 2162-float row is 8648 B, so dense rows start off the 128 B line grid and the 1 KB row segments a workgroup
 * stores straddle lines: on MI355X the materialised store stream then reaches 3.3 TB/s, with rows padded to a
 * multiple of 64 elements (2176 / 2048) 5.9 TB/s (tools/ubench/write_pattern.hip).  No reference counterpart. */
int spart_ctx_set_row_pitch(spart_ctx *ctx, int64_t pitch_full, int64_t pitch_optical);

/* calculate_tav (prospect_5d.py:249-311): average transmissivity of a dielectric plane surface for the cone half-angle
 * alpha_deg and n refractive indices.  HOST function on host pointers, float64, no context and no GPU: it is the routine
 * the library derives its interface tables from in spart_ctx_create, exported so that the
 * Python mirror of the reference function runs the same arithmetic. */
int spart_calculate_tav(double alpha_deg, const double *nr, int64_t n, double *out);

/* Bytes of scratch the batched entry points need for B samples (the prelude's per-sample constants, ~0.9 KB per
 * sample, + the band sums of the full-band kernel).  The same buffer may be reused by successive calls
 * on one stream.  spart_smac_batch sizes its workspace with dtype = SPART_F64. */
size_t spart_workspace_bytes(const spart_ctx *ctx, int dtype, int64_t B);

/* PROSPECT_5D (prospect_5d.py:117-246).  leaf[9] = Cab, Cdm, Cw, Cs, Cca, Cant, N, PROT, CBC.
 * Outputs (B,2001): refl, tran, kChlrel (any may be NULL). */
int spart_prospect_batch(spart_ctx *ctx, int dtype, int64_t B, const double *const leaf[9], void *refl, void *tran,
                         void *kchl, void *workspace, size_t workspace_bytes, void *stream);

/* BSM + soilwat (bsm.py:17-128).  soil[6] = B, lat, lon, SMp, SMC, film.
 * rdry_in: optional (B,2001) user dry spectra in `dtype` (the rdry_set branch, bsm.py:42-43).
 * Outputs (B,2001): refl (wet), refl_dry. */
int spart_bsm_batch(spart_ctx *ctx, int dtype, int64_t B, const double *const soil[6], const void *rdry_in,
                    void *refl, void *refl_dry, void *workspace, size_t workspace_bytes, void *stream);

/* calculate_leafangles (sailh.py:351-398): lidf (B,13) float64. */
int spart_lidf_batch(spart_ctx *ctx, int64_t B, const double *LIDFa, const double *LIDFb, double *lidf, void *stream);

/* SAILH (sailh.py:14-237).  rho, tau, rs: (B,2162) in `dtype`; canopy[4] = LAI, LIDFa, LIDFb, q;
 * angles[3] = sol, obs, rel (degrees).  out4 = rso, rdo, rsd, rdd, each (B,2162).
 * lidf_in, nlayers: the two attributes SAILH reads from the canopy OBJECT (sailh.py:48, 51), with the meaning of the
 * spart_materialize members of the same names: lidf_in optional (B,13) float64 (NULL: from LIDFa / LIDFb; given: canopy[1],
 * canopy[2] may be NULL), nlayers 0 = 60. */
int spart_sailh_batch(spart_ctx *ctx, int dtype, int64_t B, const void *rho, const void *tau, const void *rs,
                      const double *const canopy[4], const double *const angles[3], const double *lidf_in, int32_t nlayers,
                      void *const out4[4], void *workspace, size_t workspace_bytes, void *stream);

/* SMAC (smac.py:14-213).  angles[3]; atm[4] = aot550, uo3, uh2o, Pa.  out9 (B,nb) float64 in the
 * AtmosphericOptics order Ta_s, Ta_o, Tg, Ra_dd, Ra_so, Ta_ss, Ta_sd, Ta_oo, Ta_do (smac.py:209-211). */
int spart_smac_batch(spart_ctx *ctx, int64_t B, const double *const angles[3], const double *const atm[4],
                     double *const out9[9], void *workspace, size_t workspace_bytes, void *stream);

/* SPART(...).run() for a fresh object per sample (SPART.py:162-269).
 * params[27]: Cab Cdm Cw Cs Cca Cant N PROT CBC | B lat lon SMp SMC film | LAI LIDFa LIDFb q |
 *             tts tto psi | aot550 uo3 uh2o Pa | DOY          (each (B,) float64)
 * rho_thermal / tau_thermal: optional (B,) float64 (LeafBiology defaults 0.01 when NULL).
 * Outputs (B,nb) in `dtype`. */
int spart_run_batch(spart_ctx *ctx, int dtype, int64_t B, const double *const params[SPART_NPARAM],
                    const double *rho_thermal, const double *tau_thermal, void *R_TOC, void *R_TOA, void *L_TOA,
                    const spart_materialize *opt, void *workspace, size_t workspace_bytes, void *stream);

/* LUT inversion (the use LUTs are generated for.  The only nearest-index search in the reference is
 * an exact np.argmin with first-index ties, SPART.py:381-387 -- the behaviour kept here):
 * for each of M observed sensor spectra obs (M,nb) find THE row of lut (B,nb) that minimises
 *     c(b, m) = sum_j w_j (lut[b,j] - obs[m,j])^2      (weights (nb,) optional, NULL = 1; expected >= 0)
 * where c is evaluated in `dtype` as  c = 0; for j ascending: d = lut[b,j] - obs[m,j]; c = c + (w_j * d) * d  with every
 * operation rounded to `dtype` and no fused multiply-add.  best_idx (M,) int64 = the LOWEST row index attaining the minimum
 * of that c (bit-exact: the same answer as a brute-force loop, in both dtypes, for any nb), best_cost (M,) = that minimum
 * (divide by nb and take the root for an RMSE).  Rows or observations whose cost is not finite -- NaN, +inf, or -inf (negative
 * weights with overflowing products) -- never win (-1 / +inf when no row has a finite cost); so does a LUT row that holds a
 * non-finite value or whose centred weighted norm sum_j |w_j| (lut[b,j] - centre_j)^2 overflows `dtype` (values near the
 * largest finite number), even if its cost against a particular observation would be finite.
 * How: a GEMM with K = nb + 1 on the matrix cores -- exact-f32 v_mfma_f32_32x32x2_f32 for SPART_F32,
 * v_mfma_f64_16x16x4_f64 for SPART_F64 -- over the CENTRED LUT (per-band mean removed) ranks the tiles of 32 / 16 LUT rows by
 * |x'|^2 - 2 x'.y'; every tile whose minimum lies within a proven rounding bound of the best one is then evaluated row by
 * row with c itself, and an observation for which the scan may have missed such a tile is re-done by a brute-force kernel
 * (csrc/spart_lut.h derives the bound).  The data only decide how much of that extra work there is, never the result.
 * All pointers are device memory in `dtype`; nb <= 31; B, M <= 2e9. */
size_t spart_lut_workspace_bytes(int dtype, int64_t B, int nb, int64_t M);
int spart_lut_nearest(spart_ctx *ctx, int dtype, int64_t B, int nb, const void *lut, int64_t M, const void *obs,
                      const void *weights, int64_t *best_idx, void *best_cost, void *workspace, size_t workspace_bytes,
                      void *stream);
/* Diagnostics of the LAST spart_lut_nearest call that used `workspace` (same dtype, B, nb, M): the number of observations
 * that took the brute-force path and Nmax = max_b sum_j |w_j| (lut[b,j] - centre_j)^2, the scale of the rounding bound.
 * Synchronises the device (a blocking copy of 16 bytes).  No reference counterpart. */
int spart_lut_stats(spart_ctx *ctx, int dtype, int64_t B, int nb, int64_t M, const void *workspace, int64_t *n_brute_force,
                    double *nmax);

/* Measurement aid (bench.py): when enabled, spart_run_batch brackets each of its kernels with HIP events recorded on
 * the stream the kernel runs on, for up to max_calls calls (max_calls = 0 disables).  spart_profile_read_stages waits for
 * them and returns the summed milliseconds per stage -- [0] prelude (per-sample constants), [1] the fused full-band kernel
 * (PROSPECT + BSM + SAILH over all 2162 bands, the dominant one; includes the band-mean reduction when requested),
 * [2] the column kernel (canopy model at the sensor bands, interpolation, SMAC, TOC->TOA) -- and the number of timed
 * calls; spart_profile_read returns stage [1] only. */
#define SPART_NSTAGE 3
int spart_profile_enable(spart_ctx *ctx, int max_calls);
int spart_profile_read(spart_ctx *ctx, double *total_ms, int *ncalls);
int spart_profile_read_stages(spart_ctx *ctx, double stage_ms[SPART_NSTAGE], int *ncalls);

/* Knobs.  None of them changes a result (the one exception is marked); defaults are what every number in DESIGN.md was
 * measured with.
 *   environment, read by the library:
 *     SPART_ROCTX=1          push / pop ROCTX ranges around the stages of spart_run_batch when a roctx library can be dlopen'ed
 *     SPART_SIDE_STREAM=0    read at spart_ctx_create: run the column kernels on the caller's stream instead of a side stream
 *     SPART_CHUNK=<n>        samples per workgroup of the band kernels (tuning sweeps; default: chosen from B)
 *   environment, read by the Python loader / build script (spart_amd/_lib.py, build.py):
 *     SPART_HIP_LIB=<path>   load another build of this ABI (A/B timing, tools/ab_bench.py)
 *     SPART_FAST_MATH=0      build with IEEE division / libm transcendentals instead of v_rcp / v_exp / v_log + Newton steps
 *                            (CHANGES results at the 1e-15 (float64) / 1e-7 (float32) level; parity is tested with the default)
 *   compile-time macros (A/B variants built through build.py's `extra` flags, which are hashed into spart_build_id):
 *     SPART_PRELUDE_WAVES (3)   occupancy the prelude kernel is compiled for
 *     SPART_HOIST_FILM          hoist the water-film transmittance out of the sample loop (default: float64 kernels only)
 *     SPART_FRESH_COEF (1)      re-materialise the plate-model polynomial coefficients per use instead of holding them in VGPRs
 *     SPART_LIDF_JUMP (1)       skip ahead in the LIDF fixed-point iteration by its contraction rate (same iterate sequence end)
 *     SPART_LIDF_ROTATE (1)     sin / cos of an LIDF iterate by rotating the previous iterate's pair through the (small) step
 *     SPART_LIDF_KJUMP (2e-2), SPART_LIDF_GATE (1e-2)   how early that skip takes over from the literal passes (expansion parameter, step size)
 *     SPART_HOTSPOT_SERIES (1)  closed-form series for the hot-spot integrals where it converges, panels elsewhere
 *     SPART_LUT_TO (8)          observation blocks per wave of the float32 LUT scan
 *     SPART_BANDS_PINGPONG      double-buffered constant staging in the full-band kernel (default: float32 columns-only kernel)
 *     SPART_BANDS_SUB (32)      samples per staged copy of that kernel (64 needs the double buffer)
 *     SPART_PRELUDE_SORT (1)    deal a workgroup's 256 samples to its waves in the order of |LIDFa| + |LIDFb|
 *     SPART_COLUMNS_WAVES (4)   occupancy the column kernel is compiled for
 *     SPART_COLUMNS_DIRECT (1)  column kernel stores its results directly (0: through an LDS transpose)
 *     SPART_SMAC_LIBM (1)       library exp / sqrt in the SMAC arithmetic (0: the table-driven float64 exp of the band kernels)
 *     SPART_SMAC_SHARE_EXP (1)  SMAC's three aerosol exponentials as products of exponentials it needs anyway
 *     SPART_EXPERIMENT          1 / 2: arithmetic-only / store-only measurement variants of k_prospect (tools/prospect_split.sh).
 *                               NOT a product configuration: variant 2 does not compute leaf spectra.  Never defined by build.py.
 */

#ifdef __cplusplus
}
#endif
#endif /* SPART_HIP_H */

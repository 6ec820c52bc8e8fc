// libspart_hip.so: the C ABI declared in include/spart_hip.h (host side: context, table
// derivation, workspace carving, kernel launches).  No torch types, no allocation on the
// call path (graph-capturable), caller owns every buffer.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/spart_hip.h"
#include "spart_bands_f32.h"
#include "spart_kernels.h"
#include "spart_lut.h"

using namespace spart;

struct SideLane {
  hipStream_t caller = nullptr, side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
};
struct WsUse {
  const char* base = nullptr;
  size_t bytes = 0;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  bool captured = false;             // `done` was recorded into a stream capture: it orders nothing outside that graph
};
constexpr size_t MAX_LANES = 32;     // caller streams with their own side stream (further streams run the columns in line;
                                     // stated in include/spart_hip.h)
constexpr size_t WS_PRUNE_AT = 16;   // (workspace, stream) records kept before completed ones are looked for and recycled

struct spart_ctx {
  int device = 0;
  float* tabF = nullptr;    // (NTAB, NWL)
  double* tabD = nullptr;   // (NTAB, NWL)
  double* Ea = nullptr;     // (NWL)
  int nb = 0;
  int pf = NWLS, po = NWL;  // row pitch (elements) of the 2162- / 2001-wide spectrum arrays (spart_ctx_set_row_pitch)
  int* band0 = nullptr;      // (nb) evaluation index of the np.interp support point at / below the band centre (k_columns)
  int* band1 = nullptr;      // (nb) ... of the next one (== band0 when the centre is a grid point)
  double* frac = nullptr;    // (nb)
  double* coef = nullptr;    // (48, nb)
  double* econv = nullptr;   // (nb)
  std::vector<double> econv_host;
  // Everything below is mutable per-call state, guarded by `mu`: an entry point holds it while it checks the workspace
  // and issues its launches (microseconds; the GPU work itself stays asynchronous), so calls on ONE context may come from
  // any number of host threads and streams.
  std::mutex mu;
  // The column kernel (k_columns) runs on a side stream beside the full-band kernel (fork / join
  // with two events; spart_run_batch stays asynchronous on the caller's stream).  One lane PER CALLER STREAM, created on
  // first use: two caller streams never share a side stream or an event pair, and a stream under HIP-graph capture
  // pulls only its own side stream into the capture.
  bool side_enabled = true;
  std::vector<SideLane> lanes;
  // Last use of every workspace recently handed in: (range, stream, completion event).  A call that gets a workspace
  // still owned by a call on ANOTHER stream is ordered after it (hipStreamWaitEvent), so sharing one workspace between
  // streams is slow but never a race; if the order cannot be expressed the call fails with SPART_ERR_INVALID.
  std::vector<WsUse> ws_uses;
  // optional timing of the dominant kernel (k_bands): event pairs recorded on the caller's stream
  bool profile = false;
  std::vector<hipEvent_t> ev;   // NEV events per timed call (run_impl)
  size_t ev_used = 0;
  std::vector<char> ev_forked;  // per timed call: the column kernel ran on the side stream
};

constexpr size_t NEV = 4;        // events per timed spart_run_batch call: 3 stage intervals

// the text of the last error raised ON THE CALLING THREAD (spart_last_error): per thread, so that concurrent calls on one
// context cannot garble each other's message
static thread_local char g_err[512] = {0};

static int fail(const spart_ctx*, int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, 512, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(ctx, call)                                                                              \
  do {                                                                                                  \
    hipError_t e_ = (call);                                                                             \
    if (e_ != hipSuccess) return fail(ctx, SPART_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));      \
  } while (0)

namespace {

// Optional ROCTX ranges around the stages of spart_run_batch (the counterpart of the reference's NVTX
// annotations, SPART.py:191-227).  Never a hard dependency: the marker library is looked up with dlopen
// only when SPART_ROCTX=1 is set, and its absence is silent.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char* e = std::getenv("SPART_ROCTX");
    if (!e || e[0] != '1') return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (push && pop) return;
      push = nullptr;
      pop = nullptr;
    }
  }
};
Roctx& roctx() {
  static Roctx r;
  return r;
}
struct Range {
  bool on;
  explicit Range(const char* name) : on(roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~Range() {
    if (on) roctx().pop();
  }
};

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) ok = false;
    if (ok && prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// ---- per-call state of a context (all under ctx->mu)
// the side stream + event pair of caller stream `st`, created on first use; nullptr = run the columns in line
SideLane* lane_for(spart_ctx* ctx, hipStream_t st) {
  if (!ctx->side_enabled) return nullptr;
  for (SideLane& l : ctx->lanes)
    if (l.caller == st) return &l;
  if (ctx->lanes.size() >= MAX_LANES) return nullptr;
  SideLane l;
  l.caller = st;
  if (hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&l.fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&l.join, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    if (l.fork) (void)hipEventDestroy(l.fork);
    if (l.join) (void)hipEventDestroy(l.join);
    if (l.side) (void)hipStreamDestroy(l.side);
    return nullptr;
  }
  ctx->lanes.push_back(l);
  return &ctx->lanes.back();
}

// Before a call's first launch: order it after every call that used an overlapping workspace range on ANOTHER stream.
int ws_acquire(spart_ctx* ctx, const char* base, size_t bytes, hipStream_t st, const char* who) {
  for (const WsUse& u : ctx->ws_uses) {
    if (u.stream == st || base >= u.base + u.bytes || u.base >= base + bytes) continue;
    const hipError_t e = hipStreamWaitEvent(st, u.done, 0);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, SPART_ERR_INVALID, "%s: the workspace is in use by a call on another stream and this call cannot be "
                  "ordered after it (%s): give every stream its own workspace", who, hipGetErrorString(e));
    }
  }
  return SPART_OK;
}
// After a call's last launch: remember (range, stream) and record its completion event.  A record is only ever recycled
// when the use it stands for HAS COMPLETED (hipEventQuery) or can order nothing (its event went into a stream capture);
// while every remembered use is still in flight the list simply grows -- there is no cap that could drop a live one.
int ws_release(spart_ctx* ctx, const char* base, size_t bytes, hipStream_t st, const char* who) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
  const bool capturing = cs != hipStreamCaptureStatusNone;
  WsUse* slot = nullptr;
  for (WsUse& u : ctx->ws_uses)
    if (u.base == base && u.stream == st) slot = &u;      // same pair again: the stream orders the two uses itself
  if (!slot && ctx->ws_uses.size() >= WS_PRUNE_AT) {
    // An event query is "potentially unsafe" under a GLOBAL-mode stream capture anywhere in the process (torch.cuda.graph's
    // default): it would fail and invalidate that capture.  So: none at all while THIS stream is being captured, and
    // otherwise this thread is switched to relaxed capture interaction around the queries (what PyTorch's caching allocator
    // does around its own event queries); only events recorded outside any capture are ever queried.
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool relaxed = !capturing && hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    for (WsUse& u : ctx->ws_uses) {
      bool idle = u.captured;
      if (!idle && relaxed) {
        const hipError_t q = hipEventQuery(u.done);
        if (q == hipSuccess) idle = true;
        else (void)hipGetLastError();                     // hipErrorNotReady (or anything else): treat as in flight
      }
      if (idle) { slot = &u; break; }
    }
    if (relaxed) (void)hipThreadExchangeStreamCaptureMode(&mode);      // (back to the thread's previous mode)
    else if (!capturing) (void)hipGetLastError();
    if (slot) slot->bytes = 0;
  }
  if (!slot) {
    WsUse u;
    if (hipEventCreateWithFlags(&u.done, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      return fail(ctx, SPART_ERR_HIP, "%s: cannot create the workspace completion event", who);
    }
    ctx->ws_uses.push_back(u);
    slot = &ctx->ws_uses.back();
  }
  slot->base = base;
  slot->bytes = bytes > slot->bytes ? bytes : slot->bytes;
  slot->stream = st;
  slot->captured = capturing;
  const hipError_t e = hipEventRecord(slot->done, st);
  if (e != hipSuccess) return fail(ctx, SPART_ERR_HIP, "%s: recording the workspace completion event: %s", who, hipGetErrorString(e));
  return SPART_OK;
}

// the band kernels address 8 rows of the float64 constant block with a 32-bit byte offset (stage_constants)
constexpr int64_t SPART_MAX_BATCH = 60000000;

inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

struct Workspace {
  size_t cstf_off, cstd_off, atm_off, bs_off, total;
  int64_t Bp;       // row pitch of the structure-of-arrays blocks (spart_kernels.h)
};

// the band kernels address a chunk's rows with a 32-bit byte offset per lane
bool chunk_fits_32bit(int chunk, int pitch, size_t es) {
  return (int64_t)chunk * (int64_t)pitch * (int64_t)es <= (int64_t)3900000000LL;
}

// samples per workgroup of the band kernels.  Large batches: ~8192 chunk rows (x 8 tiles = 65k workgroups over
// 256 CUs: the finer grain costs nothing per workgroup -- 17 table loads per lane against >= 120 samples -- and
// shortens the tail of the launch, 1.4 % at B = 1M against 2048 rows), so the per-chunk band sums stay <= 256 MB
// (fp32) whatever B is.  Small batches: at least min(32, B/256) samples per workgroup so that the table loads are
// amortised while ~2000 workgroups remain.
int pick_chunk(int64_t B) {
  static const int forced = [] { const char* e = std::getenv("SPART_CHUNK"); return e ? std::atoi(e) : 0; }();   // tuning knob
  if (forced > 0) return forced;
  int64_t c = (B + 8191) / 8192;
  int64_t small = (B + 255) / 256;
  if (small > 32) small = 32;
  if (c < small) c = small;
  return (int)(c < 1 ? 1 : c);
}

Workspace carve(int dtype, int64_t B) {
  size_t es = dtype == SPART_F64 ? 8 : 4;
  Workspace w;
  w.Bp = row_pitch_of(B);
  const size_t Bp = (size_t)w.Bp;
  size_t o = 0;
  // float32 constants only in the float32 modes; the float64 constants are there in both (the default float32 mode's
  // column kernel is float64: k_columns<double, float>).  ~0.9 KB per sample.
  w.cstf_off = o; o = align_up(o + Bp * NCONST * 4);     // (float64 calls use it with spart_materialize.f32_bands)
  w.cstd_off = o; o = align_up(o + Bp * NCONST * 8);
  w.atm_off = o;  o = align_up(o + Bp * NATM * 8);
  int chunk = pick_chunk(B);
  size_t nchunk = (size_t)((B + chunk - 1) / chunk);
  w.bs_off = o;  o = align_up(o + nchunk * (size_t)(NTILE * TILE) * 4 * es);
  w.total = o;
  return w;
}

// np.interp(x, wlS, .) support points (SPART.py:220-223) on the 2162-point grid
void wl_solar(std::vector<double>& wl) {
  wl.clear();
  for (int i = 400; i <= 2400; ++i) wl.push_back(i);
  for (int i = 2500; i <= 15000; i += 100) wl.push_back(i);
  for (int i = 16000; i <= 50000; i += 1000) wl.push_back(i);
}

template <typename T> int upload(const spart_ctx* ctx, T** dst, const std::vector<T>& src) {
  HIP_TRY(ctx, hipMalloc((void**)dst, src.size() * sizeof(T)));
  HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return SPART_OK;
}

// fast: Newton LIDF / 8-point hot-spot rule (legacy float32 columns and the float32 stage-level entry points)
int launch_prelude(spart_ctx* ctx, bool fast, const ParamPtrs& pp, int mask, int64_t B, int64_t Bp, float* cstF,
                   double* cstD, double* atm, hipStream_t st) {
  unsigned grid = (unsigned)((B + 255) / 256);
  const bool user = pp.lidf != nullptr || (pp.nlayers > 0 && pp.nlayers != NLAYER);   // canopy state of the caller's own
  if (user) {
    ParamPtrs pu = pp;
    if (pu.nlayers <= 0) pu.nlayers = NLAYER;
    if (fast) hipLaunchKernelGGL((k_prelude<true, true>), dim3(grid), dim3(256), 0, st, pu, mask, B, Bp, cstF, cstD, atm);
    else hipLaunchKernelGGL((k_prelude<false, true>), dim3(grid), dim3(256), 0, st, pu, mask, B, Bp, cstF, cstD, atm);
  } else if (fast) hipLaunchKernelGGL((k_prelude<true>), dim3(grid), dim3(256), 0, st, pp, mask, B, Bp, cstF, cstD, atm);
  else hipLaunchKernelGGL((k_prelude<false>), dim3(grid), dim3(256), 0, st, pp, mask, B, Bp, cstF, cstD, atm);
  HIP_TRY(ctx, hipGetLastError());
  return SPART_OK;
}

// the stage-level entry points: constants in the call's dtype
template <typename T> T* stage_cst(char* wsp, const Workspace& ws) { return (T*)(wsp + (sizeof(T) == 4 ? ws.cstf_off : ws.cstd_off)); }
template <typename T>
int launch_stage_prelude(spart_ctx* ctx, const ParamPtrs& pp, int mask, int64_t B, char* wsp, const Workspace& ws, hipStream_t st) {
  return launch_prelude(ctx, sizeof(T) == 4, pp, mask, B, ws.Bp, sizeof(T) == 4 ? (float*)(wsp + ws.cstf_off) : nullptr,
                        sizeof(T) == 8 ? (double*)(wsp + ws.cstd_off) : nullptr, nullptr, st);
}

}  // namespace

template <typename T>
static int prospect_impl(spart_ctx* ctx, int64_t B, const double* const leaf[9], void* refl, void* tran, void* kchl,
                         char* wsp, const Workspace& ws, hipStream_t st) {
  ParamPtrs pp;
  std::memset(&pp, 0, sizeof(pp));
  for (int i = 0; i < 9; ++i) pp.p[i] = leaf[i];
  T* cst = stage_cst<T>(wsp, ws);
  int rc = launch_stage_prelude<T>(ctx, pp, PRE_LEAF, B, wsp, ws, st);
  if (rc) return rc;
  int chunk = pick_chunk(B);
  int64_t nchunk = (B + chunk - 1) / chunk;
  const T* tab = sizeof(T) == 4 ? (const T*)ctx->tabF : (const T*)ctx->tabD;
  if (!chunk_fits_32bit(chunk, ctx->po, sizeof(T))) return fail(ctx, SPART_ERR_INVALID, "batch too large for one call");
  if (nt_ok(ctx->po, sizeof(T)))
    hipLaunchKernelGGL((k_prospect<T, true>), dim3(xcd_grid(nchunk)), dim3(TILE), 0, st, tab, (const T*)cst, ws.Bp, B, chunk,
                       ctx->po, (T*)refl, (T*)tran, (T*)kchl);
  else
    hipLaunchKernelGGL((k_prospect<T, false>), dim3(xcd_grid(nchunk)), dim3(TILE), 0, st, tab, (const T*)cst, ws.Bp, B, chunk,
                       ctx->po, (T*)refl, (T*)tran, (T*)kchl);
  HIP_TRY(ctx, hipGetLastError());
  return SPART_OK;
}

template <typename T>
static int bsm_impl(spart_ctx* ctx, int64_t B, const double* const soil[6], const void* rdry_in, void* refl, void* dry,
                    char* wsp, const Workspace& ws, hipStream_t st) {
  ParamPtrs pp;
  std::memset(&pp, 0, sizeof(pp));
  for (int i = 0; i < 6; ++i) pp.p[9 + i] = soil[i];
  T* cst = stage_cst<T>(wsp, ws);
  int rc = launch_stage_prelude<T>(ctx, pp, PRE_SOIL, B, wsp, ws, st);
  if (rc) return rc;
  int chunk = pick_chunk(B);
  int64_t nchunk = (B + chunk - 1) / chunk;
  const T* tab = sizeof(T) == 4 ? (const T*)ctx->tabF : (const T*)ctx->tabD;
  if (!chunk_fits_32bit(chunk, ctx->po, sizeof(T))) return fail(ctx, SPART_ERR_INVALID, "batch too large for one call");
  if (nt_ok(ctx->po, sizeof(T)))
    hipLaunchKernelGGL((k_bsm<T, true>), dim3(xcd_grid(nchunk)), dim3(TILE), 0, st, tab, (const T*)cst, ws.Bp, B, chunk,
                       ctx->po, (const T*)rdry_in, (T*)refl, (T*)dry);
  else
    hipLaunchKernelGGL((k_bsm<T, false>), dim3(xcd_grid(nchunk)), dim3(TILE), 0, st, tab, (const T*)cst, ws.Bp, B, chunk,
                       ctx->po, (const T*)rdry_in, (T*)refl, (T*)dry);
  HIP_TRY(ctx, hipGetLastError());
  return SPART_OK;
}

template <typename T>
static int sailh_impl(spart_ctx* ctx, int64_t B, const void* rho, const void* tau, const void* rs,
                      const double* const canopy[4], const double* const angles[3], const double* lidf_in, int nlayers,
                      void* const out4[4], char* wsp, const Workspace& ws, hipStream_t st) {
  ParamPtrs pp;
  std::memset(&pp, 0, sizeof(pp));
  for (int i = 0; i < 4; ++i) pp.p[15 + i] = canopy[i];
  for (int i = 0; i < 3; ++i) pp.p[19 + i] = angles[i];
  pp.lidf = lidf_in;
  pp.nlayers = nlayers;
  T* cst = stage_cst<T>(wsp, ws);
  int rc = launch_stage_prelude<T>(ctx, pp, PRE_CANOPY, B, wsp, ws, st);
  if (rc) return rc;
  int chunk = pick_chunk(B);
  int64_t nchunk = (B + chunk - 1) / chunk;
  if (!chunk_fits_32bit(chunk, ctx->pf, sizeof(T))) return fail(ctx, SPART_ERR_INVALID, "batch too large for one call");
  if (nt_ok(ctx->pf, sizeof(T)))
    hipLaunchKernelGGL((k_sailh<T, true>), dim3((unsigned)(nchunk * NTILE_FULL)), dim3(TILE), 0, st, (const T*)cst, ws.Bp, B, chunk,
                       ctx->pf, (const T*)rho, (const T*)tau, (const T*)rs, (T*)out4[0], (T*)out4[1], (T*)out4[2], (T*)out4[3]);
  else
    hipLaunchKernelGGL((k_sailh<T, false>), dim3((unsigned)(nchunk * NTILE_FULL)), dim3(TILE), 0, st, (const T*)cst, ws.Bp, B, chunk,
                       ctx->pf, (const T*)rho, (const T*)tau, (const T*)rs, (T*)out4[0], (T*)out4[1], (T*)out4[2], (T*)out4[3]);
  HIP_TRY(ctx, hipGetLastError());
  return SPART_OK;
}

// T = dtype of the full-band kernel (band sums, materialised spectra); TG = dtype of the column path: the prelude's
// constants and the canopy model inside the column kernel (double, except spart_materialize.f32_columns); TO = dtype
// of the (B, nb) outputs.  <double,double,double> = float64 mode; <float,double,float> = the default float32 mode;
// <float,double,double> = f32_bands; <float,float,float> = f32_columns.
// In EVERY mode the columns come from prelude -> k_columns<TG, TO>, i.e. from the <= 2 nb bands they depend on; the
// full-band kernel runs beside that on the caller's stream whenever full spectra are asked for (opt = NULL: band sums).
template <typename T, typename TG, typename TO = T>
static int run_impl(spart_ctx* ctx, int64_t B, const double* const params[SPART_NPARAM], const double* rho_th,
                    const double* tau_th, void* R_TOC, void* R_TOA, void* L_TOA, const spart_materialize* opt, char* wsp,
                    const Workspace& ws, hipStream_t st) {
  ParamPtrs pp;
  for (int i = 0; i < NPARAM; ++i) pp.p[i] = params[i];
  pp.rho_th = rho_th;
  pp.tau_th = tau_th;
  pp.lidf = opt ? opt->lidf_in : nullptr;        // canopy.lidf / canopy.nlayers as the caller set them (sailh.py:48, 51)
  pp.nlayers = opt ? opt->nlayers : 0;
  const int64_t Bp = ws.Bp;
  float* cstF = (float*)(wsp + ws.cstf_off);
  double* cstD = (double*)(wsp + ws.cstd_off);
  const T* cst = sizeof(T) == 4 ? (const T*)cstF : (const T*)cstD;       // the full-band kernel's constants
  const TG* cstG = sizeof(TG) == 4 ? (const TG*)cstF : (const TG*)cstD;  // the column kernel's constants
  double* atm = (double*)(wsp + ws.atm_off);
  // ---- everything that can fail on its arguments is checked BEFORE any launch (and before the side stream is forked)
  int chunk = pick_chunk(B);
  int64_t nchunk = (B + chunk - 1) / chunk;
  MatPtrs<T> mp;
  std::memset(&mp, 0, sizeof(mp));
  mp.pf = ctx->pf; mp.po = ctx->po;
  const bool nt = nt_ok(ctx->pf, sizeof(T)) && nt_ok(ctx->po, sizeof(T));     // both row grids on the 128-byte lines
  bool mat = false;
  const bool want_rsoil = opt && opt->rsoil;
  if (opt) {
    mp.leaf_refl = (T*)opt->leaf_refl; mp.leaf_tran = (T*)opt->leaf_tran; mp.leaf_kchl = (T*)opt->leaf_kchl;
    mp.soil_refl = (T*)opt->soil_refl; mp.soil_dry = (T*)opt->soil_refl_dry;
    mp.rso = (T*)opt->rso; mp.rdo = (T*)opt->rdo; mp.rsd = (T*)opt->rsd; mp.rdd = (T*)opt->rdd;
    mp.rdry_in = (const T*)opt->rdry_in;
    mat = mp.leaf_refl || mp.leaf_tran || mp.leaf_kchl || mp.soil_refl || mp.soil_dry || mp.rso || mp.rdo || mp.rsd || mp.rdd;
  }
  if ((mat || mp.rdry_in) && !chunk_fits_32bit(chunk, ctx->pf, sizeof(T)))
    return fail(ctx, SPART_ERR_INVALID, "batch too large for materialised spectra in one call (chunk %d rows x pitch %d)", chunk, ctx->pf);
  const bool full = !(opt && opt->prune_unused_bands);
  if (opt && opt->band_mean && !full)
    return fail(ctx, SPART_ERR_INVALID, "spart_run_batch: band_mean needs prune_unused_bands = 0");
  const int narr = 3 + (want_rsoil ? 1 : 0) + ((opt && opt->La) ? 1 : 0);
  // LDS staging of the results: only the (measurement) variant of k_columns that transposes through LDS
  const size_t lds = SPART_COLUMNS_DIRECT ? 0 : (size_t)narr * 64 * ctx->nb * sizeof(TO);     // <= 5 * 64 * 64 * 8 = 160 KiB only for nb = 64 fp64
  if (lds > 64 * 1024) return fail(ctx, SPART_ERR_INVALID, "spart_run_batch: %d sensor bands need %zu B of LDS staging", ctx->nb, lds);
  int rc;
  // optional per-stage timing: four events per call (before the prelude, after the prelude, after the full-band kernel,
  // after the column kernel)
  const bool prof = ctx->profile && ctx->ev_used + NEV <= ctx->ev.size();
  hipEvent_t* ev = prof ? &ctx->ev[ctx->ev_used] : nullptr;
  if (prof) HIP_TRY(ctx, hipEventRecord(ev[0], st));
  const bool four = opt && opt->band_mean;     // the four band sums are only kept apart when their means are asked for
  const bool bands = mat || full;              // the full-band kernel runs (otherwise: pruned, column kernel only)
  {
    Range r("SPART prelude (geometry, LIDF, hot spot, soil factors)");
    // legacy float32 columns (TG = float): the fast prelude; otherwise the literal one.  cstF only when a float32
    // full-band kernel (or the float32 column kernel) will read it.
    rc = launch_prelude(ctx, sizeof(TG) == 4 || (opt && opt->fast_prelude), pp, PRE_ALL, B, Bp, (sizeof(TG) == 4 || (sizeof(T) == 4 && bands)) ? cstF : nullptr,
                        sizeof(TG) == 8 ? cstD : nullptr, atm, st);
  }
  if (rc) return rc;
  Range rb("SPART bands + sensor (BSM, PROSPECT, SAILH | interp, SMAC, TOC->TOA)");
  const T* tab = sizeof(T) == 4 ? (const T*)ctx->tabF : (const T*)ctx->tabD;
  const TG* tabG = sizeof(TG) == 4 ? (const TG*)ctx->tabF : (const TG*)ctx->tabD;
  dim3 grid(xcd_grid(nchunk));
  T* bsum = (T*)(wsp + ws.bs_off);
  if (prof) HIP_TRY(ctx, hipEventRecord(ev[1], st));
  // The columns do not depend on the full-band kernel, so the column kernel runs on the context's side stream BESIDE it
  // and fills issue slots it leaves idle; the caller's stream waits for it at the end.
  SideLane* lane = bands ? lane_for(ctx, st) : nullptr;
  const bool fork = lane != nullptr;
  hipStream_t s2 = fork ? lane->side : st;
  auto columns = [&]() -> int {                // the column kernel, on s2
    hipLaunchKernelGGL((k_columns<TG, TO, TO>), dim3((unsigned)((B + 63) / 64)), dim3(64 * COL_WAVES), lds, s2, tabG, cstG,
                       (const double*)atm, Bp, (const int*)ctx->band0, (const int*)ctx->band1, (const double*)ctx->frac,
                       (const double*)ctx->coef, (const double*)ctx->econv, ctx->nb,
                       (const TO*)(opt ? opt->rdry_in : nullptr), ctx->po, B, (TO*)R_TOC,
                       (TO*)R_TOA, (TO*)L_TOA, (TO*)(opt ? opt->rsoil : nullptr), (TO*)(opt ? opt->La : nullptr));
    HIP_TRY(ctx, hipGetLastError());
    if (prof) HIP_TRY(ctx, hipEventRecord(ev[3], s2));
    return SPART_OK;
  };
  auto band_kernels = [&]() -> int {           // the full-band kernel (+ the batch-mean reduction), on the caller's stream
    // user dry-soil spectra select the reading variant whatever is stored: band sums and band_mean must see the same soil
    // as the columns (with M = 0 the kernel would mix the GSV soil from params[9..11], which may be NULL with rdry_in)
    const int M = mp.rdry_in ? 2 : (mat ? 1 : 0);
    const int F = !full ? 0 : (four ? 2 : 1);
#define SPART_CASE(MM, FF)                                                                                                      \
  if (M == (MM) && F == (FF)) {                                                                                                 \
    if ((MM) != 0 && nt) hipLaunchKernelGGL((k_bands<T, MM, FF, (MM) != 0>), grid, dim3(TILE), 0, st, tab, cst, Bp, B, chunk, mp, bsum); \
    else hipLaunchKernelGGL((k_bands<T, MM, FF, false>), grid, dim3(TILE), 0, st, tab, cst, Bp, B, chunk, mp, bsum);            \
  }
    if constexpr (sizeof(T) == 4) {
      // the float32 kernels WITHOUT materialised spectra (the headline's dominant kernel) are compiled in their own
      // translation unit with their own scheduling strategy (spart_bands_f32.hip, build.py: TU_FLAGS)
      if (M == 0) HIP_TRY(ctx, launch_bands_f32(F, grid.x, st, (const float*)tab, (const float*)cst, Bp, B, chunk, (float*)bsum));
    } else {
      SPART_CASE(0, 1) SPART_CASE(0, 2)
    }
    SPART_CASE(1, 0) SPART_CASE(1, 1) SPART_CASE(1, 2) SPART_CASE(2, 0) SPART_CASE(2, 1) SPART_CASE(2, 2)
#undef SPART_CASE
    HIP_TRY(ctx, hipGetLastError());
    if (prof) HIP_TRY(ctx, hipEventRecord(ev[2], st));
    if (opt && opt->band_mean) {
      hipLaunchKernelGGL((k_bandmean<T>), dim3((4 * NWLS + 255) / 256), dim3(256), 0, st, (const T*)bsum, nchunk, B,
                         (T*)opt->band_mean);
      HIP_TRY(ctx, hipGetLastError());
    }
    return SPART_OK;
  };
  if (!bands) {                                // pruned: the column path alone
    if (prof) HIP_TRY(ctx, hipEventRecord(ev[2], st));
    rc = columns();
  } else if (!fork) {
    if ((rc = band_kernels()) == SPART_OK) rc = columns();
  } else {                                     // side stream first: its kernels are queued before the 65k workgroups of k_bands
    HIP_TRY(ctx, hipEventRecord(lane->fork, st));
    HIP_TRY(ctx, hipStreamWaitEvent(lane->side, lane->fork, 0));
    rc = columns();
    const int rc2 = band_kernels();
    // whatever happened after the fork, the caller's stream is ordered after the side stream's work again (and a HIP-graph
    // capture in progress gets its join): an error code is returned only after the join has been recorded
    const hipError_t ej = hipEventRecord(lane->join, lane->side);
    const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(st, lane->join, 0) : ej;
    if (rc == SPART_OK) rc = rc2;
    if (rc == SPART_OK && ew != hipSuccess) rc = fail(ctx, SPART_ERR_HIP, "spart_run_batch: joining the side stream: %s", hipGetErrorString(ew));
  }
  if (rc) return rc;
  if (prof) {
    ctx->ev_forked.push_back(fork ? 1 : 0);
    ctx->ev_used += NEV;
  }
  return SPART_OK;
}

// ---- LUT inversion (csrc/spart_lut.h): GEMM + argmin on the matrix cores as a filter, exact direct evaluation of every
// candidate, brute-force fallback.  float32: K steps of 2 (v_mfma_f32_32x32x2_f32), KS = ceil((nb + 1) / 2) MFMAs per
// 32 x 32 comparisons, rounded up to one of the compiled variants; float64: K steps of 4 (v_mfma_f64_16x16x4_f64).
static int lut_ks(int nb) {
  const int need = (nb + 2) / 2;
  for (int ks : {4, 7, 8, 11, 16})
    if (ks >= need) return ks;
  return 16;
}
static int lut_ks64(int nb) {
  const int need = (nb + 4) / 4;                      // ceil((nb + 1) / 4)
  for (int ks : {2, 4, 6, 8})
    if (ks >= need) return ks;
  return 8;
}
static int lut_to64(int ks) { return ks <= 4 ? 8 : 4; }     // 16-observation blocks per wave (operand registers: 2 KS TO)
// workgroups = ceil(M / obs per workgroup) x nslice; slices are whole tiles
static int lut_slices(int64_t M, int64_t ntile, int obs_per_wg) {
  const int64_t mg = (M + obs_per_wg - 1) / obs_per_wg;
  int64_t n = (4096 + mg - 1) / mg;
  if (n > ntile) n = ntile;
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

struct LutLayout {
  int ks, to, rows, nslice, npart, kfma;
  int64_t ntile, nfb;
  size_t tiles, pc, ps, pt, centre, ctl, flags, fbc, fbi, total;
};
static LutLayout lut_layout(int dtype, int64_t B, int nb, int64_t M) {
  LutLayout L;
  const size_t es = dtype == SPART_F64 ? 8 : 4;
  if (dtype == SPART_F32) {
    L.rows = 32; L.ks = lut_ks(nb); L.to = LUT_TO; L.kfma = 2 * L.ks;
  } else {
    L.rows = 16; L.ks = lut_ks64(nb); L.to = lut_to64(L.ks); L.kfma = 4 * L.ks;
  }
  L.ntile = (B + L.rows - 1) / L.rows;
  L.nslice = lut_slices(M, L.ntile, L.rows * L.to * 4);
  L.npart = (64 / L.rows) * L.nslice;
  const int64_t mgroups = (M + 63) / 64;
  L.nfb = mgroups > (int64_t)LUT_FB_BLOCKS * 4 ? mgroups : (int64_t)LUT_FB_BLOCKS * 4;
  size_t o = 0;
  L.tiles = o;  o = align_up(o + (size_t)L.ntile * L.ks * 64 * es);
  L.pc = o;     o = align_up(o + (size_t)L.npart * M * es);
  L.ps = o;     o = align_up(o + (size_t)L.npart * M * es);
  L.pt = o;     o = align_up(o + (size_t)L.npart * M * 4);
  L.centre = o; o = align_up(o + 32 * es);
  L.ctl = o;    o = align_up(o + LUT_CTL_WORDS * 8);
  L.flags = o;  o = align_up(o + (size_t)M * 4);
  L.fbc = o;    o = align_up(o + (size_t)L.nfb * 64 * es);
  L.fbi = o;    o = align_up(o + (size_t)L.nfb * 64 * 8);
  L.total = o;
  return L;
}
// Delta = coef_ef * [(N_a + Y) + (N_b + Y)]: spart_lut.h derives (3 nb + 2 K + 13) u; + 3 and 1 % for the second-order terms.
// coef_e: the filter's own error E = (nb + 2 K + 7) u (N + Y), with the same slack
static double lut_coef_ef(int nb, int kfma, double u) { return (3.0 * nb + 2.0 * kfma + 16.0) * 1.01 * u; }
static double lut_coef_e(int nb, int kfma, double u) { return (nb + 2.0 * kfma + 10.0) * 1.01 * u; }

template <typename T>
static int lut_impl(spart_ctx* ctx, int dtype, int64_t B, int nb, const void* lut_, int64_t M, const void* obs_,
                    const void* weights, int64_t* best_idx, void* best_cost, char* wsp, hipStream_t st) {
  const LutLayout L = lut_layout(dtype, B, nb, M);
  const T *lut = (const T*)lut_, *obs = (const T*)obs_, *w = (const T*)weights;
  T* tiles = (T*)(wsp + L.tiles);
  T* pc = (T*)(wsp + L.pc);
  T* ps = (T*)(wsp + L.ps);
  int* pt = (int*)(wsp + L.pt);
  T* centre = (T*)(wsp + L.centre);
  unsigned long long* ctl = (unsigned long long*)(wsp + L.ctl);
  int* flags = (int*)(wsp + L.flags);
  T* fbc = (T*)(wsp + L.fbc);
  int64_t* fbi = (int64_t*)(wsp + L.fbi);
  hipLaunchKernelGGL((k_lut_centre<T>), dim3(nb), dim3(256), 0, st, lut, nb, B, centre, ctl);
  HIP_TRY(ctx, hipGetLastError());
  const dim3 gprep((unsigned)((L.ntile * L.rows + 255) / 256));
  const dim3 grid((unsigned)((M + L.rows * L.to * 4 - 1) / (L.rows * L.to * 4)), (unsigned)L.nslice);
  if constexpr (sizeof(T) == 4) {
#define SPART_LUT_KS(K)                                                                                                   \
  case K:                                                                                                                  \
    hipLaunchKernelGGL((k_lut_prep<float, K, 32>), gprep, dim3(256), 0, st, lut, w, (const float*)centre, nb, B, L.ntile,  \
                       tiles, ctl);                                                                                        \
    hipLaunchKernelGGL((k_lut_scan_mfma<K>), grid, dim3(256), 0, st, (const float*)tiles, obs, w, (const float*)centre,    \
                       nb, L.ntile, M, L.nslice, pc, ps, pt);                                                              \
    break;
    switch (L.ks) { SPART_LUT_KS(4) SPART_LUT_KS(7) SPART_LUT_KS(8) SPART_LUT_KS(11) SPART_LUT_KS(16) }
#undef SPART_LUT_KS
  } else {
#define SPART_LUT_KS(K, TO)                                                                                               \
  case K:                                                                                                                  \
    hipLaunchKernelGGL((k_lut_prep<double, K, 16>), gprep, dim3(256), 0, st, lut, w, (const double*)centre, nb, B,         \
                       L.ntile, tiles, ctl);                                                                               \
    hipLaunchKernelGGL((k_lut_scan_mfma64<K, TO>), grid, dim3(256), 0, st, (const double*)tiles, obs, w,                   \
                       (const double*)centre, nb, L.ntile, M, L.nslice, pc, ps, pt);                                       \
    break;
    switch (L.ks) { SPART_LUT_KS(2, 8) SPART_LUT_KS(4, 8) SPART_LUT_KS(6, 4) SPART_LUT_KS(8, 4) }
#undef SPART_LUT_KS
  }
  HIP_TRY(ctx, hipGetLastError());
  const T coef_ef = (T)lut_coef_ef(nb, L.kfma, (double)LutNum<T>::u), coef_e = (T)lut_coef_e(nb, L.kfma, (double)LutNum<T>::u);
  const dim3 gobs((unsigned)((M + 3) / 4));                // one wave per observation
  hipLaunchKernelGGL((k_lut_reduce_exact<T, (sizeof(T) == 4 ? 32 : 16)>), gobs, dim3(256), 0, st, (const T*)pc, (const T*)ps,
                     (const int*)pt, (const T*)tiles, L.ks, lut, obs, w, (const T*)centre, nb, B, M, L.npart, coef_e, coef_ef, ctl,
                     flags, best_idx, (T*)best_cost);
  HIP_TRY(ctx, hipGetLastError());
  // the flagged observations (normally a handful, possibly all of them for degenerate data): the grid is fixed, the
  // kernels read the count on the device, so the call stays asynchronous and graph-capturable
  const int nbc = (nb + 3) / 4 * 4;
  const size_t fb_lds = (size_t)4 * LUT_FB_ROWS * nbc * sizeof(T);
#define SPART_LUT_FB(N)                                                                                                   \
  case N:                                                                                                                  \
    hipLaunchKernelGGL((k_lut_fallback<T, N>), dim3(LUT_FB_BLOCKS), dim3(256), fb_lds, st, lut, obs, w, nb, B,             \
                       (const unsigned long long*)ctl, (const int*)flags, fbc, fbi);                                       \
    break;
  switch (nbc) {
    SPART_LUT_FB(4) SPART_LUT_FB(8) SPART_LUT_FB(12) SPART_LUT_FB(16) SPART_LUT_FB(20) SPART_LUT_FB(24) SPART_LUT_FB(28) SPART_LUT_FB(32)
  }
#undef SPART_LUT_FB
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL((k_lut_fallback_merge<T>), dim3(256), dim3(256), 0, st, B, LUT_FB_BLOCKS * 4, (const unsigned long long*)ctl,
                     (const int*)flags, (const T*)fbc, (const int64_t*)fbi, best_idx, (T*)best_cost);
  HIP_TRY(ctx, hipGetLastError());
  return SPART_OK;
}

// lock the context, order the call after other streams' use of the workspace, run `body` (the launches), record the
// workspace's completion event -- also after a failed body: some of its kernels may already be queued
template <typename F>
static int guarded(spart_ctx* ctx, const char* who, const void* wsp, size_t bytes, hipStream_t st, F&& body) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  int rc = ws_acquire(ctx, (const char*)wsp, bytes, st, who);
  if (rc) return rc;
  rc = body();
  char keep[512];
  if (rc) std::snprintf(keep, sizeof(keep), "%s", g_err);
  const int rc2 = ws_release(ctx, (const char*)wsp, bytes, st, who);
  if (rc) std::snprintf(g_err, sizeof(g_err), "%s", keep);
  return rc ? rc : rc2;
}

#ifndef SPART_BUILD_ID
#define SPART_BUILD_ID "unidentified"      // built outside spart-python_amd/build.py
#endif
// tag + id: build.py finds the id in the file's bytes (binary_id), spart_build_id() returns the part after the tag
static const char k_build_id[] = "SPART_BUILD_ID:" SPART_BUILD_ID;

extern "C" {

const char* spart_build_id(void) { return k_build_id + 15; }

int spart_abi_version(void) { return SPART_ABI_VERSION; }

const char* spart_last_error(const spart_ctx*) { return g_err; }

int spart_ctx_nb(const spart_ctx* ctx) { return ctx ? ctx->nb : 0; }

int spart_ctx_set_row_pitch(spart_ctx* ctx, int64_t pitch_full, int64_t pitch_optical) {
  if (!ctx) return fail(ctx, SPART_ERR_INVALID, "spart_ctx_set_row_pitch: null context");
  if (pitch_full == 0) pitch_full = NWLS;
  if (pitch_optical == 0) pitch_optical = NWL;
  if (pitch_full < NWLS || pitch_optical < NWL || pitch_full > (1 << 20) || pitch_optical > (1 << 20))
    return fail(ctx, SPART_ERR_INVALID, "row pitch must be >= the row width (%d / %d elements)", NWLS, NWL);
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->pf = (int)pitch_full;
  ctx->po = (int)pitch_optical;
  return SPART_OK;
}

int spart_ctx_econv(const spart_ctx* ctx, double* host_out) {
  if (!ctx || !host_out) return fail(ctx, SPART_ERR_INVALID, "spart_ctx_econv: null argument");
  if (ctx->nb == 0) return fail(ctx, SPART_ERR_NOSENSOR, "context has no sensor");
  std::memcpy(host_out, ctx->econv_host.data(), sizeof(double) * ctx->nb);
  return SPART_OK;
}

int spart_ctx_destroy(spart_ctx* ctx) {
  if (!ctx) return SPART_OK;
  DeviceGuard g(ctx->device);
  (void)hipFree(ctx->tabF); (void)hipFree(ctx->tabD); (void)hipFree(ctx->Ea);
  (void)hipFree(ctx->band0); (void)hipFree(ctx->band1); (void)hipFree(ctx->frac); (void)hipFree(ctx->coef);
  (void)hipFree(ctx->econv);
  for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
  for (SideLane& l : ctx->lanes) {
    (void)hipEventDestroy(l.fork);
    (void)hipEventDestroy(l.join);
    (void)hipStreamDestroy(l.side);
  }
  for (WsUse& u : ctx->ws_uses) (void)hipEventDestroy(u.done);
  delete ctx;
  return SPART_OK;
}

int spart_ctx_create(spart_ctx** out, int device, const spart_tables* t) {
  if (!out || !t) return fail(nullptr, SPART_ERR_INVALID, "spart_ctx_create: null argument");
  *out = nullptr;
  const double* req[] = {t->nr, t->Kab, t->Kca, t->Kdm, t->Kw, t->Ks, t->Kant, t->cbc, t->prot, t->GSV, t->nw, t->Ea};
  for (const double* p : req)
    if (!p) return fail(nullptr, SPART_ERR_INVALID, "spart_ctx_create: a spectral table pointer is null");
  if (t->nb < 0 || t->nb > MAX_NB) return fail(nullptr, SPART_ERR_INVALID, "spart_ctx_create: nb=%d out of range", t->nb);
  if (t->nb > 0 && (!t->wl_smac || !t->coef || !t->wl_srf || !t->p_srf || t->nsrf <= 0))
    return fail(nullptr, SPART_ERR_INVALID, "spart_ctx_create: sensor block incomplete");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(nullptr, SPART_ERR_HIP, "spart_ctx_create: no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(nullptr, SPART_ERR_INVALID, "spart_ctx_create: device %d of %d", device, ndev);
  DeviceGuard guard(device);
  if (!guard.ok) return fail(nullptr, SPART_ERR_HIP, "spart_ctx_create: cannot select device %d", device);

  spart_ctx* ctx = new spart_ctx();
  ctx->device = device;
  // --- derived per-band tables, float64 on the host
  std::vector<double> tab((size_t)NTAB * NWL);
  const double tav90_2 = calculate_tav(90, 2.0);
  for (int i = 0; i < NWL; ++i) {
    tab[TAB_KAB * NWL + i] = t->Kab[i];
    tab[TAB_KCA * NWL + i] = t->Kca[i];
    tab[TAB_KDM * NWL + i] = t->Kdm[i];
    tab[TAB_KW * NWL + i] = t->Kw[i];
    tab[TAB_KS * NWL + i] = t->Ks[i];
    tab[TAB_KANT * NWL + i] = t->Kant[i];
    tab[TAB_CBC * NWL + i] = t->cbc[i];
    tab[TAB_PROT * NWL + i] = t->prot[i];
    double nr = t->nr[i], nw = t->nw[i];
    double t12 = calculate_tav(90, nr);                         // prospect_5d.py:202
    tab[TAB_TALF * NWL + i] = calculate_tav(40, nr);            // :200
    tab[TAB_T12 * NWL + i] = t12;
    tab[TAB_T21 * NWL + i] = t12 / (nr * nr);                   // :204
    tab[TAB_GSV0 * NWL + i] = t->GSV[3 * i + 0];
    tab[TAB_GSV1 * NWL + i] = t->GSV[3 * i + 1];
    tab[TAB_GSV2 * NWL + i] = t->GSV[3 * i + 2];
    tab[TAB_CBAC * NWL + i] = calculate_tav(90, 2.0 / nw) / tav90_2;   // bsm.py:111
    tab[TAB_PW * NWL + i] = 1.0 - calculate_tav(90, nw) / (nw * nw);   // bsm.py:115
    tab[TAB_RW * NWL + i] = 1.0 - calculate_tav(40, nw);               // bsm.py:119
  }
  std::vector<float> tabf(tab.begin(), tab.end());
  int rc;
  if ((rc = upload(ctx, &ctx->tabD, tab)) || (rc = upload(ctx, &ctx->tabF, tabf))) { spart_ctx_destroy(ctx); return rc; }
  std::vector<double> ea(t->Ea, t->Ea + NWL);
  if ((rc = upload(ctx, &ctx->Ea, ea))) { spart_ctx_destroy(ctx); return rc; }

  // --- sensor block
  ctx->nb = t->nb;
  if (t->nb > 0) {
    std::vector<double> wl;
    wl_solar(wl);
    std::vector<int> e0(t->nb), e1(t->nb);
    std::vector<double> fr(t->nb);
    auto eval_of = [](int grid_idx) { return grid_idx < NWL ? grid_idx : NWL; };   // every thermal grid point holds the same value
    for (int j = 0; j < t->nb; ++j) {
      double x = t->wl_smac[j];
      // i0 = last grid point <= x, clipped to [0, n-2]; np.interp clamps outside the grid
      int i0 = 0;
      while (i0 + 1 < (int)wl.size() - 1 && wl[i0 + 1] <= x) ++i0;
      int i1 = i0 + 1;
      double f = (x - wl[i0]) / (wl[i1] - wl[i0]);
      if (!(f > 0.0)) f = 0.0;
      if (f > 1.0) f = 1.0;
      e0[j] = eval_of(i0);
      e1[j] = f > 0.0 ? eval_of(i1) : e0[j];
      fr[j] = f;
    }
    std::vector<double> coef(t->coef, t->coef + (size_t)NCOEF * t->nb);
    std::vector<double> wsrf(t->wl_srf, t->wl_srf + (size_t)t->nsrf * t->nb), psrf(t->p_srf, t->p_srf + (size_t)t->nsrf * t->nb);
    double *d_w = nullptr, *d_p = nullptr;
    std::vector<double> ec(t->nb, 0.0);
    if ((rc = upload(ctx, &ctx->band0, e0)) || (rc = upload(ctx, &ctx->band1, e1)) || (rc = upload(ctx, &ctx->frac, fr)) ||
        (rc = upload(ctx, &ctx->coef, coef)) || (rc = upload(ctx, &ctx->econv, ec)) || (rc = upload(ctx, &d_w, wsrf)) ||
        (rc = upload(ctx, &d_p, psrf))) {
      (void)hipFree(d_w); (void)hipFree(d_p);
      spart_ctx_destroy(ctx);
      return rc;
    }
    // SRF convolution of the ET irradiance: one wave per sensor band (SPART.py:358-396)
    hipLaunchKernelGGL(k_econv, dim3(t->nb), dim3(64), 0, 0, ctx->Ea, d_w, d_p, t->nsrf, t->nb, ctx->econv);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    ctx->econv_host.resize(t->nb);
    if (e == hipSuccess) e = hipMemcpy(ctx->econv_host.data(), ctx->econv, sizeof(double) * t->nb, hipMemcpyDeviceToHost);
    (void)hipFree(d_w); (void)hipFree(d_p);
    if (e != hipSuccess) {
      fail(nullptr, SPART_ERR_HIP, "spart_ctx_create: SRF convolution kernel: %s", hipGetErrorString(e));
      spart_ctx_destroy(ctx);
      return SPART_ERR_HIP;
    }
  }
  {
    const char* e = std::getenv("SPART_SIDE_STREAM");            // "0" keeps every kernel on the caller's stream
    ctx->side_enabled = !(e && e[0] == '0');
    ctx->lanes.reserve(MAX_LANES);                                // (pointers into the vector stay valid)
    ctx->ws_uses.reserve(WS_PRUNE_AT);
  }
  *out = ctx;
  return SPART_OK;
}

int spart_profile_enable(spart_ctx* ctx, int max_calls) {
  if (!ctx) return fail(nullptr, SPART_ERR_INVALID, "spart_profile_enable: null context");
  DeviceGuard guard(ctx->device);
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->profile = max_calls > 0;
  ctx->ev_used = 0;
  ctx->ev_forked.clear();
  while (ctx->ev.size() < (size_t)(max_calls > 0 ? NEV * max_calls : 0)) {
    hipEvent_t e;
    HIP_TRY(ctx, hipEventCreate(&e));
    ctx->ev.push_back(e);
  }
  return SPART_OK;
}

int spart_profile_read_stages(spart_ctx* ctx, double stage_ms[SPART_NSTAGE], int* ncalls) {
  if (!ctx || !stage_ms || !ncalls) return fail(ctx, SPART_ERR_INVALID, "spart_profile_read_stages: null argument");
  DeviceGuard guard(ctx->device);
  std::lock_guard<std::mutex> lock(ctx->mu);
  static_assert(SPART_NSTAGE + 1 == NEV, "one event more than stages");
  for (int k = 0; k < SPART_NSTAGE; ++k) stage_ms[k] = 0.0;
  int n = 0;
  for (size_t i = 0; i + NEV <= ctx->ev_used; i += NEV) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[i + 2]));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[i + NEV - 1]));
    // events: 0 before the prelude, 1 after it, 2 after the full-band kernel, 3 after the column kernel.  When 3 was
    // recorded on the side stream the column kernel started at event 1, beside the band kernel.
    const bool forked = n < (int)ctx->ev_forked.size() && ctx->ev_forked[n];
    const int from[SPART_NSTAGE] = {0, 1, forked ? 1 : 2}, to[SPART_NSTAGE] = {1, 2, 3};
    for (int k = 0; k < SPART_NSTAGE; ++k) {
      float ms = 0.f;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev[i + from[k]], ctx->ev[i + to[k]]));
      stage_ms[k] += ms;
    }
    ++n;
  }
  ctx->ev_used = 0;
  ctx->ev_forked.clear();
  *ncalls = n;
  return SPART_OK;
}

int spart_profile_read(spart_ctx* ctx, double* total_ms, int* ncalls) {
  if (!ctx || !total_ms || !ncalls) return fail(ctx, SPART_ERR_INVALID, "spart_profile_read: null argument");
  double st[SPART_NSTAGE];
  int rc = spart_profile_read_stages(ctx, st, ncalls);
  *total_ms = st[1];
  return rc;
}

int spart_calculate_tav(double alpha_deg, const double* nr, int64_t n, double* out) {
  if (!nr || !out || n < 0) { std::snprintf(g_err, 512, "spart_calculate_tav: null pointer or negative length"); return SPART_ERR_INVALID; }
  for (int64_t i = 0; i < n; ++i) out[i] = calculate_tav(alpha_deg, nr[i]);
  return SPART_OK;
}

size_t spart_workspace_bytes(const spart_ctx* ctx, int dtype, int64_t B) {
  if (!ctx || B <= 0) return 0;
  return carve(dtype, B).total;
}

#define CHECK_COMMON(name)                                                                                  \
  if (!ctx) return fail(nullptr, SPART_ERR_INVALID, name ": null context");                                 \
  if (B > SPART_MAX_BATCH) return fail(ctx, SPART_ERR_INVALID, name ": at most %lld samples per call", (long long)SPART_MAX_BATCH); \
  if (dtype != SPART_F32 && dtype != SPART_F64) return fail(ctx, SPART_ERR_INVALID, name ": bad dtype %d", dtype); \
  if (B < 0) return fail(ctx, SPART_ERR_INVALID, name ": negative batch");                                  \
  if (B == 0) return SPART_OK;                                                                              \
  Workspace ws = carve(dtype, B);                                                                           \
  if (!workspace || workspace_bytes < ws.total)                                                             \
    return fail(ctx, SPART_ERR_WORKSPACE, name ": workspace of %zu bytes needed, %zu given", ws.total, workspace_bytes); \
  DeviceGuard guard(ctx->device);                                                                           \
  hipStream_t st = (hipStream_t)stream;                                                                     \
  char* wsp = (char*)workspace;


int spart_prospect_batch(spart_ctx* ctx, int dtype, int64_t B, const double* const leaf[9], void* refl, void* tran,
                         void* kchl, void* workspace, size_t workspace_bytes, void* stream) {
  CHECK_COMMON("spart_prospect_batch")
  if (!leaf) return fail(ctx, SPART_ERR_INVALID, "spart_prospect_batch: null leaf");
  for (int i = 0; i < 9; ++i)
    if (!leaf[i]) return fail(ctx, SPART_ERR_INVALID, "spart_prospect_batch: leaf[%d] is null", i);
  return guarded(ctx, "spart_prospect_batch", wsp, ws.total, st, [&] {
    return dtype == SPART_F32 ? prospect_impl<float>(ctx, B, leaf, refl, tran, kchl, wsp, ws, st)
                              : prospect_impl<double>(ctx, B, leaf, refl, tran, kchl, wsp, ws, st);
  });
}


int spart_bsm_batch(spart_ctx* ctx, int dtype, int64_t B, const double* const soil[6], const void* rdry_in, void* refl,
                    void* refl_dry, void* workspace, size_t workspace_bytes, void* stream) {
  CHECK_COMMON("spart_bsm_batch")
  if (!soil) return fail(ctx, SPART_ERR_INVALID, "spart_bsm_batch: null soil");
  for (int i = 0; i < 6; ++i)
    if (!soil[i] && !(rdry_in && i < 3)) return fail(ctx, SPART_ERR_INVALID, "spart_bsm_batch: soil[%d] is null", i);
  return guarded(ctx, "spart_bsm_batch", wsp, ws.total, st, [&] {
    return dtype == SPART_F32 ? bsm_impl<float>(ctx, B, soil, rdry_in, refl, refl_dry, wsp, ws, st)
                              : bsm_impl<double>(ctx, B, soil, rdry_in, refl, refl_dry, wsp, ws, st);
  });
}

int spart_lidf_batch(spart_ctx* ctx, int64_t B, const double* LIDFa, const double* LIDFb, double* lidf, void* stream) {
  if (!ctx) return fail(nullptr, SPART_ERR_INVALID, "spart_lidf_batch: null context");
  if (B < 0 || !LIDFa || !LIDFb || !lidf) return fail(ctx, SPART_ERR_INVALID, "spart_lidf_batch: bad argument");
  if (B == 0) return SPART_OK;
  DeviceGuard guard(ctx->device);
  hipLaunchKernelGGL(k_lidf, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, LIDFa, LIDFb, B, lidf);
  HIP_TRY(ctx, hipGetLastError());
  return SPART_OK;
}


int spart_sailh_batch(spart_ctx* ctx, int dtype, int64_t B, const void* rho, const void* tau, const void* rs,
                      const double* const canopy[4], const double* const angles[3], const double* lidf_in, int32_t nlayers,
                      void* const out4[4], void* workspace, size_t workspace_bytes, void* stream) {
  CHECK_COMMON("spart_sailh_batch")
  if (!rho || !tau || !rs || !canopy || !angles || !out4) return fail(ctx, SPART_ERR_INVALID, "spart_sailh_batch: null argument");
  if (nlayers < 0 || nlayers > SPART_MAX_NLAYERS) return fail(ctx, SPART_ERR_INVALID, "spart_sailh_batch: nlayers = %d (0 = the default 60, else 1 ... %d)", nlayers, SPART_MAX_NLAYERS);
  for (int i = 0; i < 4; ++i)
    if ((!canopy[i] && !(lidf_in && (i == 1 || i == 2))) || !out4[i])      // LIDFa, LIDFb are unused with a given lidf
      return fail(ctx, SPART_ERR_INVALID, "spart_sailh_batch: canopy/out4[%d] is null", i);
  for (int i = 0; i < 3; ++i)
    if (!angles[i]) return fail(ctx, SPART_ERR_INVALID, "spart_sailh_batch: angles[%d] is null", i);
  return guarded(ctx, "spart_sailh_batch", wsp, ws.total, st, [&] {
    return dtype == SPART_F32 ? sailh_impl<float>(ctx, B, rho, tau, rs, canopy, angles, lidf_in, nlayers, out4, wsp, ws, st)
                              : sailh_impl<double>(ctx, B, rho, tau, rs, canopy, angles, lidf_in, nlayers, out4, wsp, ws, st);
  });
}

int spart_smac_batch(spart_ctx* ctx, int64_t B, const double* const angles[3], const double* const atm[4],
                     double* const out9[9], void* workspace, size_t workspace_bytes, void* stream) {
  int dtype = SPART_F64;
  CHECK_COMMON("spart_smac_batch")
  if (ctx->nb == 0) return fail(ctx, SPART_ERR_NOSENSOR, "spart_smac_batch: context has no sensor");
  if (!angles || !atm || !out9) return fail(ctx, SPART_ERR_INVALID, "spart_smac_batch: null argument");
  for (int i = 0; i < 3; ++i) if (!angles[i]) return fail(ctx, SPART_ERR_INVALID, "spart_smac_batch: angles[%d] is null", i);
  for (int i = 0; i < 4; ++i) if (!atm[i]) return fail(ctx, SPART_ERR_INVALID, "spart_smac_batch: atm[%d] is null", i);
  for (int i = 0; i < 9; ++i) if (!out9[i]) return fail(ctx, SPART_ERR_INVALID, "spart_smac_batch: out9[%d] is null", i);
  return guarded(ctx, "spart_smac_batch", wsp, ws.total, st, [&]() -> int {
    ParamPtrs pp;
    std::memset(&pp, 0, sizeof(pp));
    for (int i = 0; i < 3; ++i) pp.p[19 + i] = angles[i];
    for (int i = 0; i < 4; ++i) pp.p[22 + i] = atm[i];
    double* a = (double*)(wsp + ws.atm_off);
    int rc = launch_prelude(ctx, false, pp, PRE_ATM, B, ws.Bp, nullptr, nullptr, a, st);
    if (rc) return rc;
    Out9 o;
    for (int i = 0; i < 9; ++i) o.o[i] = out9[i];
    int64_t n = B * ctx->nb;
    hipLaunchKernelGGL(k_smac, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)ctx->coef, ctx->nb,
                       (const double*)a, ws.Bp, B, o);
    HIP_TRY(ctx, hipGetLastError());
    return SPART_OK;
  });
}


int spart_run_batch(spart_ctx* ctx, int dtype, int64_t B, const double* const params[SPART_NPARAM],
                    const double* rho_thermal, const double* tau_thermal, void* R_TOC, void* R_TOA, void* L_TOA,
                    const spart_materialize* opt, void* workspace, size_t workspace_bytes, void* stream) {
  CHECK_COMMON("spart_run_batch")
  if (ctx->nb == 0) return fail(ctx, SPART_ERR_NOSENSOR, "spart_run_batch: context has no sensor");
  if (!params || !R_TOC || !R_TOA || !L_TOA) return fail(ctx, SPART_ERR_INVALID, "spart_run_batch: null argument");
  for (int i = 0; i < SPART_NPARAM; ++i)
    if (!params[i] && !(opt && opt->rdry_in && i >= 9 && i <= 11)    // B, lat, lon are unused with user dry spectra
        && !(opt && opt->lidf_in && (i == 16 || i == 17)))           // LIDFa, LIDFb are unused with a given lidf
      return fail(ctx, SPART_ERR_INVALID, "spart_run_batch: params[%d] is null", i);
  if (opt && (opt->nlayers < 0 || opt->nlayers > SPART_MAX_NLAYERS))
    return fail(ctx, SPART_ERR_INVALID, "spart_run_batch: nlayers = %d (0 = the default 60, else 1 ... %d)", opt->nlayers, SPART_MAX_NLAYERS);
  if (dtype == SPART_F64 && opt && opt->f32_bands &&
      (opt->leaf_refl || opt->leaf_tran || opt->leaf_kchl || opt->soil_refl || opt->soil_refl_dry || opt->rso || opt->rdo ||
       opt->rsd || opt->rdd || opt->band_mean || opt->rdry_in || opt->f32_columns))
    // float64 columns (identical to the float64 mode's) over a float32 full-band pass: nothing the float32 kernel
    // would have to write in float64 may be requested
    return fail(ctx, SPART_ERR_INVALID, "spart_run_batch: f32_bands goes with the sensor columns (and rsoil / La) only");
  return guarded(ctx, "spart_run_batch", wsp, ws.total, st, [&] {
    if (dtype == SPART_F64 && opt && opt->f32_bands)
      return run_impl<float, double, double>(ctx, B, params, rho_thermal, tau_thermal, R_TOC, R_TOA, L_TOA, opt, wsp, ws, st);
    if (dtype == SPART_F64)
      return run_impl<double, double, double>(ctx, B, params, rho_thermal, tau_thermal, R_TOC, R_TOA, L_TOA, opt, wsp, ws, st);
    return (opt && opt->f32_columns)
               ? run_impl<float, float, float>(ctx, B, params, rho_thermal, tau_thermal, R_TOC, R_TOA, L_TOA, opt, wsp, ws, st)
               : run_impl<float, double, float>(ctx, B, params, rho_thermal, tau_thermal, R_TOC, R_TOA, L_TOA, opt, wsp, ws, st);
  });
}

size_t spart_lut_workspace_bytes(int dtype, int64_t B, int nb, int64_t M) {
  if (B <= 0 || M <= 0 || nb < 1 || nb > 31 || (dtype != SPART_F32 && dtype != SPART_F64)) return 0;
  return lut_layout(dtype, B, nb, M).total;
}

int spart_lut_nearest(spart_ctx* ctx, int dtype, int64_t B, int nb, const void* lut, int64_t M, const void* obs,
                      const void* weights, int64_t* best_idx, void* best_cost, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!ctx) return fail(nullptr, SPART_ERR_INVALID, "spart_lut_nearest: null context");
  if (dtype != SPART_F32 && dtype != SPART_F64) return fail(ctx, SPART_ERR_INVALID, "spart_lut_nearest: bad dtype %d", dtype);
  if (B < 0 || M < 0 || nb < 1 || nb > 31 || B > 2000000000LL || M > 2000000000LL)
    return fail(ctx, SPART_ERR_INVALID, "spart_lut_nearest: bad sizes (B=%lld M=%lld nb=%d; nb <= 31, B and M <= 2e9)", (long long)B, (long long)M, nb);
  if (M == 0) return SPART_OK;
  if (B == 0) return fail(ctx, SPART_ERR_INVALID, "spart_lut_nearest: empty LUT");
  if (!lut || !obs || !best_idx || !best_cost) return fail(ctx, SPART_ERR_INVALID, "spart_lut_nearest: null argument");
  size_t need = spart_lut_workspace_bytes(dtype, B, nb, M);
  if (!workspace || workspace_bytes < need)
    return fail(ctx, SPART_ERR_WORKSPACE, "spart_lut_nearest: workspace of %zu bytes needed, %zu given", need, workspace_bytes);
  DeviceGuard guard(ctx->device);
  hipStream_t st = (hipStream_t)stream;
  return guarded(ctx, "spart_lut_nearest", workspace, need, st, [&] {
    return dtype == SPART_F32 ? lut_impl<float>(ctx, dtype, B, nb, lut, M, obs, weights, best_idx, best_cost, (char*)workspace, st)
                              : lut_impl<double>(ctx, dtype, B, nb, lut, M, obs, weights, best_idx, best_cost, (char*)workspace, st);
  });
}

int spart_lut_stats(spart_ctx* ctx, int dtype, int64_t B, int nb, int64_t M, const void* workspace, int64_t* n_brute_force,
                    double* nmax) {
  if (!ctx) return fail(nullptr, SPART_ERR_INVALID, "spart_lut_stats: null context");
  if (!workspace || !n_brute_force || !nmax || spart_lut_workspace_bytes(dtype, B, nb, M) == 0)
    return fail(ctx, SPART_ERR_INVALID, "spart_lut_stats: bad argument");
  DeviceGuard guard(ctx->device);
  unsigned long long ctl[LUT_CTL_WORDS];
  HIP_TRY(ctx, hipMemcpy(ctl, (const char*)workspace + lut_layout(dtype, B, nb, M).ctl, sizeof(ctl), hipMemcpyDeviceToHost));
  *n_brute_force = (int64_t)ctl[1];
  if (dtype == SPART_F32) {
    const unsigned b = (unsigned)ctl[0];
    float f;
    std::memcpy(&f, &b, 4);
    *nmax = f;
  } else {
    std::memcpy(nmax, &ctl[0], 8);
  }
  return SPART_OK;
}

}  // extern "C"

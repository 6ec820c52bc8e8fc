// TEST INFRASTRUCTURE ONLY: compiles the device arithmetic header (csrc/spart_math.h) with
// g++ so that the float32 / float64 formulations can be checked against the oracle on a
// machine without a GPU (tests/test_hostmath.py).  Nothing in spart-python_amd/ loads this.
#include <cstdint>
#include <cstring>

#include "../../spart-python_amd/csrc/spart_math.h"

using namespace spart;

template <typename T> static BandTab<T> tab_at(const double* tab, int i) {
  BandTab<T> t;
  t.kab = (T)tab[TAB_KAB * NWL + i]; t.kca = (T)tab[TAB_KCA * NWL + i]; t.kdm = (T)tab[TAB_KDM * NWL + i];
  t.kw = (T)tab[TAB_KW * NWL + i]; t.ks = (T)tab[TAB_KS * NWL + i]; t.kant = (T)tab[TAB_KANT * NWL + i];
  t.kcbc = (T)tab[TAB_CBC * NWL + i]; t.kprot = (T)tab[TAB_PROT * NWL + i]; t.talf = (T)tab[TAB_TALF * NWL + i];
  t.t12 = (T)tab[TAB_T12 * NWL + i]; t.t21 = (T)tab[TAB_T21 * NWL + i]; t.g0 = (T)tab[TAB_GSV0 * NWL + i];
  t.g1 = (T)tab[TAB_GSV1 * NWL + i]; t.g2 = (T)tab[TAB_GSV2 * NWL + i]; t.cbac = (T)tab[TAB_CBAC * NWL + i];
  t.pw = (T)tab[TAB_PW * NWL + i]; t.rw = (T)tab[TAB_RW * NWL + i];
  return t;
}

// lidf_in (B, 13) / nl: the caller's canopy.lidf / canopy.nlayers (nullptr / 0 = the defaults), as k_prelude<., true> takes them
template <typename T>
static void bands_impl(int64_t B, const double* tab, const double* P, double* out /* (B, NEVAL, 10) */,
                       double* atm_out, double* lidf_out, const double* lidf_in = nullptr, int nl = 0) {
  for (int64_t s = 0; s < B; ++s) {
    T c[NCONST];
    double a[NATM], li[NLINCL];
    sample_prelude<T, (sizeof(T) == 4)>(P + s * NPARAM, 0.01, 0.01, PRE_ALL, c, a, li, lidf_in ? lidf_in + s * NLINCL : nullptr, nl);
    std::memcpy(atm_out + s * NATM, a, sizeof(a));
    std::memcpy(lidf_out + s * NLINCL, li, sizeof(li));
    CanopyPar<T> cp;
    cp.sob = c[C_SOB]; cp.sof = c[C_SOF]; cp.hbf = c[C_HBF]; cp.ks = c[C_KS]; cp.ko = c[C_KO]; cp.lai = c[C_LAI]; cp.lai2 = c[C_LAI2];
    cp.tss = c[C_TSS]; cp.too = c[C_TOO]; cp.Z = c[C_Z]; cp.hot = c[C_HOT]; cp.pso2w = c[C_PSO2W];
    for (int band = 0; band < NEVAL; ++band) {
      bool thermal = band == NWL;
      BandTab<T> tb = tab_at<T>(tab, thermal ? NWL - 1 : band);
      T refl, tran, absb, K;
      leaf_band<T>(tb, c[C_CAB], c[C_CCA], c[C_CDM], c[C_CW], c[C_CS], c[C_CANT], c[C_CBC], c[C_PROT], c[C_NM1], refl, tran,
                   absb, K);
      T rdry = soil_dry<T>(tb, c[C_F1], c[C_F2], c[C_F3]);
      T fm[7] = {c[C_FM0], c[C_FM1], c[C_FM2], c[C_FM3], c[C_FM4], c[C_FM5], c[C_FM6]};
      T rwet;
      soil_band<T>(tb, rdry, c[C_WET], fm, c[C_FMSUM], c[C_FILM2L], rwet);
      T rho = thermal ? c[C_RHO_TH] : refl, tau = thermal ? c[C_TAU_TH] : tran;
      T ab = thermal ? (T(1) - c[C_RHO_TH] - c[C_TAU_TH]) : absb;
      T rso, rdo, rsd, rdd;
      canopy_band<T>(cp, rho, tau, ab, rwet, rso, rdo, rsd, rdd);
      double* o = out + ((size_t)s * NEVAL + band) * 10;
      o[0] = rho; o[1] = tau; o[2] = (K > T(0)) ? (double)(c[C_CAB] * tb.kab / K) : 0.0; o[3] = rdry; o[4] = rwet;
      o[5] = rso; o[6] = rdo; o[7] = rsd; o[8] = rdd; o[9] = ab;
    }
  }
}

extern "C" {

void hm_derive_tables(const double* nr, const double* nw, const double* Kab, const double* Kca, const double* Kdm,
                      const double* Kw, const double* Ks, const double* Kant, const double* cbc, const double* prot,
                      const double* GSV, double* tab) {
  const double tav90_2 = calculate_tav(90, 2.0);
  for (int i = 0; i < NWL; ++i) {
    tab[TAB_KAB * NWL + i] = Kab[i]; tab[TAB_KCA * NWL + i] = Kca[i]; tab[TAB_KDM * NWL + i] = Kdm[i];
    tab[TAB_KW * NWL + i] = Kw[i]; tab[TAB_KS * NWL + i] = Ks[i]; tab[TAB_KANT * NWL + i] = Kant[i];
    tab[TAB_CBC * NWL + i] = cbc[i]; tab[TAB_PROT * NWL + i] = prot[i];
    double t12 = calculate_tav(90, nr[i]);
    tab[TAB_TALF * NWL + i] = calculate_tav(40, nr[i]); tab[TAB_T12 * NWL + i] = t12; tab[TAB_T21 * NWL + i] = t12 / (nr[i] * nr[i]);
    tab[TAB_GSV0 * NWL + i] = GSV[3 * i]; tab[TAB_GSV1 * NWL + i] = GSV[3 * i + 1]; tab[TAB_GSV2 * NWL + i] = GSV[3 * i + 2];
    tab[TAB_CBAC * NWL + i] = calculate_tav(90, 2.0 / nw[i]) / tav90_2;
    tab[TAB_PW * NWL + i] = 1.0 - calculate_tav(90, nw[i]) / (nw[i] * nw[i]);
    tab[TAB_RW * NWL + i] = 1.0 - calculate_tav(40, nw[i]);
  }
}

// out: (B, 2002, 10) = rho, tau, kchl, rdry, rwet, rso, rdo, rsd, rdd, absorptance
void hm_bands(int dtype, int64_t B, const double* tab, const double* P, double* out, double* atm, double* lidf) {
  if (dtype == 0) bands_impl<float>(B, tab, P, out, atm, lidf);
  else bands_impl<double>(B, tab, P, out, atm, lidf);
}

// the same with the caller's canopy state (sailh.py:48, 51): lidf_in (B, 13) or NULL, nl = canopy.nlayers or 0
void hm_bands_state(int dtype, int64_t B, const double* tab, const double* P, const double* lidf_in, int nl, double* out,
                    double* atm, double* lidf) {
  if (dtype == 0) bands_impl<float>(B, tab, P, out, atm, lidf, lidf_in, nl);
  else bands_impl<double>(B, tab, P, out, atm, lidf, lidf_in, nl);
}

// SMAC + TOC->TOA for (B, nb): rv (B, nb, 4) = rso, rdo, rsd, rdd at the band centres
void hm_sensor(int64_t B, int nb, const double* atm, const double* coef, const double* econv, const double* rv,
               double* smac9 /* (B,nb,9) */, double* toa3 /* (B,nb,3) */) {
  for (int64_t s = 0; s < B; ++s)
    for (int j = 0; j < nb; ++j) {
      SmacOut so = smac_band(atm + s * NATM, coef + j, nb);
      double* o = smac9 + ((size_t)s * nb + j) * 9;
      o[0] = so.Ta_s; o[1] = so.Ta_o; o[2] = so.Tg; o[3] = so.Ra_dd; o[4] = so.Ra_so; o[5] = so.Ta_ss; o[6] = so.Ta_sd;
      o[7] = so.Ta_oo; o[8] = so.Ta_do;
      const double* v = rv + ((size_t)s * nb + j) * 4;
      double La = atm[s * NATM + A_LAF] * econv[j];
      double* t = toa3 + ((size_t)s * nb + j) * 3;
      toc_to_toa(so, v[0], v[1], v[3], v[2], La, t[0], t[1], t[2]);
    }
}

// cumulative LIDF at the 12 class boundaries: mode 0 = the reference's x-form with library sin / cos (lidf_dcum),
// 1 = the same iteration in u with the Taylor rotation, 2 = with the closed-form jump over its linear tail;
// jumped[i] = 1 where mode 2 took the jump
void hm_lidf_dcum(int mode, int64_t n, const double* a, const double* b, double* F /* (n, 12) */, int* jumped /* (n, 12) */) {
  for (int64_t s = 0; s < n; ++s)
    for (int i = 0; i < NLINCL - 1; ++i) {
      int j = 0;
      double f = mode == 0 ? lidf_dcum(a[s], b[s], lidf_theta(i))
                           : (mode == 1 ? lidf_dcum_lit_impl<false>(a[s], b[s], i) : lidf_dcum_lit_impl<true>(a[s], b[s], i, &j));
      F[s * (NLINCL - 1) + i] = f;
      jumped[s * (NLINCL - 1) + i] = j;
    }
}

// hot-spot integrals (sailh.py:115-135): taken[i] = 1 where the closed-form series applies (hotspot_series); ser (n,2) its
// two integrals; pan (n,2) the Gauss-Legendre panels with one halving MORE than the kernel uses (a tighter reference) and the
// below-canopy stretch in two panels
void hm_hotspot(int64_t n, const double* K, const double* k, const double* LAI, const double* q, const double* dso, int* taken,
                double* ser, double* pan) {
  const double dx = 1.0 / NLAYER;
  for (int64_t i = 0; i < n; ++i) {
    PsoFn f;
    f.hot = false;
    f.alpha = (dso[i] / q[i]) * 2.0 / (k[i] + K[i]);
    f.A = (K[i] + k[i]) * LAI[i];
    f.C = std::sqrt(K[i] * k[i]) * LAI[i] / f.alpha;
    ser[2 * i] = ser[2 * i + 1] = 0.0;
    taken[i] = hotspot_series(f.A, f.C, f.alpha, ser[2 * i], ser[2 * i + 1]) ? 1 : 0;
    double rate = std::fmax(f.alpha, f.A + std::sqrt(K[i] * k[i]) * LAI[i]), hw = 1.0, tot = 0.0, lo = -1.0;
    int m = 0;
    while (rate * hw > 1.0 && m < 44) { hw *= 0.5; ++m; }
    for (int j = 0; j < m; ++j) { tot += gl_panel<false>(f, lo, 0.5 * lo); lo *= 0.5; }
    pan[2 * i] = tot + gl_panel<false>(f, lo, 0.0);
    pan[2 * i + 1] = (gl_panel<false>(f, -1.0 - dx, -1.0 - 0.5 * dx) + gl_panel<false>(f, -1.0 - 0.5 * dx, -1.0)) / dx;
  }
}

void hm_log1p(int dtype, int64_t n, const double* x, double* out) {
  for (int64_t i = 0; i < n; ++i) out[i] = dtype == 0 ? (double)Mx<float>::log1p((float)x[i]) : Mx<double>::log1p(x[i]);
}

void hm_plate_tau(int dtype, int64_t n, const double* K, double* tau, double* u) {
  for (int64_t i = 0; i < n; ++i) {
    if (dtype == 0) { float t, uu; plate_tau<float>((float)K[i], t, uu); tau[i] = t; u[i] = uu; }
    else { double t, uu; plate_tau<double>(K[i], t, uu); tau[i] = t; u[i] = uu; }
  }
}
}

// Host side of libce_hip.so: descriptor validation, float64 derivation of every per-plan table,
// upload, launch and HIP-event timing.  C ABI declared in include/ce_hip.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <complex>
#include <string>
#include <vector>

#include "ce_plan.h"

#ifndef CE_LDS_BIG_BYTES
#define CE_LDS_BIG_BYTES (80 * 1024)   // two workgroups per CU
#endif
#ifndef CE_LDS_BIG_ITEMS
#define CE_LDS_BIG_ITEMS 8192          // work items from which a launch of the wide none / mean kernel asks for it
#endif

struct ce_plan {
  ce_plan_desc desc;
  CeDevPlan host;
  CeDevPlan* dev_plan = nullptr;
  uint16_t* dev_re_idx = nullptr;
  uint16_t* dev_ta_inv = nullptr;
  float2* dev_tw = nullptr;
  ce_plan_info info;
  int device = 0;
  int lds_big = 0;    // > 0: dynamic LDS requested for launches of >= CE_LDS_BIG_ITEMS work items (fewer workgroups per CU)
  std::vector<float> mmse_w;  // extension: Re | Im of W[m][k], CE_MMSE_BLOCK^2 each
};

thread_local std::string g_err;

// Tuning / A-B knobs (environment variables read when a plan is created).  They exist only in the diagnostic library
// (libce_hip_knobs.so, built with -DCE_TUNING_KNOBS; tests/test_hip_tiers.py and the dev tools load it through CE_HIP_LIB):
// the shipped libce_hip.so never looks at the environment, so a stray variable cannot change its kernel selection or LDS
// sizing.  Listed in include/ce_hip.h.
static inline const char* ce_knob(const char* name) {
#ifdef CE_TUNING_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

namespace {

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

}  // namespace

int ce_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

namespace {

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return fail(CE_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Raised-cosine taps: rcosdesign(0.2, n_rbs, 10) sampled every `stride`, unit sum (T:143-234).
std::vector<double> rc_taps(int stride, int n_rbs) {
  const double beta = 0.2;
  const int sps = 10, half_n = n_rbs * sps / 2, len = 2 * half_n + 1;
  std::vector<double> ff(len);
  for (int i = 0; i < len; ++i) {
    const double t = (double)(i - half_n) / sps;
    const double sinc = (t == 0.0) ? 1.0 : sin(M_PI * t) / (M_PI * t);
    const double den = 1.0 - (2.0 * beta * t) * (2.0 * beta * t);
    double h = sinc * cos(M_PI * beta * t) / den;
    if (!isfinite(h) || fabs(fabs(t) - 1.0 / (2.0 * beta)) < (1.0 / sps) * 1e-6)
      h = (M_PI * beta / 2.0) * sin(1.0 / (2.0 * beta));
    ff[i] = h;
  }
  const int half = len / 2, kmax = (half / stride) * stride, center = (len - 1) / 2;
  std::vector<double> taps;
  double sum = 0.0;
  for (int k = -kmax; k <= kmax; k += stride) {
    taps.push_back(ff[k + center]);
    sum += ff[k + center];
  }
  for (double& v : taps) v /= sum;
  return taps;
}

int popcount12(unsigned m) { return __builtin_popcount(m & 0xFFFu); }

// EXTENSION (CE_SMOOTH_MMSE): W = R (R + nsr I)^-1 for one block of m pilots at subcarriers sc[0..m), R from a
// uniform power-delay profile on [0, tau]: r(d) = sinc(d tau) exp(-j pi d tau), d = (sc_i - sc_j) * scs.
// A = R + nsr I is Hermitian, so W = (A^-1 R)^H: solve A Z = R by LU with partial pivoting, W = Z^H.
// w_out: Re then Im, each [CE_MMSE_BLOCK][CE_MMSE_BLOCK] row-major W[row m][col k], zero-padded.
bool mmse_matrix(const uint16_t* sc, int m, double scs, double tau, double nsr, float* w_out) {
  typedef std::complex<double> cd;
  std::vector<cd> A((size_t)m * m), Z((size_t)m * m);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      const double x = (double)((int)sc[i] - (int)sc[j]) * scs * tau;
      const double sinc = (x == 0.0) ? 1.0 : sin(M_PI * x) / (M_PI * x);
      const cd r = sinc * std::exp(cd(0.0, -M_PI * x));
      Z[(size_t)i * m + j] = r;
      A[(size_t)i * m + j] = r + (i == j ? nsr : 0.0);
    }
  for (int c = 0; c < m; ++c) {  // forward elimination on [A | Z]
    int piv = c;
    for (int r = c + 1; r < m; ++r)
      if (std::abs(A[(size_t)r * m + c]) > std::abs(A[(size_t)piv * m + c])) piv = r;
    if (std::abs(A[(size_t)piv * m + c]) < 1e-300) return false;
    if (piv != c)
      for (int j = 0; j < m; ++j) { std::swap(A[(size_t)c * m + j], A[(size_t)piv * m + j]); std::swap(Z[(size_t)c * m + j], Z[(size_t)piv * m + j]); }
    const cd inv = 1.0 / A[(size_t)c * m + c];
    for (int r = c + 1; r < m; ++r) {
      const cd f = A[(size_t)r * m + c] * inv;
      if (f == cd(0.0)) continue;
      for (int j = c; j < m; ++j) A[(size_t)r * m + j] -= f * A[(size_t)c * m + j];
      for (int j = 0; j < m; ++j) Z[(size_t)r * m + j] -= f * Z[(size_t)c * m + j];
    }
  }
  for (int c = m - 1; c >= 0; --c) {  // back substitution
    const cd inv = 1.0 / A[(size_t)c * m + c];
    for (int j = 0; j < m; ++j) {
      cd v = Z[(size_t)c * m + j];
      for (int k = c + 1; k < m; ++k) v -= A[(size_t)c * m + k] * Z[(size_t)k * m + j];
      Z[(size_t)c * m + j] = v * inv;
    }
  }
  memset(w_out, 0, sizeof(float) * 2 * CE_MMSE_BLOCK * CE_MMSE_BLOCK);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {  // W = Z^H
      const cd w = std::conj(Z[(size_t)j * m + i]);
      w_out[i * CE_MMSE_BLOCK + j] = (float)w.real();
      w_out[CE_MMSE_BLOCK * CE_MMSE_BLOCK + i * CE_MMSE_BLOCK + j] = (float)w.imag();
    }
  return true;
}

}  // namespace

// Routes a launch / prepare to the translation unit holding the plan's instantiation (ce_inst_*.hip).
static int kernel_op(int op, const CeDevPlan& P, const CeLaunchCtx& c) {
  if (P.narrow) return ce_tu_narrow(op, P.n_layers * 10 + P.n_hops, c);
  const int key = CE_KERNEL_KEY(P.feat, P.n_layers, P.reg_nd, P.reg_kpt);
  const bool two = P.n_hops == 2;
  if (P.reg_nd == 0 || P.feat == 3) return two ? ce_tu_gen_h2(op, key, c) : ce_tu_gen_h1(op, key, c);
  if (P.feat == 0) return two ? ce_tu_reg_h2_f0(op, key, c) : ce_tu_reg_h1_f0(op, key, c);
  return two ? ce_tu_reg_h2_f1(op, key, c) : P.reg_kpt >= 4 ? ce_tu_reg_h1_f1w(op, key, c) : ce_tu_reg_h1_f1(op, key, c);
}

#if defined(CE_STAMPS)
static unsigned long long* g_stamps = nullptr;  // diagnostic builds only (tools/stamps.py)
extern "C" int ce_debug_set_stamps(void* p) { g_stamps = (unsigned long long*)p; return 0; }
#endif

extern "C" {

const char* ce_last_error(void) { return g_err.c_str(); }
int ce_abi_version(void) { return CE_ABI_VERSION; }

static int plan_build(const ce_plan_desc* d, ce_plan** out, bool upload) {
  if (!d || !out) return fail(CE_ERR_INVALID, "null argument");
  *out = nullptr;
  if (d->abi_version != CE_ABI_VERSION) return fail(CE_ERR_INVALID, "ABI version %d != %d", d->abi_version, CE_ABI_VERSION);
  if (d->n_layers < 1 || d->n_layers > CE_MAX_LAYERS) return fail(CE_ERR_UNSUPPORTED, "n_layers=%d outside 1..%d", d->n_layers, CE_MAX_LAYERS);
  if (d->n_hops < 1 || d->n_hops > CE_MAX_HOPS) return fail(CE_ERR_INVALID, "n_hops=%d outside 1..2", d->n_hops);
  if (d->n_prb_grid < 1 || 12 * d->n_prb_grid > CE_FFT_SIZE)
    return fail(CE_ERR_UNSUPPORTED, "grid of %d PRB: the time-alignment IFFT (T:679) needs 12*n_prb <= %d", d->n_prb_grid, CE_FFT_SIZE);
  if (d->n_sym < 1 || d->n_sym > CE_MAX_SYMBOLS) return fail(CE_ERR_UNSUPPORTED, "n_sym=%d outside 1..%d", d->n_sym, CE_MAX_SYMBOLS);
  if (d->smoothing < CE_SMOOTH_NONE || d->smoothing > CE_SMOOTH_MMSE) return fail(CE_ERR_INVALID, "Unknown smoothing strategy %d.", d->smoothing);
  if (d->interp != CE_INTERP_LINEAR && d->interp != CE_INTERP_CNN) return fail(CE_ERR_INVALID, "unknown interp %d", d->interp);
  if (!(d->scs_hz > 0) || !(d->beta_dmrs > 0)) return fail(CE_ERR_INVALID, "scs and beta_dmrs must be positive");

  ce_plan* p = new (std::nothrow) ce_plan();
  if (!p) return fail(CE_ERR_NOMEM, "out of host memory");
  p->desc = *d;
  p->device = d->device;
  CeDevPlan& P = p->host;
  memset(&P, 0, sizeof(P));
  const int L = d->n_layers, n_cdm = (L + 1) / 2, n_sc = 12 * d->n_prb_grid;
  P.n_sc = n_sc; P.n_sym = d->n_sym; P.n_layers = L; P.n_cdm = n_cdm; P.n_hops = d->n_hops;
  P.smoothing = d->smoothing; P.cfo_comp = d->cfo_compensate ? 1 : 0; P.interp = d->interp;
  P.beta = d->beta_dmrs; P.beta_f = (float)d->beta_dmrs; P.scs = d->scs_hz; P.denom_cdm = (double)n_cdm;

  // symbolStartTime = cumsum([CPD0, CPD1..13 + 1]), CPD = cp_ms*scs/1000 (T:809-820)
  {
    double acc = 0.0;
    for (int s = 0; s < CE_MAX_SYMBOLS; ++s) {
      const double cpd = d->cp_ms[s] * d->scs_hz / 1000.0;
      acc += (s == 0) ? cpd : cpd + 1.0;
      P.sst[s] = acc;
    }
  }

  std::vector<uint16_t> re_idx;
  std::vector<uint16_t> ta_inv((size_t)d->n_hops * CE_FFT_SIZE, 0xFFFFu);
  int n_re = -1, n_dmrs_total = 0, cfo_estimated = 0;
  uint8_t seen_sym[CE_MAX_SYMBOLS] = {0};
  for (int h = 0; h < d->n_hops; ++h) {
    const ce_hop_desc& hd = d->hop[h];
    CeDevHop& H = P.hop[h];
    if (!hd.mask_prbs) { delete p; return fail(CE_ERR_INVALID, "hop %d: mask_prbs is null", h); }
    for (int s = 0; s < d->n_sym; ++s)
      if (hd.dmrs_symbols[s]) {
        if (seen_sym[s]) { delete p; return fail(CE_ERR_INVALID, "Hops should not overlap."); }
        seen_sym[s] = 1;
        H.dmrs_sym[H.n_dmrs++] = s;
      }
    if (H.n_dmrs < 1) { delete p; return fail(CE_ERR_INVALID, "hop %d has no DM-RS symbol", h); }
    if (h == 1 && (hd.re_mask[0] != d->hop[0].re_mask[0] || (n_cdm > 1 && hd.re_mask[1] != d->hop[0].re_mask[1]))) {
      delete p;
      return fail(CE_ERR_INVALID, "The DM-RS mask should be the same for the two hops.");
    }
    for (int i = 0; i < H.n_dmrs; ++i) P.sst_dmrs[h][i] = P.sst[H.dmrs_sym[i]];
    H.pil_sym0 = n_dmrs_total;
    n_dmrs_total += H.n_dmrs;
    H.has_cfo = H.n_dmrs >= 2;
    cfo_estimated |= H.has_cfo;
    if (hd.prb_start < 0 || hd.n_prbs < 1 || hd.prb_start + hd.n_prbs > d->n_prb_grid) {
      delete p;
      return fail(CE_ERR_INVALID, "hop %d: PRBstart=%d nPRBs=%d outside the %d-PRB grid", h, hd.prb_start, hd.n_prbs, d->n_prb_grid);
    }
    if (hd.start_symbol < 0 || hd.n_alloc_symbols < 0 || hd.start_symbol + hd.n_alloc_symbols > d->n_sym) {
      delete p;
      return fail(CE_ERR_INVALID, "hop %d: symbols %d..+%d outside the %d-symbol grid", h, hd.start_symbol, hd.n_alloc_symbols, d->n_sym);
    }
    H.sc0 = 12 * hd.prb_start; H.n_sc_hop = 12 * hd.n_prbs;
    H.sym0 = hd.start_symbol; H.sym1 = hd.start_symbol + hd.n_alloc_symbols;
    int n_active = 0;
    for (int q = 0; q < d->n_prb_grid; ++q) n_active += hd.mask_prbs[q] ? 1 : 0;
    if (n_active != hd.n_prbs) {
      delete p;
      return fail(CE_ERR_INVALID, "hop %d: sum(maskPRBs)=%d != nPRBs=%d (the reference's grid fill T:291-292 needs them equal)", h, n_active, hd.n_prbs);
    }
    H.prb_start = hd.prb_start; H.n_prbs = hd.n_prbs;
    H.contig = 1;
    for (int q = 0; q < d->n_prb_grid; ++q)
      if ((hd.mask_prbs[q] != 0) != (q >= hd.prb_start && q < hd.prb_start + hd.n_prbs)) H.contig = 0;
    for (int c = 0; c < n_cdm; ++c) {
      const unsigned m = hd.re_mask[c] & 0xFFFu;
      const int dpp = popcount12(m);
      if (dpp == 0) { delete p; return fail(CE_ERR_INVALID, "hop %d: DMRSREmask column %d is empty", h, c); }
      H.dpp[c] = dpp;
      H.mask12 |= m << (16 * c);
      H.div_magic[c] = dpp > 1 ? (uint32_t)(0x100000000ull / (unsigned)dpp) + 1u : 0u;  // dpp == 1: k / dpp == k (the magic would need 33 bits)
      {
        uint64_t pos = 0, ord = 0;
        int j = 0;
        for (int r = 0; r < 12; ++r) {
          if (m >> r & 1) { pos |= (uint64_t)r << (4 * j); ord |= (uint64_t)j << (4 * r); ++j; }
          else ord |= (uint64_t)15 << (4 * r);
        }
        H.pos_packed[c] = pos;
        if (c == n_cdm - 1) H.ord_packed = ord;
      }
      H.re_off[c] = (int)re_idx.size();
      // maskREs = kron(maskPRBs, DMRSREmask[:, c]) (T:572-576)
      for (int q = 0; q < d->n_prb_grid; ++q)
        if (hd.mask_prbs[q])
          for (int r = 0; r < 12; ++r)
            if (m >> r & 1) re_idx.push_back((uint16_t)(12 * q + r));
      const int cnt = (int)re_idx.size() - H.re_off[c];
      if (n_re < 0) n_re = cnt;
      if (cnt != n_re) {
        delete p;
        return fail(CE_ERR_INVALID, "hop %d CDM %d has %d pilot REs, expected %d (pilots.shape[0] is shared)", h, c, cnt, n_re);
      }
      // interpolation anchors, periodic in the PRB (T:311-338)
      int first_r = 0, last_r = 11;
      while (!(m >> first_r & 1)) ++first_r;
      while (!(m >> last_r & 1)) --last_r;
      H.last_idx[c] = 12 * (hd.n_prbs - 1) + last_r;
      for (int r = 0; r < 12; ++r) {
        int rr = r;  // right anchor: first pilot at or after r (possibly in the next PRB)
        while (rr < 12 && !(m >> rr & 1)) ++rr;
        int right_pos, ord;
        if (rr < 12) { right_pos = rr; ord = popcount12(m & ((1u << rr) - 1u)); }
        else { right_pos = 12 + first_r; ord = dpp; }
        int lp = (rr < 12 ? rr : 12) - 1;  // left anchor: the pilot before the right anchor
        while (lp >= 0 && !(m >> lp & 1)) --lp;
        const int left_pos = lp >= 0 ? lp : last_r - 12;
        H.r_ord[c][r] = ord;
        H.alpha[c][r] = (float)(r - left_pos) / (float)(right_pos - left_pos);
      }
    }
    // time-alignment scatter uses the LAST CDM group's RE list for every layer (T:672-675)
    {
      H.ta_inv_off = h * CE_FFT_SIZE;
      unsigned seen = 0;
      int pos_min = CE_FFT_SIZE, pos_max = 0;
      for (int k = 0; k < n_re; ++k) {
        const int pos = re_idx[H.re_off[n_cdm - 1] + k];
        ta_inv[(size_t)h * CE_FFT_SIZE + pos] = (uint16_t)k;
        seen |= 1u << (pos & 15);
        pos_min = std::min(pos_min, pos);
        pos_max = std::max(pos_max, pos);
      }
      for (int r = 0; r < 16; ++r)
        if (seen >> r & 1) { H.ta_res_packed |= (uint64_t)r << (4 * H.ta_nres); H.ta_res[H.ta_nres++] = r; }
      // Narrow bands: |X[k]| does not change when the band is moved down by a multiple of 16 subcarriers (a unit phase per
      // bin, the residues stay), and a band that then ends below subcarrier 256 / 512 has one / two non-zero inputs per
      // 16-point transform of the first radix-16 pass, which collapses to a twiddle multiply (ce_estimate_kernel.h).
      const int shift = pos_min & ~15, nb = (pos_max - shift) / 256 + 1;
      // (the collapsed pass probes subcarriers shift .. shift + 256 nb - 1 of this hop's subcarrier -> pilot table: a band at
      // the top of a grid wider than 3840 subcarriers must keep those inside the table's CE_FFT_SIZE entries)
      H.ta_win = (nb <= 2 && shift + 256 * nb <= CE_FFT_SIZE && !ce_knob("CE_TA_FULL")) ? (uint32_t)shift | ((uint32_t)nb << 16) : 0u;
    }
    // nSamples = nSyms + sum(CPDs(i0+1 .. i1)), CPDs = cp_ms * (scs/1000) (T:395-426, called with scs/1000 at T:599)
    if (H.has_cfo) {
      const int i0 = H.dmrs_sym[0], i1 = H.dmrs_sym[1];
      double cp_sum = 0.0;
      for (int s = i0 + 1; s <= i1 && s < CE_MAX_SYMBOLS; ++s) cp_sum += d->cp_ms[s] * (d->scs_hz / 1000.0);
      H.two_pi_nsamples = 2.0 * M_PI * ((double)(i1 - i0) + cp_sum);
      H.inv_two_pi_nsamples = 1.0 / H.two_pi_nsamples;
    }
  }
  P.n_re = n_re; P.n_re_pad = (n_re + 1) & ~1;
  P.cfo_estimated = cfo_estimated;
  if (P.cfo_comp && cfo_estimated && d->n_sym != CE_MAX_SYMBOLS) {
    delete p;
    return fail(CE_ERR_INVALID, "CFO compensation needs a 14-symbol grid (T:928-929), got %d", d->n_sym);
  }

  // nPilots = hop1.nPRBs * sum(hop1.DMRSREmask(:,1)) * nDMRSsymbols (T:898-915)
  const double n_pilots = (double)(d->hop[0].n_prbs * P.hop[0].dpp[0] * n_dmrs_total);
  P.n_pilots = n_pilots;
  P.noise_den = (double)n_cdm * n_pilots - 1.0;
  P.inv_n_pilots = 1.0 / n_pilots; P.inv_layers = 1.0 / (double)L; P.inv_noise_den = 1.0 / P.noise_den;
  P.inv_denom_cdm = 1.0 / (double)n_cdm;

  if (d->smoothing == CE_SMOOTH_FILTER) {
    const int dpp0 = P.hop[0].dpp[0];
    // stride = 12 // pilots-per-PRB, floor division as the reference does (T:640) even when it does not divide 12
    const int n_active = d->hop[0].n_prbs;
    std::vector<double> rc = rc_taps(12 / dpp0, n_active < 3 ? n_active : 3);
    if ((int)rc.size() > CE_MAX_RC_TAPS) { delete p; return fail(CE_ERR_UNSUPPORTED, "%zu RC taps", rc.size()); }
    P.rc_len = (int)rc.size();
    for (size_t i = 0; i < rc.size(); ++i) P.rc[i] = rc[i];
    P.n_pils = n_active > 1 ? ((int)rc.size() / 2 < 12 ? (int)rc.size() / 2 : 12) : dpp0;  // T:644-647
    // n_pils == 0 (a one-tap filter: stride 12 over two PRBs) is valid: no virtual pilots, identity FIR (T:649-664)
    if (P.n_pils > n_re) { delete p; return fail(CE_ERR_UNSUPPORTED, "n_pils=%d vs n_re=%d", P.n_pils, n_re); }
    P.ext_len = n_re + 2 * P.n_pils;
    for (size_t i = 0; i < rc.size(); ++i) P.rcz[i + CE_CONV_C - 1] = rc[i];
    {
      const double n = (double)P.n_pils;
      double sxx = 0.0;
      for (int i = 0; i < P.n_pils; ++i) sxx += (double)i * (double)i;
      P.vp_mx = (n - 1.0) / 2.0;
      P.vp_inv_n = P.n_pils > 0 ? 1.0 / n : 0.0;
      P.vp_inv_denom = P.n_pils > 1 ? 1.0 / (sxx - n * P.vp_mx * P.vp_mx) : 0.0;
    }
    // windowed FIR: band fits 9 outputs x 192 threads, both edge zones (len(rc)/2 outputs each) are disjoint
    P.filt_windowed = (n_re <= (CE_THREADS - 64) * CE_CONV_C && P.n_pils <= 12 && n_re >= 2 * ((int)rc.size() / 2) &&
                       (int)rc.size() == 15) ? 1 : 0;
  }

  if (d->smoothing == CE_SMOOTH_MMSE) {
    // every block of CE_MMSE_BLOCK pilots must see the same pilot spacing pattern (one W serves them all)
    const int m = n_re < CE_MMSE_BLOCK ? n_re : CE_MMSE_BLOCK;
    P.mmse_nb = (n_re + m - 1) / m;
    P.mmse_nbp = (P.mmse_nb + 15) & ~15;
    for (int h = 0; h < d->n_hops; ++h)
      for (int c = 0; c < n_cdm; ++c) {
        const uint16_t* sc = re_idx.data() + P.hop[h].re_off[c];
        for (int b = 0; b < P.mmse_nb; ++b) {
          const int s0 = b * m < n_re - m ? b * m : n_re - m;
          for (int i = 0; i < m; ++i)
            if ((int)sc[s0 + i] - (int)sc[s0] != (int)re_idx[P.hop[0].re_off[0] + i] - (int)re_idx[P.hop[0].re_off[0]]) {
              delete p;
              return fail(CE_ERR_UNSUPPORTED, "mmse smoothing needs the same pilot spacing in every block of %d pilots", m);
            }
        }
      }
    if (!(d->mmse_delay_spread_s >= 0.0) || !(d->mmse_noise_to_signal > 0.0)) { delete p; return fail(CE_ERR_INVALID, "mmse: delay spread must be >= 0 and noise-to-signal > 0"); }
    p->mmse_w.resize(2 * CE_MMSE_BLOCK * CE_MMSE_BLOCK);
    if (!mmse_matrix(re_idx.data() + P.hop[0].re_off[0], m, d->scs_hz, d->mmse_delay_spread_s, d->mmse_noise_to_signal, p->mmse_w.data())) {
      delete p;
      return fail(CE_ERR_INVALID, "mmse: singular correlation matrix");
    }
  }

  // LDS scratch: TA residue blocks | virtual-pilot-extended band for the RC FIR | writer's H chunk
  {
    int need = 0;
    for (int h = 0; h < d->n_hops; ++h) need = P.hop[h].ta_nres * CE_TA_ROW * 8 > need ? P.hop[h].ta_nres * CE_TA_ROW * 8 : need;  // (the unpadded layout needs less: ta_lp below)
    if (d->smoothing == CE_SMOOTH_FILTER && P.ext_len * 8 > need) need = P.ext_len * 8;
    if (P.n_hops * L * 256 * 8 > need) need = P.n_hops * L * 256 * 8;
    if (d->smoothing == CE_SMOOTH_MMSE) {  // W^T (Re, Im) + X^T (Re, Im): [32][32] and [32][nbp] floats each
      const int mm = (2 * CE_MMSE_BLOCK * CE_MMSE_BLOCK + 2 * CE_MMSE_BLOCK * P.mmse_nbp) * 4;
      if (mm > need) need = mm;
    }
    P.scratch_bytes = need;
    if (d->smoothing == CE_SMOOTH_FILTER) {
      P.filt_lpp = need / (P.ext_len * 8);
      if (P.filt_lpp > L) P.filt_lpp = L;
    }
    int lg = 8;
    while (lg < 12 && P.n_hops * L * (2 << lg) * 8 <= need) ++lg;
    P.wr_ch_log2 = lg;
    // A comb-2 DM-RS (every other RE, either offset) makes the partial-convolution in-painting of C:473-508 reach its
    // fixed point -- the mean of the two neighbouring pilots -- after two iterations (the float32 round trip of C:501
    // absorbs the 1/(1+1e-12) factor of the remaining max(6, n/8) - 2), and the two low-pass passes then give
    // (P[k-1] + 15 P[k] + 15 P[k+1] + P[k+2]) / 32 with reflected pilot indices at the band edges: the writer evaluates
    // that straight from P, no whole-band staging (measured against the real ce_dl_cnn.py fixtures like the general form).
    // Any other mask: the iteration x <- (x[i-1] + 2 x[i] + x[i+1]) / 4 on the unknown REs converges to the straight line
    // between the neighbouring pilots (flat beyond the first / last one: reflect padding), with factor
    // cos^2(pi / (2 (g + 1))) per iteration for a run of g unknowns (an edge run of e counts as 2 e - 1, mirrored).  When
    // max(6, n / 8) iterations bring that below 1e-6 (20x inside the parity tolerance) the reference sits on the fixed point, and
    // in-painting + low-pass is the 5-tap binomial [1 4 6 4 1] / 16 over the linear fill of T:311-338, pilots restored
    // (mode 2).  Shorter bands, or sparser masks, depend on the exact iteration count and are iterated as before.
    P.cnn_comb2 = 0;
    if (d->interp == CE_INTERP_CNN && !ce_knob("CE_CNN_GENERAL")) {
      bool comb2 = true, converges = true;
      for (int h = 0; h < d->n_hops; ++h)
        for (int c = 0; c < n_cdm; ++c) {
          const unsigned m12 = (P.hop[h].mask12 >> (16 * c)) & 0xFFFu;
          if (m12 != 0x555u && m12 != 0xAAAu) comb2 = false;
          if (m12 == 0xFFFu) { converges = false; continue; }   // every RE a pilot: low-pass only (C:487-488), general form
          int first = 0, last = 11, g = 0, run = 0;
          while (!((m12 >> first) & 1u)) ++first;
          while (!((m12 >> last) & 1u)) --last;
          for (int r = first; r <= last; ++r) {
            if ((m12 >> r) & 1u) run = 0; else if (++run > g) g = run;
          }
          const int wrap = (11 - last) + first;                   // run across a PRB boundary
          if (P.hop[h].n_prbs > 1 && wrap > g) g = wrap;
          if (2 * first - 1 > g) g = 2 * first - 1;               // band edges, mirrored
          if (2 * (11 - last) - 1 > g) g = 2 * (11 - last) - 1;
          const int n_it = P.hop[h].n_sc_hop / 8 > 6 ? P.hop[h].n_sc_hop / 8 : 6;
          const double cs = cos(M_PI / (2.0 * (g + 1)));
          if (g > 0 && log(1e-6) / log(cs * cs) > (double)n_it) converges = false;
        }
      P.cnn_comb2 = comb2 ? 1 : converges ? 2 : 0;
    }
    if (CE_CNNFP_STAGED && P.cnn_comb2 == 2) {
      // the staged writer evaluates the binomial from a staged linear fill: half the scratch for the H chunk, the other half
      // (+ two subcarriers either side, per hop and layer) for the fill (ce_estimate_kernel.h: the staged writer)
      P.wr_ch_log2 = lg - 1;
      P.scratch_bytes = std::max(P.scratch_bytes, P.n_hops * L * ((2 << P.wr_ch_log2) + 4) * 8);
    }
    if (d->interp == CE_INTERP_CNN && !P.cnn_comb2) {
      // band-relative H rows for every (hop, layer) + a second x buffer + two mask byte arrays; when all rows together
      // would not fit the LDS (many layers of wide hops) the writer in-paints and stores one row at a time
      int n_max = 0;
      for (int h = 0; h < d->n_hops; ++h) n_max = P.hop[h].n_sc_hop > n_max ? P.hop[h].n_sc_hop : n_max;
      P.cnn_n_max = n_max;
      P.cnn_h_stride = (n_max + 1) & ~1;
      const int aux = ((n_max + 1) & ~1) * 8;
      int rows = P.n_hops * L;
      const int fixed = ce_lds_layout(P.n_hops, L, (n_re + 1) & ~1, 0).total;
      if (fixed + rows * P.cnn_h_stride * 8 + aux > 160 * 1024 - 256) { rows = 1; P.cnn_rowwise = 1; }
      const int h_bytes = rows * P.cnn_h_stride * 8;
      P.cnn_pong_off = h_bytes;
      const int cnn_need = h_bytes + aux;
      if (cnn_need > P.scratch_bytes) P.scratch_bytes = cnn_need;
      // longest run of unknown REs between pilots, across a PRB boundary or at a band edge: the in-painting iterates one
      // run per thread in registers (ce_estimate_kernel.h: cnn_inpaint_runs)
      int gmax = 0;
      for (int h = 0; h < d->n_hops; ++h)
        for (int c = 0; c < n_cdm; ++c) {
          const unsigned m12 = (P.hop[h].mask12 >> (16 * c)) & 0xFFFu;
          int first = 0, last = 11, run = 0;
          while (!((m12 >> first) & 1u)) ++first;
          while (!((m12 >> last) & 1u)) --last;
          for (int r = first; r <= last; ++r) {
            if ((m12 >> r) & 1u) run = 0; else if (++run > gmax) gmax = run;
          }
          if ((11 - last) + first > gmax) gmax = (11 - last) + first;
        }
      P.cnn_gmax = gmax;
    }
    if (d->interp == CE_INTERP_CNN) {
      double a = d->cnn_smoothing_alpha;
      P.cnn_alpha = (float)(a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a));
      for (int c = 0; c < 5; ++c) P.cnn_rcp[c] = 1.0 / (0.25 * c + 1e-12);
    }
  }
  // Two hops whose fill rectangles share OFDM symbols (the reference harness describes both hops of a hopping
  // allocation with the slot's whole symbol range, scripts/validation/validate_case4.py:85-103): which hop an
  // element belongs to then depends on its subcarrier as well, which only the element-wise writer resolves (for
  // either interpolator: src/ce_dl_cnn.py:233-352 overwrites the same way)
  P.sym_overlap = (d->n_hops == 2 && std::max(P.hop[0].sym0, P.hop[1].sym0) < std::min(P.hop[0].sym1, P.hop[1].sym1)) ? 1 : 0;
  // register path: one layer (two layers' pilots would spill: measured slower than re-reading), every hop
  // with the same DM-RS symbol count, band fits CE_KPT pilot REs per thread
  P.reg_kpt = ce_knob("CE_FORCE_WIDE") ? CE_KPT : (n_re <= CE_THREADS ? 1 : n_re <= 2 * CE_THREADS ? 2 : n_re <= 4 * CE_THREADS ? 4 : CE_KPT);
  P.reg_nd = 0;
  if (L == 1 && n_re <= CE_KPT * CE_THREADS) {
    const int nd = P.hop[0].n_dmrs;
    // received pilots of nd symbols x reg_kpt REs per thread stay in registers (the DM-RS symbols too while they
    // fit, ce_estimate_kernel.h: PREG): up to 4 symbols in the narrow-band kernels, 3 in the wide one
    bool same = nd <= (P.reg_kpt < CE_KPT ? 4 : 3);
    for (int h = 1; h < d->n_hops; ++h) same = same && P.hop[h].n_dmrs == nd;
    if (same) P.reg_nd = nd;
  }
  if (ce_knob("CE_FORCE_GENERIC")) P.reg_nd = 0;  // always take the re-read path
  // feature set the kernel must carry (ce_estimate_kernel.h): the register-path kernels are built without the
  // extensions (one exception: the wide 2-symbol shape), plans that need them take the re-read path
  P.feat = d->smoothing == CE_SMOOTH_FILTER ? 1 : 0;
  if (d->smoothing == CE_SMOOTH_MMSE || (d->interp == CE_INTERP_CNN && !P.cnn_comb2)) P.feat = 3;
  if (P.feat == 3 && P.reg_nd > 0 && !ce_reg_has_ext(P.n_hops, P.reg_nd, P.reg_kpt)) P.reg_nd = 0;

  // Two TA transforms side by side -- the layers of a multi-layer hop, or the two hops of a one-layer item -- when at most 8
  // residues carry pilots (threads 128-255 are idle in the radix-16 passes then) and the second set of residue blocks
  // fits the LDS share that the kernel's register budget (ce_min_waves) leaves to a workgroup anyway
  P.ta_lp = 1;
  {
    int nres_max = 0;
    for (int h = 0; h < d->n_hops; ++h) nres_max = std::max(nres_max, (int)P.hop[h].ta_nres);
    const int kpt = P.reg_nd ? P.reg_kpt : CE_KPT;
    const bool late = ce_ta_late(L, P.n_hops, P.reg_nd, kpt, P.feat & 1);
    const int sb2 = std::max(P.scratch_bytes, 2 * 8 * ce_ta_row(late) * 8);
    const int waves = ce_min_waves(P.n_hops, P.reg_nd, kpt, P.feat & 1, L);
    const bool shape = L >= 2 || (d->n_hops == 2 && late);
    if (shape && !ce_knob("CE_TA_LP1") && nres_max <= 8 &&
        ((ce_lds_layout(P.n_hops, L, P.n_re_pad, sb2).total + 2047) & ~2047) * waves <= 160 * 1024) {  // LDS is granted in 2 KB steps (measured: 3 x 52 128 B fit a CU, 3 x 54 176 B do not)
      P.ta_lp = 2;
      P.scratch_bytes = sb2;
    } else if (CE_TA_OVER_P && L >= 2 && d->n_hops == 2 && late && !ce_knob("CE_TA_LP1") && nres_max <= 8 && P.scratch_bytes >= 8 * ce_ta_row(late) * 8 &&
               L * P.n_re_pad * 8 >= 8 * ce_ta_row(late) * 8) {
      P.ta_over_p = 1;   // (e.g. 4 layers x 2 hops x 136 PRB: 52 KB of P, two workgroups per CU whatever the scratch -- six transform rounds instead of eight)
    }
  }
  // Register-path kernels that cannot also hold the DM-RS symbols in registers (ce_pilots_in_regs) park the current hop's
  // in the scratch behind whatever the smoothing stage uses there, when that fits the kernel's LDS share
  P.pil_stash = 0;
  if (P.reg_nd >= 2 && !ce_pilots_in_regs(P.n_hops, P.reg_nd, P.reg_kpt) && !ce_knob("CE_NO_PIL_STASH")) {
    int smooth_need = 512;  // windowed FIR: virtual pilots of up to two rows
    if (d->smoothing == CE_SMOOTH_FILTER && !P.filt_windowed) smooth_need = P.filt_lpp * P.ext_len * 8;
    if (d->interp == CE_INTERP_CNN && P.cnn_alpha > 0.f) smooth_need = std::max(smooth_need, n_re * 8);
    smooth_need = (smooth_need + 15) & ~15;
    const int waves = ce_min_waves(P.n_hops, P.reg_nd, P.reg_kpt, P.feat & 1, L);
    for (int nd_st = P.reg_nd; nd_st >= 1; --nd_st) {  // as many of the hop's symbols as fit; the rest is re-read in the residual stage
      const int sb = std::max(P.scratch_bytes, smooth_need + nd_st * L * P.n_re_pad * 8);
      if (((ce_lds_layout(P.n_hops, L, P.n_re_pad, sb).total + 2047) & ~2047) * waves <= 160 * 1024) {
        P.pil_stash = (smooth_need / 8) | (nd_st << 24);
        P.scratch_bytes = sb;
        break;
      }
    }
  }
  // Narrow allocations run on the wave-per-item kernel (ce_narrow_kernel.h): four items per workgroup, one wave each.
  // What it covers: linear interpolation and the closed forms of ce_dl_cnn's in-painting (cnn_comb2 != 0), none / mean / filter, 14- and
  // 12-symbol grids, at most CE_NARROW_MAX_RE pilots per symbol, at most four DM-RS symbols per hop, every hop's band inside the collapsed time-alignment window (ta_win: scattered PRB masks may span more), and a
  // workgroup's LDS (plan + twiddles + 4 x {staged hop, P, tables}) within CE_NARROW_LDS_LIMIT.  Everything else -- and
  // every plan when the diagnostic build sees CE_NO_NARROW -- takes the workgroup-per-item kernels.
  {
    int nd_max = 0;
    bool win_ok = true;
    for (int h = 0; h < d->n_hops; ++h) {
      nd_max = std::max(nd_max, (int)P.hop[h].n_dmrs);
      win_ok = win_ok && P.hop[h].ta_win != 0u;
    }
    int h_max = 0;
    for (int h = 0; h < d->n_hops; ++h) h_max = std::max(h_max, (int)P.hop[h].n_sc_hop);
    P.nrw_nd_max = nd_max;
    P.nrw_h_stride = (h_max + 1) & ~1;
    P.nrw_magic_nre = (uint32_t)(0x100000000ull / (unsigned)n_re) + 1u;
    // halo of a P row = what the RC filter reaches beyond the band: len(rc) / 2 taps, of which n_pils carry virtual pilots (even, so rows stay 16-byte aligned)
    P.nrw_halo = d->smoothing == CE_SMOOTH_FILTER ? ((std::max(P.rc_len / 2, P.n_pils) + 1) & ~1) : 0;
    const CeNarrowLayout nl = ce_narrow_layout(P.n_hops, L, nd_max, P.n_re_pad, P.nrw_h_stride, P.nrw_halo);
    // Where it pays (in-process A/B over tools/perf_cases.py, profiles/round3_narrow_kernel_ab.txt): every two-hop and every
    // multi-layer narrow shape (-13 ... -52 %), and one-hop one-layer allocations of a few PRB (3 PRB: -6 %); from about 6 PRB
    // on, the one-hop one-layer register tiers of ce_estimate_kernel.h (five workgroups per CU) are 3-8 % faster and keep the plan.
    // 12-symbol grids (extended CP): every narrow plan (one hop, one layer, 25 PRB: 0.422 vs 0.435 ms through the workgroup kernels' 12-symbol writers).
    const bool pays = d->n_hops == 2 || L >= 2 || n_re <= CE_NARROW_1H1L_MAX_RE || d->n_sym == 12 || ce_knob("CE_FORCE_NARROW");
    P.narrow = ((d->interp == CE_INTERP_LINEAR || P.cnn_comb2 != 0) && d->smoothing != CE_SMOOTH_MMSE && (d->n_sym == CE_MAX_SYMBOLS || d->n_sym == 12) && n_re <= CE_NARROW_MAX_RE &&
                nd_max <= 4 && pays && win_ok && nl.total <= CE_NARROW_LDS_LIMIT && !ce_knob("CE_NO_NARROW")) ? 1 : 0;   // (its per-hop phasor tables hold four DM-RS symbols)
  }
  CeLdsLayout lay = ce_lds_layout(P.n_hops, L, P.n_re_pad, P.scratch_bytes);
  if (P.narrow) lay.total = ce_narrow_layout(P.n_hops, L, P.nrw_nd_max, P.n_re_pad, P.nrw_h_stride, P.nrw_halo).total;
#ifdef CE_LDS_PAD_DEFAULT   // A/B builds (tools/ab_inproc.py loads several libraries into one process, which share the environment)
  lay.total += CE_LDS_PAD_DEFAULT & ~15;
#endif
  if (const char* pad = ce_knob("CE_LDS_PAD_BYTES")) lay.total += atoi(pad) & ~15;  // lowers the workgroups resident per CU
  if (lay.total > 160 * 1024) { delete p; return fail(CE_ERR_UNSUPPORTED, "plan needs %d B of LDS (> 160 KiB)", lay.total); }

  ce_plan_info& I = p->info;
  I.n_sc = n_sc; I.n_re = n_re; I.n_dmrs_total = n_dmrs_total; I.cfo_estimated = cfo_estimated;
  I.lds_bytes = lay.total; I.threads = CE_THREADS;
  I.alg_bytes_per_item = (int64_t)n_re * n_dmrs_total * 8 * n_cdm + (int64_t)n_sc * d->n_sym * L * 8;
  I.pilot_bytes_per_slot = (int64_t)n_re * n_dmrs_total * L * 8;

  // IFFT twiddles exp(+j*2*pi*m/4096), float64 -> float32
  std::vector<float2> tw(CE_TW_TOTAL);  // + W^T (Re | Im) for the mmse extension + the TA transform's 272, contiguous (ce_plan.h)
  for (int m = 0; m < CE_FFT_SIZE; ++m) {
    const double a = 2.0 * M_PI * (double)m / (double)CE_FFT_SIZE;
    tw[m] = make_float2((float)cos(a), (float)sin(a));
  }
  for (int j = 0; j < 256; ++j) tw[CE_TWC_OFF + j] = tw[16 * j];
  for (int i = 0; i < 16; ++i) tw[CE_TWC_OFF + 256 + i] = tw[i];
  if (!p->mmse_w.empty()) {
    float* wt = reinterpret_cast<float*>(tw.data() + CE_FFT_SIZE);  // [2][k][m]
    for (int part = 0; part < 2; ++part)
      for (int k = 0; k < CE_MMSE_BLOCK; ++k)
        for (int m = 0; m < CE_MMSE_BLOCK; ++m)
          wt[part * CE_MMSE_BLOCK * CE_MMSE_BLOCK + k * CE_MMSE_BLOCK + m] = p->mmse_w[part * CE_MMSE_BLOCK * CE_MMSE_BLOCK + m * CE_MMSE_BLOCK + k];
  }

  if (!upload) {  // host-only derivation (ce_plan_derive_host): no HIP call at all
    *out = p;
    return CE_OK;
  }
  CeDeviceScope scope(d->device);  // the caller's current device is restored on return
  hipError_t e = scope.err;
  if (e == hipSuccess) e = hipMalloc(&p->dev_plan, sizeof(CeDevPlan));
  if (e == hipSuccess) e = hipMalloc(&p->dev_re_idx, re_idx.size() * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMalloc(&p->dev_tw, tw.size() * sizeof(float2));
  if (e == hipSuccess) e = hipMalloc(&p->dev_ta_inv, ta_inv.size() * sizeof(uint16_t));
  if (e == hipSuccess) e = hipMemcpy(p->dev_ta_inv, ta_inv.data(), ta_inv.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->dev_plan, &P, sizeof(CeDevPlan), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->dev_re_idx, re_idx.data(), re_idx.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->dev_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice);
  int blocks_per_cu = 1;
  if (e == hipSuccess) {
    CeLaunchCtx c = {};
    c.lds = lay.total;
    c.blocks_per_cu = &blocks_per_cu;
    const int kr = kernel_op(CE_OP_PREPARE, P, c);
    if (kr < 0) {
      fail(CE_ERR_UNSUPPORTED, "no kernel for (layers %d, hops %d, reg_nd %d, kpt %d, feat %d, narrow %d)", L, P.n_hops, P.reg_nd, P.reg_kpt, P.feat, P.narrow);
      ce_plan_destroy(p);
      return CE_ERR_UNSUPPORTED;
    }
    e = (hipError_t)kr;
  }
  // The wide single-hop none / mean kernel needs 114 VGPRs and little LDS: four workgroups fit a CU, but from a few rounds of
  // work on it runs 2-4 % faster with two (in-process A/B: 2048 x 4 items -4.3 %, 8192 x 4 -2.1 %; 1024 x 1 +2.5 %), so large
  // launches request as much dynamic LDS as leaves room for two.  Placement only: results are unaffected.
  if (e == hipSuccess && !P.narrow && P.n_hops == 1 && L == 1 && P.reg_nd == 2 && P.reg_kpt == CE_KPT && P.feat == 0 && lay.total <= CE_LDS_BIG_BYTES &&
      !ce_knob("CE_NO_LDS_BIG")) {
    int nb2 = 1;
    CeLaunchCtx c2 = {};
    c2.lds = CE_LDS_BIG_BYTES;
    c2.blocks_per_cu = &nb2;
    if (kernel_op(CE_OP_PREPARE, P, c2) == 0) p->lds_big = CE_LDS_BIG_BYTES;   // (raises the kernel's dynamic-LDS limit on this device)
  }
  if (e != hipSuccess) {
    fail(CE_ERR_HIP, "plan upload failed: %s", hipGetErrorString(e));
    ce_plan_destroy(p);
    return CE_ERR_HIP;
  }
  *out = p;
  return CE_OK;
}

int ce_plan_create(const ce_plan_desc* d, ce_plan** out) { return plan_build(d, out, true); }

int ce_plan_derive_host(const ce_plan_desc* d, ce_plan_host_view* v) {
  if (!v) return fail(CE_ERR_INVALID, "null argument");
  ce_plan* p = nullptr;
  const int rc = plan_build(d, &p, false);
  if (rc != CE_OK) return rc;
  const CeDevPlan& P = p->host;
  memset(v, 0, sizeof(*v));
  v->n_re = P.n_re; v->n_dmrs_total = p->info.n_dmrs_total; v->n_pils = P.n_pils; v->rc_len = P.rc_len;
  v->reg_nd = P.reg_nd; v->lds_bytes = p->info.lds_bytes; v->scratch_bytes = P.scratch_bytes;
  v->filt_windowed = P.filt_windowed; v->cfo_estimated = P.cfo_estimated; v->narrow = P.narrow;
  v->n_pilots = P.n_pilots; v->noise_den = P.noise_den;
  for (int i = 0; i < CE_MAX_RC_TAPS; ++i) v->rc[i] = P.rc[i];
  for (int i = 0; i < CE_MAX_SYMBOLS; ++i) v->sst[i] = P.sst[i];
  for (int h = 0; h < P.n_hops; ++h) {
    v->two_pi_nsamples[h] = P.hop[h].two_pi_nsamples;
    v->ta_nres[h] = P.hop[h].ta_nres;
    v->contig[h] = P.hop[h].contig;
    for (int c = 0; c < CE_MAX_CDM; ++c) {
      v->last_idx[h][c] = P.hop[h].last_idx[c];
      for (int r = 0; r < 12; ++r) { v->r_ord[h][c][r] = P.hop[h].r_ord[c][r]; v->alpha[h][c][r] = P.hop[h].alpha[c][r]; }
    }
  }
  if (!p->mmse_w.empty()) memcpy(v->mmse_w, p->mmse_w.data(), sizeof(v->mmse_w));
  delete p;
  return CE_OK;
}

void ce_plan_destroy(ce_plan* p) {
  if (!p) return;
  if (p->dev_plan) (void)hipFree(p->dev_plan);
  if (p->dev_re_idx) (void)hipFree(p->dev_re_idx);
  if (p->dev_tw) (void)hipFree(p->dev_tw);
  if (p->dev_ta_inv) (void)hipFree(p->dev_ta_inv);
  delete p;
}

int ce_plan_get_info(const ce_plan* plan, ce_plan_info* info) {
  if (!plan || !info) return fail(CE_ERR_INVALID, "null argument");
  *info = plan->info;
  return CE_OK;
}

static int check_batch(const ce_plan* plan, const void* rx, const int64_t* rs, const void* pilots, const int64_t* ps,
                       int64_t n_slots, int32_t n_ports, void* ch_est, double* noise, double* rsrp, double* epre,
                       double* ta, double* cfo, CeKernelArgs* a) {
  if (!plan || !rx || !rs || !pilots || !ps || !ch_est || !noise || !rsrp || !epre || !ta || !cfo)
    return fail(CE_ERR_INVALID, "null argument");
  if (n_slots < 0 || n_ports < 1) return fail(CE_ERR_INVALID, "n_slots=%lld n_ports=%d", (long long)n_slots, n_ports);
  if (n_slots * n_ports > 0x7FFFFFFFll) return fail(CE_ERR_UNSUPPORTED, "more than 2^31-1 work items in one launch");
  for (int i = 0; i < 4; ++i)
    if (rs[i] < 0 || ps[i] < 0) return fail(CE_ERR_INVALID, "negative strides are not supported");
  if ((plan->info.n_sc - 1) * rs[2] >= 0x7FFFFFFFll || (int64_t)(plan->info.n_re - 1) * ps[1] >= 0x7FFFFFFFll)
    return fail(CE_ERR_UNSUPPORTED, "subcarrier / pilot strides too large for 32-bit in-item offsets");
  a->rx = (const float2*)rx; a->rs_b = rs[0]; a->rs_r = rs[1]; a->rs_sc = rs[2]; a->rs_sym = rs[3];
  a->pil = (const float2*)pilots; a->ps_b = ps[0]; a->ps_re = ps[1]; a->ps_sym = ps[2]; a->ps_l = ps[3];
  a->out = (float2*)ch_est; a->noise = noise; a->rsrp = rsrp; a->epre = epre; a->ta = ta; a->cfo = cfo;
  a->n_items = n_slots * n_ports; a->n_ports = n_ports;
  a->item0 = 0; a->n_local = a->n_items;
  a->stage_p = nullptr; a->stage_s = nullptr;
  a->stamps = nullptr;
#if defined(CE_STAMPS)
  a->stamps = g_stamps;
#endif
  return CE_OK;
}

static int launch_batch(const ce_plan* plan, const void* rx, const int64_t rx_strides[4], const void* pilots,
                        const int64_t pil_strides[4], int64_t n_slots, int32_t n_ports, void* ch_est, double* noise,
                        double* rsrp, double* epre, double* ta, double* cfo_hz, void* stage_p, double* stage_s, void* stream) {
  CeKernelArgs a;
  int rc = check_batch(plan, rx, rx_strides, pilots, pil_strides, n_slots, n_ports, ch_est, noise, rsrp, epre, ta, cfo_hz, &a);
  if (rc != CE_OK) return rc;
  a.stage_p = (float2*)stage_p; a.stage_s = stage_s;
  if (a.n_items == 0) return CE_OK;
  CeDeviceScope scope(plan->device);  // `stream` belongs to the plan's device
  if (scope.err != hipSuccess) return fail(CE_ERR_HIP, "device %d: %s", plan->device, hipGetErrorString(scope.err));
  CeLaunchCtx c = {};
  c.dplan = plan->dev_plan; c.re_idx = plan->dev_re_idx; c.ta_inv = plan->dev_ta_inv; c.tw = plan->dev_tw;
  c.args = &a; c.lds = (plan->lds_big && a.n_items >= CE_LDS_BIG_ITEMS) ? plan->lds_big : plan->info.lds_bytes; c.stream = (hipStream_t)stream;
  int e = kernel_op(CE_OP_LAUNCH, plan->host, c);
  if (e != 0) return fail(CE_ERR_HIP, "kernel launch failed: %s", e > 0 ? hipGetErrorString((hipError_t)e) : "no kernel for this (layers, hops)");
  return CE_OK;
}

int ce_estimate_batch(const ce_plan* plan, const void* rx, const int64_t rx_strides[4], const void* pilots,
                      const int64_t pil_strides[4], int64_t n_slots, int32_t n_ports, void* ch_est, double* noise,
                      double* rsrp, double* epre, double* ta, double* cfo_hz, void* stream) {
  return launch_batch(plan, rx, rx_strides, pilots, pil_strides, n_slots, n_ports, ch_est, noise, rsrp, epre, ta, cfo_hz, nullptr, nullptr, stream);
}

int ce_estimate_batch_stages(const ce_plan* plan, const void* rx, const int64_t rx_strides[4], const void* pilots,
                             const int64_t pil_strides[4], int64_t n_slots, int32_t n_ports, void* ch_est, double* noise,
                             double* rsrp, double* epre, double* ta, double* cfo_hz, void* stage_estimates,
                             double* stage_scalars, void* stream) {
  if (!stage_estimates || !stage_scalars) return fail(CE_ERR_INVALID, "null stage buffers");
  return launch_batch(plan, rx, rx_strides, pilots, pil_strides, n_slots, n_ports, ch_est, noise, rsrp, epre, ta, cfo_hz, stage_estimates, stage_scalars, stream);
}

int ce_time_batch(const ce_plan* plan, const void* rx, const int64_t rx_strides[4], const void* pilots,
                  const int64_t pil_strides[4], int64_t n_slots, int32_t n_ports, void* ch_est, double* noise,
                  double* rsrp, double* epre, double* ta, double* cfo_hz, void* stream, int32_t warmup, int32_t iters,
                  double* avg_ms) {
  if (!avg_ms || iters < 1 || warmup < 0) return fail(CE_ERR_INVALID, "bad timing arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < warmup; ++i) {
    int rc = ce_estimate_batch(plan, rx, rx_strides, pilots, pil_strides, n_slots, n_ports, ch_est, noise, rsrp, epre, ta, cfo_hz, stream);
    if (rc != CE_OK) return rc;
  }
  CeDeviceScope scope(plan ? plan->device : 0);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    return fail(CE_ERR_HIP, "hipEventCreate failed");
  }
  hipError_t he = hipEventRecord(e0, st);
  int rc = CE_OK;
  for (int i = 0; i < iters && rc == CE_OK && he == hipSuccess; ++i)
    rc = ce_estimate_batch(plan, rx, rx_strides, pilots, pil_strides, n_slots, n_ports, ch_est, noise, rsrp, epre, ta, cfo_hz, stream);
  float ms = 0.f;
  if (he == hipSuccess) he = hipEventRecord(e1, st);
  if (he == hipSuccess) he = hipEventSynchronize(e1);
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (rc != CE_OK) return rc;
  if (he != hipSuccess) return fail(CE_ERR_HIP, "timing events: %s", hipGetErrorString(he));
  *avg_ms = (double)ms / iters;
  return CE_OK;
}

}  // extern "C"

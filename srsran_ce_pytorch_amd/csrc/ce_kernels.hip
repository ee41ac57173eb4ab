// Fused PUSCH DM-RS channel-estimation kernel for gfx950 (MI355X): one workgroup per
// slot x Rx-port work item; everything between the received grid and the channel grid stays
// in LDS/registers, the (n_sc, n_sym, L) output block is written once with 16-byte stores.
//
// Stage map (reference: src/ce_rule_tensorized.py, "T"):
//   cfo_pass        S4  T:357-426   inner products of the first two DM-RS symbols -> CFO per hop
//   ls_pass         S1-S3,S5 T:571-613  pilot gather, EPRE, LS (x conj(pilot)), de-rotation, DM-RS average
//   despread        S6  T:620-628
//   smooth_*        S7  T:633-668   mean | virtual pilots (T:69-140) + RC FIR (T:459-493)
//   time_alignment  S8  T:670-698   4096-point inverse FFT in LDS, arg-max over +-144 bins
//   residual_pass   S9,S11 T:700-730 reconstructed pilots, noise, RSRP
//   write_grid      S10 + epilogue T:237-354, T:921-929  linear interpolation, symbol replicate, CFO ramp
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "ce_plan.h"

namespace {

constexpr int NT = CE_THREADS;
constexpr int NW = NT / 64;
constexpr double kTwoPi = 6.283185307179586476925286766559;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {  // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum N per-thread doubles over the workgroup; every thread receives the totals.
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = wave_sum(v[i]);
    if (lane == 0) red[wave * 16 + i] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * 16 + i];
    v[i] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ int digit_reverse4_12(int n) {  // base-4 digit reversal of a 12-bit index
  int r = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    r = (r << 2) | (n & 3);
    n >>= 2;
  }
  return r;
}

// In-place radix-4 decimation-in-frequency inverse FFT of 4096 complex64 values in LDS.
// Natural-order input; bin n ends up at digit_reverse4_12(n).  No 1/N scale (arg-max only).
__device__ void ifft4096_lds(float2* x, const float2* __restrict__ tw) {
#pragma unroll 1
  for (int s = 0; s < 6; ++s) {
    const int span = 1024 >> (2 * s);
    const int tws = 1 << (2 * s);
    for (int b = threadIdx.x; b < 1024; b += NT) {
      const int i = b & (span - 1);
      const int base = ((b - i) << 2) + i;
      float2 a0 = x[base], a1 = x[base + span], a2 = x[base + 2 * span], a3 = x[base + 3 * span];
      float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
      float2 jd13 = make_float2(-d13.y, d13.x);  // +j * (a1 - a3)
      float2 y0 = cadd(s02, s13), y2 = csub(s02, s13), y1 = cadd(d02, jd13), y3 = csub(d02, jd13);
      if (span > 1) {
        const float2 w1 = tw[i * tws], w2 = tw[2 * i * tws], w3 = tw[3 * i * tws];
        y1 = cmul(y1, w1);
        y2 = cmul(y2, w2);
        y3 = cmul(y3, w3);
      }
      x[base] = y0;
      x[base + span] = y1;
      x[base + 2 * span] = y2;
      x[base + 3 * span] = y3;
    }
    __syncthreads();
  }
}

struct Ctx {
  const CeDevPlan* plan;
  const uint16_t* re_idx;
  const float2* rx;   // item base
  const float2* pil;  // slot base
  int64_t rs_sc, rs_sym, ps_re, ps_sym, ps_l;
};

// One 16-lane group per (layer, band edge): straight-line fit of modulus and unwrapped phase of
// the n_pils pilots next to the edge, extrapolated n_pils positions outwards (T:69-140, T:35-66).
__device__ void virtual_pilots(const float2* Pl, int n_re, int n_pils, bool tail, float2* ext_l, int j) {
  const double PI = 3.14159265358979323846;
  float2 v = make_float2(0.f, 0.f);
  if (j < n_pils) v = tail ? Pl[n_re - 1 - j] : Pl[j];
  double amp = (double)hypotf(v.x, v.y);
  double ang = (double)atan2f(v.y, v.x);
  double outr, outi;
  if (n_pils == 1) {  // T:95-101
    double sn, cs;
    sincos(ang, &sn, &cs);
    outr = amp * cs;
    outi = amp * sn;
  } else {
    // unwrap: per-gap correction then inclusive prefix sum over the 16-lane group
    double prev = __shfl_up(ang, 1, 16);
    double corr = 0.0;
    if (j >= 1 && j < n_pils) {
      double dd = ang - prev;
      double ddmod = fmod(dd + PI, 2.0 * PI);
      if (ddmod < 0.0) ddmod += 2.0 * PI;  // torch.remainder: result has the divisor's sign
      ddmod -= PI;
      if (ddmod == -PI && dd > 0.0) ddmod += 2.0 * PI;
      corr = fabs(dd) < PI ? 0.0 : ddmod - dd;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      double t = __shfl_up(corr, o, 16);
      if (j >= o) corr += t;
    }
    double ph = ang + corr;
    const double x = (double)j;
    double sx = j < n_pils ? x : 0.0, sxx = j < n_pils ? x * x : 0.0;
    double sa = j < n_pils ? amp : 0.0, sxa = j < n_pils ? x * amp : 0.0;
    double sp = j < n_pils ? ph : 0.0, sxp = j < n_pils ? x * ph : 0.0;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      sx += __shfl_xor(sx, o, 16);
      sxx += __shfl_xor(sxx, o, 16);
      sa += __shfl_xor(sa, o, 16);
      sxa += __shfl_xor(sxa, o, 16);
      sp += __shfl_xor(sp, o, 16);
      sxp += __shfl_xor(sxp, o, 16);
    }
    const double n = (double)n_pils;
    const double mx = sx / n, denom = sxx - n * mx * mx;
    const double ma = sa / n, mp = sp / n;
    const double a_amp = (sxa - n * mx * ma) / denom, b_amp = ma - a_amp * mx;
    const double a_ph = (sxp - n * mx * mp) / denom, b_ph = mp - a_ph * mx;
    const double k = (double)(j - n_pils);  // positions -nV .. -1
    const double va = a_amp * k + b_amp, vp = a_ph * k + b_ph;
    double sn, cs;
    sincos(vp, &sn, &cs);
    outr = va * cs;
    outi = va * sn;
  }
  if (j < n_pils) {
    // head: ext[j] (position j - nV); tail: flipped, ext[nP + n_re + (nV-1-j)]
    const int idx = tail ? (n_pils + n_re + (n_pils - 1 - j)) : j;
    ext_l[idx] = make_float2((float)outr, (float)outi);
  }
}

template <int L, int NH>
__global__ __launch_bounds__(NT) void ce_estimate_kernel(const CeDevPlan* __restrict__ plan,
                                                         const uint16_t* __restrict__ re_idx,
                                                         const float2* __restrict__ tw, CeKernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int64_t item = blockIdx.x;
  if (item >= a.n_items) return;
  const int64_t slot = item / a.n_ports;
  const int port = (int)(item - slot * a.n_ports);

  const int n_re = plan->n_re, n_re_pad = plan->n_re_pad, n_cdm = plan->n_cdm;
  const CeLdsLayout lay = ce_lds_layout(NH, L, n_re_pad);
  float2* P = reinterpret_cast<float2*>(smem + lay.off_p);              // [NH][L][n_re_pad]
  float2* scratch = reinterpret_cast<float2*>(smem + lay.off_scratch);   // 4096 complex
  double* red = reinterpret_cast<double*>(smem + lay.off_red);
  float2* rot_final = reinterpret_cast<float2*>(smem + lay.off_rot);     // [16]
  float2* rot_neg = rot_final + 16;                                      // [NH][16] exp(-j ph) at DM-RS symbols
  float2* rot_pos = rot_neg + CE_MAX_HOPS * 16;                          // [NH][16] exp(+j ph)
  float2* tab = reinterpret_cast<float2*>(smem + lay.off_tab);           // [NH][CDM][12] {alpha, bits(r_ord)}
  double* misc = reinterpret_cast<double*>(smem + lay.off_misc);         // [0..1] cfo_hop, [2] cfo_final

  const float2* rx = a.rx + slot * a.rs_b + port * a.rs_r;
  const float2* pil = a.pil + slot * a.ps_b;
  const float beta_f = plan->beta_f;
  const bool cfo_comp = plan->cfo_comp != 0;

  // interpolation tables -> LDS
  for (int i = tid; i < NH * CE_MAX_CDM * 12; i += NT) {
    const int h = i / (CE_MAX_CDM * 12), c = (i / 12) % CE_MAX_CDM, r = i % 12;
    tab[i] = make_float2(plan->hop[h].alpha[c][r], __int_as_float(plan->hop[h].r_ord[c][r]));
  }

  // ---------------------------------------------------------------- CFO per hop (S4)
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    const CeDevHop& hp = plan->hop[h];
    if (!hp.has_cfo) continue;  // uniform
    double acc[2 * L];
#pragma unroll
    for (int i = 0; i < 2 * L; ++i) acc[i] = 0.0;
    const int64_t o0 = hp.dmrs_sym[0] * a.rs_sym, o1 = hp.dmrs_sym[1] * a.rs_sym;
    const int64_t p0 = hp.pil_sym0 * a.ps_sym, p1 = (hp.pil_sym0 + 1) * a.ps_sym;
    for (int k = tid; k < n_re; k += NT) {
#pragma unroll
      for (int c = 0; c < (L + 1) / 2; ++c) {
        const int64_t sc = re_idx[hp.re_off[c] + k];
        const float2 x0 = rx[sc * a.rs_sc + o0], x1 = rx[sc * a.rs_sc + o1];
#pragma unroll
        for (int l = 2 * c; l < 2 * c + 2 && l < L; ++l) {
          const float2 q0 = pil[k * a.ps_re + p0 + l * a.ps_l], q1 = pil[k * a.ps_re + p1 + l * a.ps_l];
          const float2 r0 = cmul_conj(x0, q0), r1 = cmul_conj(x1, q1);
          const float2 in = cmul_conj(r1, r0);  // conj(r0) * r1
          acc[2 * l] += (double)in.x;
          acc[2 * l + 1] += (double)in.y;
        }
      }
    }
    block_sum<2 * L>(acc, red);
    if (tid == 0) {
      double ang = 0.0;
#pragma unroll
      for (int l = 0; l + 1 < L; l += 2)  // CDM pairs are summed before the angle (T:410-413)
        ang += (double)atan2f((float)(acc[2 * l + 1] + acc[2 * l + 3]), (float)(acc[2 * l] + acc[2 * l + 2]));
      if (L & 1) ang += (double)atan2f((float)acc[2 * L - 1], (float)acc[2 * L - 2]);
      misc[h] = ang / hp.two_pi_nsamples / plan->denom_cdm;
    }
  }
  __syncthreads();
  if (tid == 0) {  // running mean over hops (T:605-609)
    double cfo = 0.0;
    bool have = false;
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (plan->hop[h].has_cfo) {
        cfo = have ? (cfo + misc[h]) / 2 : misc[h];
        have = true;
      }
    misc[2] = cfo;
  }
  __syncthreads();
  const bool apply_rot = cfo_comp && plan->cfo_estimated;
  if (tid < 16) {
    float2 r = make_float2(1.f, 0.f);
    if (apply_rot && tid < CE_MAX_SYMBOLS) {
      double sn, cs;
      sincos(kTwoPi * plan->sst[tid] * misc[2], &sn, &cs);
      r = make_float2((float)cs, (float)sn);
    }
    rot_final[tid] = r;
  } else if (tid >= 64 && tid < 64 + NH * 16) {
    const int h = (tid - 64) >> 4, s = (tid - 64) & 15;
    const CeDevHop& hp = plan->hop[h];
    float2 rn = make_float2(1.f, 0.f), rp = make_float2(1.f, 0.f);
    if (cfo_comp && hp.has_cfo && s < hp.n_dmrs) {
      double sn, cs;
      sincos(kTwoPi * plan->sst[hp.dmrs_sym[s]] * misc[h], &sn, &cs);
      rn = make_float2((float)cs, (float)(-sn));
      rp = make_float2((float)cs, (float)sn);
    }
    rot_neg[h * 16 + s] = rn;
    rot_pos[h * 16 + s] = rp;
  }
  __syncthreads();

  double tot_epre = 0.0, tot_noise = 0.0, tot_rsrp = 0.0, tot_ta = 0.0;

#pragma unroll 1
  for (int h = 0; h < NH; ++h) {
    const CeDevHop& hp = plan->hop[h];
    float2* Ph = P + h * L * n_re_pad;
    const int n_dmrs = hp.n_dmrs;
    const float n_dmrs_f = (float)n_dmrs;

    // ------------------------------------------------------------ LS + DM-RS average (S1-S3, S5)
    float epre_part = 0.f;
    for (int k = tid; k < n_re; k += NT) {
#pragma unroll
      for (int c = 0; c < (L + 1) / 2; ++c) {
        const int64_t sc = re_idx[hp.re_off[c] + k];
        float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
        for (int s = 0; s < n_dmrs; ++s) {
          const float2 x = rx[sc * a.rs_sc + hp.dmrs_sym[s] * a.rs_sym];
          epre_part += x.x * x.x + x.y * x.y;
          const float2 rn = rot_neg[h * 16 + s];
          const int64_t pb = k * a.ps_re + (hp.pil_sym0 + s) * a.ps_sym;
          acc0 = cadd(acc0, cmul(cmul_conj(x, pil[pb + (2 * c) * a.ps_l]), rn));
          if (2 * c + 1 < L) acc1 = cadd(acc1, cmul(cmul_conj(x, pil[pb + (2 * c + 1) * a.ps_l]), rn));
        }
        Ph[(2 * c) * n_re_pad + k] = make_float2(acc0.x / beta_f / n_dmrs_f, acc0.y / beta_f / n_dmrs_f);
        if (2 * c + 1 < L)
          Ph[(2 * c + 1) * n_re_pad + k] = make_float2(acc1.x / beta_f / n_dmrs_f, acc1.y / beta_f / n_dmrs_f);
      }
    }
    __syncthreads();

    // ------------------------------------------------------------ CDM de-spread (S6)
    if (L >= 2) {
      for (int i = tid; i < n_re / 2; i += NT) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 u = Ph[l * n_re_pad + 2 * i], v = Ph[l * n_re_pad + 2 * i + 1];
          const float2 m = make_float2((u.x + v.x) / 2.f, (u.y + v.y) / 2.f);
          Ph[l * n_re_pad + 2 * i] = m;
          Ph[l * n_re_pad + 2 * i + 1] = m;
        }
      }
      __syncthreads();
    }

    // ------------------------------------------------------------ frequency smoothing (S7)
    if (plan->smoothing == CE_SMOOTH_MEAN) {
      double m[2 * L];
#pragma unroll
      for (int i = 0; i < 2 * L; ++i) m[i] = 0.0;
      for (int k = tid; k < n_re; k += NT) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 v = Ph[l * n_re_pad + k];
          m[2 * l] += (double)v.x;
          m[2 * l + 1] += (double)v.y;
        }
      }
      block_sum<2 * L>(m, red);
      for (int k = tid; k < n_re; k += NT) {
#pragma unroll
        for (int l = 0; l < L; ++l)
          Ph[l * n_re_pad + k] = make_float2((float)(m[2 * l] / (double)n_re), (float)(m[2 * l + 1] / (double)n_re));
      }
      __syncthreads();
    } else if (plan->smoothing == CE_SMOOTH_FILTER) {
      const int n_pils = plan->n_pils, rc_len = plan->rc_len, ext_len = plan->ext_len, lpp = plan->filt_lpp;
      const int pad = rc_len / 2;
#pragma unroll 1
      for (int l0 = 0; l0 < L; l0 += lpp) {
        const int nl = min(lpp, L - l0);
        // virtual pilots: one 16-lane group per (layer, edge)
        if (tid < nl * 32) {
          const int g = tid >> 4, j = tid & 15;
          virtual_pilots(Ph + (l0 + (g >> 1)) * n_re_pad, n_re, n_pils, (g & 1) != 0, scratch + (g >> 1) * ext_len, j);
        }
        for (int i = tid; i < nl * n_re; i += NT) {
          const int ll = i / n_re, k = i - ll * n_re;
          scratch[ll * ext_len + n_pils + k] = Ph[(l0 + ll) * n_re_pad + k];
        }
        __syncthreads();
        // conv(x, rc, "same") cropped by n_pils on both sides, float64 MACs (T:459-493, T:660-664)
        for (int i = tid; i < nl * n_re; i += NT) {
          const int ll = i / n_re, m = i - ll * n_re;
          const float2* x = scratch + ll * ext_len;
          double ar = 0.0, ai = 0.0;
          for (int j = 0; j < rc_len; ++j) {
            const int xi = m + n_pils + pad - j;
            if (xi >= 0 && xi < ext_len) {
              const float2 v = x[xi];
              const double w = plan->rc[j];
              ar += w * (double)v.x;
              ai += w * (double)v.y;
            }
          }
          Ph[(l0 + ll) * n_re_pad + m] = make_float2((float)ar, (float)ai);
        }
        __syncthreads();
      }
    }

    // ------------------------------------------------------------ time alignment (S8)
    {
      float pw0 = 0.f, pw1 = 0.f;  // bins tid and tid + NT of the 288 examined
      const uint16_t* ta_idx = re_idx + hp.re_off[n_cdm - 1];  // LAST CDM group's mask for every layer (T:672-675)
#pragma unroll 1
      for (int l = 0; l < L; ++l) {
        float4* s4 = reinterpret_cast<float4*>(scratch);
        for (int i = tid; i < CE_FFT_SIZE / 2; i += NT) s4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        for (int k = tid; k < n_re; k += NT) scratch[ta_idx[k]] = Ph[l * n_re_pad + k];
        __syncthreads();
        ifft4096_lds(scratch, tw);
        {
          const int n = tid < CE_TA_HALF ? tid : CE_FFT_SIZE - 2 * CE_TA_HALF + tid;
          const float2 v = scratch[digit_reverse4_12(n)];
          pw0 += v.x * v.x + v.y * v.y;
          if (tid + NT < 2 * CE_TA_HALF) {
            const int n1 = CE_FFT_SIZE - 2 * CE_TA_HALF + tid + NT;
            const float2 v1 = scratch[digit_reverse4_12(n1)];
            pw1 += v1.x * v1.x + v1.y * v1.y;
          }
        }
        __syncthreads();
      }
      // arg-max with first-index tie break on each side: key = (power bits, ~index)
      unsigned long long kh = 0ull, kt = 0ull;
      {
        const unsigned long long key0 = ((unsigned long long)__float_as_uint(pw0) << 32);
        if (tid < CE_TA_HALF) kh = key0 | (unsigned)(0xFFFFFFFFu - (unsigned)tid);
        else kt = key0 | (unsigned)(0xFFFFFFFFu - (unsigned)(tid - CE_TA_HALF));
        if (tid + NT < 2 * CE_TA_HALF) {
          const unsigned long long key1 = ((unsigned long long)__float_as_uint(pw1) << 32) |
                                          (unsigned)(0xFFFFFFFFu - (unsigned)(tid + NT - CE_TA_HALF));
          kt = key1 > kt ? key1 : kt;
        }
      }
      kh = wave_max_u64(kh);
      kt = wave_max_u64(kt);
      unsigned long long* ared = reinterpret_cast<unsigned long long*>(red);
      if ((tid & 63) == 0) {
        ared[(tid >> 6) * 2] = kh;
        ared[(tid >> 6) * 2 + 1] = kt;
      }
      __syncthreads();
      if (tid == 0) {
        unsigned long long mh = 0ull, mt = 0ull;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          mh = ared[2 * w] > mh ? ared[2 * w] : mh;
          mt = ared[2 * w + 1] > mt ? ared[2 * w + 1] : mt;
        }
        const float vd = __uint_as_float((unsigned)(mh >> 32)), va = __uint_as_float((unsigned)(mt >> 32));
        const int i_delay = (int)(0xFFFFFFFFu - (unsigned)(mh & 0xFFFFFFFFull));
        const int i_adv = (int)(0xFFFFFFFFu - (unsigned)(mt & 0xFFFFFFFFull));
        const int i_max = (vd >= va) ? i_delay : -(CE_TA_HALF - i_adv);
        tot_ta += (double)i_max / (double)CE_FFT_SIZE / plan->scs;
      }
      __syncthreads();
    }

    // ------------------------------------------------------------ residual noise, RSRP (S9, S11)
    {
      float noise_part = 0.f, rsrp_part = 0.f;
      for (int k = tid; k < n_re; k += NT) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 v = Ph[l * n_re_pad + k];
          rsrp_part += v.x * v.x + v.y * v.y;
        }
#pragma unroll
        for (int c = 0; c < (L + 1) / 2; ++c) {
          const int64_t sc = re_idx[hp.re_off[c] + k];
          const float2 h0 = Ph[(2 * c) * n_re_pad + k];
          const float2 h1 = (2 * c + 1 < L) ? Ph[(2 * c + 1) * n_re_pad + k] : make_float2(0.f, 0.f);
          for (int s = 0; s < n_dmrs; ++s) {
            const float2 x = rx[sc * a.rs_sc + hp.dmrs_sym[s] * a.rs_sym];
            const float2 rp = rot_pos[h * 16 + s];
            const int64_t pb = k * a.ps_re + (hp.pil_sym0 + s) * a.ps_sym;
            float2 est = cmul(pil[pb + (2 * c) * a.ps_l], cmul(h0, rp));
            if (2 * c + 1 < L) est = cadd(est, cmul(pil[pb + (2 * c + 1) * a.ps_l], cmul(h1, rp)));
            const float dr = x.x - beta_f * est.x, di = x.y - beta_f * est.y;
            noise_part += dr * dr + di * di;
          }
        }
      }
      double v[3] = {(double)epre_part, (double)noise_part, (double)rsrp_part};
      block_sum<3>(v, red);
      tot_epre += v[0];
      tot_noise += v[1];
      tot_rsrp += plan->beta * plan->beta * v[2] * (double)n_dmrs;
    }
  }

  // ---------------------------------------------------------------- scalars (T:898-937)
  if (tid == 0) {
    const double np = plan->n_pilots;
    a.rsrp[item] = tot_rsrp / np / (double)L;
    a.epre[item] = tot_epre / np;
    a.noise[item] = tot_noise / plan->noise_den;
    a.ta[item] = (NH == 2) ? tot_ta / 2.0 : tot_ta;
    a.cfo[item] = plan->cfo_estimated ? misc[2] * plan->scs : __longlong_as_double(0x7FF8000000000000ll);
  }

  // ---------------------------------------------------------------- interpolate + replicate + CFO ramp (S10)
  const int n_sym = plan->n_sym;
  const int row = n_sym * L;  // complex values per subcarrier
  const int64_t total = (int64_t)plan->n_sc * row;
  float2* out = a.out + item * total;

  auto elem = [&](int sc, int rem) -> float2 {
    const int sym = rem / L, l = rem - sym * L;
    float2 val = make_float2(0.f, 0.f);
#pragma unroll
    for (int h = NH - 1; h >= 0; --h) {  // a later hop overwrites an earlier one (T:872-896)
      const CeDevHop& hp = plan->hop[h];
      const int p = sc - hp.sc0;
      if (sym >= hp.sym0 && sym < hp.sym1 && p >= 0 && p < hp.n_sc_hop) {
        const int c = l >> 1;
        const int q = p / 12, r = p - 12 * q;
        const float2 t = tab[(h * CE_MAX_CDM + c) * 12 + r];
        int ro = q * hp.dpp[c] + __float_as_int(t.y);
        int lo = ro - 1;
        if (p >= hp.last_idx[c]) lo = ro = n_re - 1;
        lo = lo < 0 ? 0 : lo;
        const float2* Pl = P + (h * L + l) * n_re_pad;
        const float2 u = Pl[lo], v = Pl[ro];
        val = make_float2(u.x + t.x * (v.x - u.x), u.y + t.x * (v.y - u.y));
        if (apply_rot) val = cmul(val, rot_final[sym]);
        break;
      }
    }
    return val;
  };

  if ((row & 1) == 0) {
    float4* out4 = reinterpret_cast<float4*>(out);
    const int64_t npairs = total >> 1;
    int sc = (2 * tid) / row, rem = (2 * tid) - sc * row;
    const int dsc = (2 * NT) / row, drem = (2 * NT) - dsc * row;
    for (int64_t f = tid; f < npairs; f += NT) {
      const float2 v0 = elem(sc, rem), v1 = elem(sc, rem + 1);
      out4[f] = make_float4(v0.x, v0.y, v1.x, v1.y);
      rem += drem;
      sc += dsc;
      if (rem >= row) {
        rem -= row;
        ++sc;
      }
    }
  } else {
    for (int64_t e = tid; e < total; e += NT) {
      const int sc = (int)(e / row);
      out[e] = elem(sc, (int)(e - (int64_t)sc * row));
    }
  }
}

template <int L, int NH>
int launch_t(const CeDevPlan* dplan, const uint16_t* re_idx, const float2* tw, const CeKernelArgs& args, int lds,
             hipStream_t stream) {
  hipLaunchKernelGGL((ce_estimate_kernel<L, NH>), dim3((unsigned)args.n_items), dim3(NT), lds, stream, dplan, re_idx,
                     tw, args);
  return (int)hipGetLastError();
}

template <int L, int NH>
int prepare_t(int lds) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&ce_estimate_kernel<L, NH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

#define CE_DISPATCH(FN, ...)                                      \
  switch (n_layers * 10 + n_hops) {                               \
    case 11: return FN<1, 1>(__VA_ARGS__);                        \
    case 12: return FN<1, 2>(__VA_ARGS__);                        \
    case 21: return FN<2, 1>(__VA_ARGS__);                        \
    case 22: return FN<2, 2>(__VA_ARGS__);                        \
    case 31: return FN<3, 1>(__VA_ARGS__);                        \
    case 32: return FN<3, 2>(__VA_ARGS__);                        \
    case 41: return FN<4, 1>(__VA_ARGS__);                        \
    case 42: return FN<4, 2>(__VA_ARGS__);                        \
    default: return -1;                                           \
  }

int ce_launch(const CeDevPlan& hplan, const CeDevPlan* dplan, const uint16_t* re_idx, const float2* tw,
              const CeKernelArgs& args, int lds_bytes, hipStream_t stream) {
  const int n_layers = hplan.n_layers, n_hops = hplan.n_hops;
  CE_DISPATCH(launch_t, dplan, re_idx, tw, args, lds_bytes, stream)
}

int ce_prepare_kernel(int n_layers, int n_hops, int lds_bytes) { CE_DISPATCH(prepare_t, lds_bytes) }

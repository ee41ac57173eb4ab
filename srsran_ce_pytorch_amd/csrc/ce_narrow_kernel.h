// Wave-per-item PUSCH DM-RS channel estimation for NARROW allocations (gfx950): a 256-thread workgroup carries four work
// items (slot x Rx port), one per 64-lane wave; after one workgroup barrier (LDS copy of the plan, TA twiddles) the waves
// never synchronise with each other again -- every later hand-over is inside a wave (LDS is in order per wave; a
// wavefront fence keeps the compiler honest).
//
// Why: an allocation of a few PRB is a latency chain, not a bandwidth problem.  One 256-thread workgroup per item
// (ce_estimate_kernel.h) spends most of such an item parked at workgroup barriers with 3-4 items resident per CU; here a
// CU holds 16 items (4 workgroups x 4 waves) that advance independently, and no stage ever waits for another wave.
//
// Same arithmetic as ce_estimate_kernel.h stage by stage (reference: src/ce_rule_tensorized.py "T"):
//   stage      S1 T:571-581   the hop's received pilots + DM-RS symbols -> per-wave LDS rows, once per hop
//   cfo        S4 T:357-426   inner products of the first two DM-RS symbols -> CFO of the hop (float64 wave sums, DPP)
//   ls         S2,S3,S5 T:584-613
//   despread   S6 T:620-628
//   smooth     S7 T:633-668   mean | virtual pilots (T:69-140; one 16-lane DPP row per band edge) + RC FIR (float64 MACs)
//   residual   S9,S11 T:700-730
//   epilogue   T:898-937
//   write      S10 + T:921-929  63 / 56 lanes own one (symbol pair, layer) float4 phase each; contiguous 1 KB wave stores
//   ta         S8 T:670-698   the pruned 4096-point inverse DFT entirely in registers: lane (residue, c) forms the
//                             collapsed first radix-16 pass and the second pass for its column, contributes its 18 examined
//                             bins, two cross-row shuffles add the residues; one arg-max key per hop
// Plans: interp = linear, smoothing none / mean / filter, 14-symbol grids, <= CE_NARROW_MAX_RE pilots per symbol, every
// hop's band inside the collapsed TA window (ce_api.hip decides; everything else runs on ce_estimate_kernel.h).
#pragma once
#include "ce_estimate_kernel.h"   // shared device helpers (DPP reductions, idft16, virtual_pilots, PilotMap, ...)

#ifndef CE_NRW_ABLATE
#define CE_NRW_ABLATE 0   // timing / register experiments only: 1 TA, 2 smoothing, 4 writer, 8 staging loads
#endif

namespace {

__device__ __forceinline__ void wave_sync() {   // orders a wave's LDS writes before its later LDS reads (other lanes' data)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// idft16 (ce_estimate_kernel.h) without its 16-element output copy: X[d] = sum_b v[b] exp(+j 2 pi b d / 16) is left at
// v[idft16_at(d)] -- the second radix-4 stage writes over its own inputs, so the result sits in digit-swapped order and
// the (compile-time) index map undoes it.  Same operations in the same order as idft16: bit-identical values.
__device__ constexpr int idft16_at(int d) { return 4 * (d & 3) + (d >> 2); }
__device__ __forceinline__ void idft16_inplace(float2 (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
  for (int j = 0; j < 4; ++j) r4inv(v[j], v[j + 4], v[j + 8], v[j + 12]);
  v[1 + 4] = cmul(v[1 + 4], make_float2(C1, S1));
  v[1 + 8] = cmul(v[1 + 8], make_float2(R2, R2));
  v[1 + 12] = cmul(v[1 + 12], make_float2(S1, C1));
  v[2 + 4] = cmul(v[2 + 4], make_float2(R2, R2));
  v[2 + 8] = make_float2(-v[2 + 8].y, v[2 + 8].x);
  v[2 + 12] = cmul(v[2 + 12], make_float2(-R2, R2));
  v[3 + 4] = cmul(v[3 + 4], make_float2(S1, C1));
  v[3 + 8] = cmul(v[3 + 8], make_float2(-R2, R2));
  v[3 + 12] = cmul(v[3 + 12], make_float2(-C1, -S1));
#pragma unroll
  for (int m = 0; m < 4; ++m) r4inv(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]);   // X[m + 4 n] at v[4 m + n]
}

__device__ __forceinline__ float shfl_xor_f(float v, int mask) {   // ds_bpermute: lane ^ mask
  return __int_as_float(__builtin_amdgcn_ds_bpermute(((int)(threadIdx.x & 63) ^ mask) << 2, __float_as_int(v)));
}

template <int L, int NH>
__global__ __launch_bounds__(NT, CE_NARROW_MIN_WAVES) void ce_narrow_kernel(const CeDevPlan* __restrict__ plan,
                                                                           const uint16_t* __restrict__ re_idx,
                                                                           const uint16_t* __restrict__ ta_inv,
                                                                           const float2* __restrict__ tw, CeKernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NC = (L + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_re = plan->n_re, n_re_pad = plan->n_re_pad;
  const CeNarrowLayout lay = ce_narrow_layout(NH, L, plan->nrw_nd_max, n_re_pad);
  const CeDevPlan* lp = reinterpret_cast<const CeDevPlan*>(smem + lay.off_plan);
  float2* tw256 = reinterpret_cast<float2*>(smem + lay.off_tw);   // [256] W256^j, then [16] W4096^i
  float2* tw16 = tw256 + 256;
  unsigned char* wbase = smem + lay.off_wave0 + wave * lay.wave_stride;
  float2* S = reinterpret_cast<float2*>(wbase + lay.stage_off);   // staged hop: rx rows [c][s], then DM-RS rows [l][s], n_re_pad each
  float2* P = reinterpret_cast<float2*>(wbase + lay.p_off);       // [NH][L][n_re_pad]
  float2* vp = reinterpret_cast<float2*>(wbase + lay.vp_off);     // [2 layers][head, tail][16]
  float2* rot_final = reinterpret_cast<float2*>(wbase + lay.rot_off);   // [16]
  float2* rot_tab = rot_final + 16;                               // per hop: [4] exp(-j ph) at its DM-RS symbols, [4] exp(+j ph)

  // ---- the one workgroup-wide step: plan + twiddles -> LDS
  constexpr int PLAN4 = (int)(sizeof(CeDevPlan) / 16), TW4 = (256 + 16) / 2;
  static_assert(sizeof(CeDevPlan) % 16 == 0 && PLAN4 <= NT && TW4 <= NT, "one float4 per thread covers the plan and the twiddles");
  {
    float4 plan_v = make_float4(0.f, 0.f, 0.f, 0.f), tw_v = plan_v;
    if (tid < PLAN4) plan_v = reinterpret_cast<const float4*>(plan)[tid];
    if (tid < TW4) tw_v = reinterpret_cast<const float4*>(tw + CE_TWC_OFF)[tid];
    if (tid < PLAN4) reinterpret_cast<float4*>(smem + lay.off_plan)[tid] = plan_v;
    if (tid < TW4) reinterpret_cast<float4*>(tw256)[tid] = tw_v;
  }
  __syncthreads();
  const int64_t local = (int64_t)blockIdx.x * NW + wave;
  if (local >= a.n_local) return;   // a dead wave of the last workgroup (no workgroup barrier follows)
  const int64_t item = a.item0 + local;
  const int64_t slot = item / a.n_ports;
  const int port = (int)(item - slot * a.n_ports);
  const float2* rx = a.rx + slot * a.rs_b + port * a.rs_r;
  const float2* pil = a.pil + slot * a.ps_b;
  const float beta_f = lp->beta_f;
  const bool cfo_comp = lp->cfo_comp != 0;
  const unsigned magic_nre = lp->nrw_magic_nre;

  // ------------------------------------------------------------------ S1: one hop's pilots -> per-wave LDS rows
  // Rows: rx of CDM group c at DM-RS symbol s -> row c * nd + s; DM-RS symbol s of layer l -> row (NC + l) * nd + s.
  // All loads of a batch are requested before the first is stored (one round trip per 8 elements per lane).
  auto stage_hop = [&](int h) __attribute__((always_inline)) {
    const CeDevHop& hp = lp->hop[h];
    const int nd = hp.n_dmrs, rows_rx = NC * nd, total = (NC + L) * nd * n_re;
    const float inv_nd = 1.0f / (float)nd;
    PilotMap pm[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) pm[c] = pilot_map(hp, c);
    const int psym0 = hp.pil_sym0;
#pragma unroll 1
    for (int base = 0; base < ((CE_NRW_ABLATE & 8) ? 0 : total); base += 64 * 8) {
      float2 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 64 + lane;
        const bool ok = idx < total;
        const int ii = ok ? idx : 0;
        const int row = n_re == 1 ? ii : (int)__umulhi((unsigned)ii, magic_nre), k = ii - row * n_re;   // (one pilot: the 33-bit magic does not exist)
        const float2* src;
        if (row < rows_rx) {
          const int c = (int)(((float)row + 0.5f) * inv_nd), s = row - c * nd;
          int sc = pilot_sc(pm[0], re_idx, k);
          if (NC > 1 && c == 1) sc = pilot_sc(pm[NC - 1], re_idx, k);
          src = rx + (int64_t)sc * a.rs_sc + (int64_t)hp.dmrs_sym[s] * a.rs_sym;
        } else {
          const int r2 = row - rows_rx, l = (int)(((float)r2 + 0.5f) * inv_nd), s = r2 - l * nd;
          src = pil + (int64_t)k * a.ps_re + (int64_t)(psym0 + s) * a.ps_sym + (int64_t)l * a.ps_l;
        }
        v[u] = *src;
        dst[u] = ok ? row * n_re_pad + k : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (dst[u] >= 0) S[dst[u]] = v[u];
    }
    wave_sync();
  };

  auto dump_stage = [&](int st, int h) __attribute__((always_inline)) {   // ce_estimate_batch_stages only (a.stage_p null otherwise)
    if (a.stage_p) {
      float2* sp = a.stage_p + ((item * 2 + st) * NH + h) * (int64_t)(L * n_re);
      const float2* Ph = P + h * L * n_re_pad;
      for (int i = lane; i < L * n_re; i += 64) {
        const int l = i / n_re;
        sp[i] = Ph[l * n_re_pad + (i - l * n_re)];
      }
    }
  };

  double tot_epre = 0.0, tot_noise = 0.0, tot_rsrp = 0.0, cfo_hops[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) cfo_hops[h] = 0.0;

#pragma unroll 1
  for (int h = 0; h < NH; ++h) {
    const CeDevHop& lh = lp->hop[h];
    float2* Ph = P + h * L * n_re_pad;
    float2* rot_neg = rot_tab + h * 8;
    float2* rot_pos = rot_neg + 4;
    const int nd = lh.n_dmrs;
    const float nd_f = (float)nd;
    const bool has_cfo = lh.has_cfo != 0;
    stage_hop(h);
    const float2* X = S;                         // X[(c * nd + s) * n_re_pad + k]
    const float2* D = S + NC * nd * n_re_pad;    // D[(l * nd + s) * n_re_pad + k]

    // ---------------------------------------------------------------- S4: CFO of the hop
    double cfo_hop = 0.0;
    if (has_cfo) {
      float part[2 * L];
#pragma unroll
      for (int i = 0; i < 2 * L; ++i) part[i] = 0.f;
      for (int k = lane; k < n_re; k += 64) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float2 x0 = X[(c * nd) * n_re_pad + k], x1 = X[(c * nd + 1) * n_re_pad + k];
#pragma unroll
          for (int l = 2 * c; l < 2 * c + 2 && l < L; ++l) {
            const float2 q0 = D[(l * nd) * n_re_pad + k], q1 = D[(l * nd + 1) * n_re_pad + k];
            const float2 r0 = cmul_conj(x0, q0), r1 = cmul_conj(x1, q1);
            const float2 in = cmul_conj(r1, r0);  // conj(r0) * r1
            part[2 * l] += in.x;
            part[2 * l + 1] += in.y;
          }
        }
      }
      double acc[2 * L];
#pragma unroll
      for (int i = 0; i < 2 * L; ++i) acc[i] = wave_sum((double)part[i]);
      double ang = 0.0;
#pragma unroll
      for (int l = 0; l + 1 < L; l += 2)  // CDM pairs are summed before the angle (T:410-413)
        ang += (double)atan2f((float)(acc[2 * l + 1] + acc[2 * l + 3]), (float)(acc[2 * l] + acc[2 * l + 2]));
      if (L & 1) ang += (double)atan2f((float)acc[2 * L - 1], (float)acc[2 * L - 2]);
      cfo_hop = ang * lh.inv_two_pi_nsamples * lp->inv_denom_cdm;  // T:426
      if (lane == 0 && a.stage_s) a.stage_s[(item * NH + h) * 2] = cfo_hop;
    }
    cfo_hops[h] = cfo_hop;
    if (lane < 4) {  // de-rotation / re-rotation phasors of the hop's DM-RS symbols (T:439-447, T:713-718)
      float2 rn = make_float2(1.f, 0.f), rp = make_float2(1.f, 0.f);
      if (cfo_comp && has_cfo && lane < nd) {
        float sn, cs;
        sincospi_f64arg(2.0 * lp->sst_dmrs[h][lane] * cfo_hop, &sn, &cs);
        rn = make_float2(cs, -sn);
        rp = make_float2(cs, sn);
      }
      rot_neg[lane] = rn;
      rot_pos[lane] = rp;
    }
    wave_sync();

    // ---------------------------------------------------------------- S2, S3, S5: EPRE, LS, DM-RS average
    float epre_part = 0.f;
    for (int k = lane; k < n_re; k += 64) {
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
        for (int s = 0; s < nd; ++s) {
          const float2 x = X[(c * nd + s) * n_re_pad + k];
          epre_part += x.x * x.x + x.y * x.y;
          const float2 rn = rot_neg[s];
          acc0 = cadd(acc0, cmul(cmul_conj(x, D[((2 * c) * nd + s) * n_re_pad + k]), rn));
          if (2 * c + 1 < L) acc1 = cadd(acc1, cmul(cmul_conj(x, D[((2 * c + 1) * nd + s) * n_re_pad + k]), rn));
        }
        Ph[(2 * c) * n_re_pad + k] = make_float2(acc0.x / beta_f / nd_f, acc0.y / beta_f / nd_f);
        if (2 * c + 1 < L) Ph[(2 * c + 1) * n_re_pad + k] = make_float2(acc1.x / beta_f / nd_f, acc1.y / beta_f / nd_f);
      }
    }
    wave_sync();

    // ---------------------------------------------------------------- S6: CDM de-spread
    if (L >= 2) {
      for (int i = lane; i < n_re / 2; i += 64) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 u = Ph[l * n_re_pad + 2 * i], v = Ph[l * n_re_pad + 2 * i + 1];
          const float2 m = make_float2((u.x + v.x) / 2.f, (u.y + v.y) / 2.f);
          Ph[l * n_re_pad + 2 * i] = m;
          Ph[l * n_re_pad + 2 * i + 1] = m;
        }
      }
      wave_sync();
    }
    dump_stage(0, h);

    // ---------------------------------------------------------------- S7: frequency smoothing
    if (CE_NRW_ABLATE & 2) {
    } else if (lp->smoothing == CE_SMOOTH_MEAN) {
#pragma unroll 1
      for (int l = 0; l < L; ++l) {
        double mr = 0.0, mi = 0.0;
        for (int k = lane; k < n_re; k += 64) {
          const float2 v = Ph[l * n_re_pad + k];
          mr += (double)v.x;
          mi += (double)v.y;
        }
        mr = wave_sum(mr);
        mi = wave_sum(mi);
        const float2 m = make_float2((float)(mr / (double)n_re), (float)(mi / (double)n_re));
        for (int k = lane; k < n_re; k += 64) Ph[l * n_re_pad + k] = m;
      }
      wave_sync();
    } else if (lp->smoothing == CE_SMOOTH_FILTER) {
      const int n_pils = lp->n_pils, rc_len = lp->rc_len, pad = rc_len / 2;
      const double vmx = lp->vp_mx, vin = lp->vp_inv_n, vid = lp->vp_inv_denom;
      const double* rc = lp->rc;
#pragma unroll 1
      for (int l0 = 0; l0 < L; l0 += 2) {
        {  // virtual pilots of layers l0 (lanes 0-31) and l0 + 1 (lanes 32-63): one 16-lane DPP row per band edge
          const int sub = lane >> 5, ll = l0 + sub < L ? l0 + sub : l0, e = (lane >> 4) & 1;
          virtual_pilots(Ph + ll * n_re_pad, n_re, n_pils, e != 0, lane & 15, vmx, vin, vid,
                         [&](int dist, float2 val) { vp[sub * 32 + e * 16 + dist] = val; });
        }
        wave_sync();
#pragma unroll 1
        for (int sub = 0; sub < 2 && l0 + sub < L; ++sub) {
          float2* Pl = Ph + (l0 + sub) * n_re_pad;
          const float2* vpl = vp + sub * 32;
          // conv([virtual head ; P ; virtual tail], rc, "same") cropped back to P (T:649-664), float64 MACs (T:477-490);
          // every output of the row is formed before the first is written
          double yr[3], yi[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int m = lane + 64 * i;
            double ar = 0.0, ai = 0.0;
            if (m < n_re) {
              for (int j = 0; j < rc_len; ++j) {
                const int idx = m + pad - j;
                float2 x = make_float2(0.f, 0.f);
                if (idx < 0) {
                  if (-1 - idx < n_pils) x = vpl[-1 - idx];
                } else if (idx >= n_re) {
                  if (idx - n_re < n_pils) x = vpl[16 + idx - n_re];
                } else {
                  x = Pl[idx];
                }
                const double w = rc[j];
                ar += w * (double)x.x;
                ai += w * (double)x.y;
              }
            }
            yr[i] = ar;
            yi[i] = ai;
          }
          wave_sync();
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int m = lane + 64 * i;
            if (m < n_re) Pl[m] = make_float2((float)yr[i], (float)yi[i]);
          }
        }
        wave_sync();
      }
    }
    dump_stage(1, h);

    // ---------------------------------------------------------------- S9, S11: residual noise, RSRP
    {
      float noise_part = 0.f, rsrp_part = 0.f;
      for (int k = lane; k < n_re; k += 64) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 v = Ph[l * n_re_pad + k];
          rsrp_part += v.x * v.x + v.y * v.y;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float2 h0 = Ph[(2 * c) * n_re_pad + k];
          const float2 h1 = (2 * c + 1 < L) ? Ph[(2 * c + 1) * n_re_pad + k] : make_float2(0.f, 0.f);
          for (int s = 0; s < nd; ++s) {
            const float2 x = X[(c * nd + s) * n_re_pad + k];
            const float2 rp = rot_pos[s];
            float2 est = cmul(D[((2 * c) * nd + s) * n_re_pad + k], cmul(h0, rp));
            if (2 * c + 1 < L) est = cadd(est, cmul(D[((2 * c + 1) * nd + s) * n_re_pad + k], cmul(h1, rp)));
            const float dr = x.x - beta_f * est.x, di = x.y - beta_f * est.y;
            noise_part += dr * dr + di * di;
          }
        }
      }
      tot_epre += wave_sum((double)epre_part);
      tot_noise += wave_sum((double)noise_part);
      tot_rsrp += lp->beta * lp->beta * wave_sum((double)rsrp_part) * (double)nd;
    }
    wave_sync();   // the next hop's staging overwrites S
  }

  // ------------------------------------------------------------------ slot-level epilogue (T:898-937)
  const bool apply_rot = cfo_comp && lp->cfo_estimated;
  {
    double cfo = 0.0;  // running mean over the hops that estimated one (T:605-609)
    bool have = false;
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (lp->hop[h].has_cfo) {
        cfo = have ? (cfo + cfo_hops[h]) / 2 : cfo_hops[h];
        have = true;
      }
    if (lane < 16) {
      float2 r = make_float2(1.f, 0.f);
      if (apply_rot && lane < CE_MAX_SYMBOLS) {
        float sn, cs;
        sincospi_f64arg(2.0 * lp->sst[lane] * cfo, &sn, &cs);
        r = make_float2(cs, sn);
      }
      rot_final[lane] = r;
    }
    if (lane == 0) {
      const double np = lp->inv_n_pilots;
      a.rsrp[item] = tot_rsrp * np * lp->inv_layers;
      a.epre[item] = tot_epre * np;
      a.noise[item] = tot_noise * lp->inv_noise_den;
      a.cfo[item] = lp->cfo_estimated ? cfo * lp->scs : __longlong_as_double(0x7FF8000000000000ll);
    }
  }
  wave_sync();

  // ------------------------------------------------------------------ S10: interpolate + replicate + CFO ramp
  // A subcarrier's (14 symbols x L layers) is 7L float4; SCS = 64 / 7L subcarriers per wave iteration, each active lane
  // owns ONE (symbol pair, layer) float4 phase for the whole item (its two phasors and hop candidates are lane
  // constants) and a wave iteration stores ACTIVE * 16 contiguous bytes.  Linear interpolation straight from P
  // (left + alpha (right - left), also AT pilots, as the reference does; T:311-338); the hop of an element is the last hop
  // whose symbol AND subcarrier range cover it (T:872-896).
  if (!(CE_NRW_ABLATE & 4)) {
    constexpr int ROW4 = 7 * L, SCS = 64 / ROW4, ACTIVE = SCS * ROW4;
    const int ph = lane % ROW4, sc_off = lane / ROW4;
    const int n_sc = lp->n_sc;
    const CeDevHop& g0 = lp->hop[0];
    const CeDevHop& g1 = lp->hop[NH - 1];
    const int b0 = g0.sc0, n0 = g0.n_sc_hop, b1 = g1.sc0, n1 = g1.n_sc_hop;
    float2 rsel[2];
    unsigned cand[2];
    int lsel[2], dppe[2], lastp[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int rem = 2 * ph + e, sym = rem / L, l = rem - sym * L, c = l >> 1;
      cand[e] = (sym >= g0.sym0 && sym < g0.sym1) ? 1u : 0u;
      if (NH == 2 && sym >= g1.sym0 && sym < g1.sym1) cand[e] |= 2u;
      rsel[e] = cand[e] ? rot_final[sym < 16 ? sym : 0] : make_float2(0.f, 0.f);   // rot_final == 1 when no CFO ramp applies
      lsel[e] = l;
      dppe[e] = g0.dpp[c];
      lastp[e] = g0.last_idx[c];       // both hops carry the same RE mask and PRB count (T:869; pilots.shape[0] is shared)
    }
    const bool same = L == 1 && cand[0] == cand[1];
    float4* o = reinterpret_cast<float4*>(a.out + item * ((int64_t)n_sc * 14 * L)) + lane;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_it = (n_sc + SCS - 1) / SCS;   // wave-uniform trip count
#pragma unroll 2
    for (int it = 0; it < n_it; ++it, o += ACTIVE) {
      const int sc = sc_off + it * SCS;
      const bool live = lane < ACTIVE && sc < n_sc;
      const bool in0 = (unsigned)(sc - b0) < (unsigned)n0, in1 = NH == 2 && (unsigned)(sc - b1) < (unsigned)n1;
      if (__builtin_amdgcn_ballot_w64(live && (in0 || in1)) == 0ull) {   // no lane of the wave inside a hop's band: zeros
        if (live) *o = z4;
        continue;
      }
      float2 y[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e == 1 && same) {
          y[1] = y[0];
        } else {
          const bool use1 = in1 && (cand[e] & 2u), use0 = !use1 && in0 && (cand[e] & 1u), valid = live && (use1 || use0);
          const int p = valid ? sc - (use1 ? b1 : b0) : 0;
          const int q = (int)(((unsigned)p * 0xAAABu) >> 19), r = p - 12 * q;   // p / 12 for p < 2^15
          const int c = lsel[e] >> 1;
          const float al = g0.alpha[c][r];
          int ro = q * dppe[e] + g0.r_ord[c][r], lo = ro - 1;
          if (p >= lastp[e]) lo = ro = n_re - 1;     // at/after the last pilot: hold (T:316,321)
          lo = lo < 0 ? 0 : lo;                       // at/before the first pilot: hold (T:315,320)
          const float2* Pl = P + ((use1 ? L : 0) + lsel[e]) * n_re_pad;
          const float2 u = Pl[lo], v = Pl[ro];
          const float2 w = make_float2(u.x + al * (v.x - u.x), u.y + al * (v.y - u.y));
          y[e] = make_float2(valid ? w.x : 0.f, valid ? w.y : 0.f);
        }
      }
      const float2 ya = cmul(y[0], rsel[0]), yb = cmul(y[1], rsel[1]);
      if (live) *o = make_float4(ya.x, ya.y, yb.x, yb.y);
    }
  }

  // ------------------------------------------------------------------ S8: time alignment of each hop, while the stores drain
  // x[n] = P[k] at the pilot subcarriers of the LAST CDM group (for every layer, T:672-675), else 0;
  // X[k] = sum_n x[n] W^(nk), W = exp(+j 2 pi / 4096), wanted for k in [0,144) U [3952,4096).  n = r + 16 n':
  // X[k] = sum_r W^(rk) Y_r[k mod 256], Y_r = 256-point inverse DFT of x[r + 16 n'].  The band, moved down by `shift`
  // (a multiple of 16: a unit phase per bin, |X| unchanged), ends below subcarrier 512, so Y_r[c + 16 d] =
  // sum_a (x0[a] + x1[a] W16^c) W256^(ac) W16^(ad): lane (r, c) forms the 16 products and one 16-point transform in
  // registers and then owns Y_r[c + 16 d], d = 0..15 -- exactly the 18 examined bins k = c + 16 m (d = m) and
  // k = 3952 + c + 16 m (d = 7 + m), m = 0..8, of its column.  Four residues per round (one per 16-lane row).
  double tot_ta = 0.0;
#pragma unroll 1
  for (int h = 0; h < ((CE_NRW_ABLATE & 1) ? 0 : NH); ++h) {
    const CeDevHop& lh = lp->hop[h];
    const int nres = lh.ta_nres, shift = (int)(lh.ta_win & 0xFFFFu);
    const bool two = (lh.ta_win >> 16) == 2u;
    const unsigned long long resp = lh.ta_res_packed, ord_packed = lh.ord_packed;
    const int contig = lh.contig, dpp_last = lh.dpp[NC - 1], prb0 = lh.prb_start, nprb = lh.n_prbs;
    const uint16_t* inv = ta_inv + lh.ta_inv_off;
    auto pilot_at = [&](int n) -> int {   // subcarrier n -> ordinal of the pilot it carries (last CDM group), or -1
      if (contig) {
        const int q = n / 12, rem = n - 12 * q, pq = q - prb0;
        const int o = (int)((ord_packed >> (4 * rem)) & 15u);
        return (pq >= 0 && pq < nprb && o != 15) ? pq * dpp_last + o : -1;
      }
      const unsigned idx = inv[n];
      return idx == 0xFFFFu ? -1 : (int)idx;
    };
    // After the residues of a round are transformed, the four rows' contributions to a bin are added by a reduce-scatter
    // over two shuffles: rows 0-1 end up owning the delay-side bins of their column, rows 2-3 the advance-side ones; of
    // those nine the even row keeps m = 0..4, the odd row m = 5..8 -- five complex accumulators per lane instead of
    // eighteen, and every lane's bins are distinct, so the arg-max key needs no de-duplication.
    const int c = lane & 15, row = lane >> 4;
    const bool adv = (row & 2) != 0, upper = (row & 1) != 0;
    float2* xs = vp;   // [4 rows][x0: 16 | x1: 16] -- the band's pilots of the round's residues (the virtual-pilot buffer is free now)
    float pw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) pw[i] = 0.f;
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
      const float2* Pl = P + (h * L + l) * n_re_pad;
      float2 acc[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) acc[i] = make_float2(0.f, 0.f);
#pragma unroll 1
      for (int g = 0; g < nres; g += 4) {
        const bool unit = g + row < nres;
        const int r = (int)((resp >> (4 * (unit ? g + row : 0))) & 15u);
        // everything below that depends on the column alone is loop-invariant; left alone the compiler hoists it (16 twiddle
        // addresses, the bins' indices ...) out of the residue and layer loops and keeps it live across them: an opaque copy
        // of the column keeps each round's temporaries local to the round
        int co = c;
        asm volatile("" : "+v"(co));
        {  // lane (row, c) fetches the pilot at subcarrier shift + r + 16 c (and + 256): one position look-up per lane
          const int n0 = shift + r + 16 * c;
          const int i0 = unit ? pilot_at(n0) : -1, i1 = (unit && two) ? pilot_at(n0 + 256) : -1;
          xs[row * 32 + c] = i0 >= 0 ? Pl[i0] : make_float2(0.f, 0.f);
          xs[row * 32 + 16 + c] = i1 >= 0 ? Pl[i1] : make_float2(0.f, 0.f);
        }
        wave_sync();
        float2 v[16];
        const float2 wc = tw256[16 * co];   // W16^c
#pragma unroll
        for (int aq = 0; aq < 16; ++aq) {
          const float2 x0 = xs[row * 32 + aq], x1 = xs[row * 32 + 16 + aq];
          v[aq] = cmul(cadd(x0, cmul(x1, wc)), tw256[(aq * co) & 255]);
        }
        wave_sync();   // (the next round overwrites xs)
        idft16_inplace(v);   // Y_r[c + 16 d] at v[idft16_at(d)]
        float2 s1[9];
#pragma unroll
        for (int m = 0; m < 9; ++m) {
          const int kd = co + 16 * m, ka = CE_FFT_SIZE - CE_TA_HALF + co + 16 * m;
          const int md = (r * kd) & (CE_FFT_SIZE - 1), ma = (r * ka) & (CE_FFT_SIZE - 1);   // W4096^(r k)
          const float2 td = cmul(cmul(tw256[md >> 4], tw16[md & 15]), v[idft16_at(m)]);
          const float2 ta = cmul(cmul(tw256[ma >> 4], tw16[ma & 15]), v[idft16_at(7 + m)]);
          const float2 keep = adv ? ta : td, send = adv ? td : ta;   // rows 2-3 collect the advance side, rows 0-1 the delay side
          s1[m] = make_float2(keep.x + shfl_xor_f(send.x, 32), keep.y + shfl_xor_f(send.y, 32));
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const float2 lo = s1[i], hi = i < 4 ? s1[5 + i] : make_float2(0.f, 0.f);
          const float2 keep = upper ? hi : lo, send = upper ? lo : hi;
          acc[i].x += keep.x + shfl_xor_f(send.x, 16);
          acc[i].y += keep.y + shfl_xor_f(send.y, 16);
        }
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) pw[i] += acc[i].x * acc[i].x + acc[i].y * acc[i].y;
    }
    // arg-max: T:683-696 takes the first maximum of the delay side (bins 0..143), the first maximum of the advance side
    // (bins 3952..4095) and prefers the delay side when the two are equal: one key orders all 288 bins that way
    unsigned long long key = 0ull;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int m = (upper ? 5 : 0) + i;
      if (m < 9) {
        const unsigned bd = (unsigned)(c + 16 * m);   // index on its side, 0..143
        const unsigned long long k = ((unsigned long long)__float_as_uint(pw[i]) << 32) | ((adv ? 0x7FFFFFFFu : 0xFFFFFFFFu) - bd);
        key = k > key ? k : key;
      }
    }
    key = wave_max_u64(key);
    const unsigned low = (unsigned)(key & 0xFFFFFFFFull);
    const int i_max = (low & 0x80000000u) ? (int)(0xFFFFFFFFu - low) : -(CE_TA_HALF - (int)(0x7FFFFFFFu - low));
    tot_ta += (double)i_max / (double)CE_FFT_SIZE / lp->scs;  // T:698, the reference's two float64 divisions
    if (lane == 0 && a.stage_s) a.stage_s[(item * NH + h) * 2 + 1] = (double)i_max;
  }
  if (lane == 0) a.ta[item] = (NH == 2) ? tot_ta / 2.0 : tot_ta;  // T:918-919
}

template <int L, int NH>
int narrow_launch_t(const CeLaunchCtx& c) {
  const unsigned grid = (unsigned)((c.args->n_local + NW - 1) / NW);
  hipLaunchKernelGGL((ce_narrow_kernel<L, NH>), dim3(grid), dim3(NT), c.lds, c.stream, c.dplan, c.re_idx, c.ta_inv, c.tw, *c.args);
  return (int)hipGetLastError();
}

template <int L, int NH>
int narrow_prepare_t(const CeLaunchCtx& c) {   // see prepare_t (ce_estimate_kernel.h): the dynamic-LDS limit only grows, under a mutex
  const void* fn = reinterpret_cast<const void*>(&ce_narrow_kernel<L, NH>);
  static std::mutex mu;
  static int lds_limit[CE_MAX_DEVICES];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= CE_MAX_DEVICES) return (int)hipErrorInvalidDevice;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (c.lds > lds_limit[dev]) {
      e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds);
      if (e != hipSuccess) return (int)e;
      lds_limit[dev] = c.lds;
    }
  }
  int nb = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, NT, c.lds);
  *c.blocks_per_cu = nb > 0 ? nb : 1;
  return (int)e;
}

template <int L, int NH>
int narrow_run_t(int op, const CeLaunchCtx& c) {
  return op == CE_OP_LAUNCH ? narrow_launch_t<L, NH>(c) : narrow_prepare_t<L, NH>(c);
}

}  // namespace

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
// Plans: linear interpolation or ce_dl_cnn.py's in-painting in its closed forms (+ the CNNSmoothingAlpha blend), smoothing none /
// mean / filter, 14- and 12-symbol grids, <= CE_NARROW_MAX_RE pilots per symbol, every hop's band inside the collapsed TA window
// (ce_api.hip decides; everything else -- the iterated in-painting among it -- runs on ce_estimate_kernel.h).
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
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the item, its pointers and the wave's LDS region are wave-uniform
  const int n_re = plan->n_re, n_re_pad = plan->n_re_pad;
  const int hs = plan->nrw_h_stride, halo = plan->nrw_halo;
  const CeNarrowLayout lay = ce_narrow_layout(NH, L, plan->nrw_nd_max, n_re_pad, hs, halo);
  const int prs = n_re_pad + 2 * halo;   // P row stride: [halo | n_re_pad | halo]; halo = the RC filter's reach (0 without the filter)
  const CeDevPlan* lp = reinterpret_cast<const CeDevPlan*>(smem + lay.off_plan);
  float2* tw256 = reinterpret_cast<float2*>(smem + lay.off_tw);   // [256] W256^j, then [16] W4096^i
  float2* tw16 = tw256 + 256;
  unsigned char* wbase = smem + lay.off_wave0 + wave * lay.wave_stride;
  float2* S = reinterpret_cast<float2*>(wbase + lay.stage_off);   // staged hop: rx rows [c][s], then DM-RS rows [l][s], n_re_pad each
  float2* P = reinterpret_cast<float2*>(wbase + lay.p_off) + halo;   // [NH][L][prs]: row (h, l) starts at P + (h L + l) prs; indices
                                                                     // -halo..-1 and n_re..n_re+halo-1 hold the virtual pilots / zeros
  float2* xs = reinterpret_cast<float2*>(wbase + lay.vp_off);     // the TA's [4 rows][x0: 16 | x1: 16] (over the staging bytes: free after the writer)
  float2* Hb = S;                                                 // after the hops: interpolated response [NH][L][hs]
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

  // ------------------------------------------------------------------ S1: one hop's pilots -> per-wave LDS rows
  // Rows: rx of CDM group c at DM-RS symbol s -> row c * nd + s; DM-RS symbol s of layer l -> row (NC + l) * nd + s.
  // All loads of a batch are requested before the first is stored (one round trip per 8 elements per lane).
  auto stage_hop = [&](int h) __attribute__((always_inline)) {
    const CeDevHop& hp = lp->hop[h];
    const int nd = hp.n_dmrs, rows_rx = NC * nd, rows = (NC + L) * nd;
    const int kp = (n_re + 63) >> 6, total = rows * kp;   // (row, 64-pilot chunk) pairs: both wave-uniform, decoded on the scalar unit
    PilotMap pm[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) pm[c] = pilot_map(hp, c);
    const int psym0 = hp.pil_sym0;
#pragma unroll 1
    for (int j0 = 0; j0 < ((CE_NRW_ABLATE & 8) ? 0 : total); j0 += 8) {
      float2 v[8];
      int dst[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        dst[u] = -1;
        const int j = j0 + u;
        if (j >= total) continue;   // wave-uniform: the rest of the batch is empty
        const int row = kp == 1 ? j : kp == 2 ? (j >> 1) : (int)(((unsigned)j * 43691u) >> 17), kk = j - row * kp;   // j / kp, kp <= 3, j < 2^15
        const int k = lane + 64 * kk;
        const bool ok = k < n_re;
        const int kc = ok ? k : 0;
        const float2* src;
        if (row < rows_rx) {
          const int c = (NC > 1 && row >= nd) ? 1 : 0, sy = row - c * nd;
          const int sc = pilot_sc(pm[NC > 1 ? c : 0], re_idx, kc);
          src = rx + (int64_t)hp.dmrs_sym[sy] * a.rs_sym + (int64_t)sc * a.rs_sc;
        } else {
          const int r2 = row - rows_rx, l = (r2 >= nd ? 1 : 0) + (r2 >= 2 * nd ? 1 : 0) + (r2 >= 3 * nd ? 1 : 0), sy = r2 - l * nd;
          src = pil + ((int64_t)(psym0 + sy) * a.ps_sym + (int64_t)l * a.ps_l) + (int64_t)kc * a.ps_re;
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
      const float2* Ph = P + h * L * prs;
      for (int i = lane; i < L * n_re; i += 64) {
        const int l = i / n_re;
        sp[i] = Ph[l * prs + (i - l * n_re)];
      }
    }
  };

  double tot_epre = 0.0, tot_noise = 0.0, tot_rsrp = 0.0, cfo_hops[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) cfo_hops[h] = 0.0;

#pragma unroll 1
  for (int h = 0; h < NH; ++h) {
    const CeDevHop& lh = lp->hop[h];
    float2* Ph = P + h * L * prs;
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
        Ph[(2 * c) * prs + k] = make_float2(acc0.x / beta_f / nd_f, acc0.y / beta_f / nd_f);
        if (2 * c + 1 < L) Ph[(2 * c + 1) * prs + k] = make_float2(acc1.x / beta_f / nd_f, acc1.y / beta_f / nd_f);
      }
    }
    wave_sync();

    // ---------------------------------------------------------------- S6: CDM de-spread
    if (L >= 2) {
      for (int i = lane; i < n_re / 2; i += 64) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 u = Ph[l * prs + 2 * i], v = Ph[l * prs + 2 * i + 1];
          const float2 m = make_float2((u.x + v.x) / 2.f, (u.y + v.y) / 2.f);
          Ph[l * prs + 2 * i] = m;
          Ph[l * prs + 2 * i + 1] = m;
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
          const float2 v = Ph[l * prs + k];
          mr += (double)v.x;
          mi += (double)v.y;
        }
        mr = wave_sum(mr);
        mi = wave_sum(mi);
        const float2 m = make_float2((float)(mr / (double)n_re), (float)(mi / (double)n_re));
        for (int k = lane; k < n_re; k += 64) Ph[l * prs + k] = m;
      }
      wave_sync();
    } else if (lp->smoothing == CE_SMOOTH_FILTER) {
      const int n_pils = lp->n_pils, rc_len = lp->rc_len, pad = rc_len / 2;
      const double vmx = lp->vp_mx, vin = lp->vp_inv_n, vid = lp->vp_inv_denom;
      const double* rc = lp->rc;
      // conv([virtual head ; P ; virtual tail], rc, "same") cropped back to P (T:649-664), float64 MACs (T:477-490).  The
      // virtual pilots are written INTO the row's halo (index -1 - e and n_re + e for distance e + 1 from the band; zeros
      // beyond them, up to the filter's reach), so the taps run over one contiguous piece of LDS without a branch.
      for (int i = lane; i < L * 2 * halo; i += 64) {
        const int l = i / (2 * halo), j = i - l * (2 * halo);
        Ph[l * prs + (j < halo ? j - halo : n_re + j - halo)] = make_float2(0.f, 0.f);
      }
      wave_sync();
#pragma unroll 1
      for (int l0 = 0; l0 < L; l0 += 2) {
        {  // virtual pilots of layers l0 (lanes 0-31) and l0 + 1 (lanes 32-63): one 16-lane DPP row per band edge
          const int sub = lane >> 5, e = (lane >> 4) & 1;
          const bool on = l0 + sub < L;
          float2* Pl = Ph + (on ? l0 + sub : l0) * prs;
          virtual_pilots(Pl, n_re, n_pils, e != 0, lane & 15, vmx, vin, vid,
                         [&](int dist, float2 val) { if (on) Pl[e ? n_re + dist : -1 - dist] = val; });
        }
        wave_sync();
      }
#pragma unroll 1
      for (int l = 0; l < L; ++l) {
        float2* Pl = Ph + l * prs;
        double yr[3], yi[3];   // every output of the row is formed before the first is written
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          yr[i] = yi[i] = 0.0;
          if (64 * i >= n_re) continue;   // wave-uniform
          const int m = lane + 64 * i < n_re ? lane + 64 * i : n_re - 1;   // (idle lanes repeat the last output: in-range reads, result unused)
          const float2* x = Pl + m + pad;   // tap j meets x[-j]
          double ar = 0.0, ai = 0.0;
          if (rc_len == 15) {   // the usual case (>= 3 PRB of a comb-2 DM-RS): constant trip count, immediate LDS offsets
#pragma unroll
            for (int j = 0; j < 15; ++j) {
              const float2 v = x[-j];
              const double w = rc[j];
              ar += w * (double)v.x;
              ai += w * (double)v.y;
            }
          } else {
            for (int j = 0; j < rc_len; ++j) {
              const float2 v = x[-j];
              const double w = rc[j];
              ar += w * (double)v.x;
              ai += w * (double)v.y;
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
        wave_sync();
      }
      if (lp->interp == CE_INTERP_CNN && lp->cnn_alpha > 0.f) {
        // optional blend with one low-pass pass over the smoothed pilots (src/ce_dl_cnn.py:712-715)
        const float al = lp->cnn_alpha;
#pragma unroll 1
        for (int l = 0; l < L; ++l) {
          float2* Pl = Ph + l * prs;
          float2 b[3];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int k = lane + 64 * i;
            b[i] = make_float2(0.f, 0.f);
            if (k < n_re) {
              const float2 rcv = Pl[k];
              float2 sm = rcv;
              if (n_re > 2) {
                double yr2, yi2;
                lp3(Pl, k, n_re, &yr2, &yi2);
                sm = make_float2((float)yr2, (float)yi2);
              }
              b[i] = make_float2(rcv.x + al * (sm.x - rcv.x), rcv.y + al * (sm.y - rcv.y));
            }
          }
          wave_sync();
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int k = lane + 64 * i;
            if (k < n_re) Pl[k] = b[i];
          }
          wave_sync();
        }
      }
    }
    dump_stage(1, h);

    // ---------------------------------------------------------------- S9, S11: residual noise, RSRP
    {
      float noise_part = 0.f, rsrp_part = 0.f;
      for (int k = lane; k < n_re; k += 64) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const float2 v = Ph[l * prs + k];
          rsrp_part += v.x * v.x + v.y * v.y;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float2 h0 = Ph[(2 * c) * prs + k];
          const float2 h1 = (2 * c + 1 < L) ? Ph[(2 * c + 1) * prs + k] : make_float2(0.f, 0.f);
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
  // (1) Linear interpolation of every (hop, layer) over the hop's band -> H rows in LDS (the staging bytes, free now):
  //     left + alpha (right - left), also AT pilots, as the reference does; flat beyond the first / last pilot (T:311-338).
  // (2) A subcarrier's (14 symbols x L layers) is 7L float4; SCS = 64 / 7L subcarriers per wave iteration, each active lane
  //     owns ONE (symbol pair, layer) float4 phase for the whole item (its two phasors and hop candidates are lane
  //     constants) and a wave iteration stores ACTIVE * 16 contiguous bytes: one H read, two complex multiplies, one store.
  //     The hop of an element is the last hop whose symbol AND subcarrier range cover it (T:872-896).
  if (!(CE_NRW_ABLATE & 4)) {
#pragma unroll 1
    for (int hl = 0; hl < NH * L; ++hl) {
      const int h = hl / L, l = hl - h * L, c = l >> 1;
      const CeDevHop& lh = lp->hop[h];
      const float2* Pl = P + hl * prs;
      const int dpp = lh.dpp[c], lastp = lh.last_idx[c], nb = lh.n_sc_hop;
      auto lin_at = [&](int p) __attribute__((always_inline)) -> float2 {   // T:311-338
        const int q = (int)(((unsigned)p * 0xAAABu) >> 19), r = p - 12 * q;   // p / 12 for p < 2^15
        const float al = lh.alpha[c][r];
        int ro = q * dpp + lh.r_ord[c][r], lo = ro - 1;
        if (p >= lastp) lo = ro = n_re - 1;     // at/after the last pilot: hold (T:316,321)
        lo = lo < 0 ? 0 : lo;                    // at/before the first pilot: hold (T:315,320)
        const float2 u = Pl[lo], v = Pl[ro];
        return make_float2(u.x + al * (v.x - u.x), u.y + al * (v.y - u.y));
      };
      const int cnn_mode = lp->interp == CE_INTERP_CNN ? lp->cnn_comb2 : 0;   // (the iterated in-painting never reaches this kernel)
      if (cnn_mode == 0) {
        for (int p = lane; p < nb; p += 64) Hb[hl * hs + p] = lin_at(p);
      } else if (cnn_mode == 1) {
        // ce_dl_cnn.py's in-painting + two low-pass passes for a comb-2 DM-RS in closed form (ce_estimate_kernel.h: cnn2_at):
        // pilots keep their value (C:507-508), an RE between pilots k and k+1 gets (P[k-1] + 15 P[k] + 15 P[k+1] + P[k+2]) / 32,
        // pilot indices outside the band reflected the way the RE-domain reflect padding (C:433-451) maps them
        const int off = ((lh.mask12 >> (16 * c)) & 1u) ? 0 : 1, K = n_re - 1;
        for (int p = lane; p < nb; p += 64) {
          const int q = p - off;
          float2 val;
          if (q >= 0 && !(q & 1)) {
            val = Pl[q >> 1];
          } else {
            const int kl = (q - 1) >> 1;  // pilot on the left (-1: the RE in front of the first pilot)
            auto refl = [&](int k) { return k < 0 ? -k - off : (k > K ? 2 * K + 1 - off - k : k); };
            const float2 a0 = Pl[refl(kl - 1)], b0 = Pl[refl(kl)], c0 = Pl[refl(kl + 1)], d0 = Pl[refl(kl + 2)];
            val = make_float2(((a0.x + d0.x) + 15.f * (b0.x + c0.x)) * (1.f / 32.f), ((a0.y + d0.y) + 15.f * (b0.y + c0.y)) * (1.f / 32.f));
          }
          Hb[hl * hs + p] = val;
        }
      } else {
        // any mask whose in-painting reaches its fixed point within the reference's iteration count (ce_estimate_kernel.h:
        // cnnfp_at): 5-tap binomial (two [1 2 1] / 4 passes, reflect padding C:433-451) over the linear fill, pilots restored
        const unsigned m12 = (lh.mask12 >> (16 * c)) & 0xFFFu;
        auto rf = [&](int i) { return i < 0 ? -i : (i >= nb ? 2 * nb - 2 - i : i); };
        for (int p = lane; p < nb; p += 64) {
          const int q12 = (int)(((unsigned)p * 0xAAABu) >> 19), r12 = p - 12 * q12;
          float2 acc = make_float2(0.f, 0.f);
          if ((m12 >> r12) & 1u) {
            acc = lin_at(p);   // a pilot RE keeps its value (C:507-508)
          } else {
#pragma unroll
            for (int d = -1; d <= 1; ++d) {
              const int i = rf(p + d);
              const float wd = d == 0 ? 0.5f : 0.25f;
#pragma unroll
              for (int e = -1; e <= 1; ++e) {
                const float2 x = lin_at(rf(i + e));
                const float w = wd * (e == 0 ? 0.5f : 0.25f);
                acc.x += w * x.x;
                acc.y += w * x.y;
              }
            }
          }
          Hb[hl * hs + p] = acc;
        }
      }
    }
    wave_sync();
#ifndef CE_NRW_SCS1
#define CE_NRW_SCS1 9   // one layer: subcarriers per wave iteration (9 = 63 active lanes, 1008 B; 8 = 56 lanes, 896 B = whole 128-byte lines: measured equal)
#endif
    auto store_rows = [&](auto ns2c) __attribute__((always_inline)) {
    constexpr int NS2 = decltype(ns2c)::value;   // symbol pairs per subcarrier: 7 (14-symbol slot) or 6 (12 symbols: extended CP)
    constexpr int ROW4 = NS2 * L, SCS = (L == 1 && NS2 == 7) ? CE_NRW_SCS1 : 64 / ROW4, ACTIVE = SCS * ROW4;
    const int ph = lane % ROW4, sc_off = lane / ROW4;
    const int n_sc = lp->n_sc;
    const CeDevHop& g0 = lp->hop[0];
    const CeDevHop& g1 = lp->hop[NH - 1];
    const int b0 = g0.sc0, n0 = g0.n_sc_hop, b1 = g1.sc0, n1 = g1.n_sc_hop;
    float2 rsel[2];
    bool c0[2], c1[2];   // element e's symbol lies in hop 0's / hop 1's symbol range (lane constants)
    int off0[2], off1[2];   // H row of (hop, layer of element e), as an element offset
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int rem = 2 * ph + e, sym = rem / L, l = rem - sym * L;
      c0[e] = sym >= g0.sym0 && sym < g0.sym1;
      c1[e] = NH == 2 && sym >= g1.sym0 && sym < g1.sym1;
      rsel[e] = (c0[e] || c1[e]) ? rot_final[sym < 16 ? sym : 0] : make_float2(0.f, 0.f);   // rot_final == 1 when no CFO ramp applies
      off0[e] = l * hs;
      off1[e] = ((NH - 1) * L + l) * hs;
    }
    // wave-uniform store base (scalar registers) + a constant per-lane offset: no per-iteration vector address arithmetic
    float4* obase = reinterpret_cast<float4*>(a.out + item * ((int64_t)n_sc * (2 * NS2) * L));
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int n_it = (n_sc + SCS - 1) / SCS;   // wave-uniform trip count
    const bool act = lane < ACTIVE;
    // one element: the H value of the last hop whose symbol AND subcarrier range cover it, else +0 -- by selects on the LDS
    // address and an AND on the bits (exact; no exec-masked regions, so the LDS read and the store of consecutive iterations overlap)
    auto pick = [&](int e, int d0, int d1, bool in0, bool in1) __attribute__((always_inline)) -> float2 {
      const bool u1 = in1 && c1[e], u0 = !u1 && in0 && c0[e];
      const int idx = u1 ? off1[e] + d1 : (u0 ? off0[e] + d0 : 0);
      const float2 v = Hb[idx];
      const int m = (u1 || u0) ? -1 : 0;
      return make_float2(__int_as_float(__float_as_int(v.x) & m), __int_as_float(__float_as_int(v.y) & m));
    };
    // one layer and both symbols of every lane's pair in the same hops: the two elements of a float4 are the same H value
    const bool twin = L == 1 && __builtin_amdgcn_ballot_w64(c0[0] != c0[1] || c1[0] != c1[1]) == 0ull;
    // Iterations whose SCS subcarriers touch a hop's band form one interval per hop (wave-uniform, scalar registers); every
    // other iteration stores zeros and does nothing else -- in the reference harness's 52-PRB grids with 3-PRB allocations
    // that is nine iterations of ten.
    const int lo0 = b0 / SCS, hi0 = (b0 + n0 + SCS - 1) / SCS;
    const int lo1 = NH == 2 ? b1 / SCS : 0, hi1 = NH == 2 ? (b1 + n1 + SCS - 1) / SCS : 0;
    const bool last_ok = act && sc_off + (n_it - 1) * SCS < n_sc;   // the last iteration may run past the grid's end
#pragma unroll 1
    for (int it = 0; it < n_it; ++it) {
      const bool live = it + 1 < n_it ? act : last_ok;
      float4* o = obase + (size_t)it * ACTIVE + lane;
      if (!((it >= lo0 && it < hi0) || (it >= lo1 && it < hi1))) {
        if (live) *o = z4;
        continue;
      }
      const int sc = sc_off + it * SCS;
      const int d0 = sc - b0, d1 = sc - b1;
      const bool in0 = live && (unsigned)d0 < (unsigned)n0, in1 = NH == 2 && live && (unsigned)d1 < (unsigned)n1;
      const float2 y0 = pick(0, d0, d1, in0, in1);
      const float2 y1 = twin ? y0 : pick(1, d0, d1, in0, in1);
      const float2 ya = cmul(y0, rsel[0]), yb = cmul(y1, rsel[1]);
      if (live) *o = make_float4(ya.x, ya.y, yb.x, yb.y);
    }
    };   // store_rows
    if (lp->n_sym == CE_MAX_SYMBOLS) store_rows(std::integral_constant<int, 7>{});
    else store_rows(std::integral_constant<int, 6>{});   // (the host sends only 14- and 12-symbol grids here)
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
  wave_sync();   // the pilot buffer `xs` lies over the H rows the writer has just read
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
    float pw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) pw[i] = 0.f;
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
      const float2* Pl = P + (h * L + l) * prs;
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
          const int i0 = unit ? pilot_at(n0) : -1;
          xs[row * 32 + c] = i0 >= 0 ? Pl[i0] : make_float2(0.f, 0.f);
          if (two) {
            const int i1 = unit ? pilot_at(n0 + 256) : -1;
            xs[row * 32 + 16 + c] = i1 >= 0 ? Pl[i1] : make_float2(0.f, 0.f);
          }
        }
        wave_sync();
        float2 v[16];
        if (two) {
          const float2 wc = tw256[16 * co];   // W16^c
#pragma unroll
          for (int aq = 0; aq < 16; ++aq) {
            const float2 x0 = xs[row * 32 + aq], x1 = xs[row * 32 + 16 + aq];
            v[aq] = cmul(cadd(x0, cmul(x1, wc)), tw256[(aq * co) & 255]);
          }
        } else {
#pragma unroll
          for (int aq = 0; aq < 16; ++aq) v[aq] = cmul(xs[row * 32 + aq], tw256[(aq * co) & 255]);
        }
        wave_sync();   // (the next round overwrites xs)
        idft16_inplace(v);   // Y_r[c + 16 d] at v[idft16_at(d)]
        // bin k = c + 16 m (delay side) / 3952 + c + 16 m (advance side): W4096^(r k) = W4096^(r c [+ 3952 r]) W256^(r m) -- one
        // lane constant per side and one table read per m, shared by the two sides
        const int md0 = (r * co) & (CE_FFT_SIZE - 1), ma0 = (r * (CE_FFT_SIZE - CE_TA_HALF + co)) & (CE_FFT_SIZE - 1);
        const float2 wd = cmul(tw256[md0 >> 4], tw16[md0 & 15]), wa = cmul(tw256[ma0 >> 4], tw16[ma0 & 15]);
        // rows 2-3 collect the advance side, rows 0-1 the delay side (shuffle over 32 lanes); of a side's nine bins the even row
        // keeps m = 0..4, the odd row m = 5..8 (shuffle over 16 lanes).  Bins are taken in the order (i, 5 + i) so that a pair
        // is folded into its accumulator as soon as both halves exist: two first-stage sums live at a time, not nine.
        auto side_sum = [&](int m) __attribute__((always_inline)) -> float2 {
          const float2 t = tw256[(r * m) & 255];
          const float2 td = cmul(cmul(wd, t), v[idft16_at(m)]);
          const float2 ta = cmul(cmul(wa, t), v[idft16_at(7 + m)]);
          const float2 keep = adv ? ta : td, send = adv ? td : ta;
          return make_float2(keep.x + shfl_xor_f(send.x, 32), keep.y + shfl_xor_f(send.y, 32));
        };
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const float2 lo = side_sum(i), hi = i < 4 ? side_sum(5 + i) : make_float2(0.f, 0.f);
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

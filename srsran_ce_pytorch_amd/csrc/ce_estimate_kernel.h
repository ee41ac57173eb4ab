// Fused PUSCH DM-RS channel-estimation kernel for gfx950 (MI355X): one workgroup per
// slot x Rx-port work item; everything between the received grid and the channel grid stays
// in LDS/registers, the (n_sc, n_sym, L) output block is written once with 16-byte stores.
//
// Stage map (reference: src/ce_rule_tensorized.py, "T"), per hop:
//   load            S1  T:571-581   pilot REs of the DM-RS symbols (+ pilots) -> registers (register path)
//   cfo             S4  T:357-426   inner products of the first two DM-RS symbols -> CFO of the hop
//   ls              S2,S3,S5 T:584-613  EPRE, LS (x conj(pilot)), de-rotation, DM-RS average -> P in LDS
//   despread        S6  T:620-628
//   smooth_*        S7  T:633-668   mean | virtual pilots (T:69-140) + RC FIR (T:459-493)
//   residual        S9,S11 T:700-730 reconstructed pilots, noise, RSRP
//   [time_alignment S8  T:670-698   pruned 4096-point inverse DFT (3 radix-16 passes, only the residues
//                                   mod 16 that carry pilots, only the 288 examined bins), arg-max -- see below]
// then once per item:
//   write_grid      S10 + epilogue T:237-354, T:921-929  linear interpolation, symbol replicate, CFO ramp
// and, while the grid's stores drain, per hop:
//   time_alignment  S8  (above) -- nothing the grid needs depends on it, so it runs after the writer
//
// This header holds the kernel template; the translation units ce_inst_*.hip instantiate slices of the
// (layers, hops, register-path shape, feature set) space so they compile in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "ce_plan.h"

namespace {

constexpr int NT = CE_THREADS;
constexpr int NW = NT / 64;
constexpr double kInvPi = 0.31830988618379067153776752674503;

#ifndef CE_ABLATE
#define CE_ABLATE 0   // timing experiments only (tools/ablate.py): 1 TA, 2 smoothing, 4 residual, 8 writer, 16 CFO, 32 input loads, 64 writer without LDS reads
#endif
// (register budgets, feature sets and the TA placement policy: ce_plan.h -- the host sizes the LDS by the same rules)
#ifndef CE_LEAN
#define CE_LEAN 0         // code-size experiment (never shipped): 2 drops the element-wise writer, 1 also the staged one where the direct writer exists
#endif
#ifndef CE_PRIO
#define CE_PRIO 0         // wave-priority experiments (0 = none): 1 writer high, 2 estimation stages high
#endif
#ifndef CE_WR_BR_UNROLL
#define CE_WR_BR_UNROLL 2 // direct writer with per-element branches: unrolled iterations
#endif
#ifndef CE_WR_WIDE
#define CE_WR_WIDE 0      // the same for the kernels built with the 2-wave bound (0 = per-element branches)
#endif
#ifndef CE_GEN_KU
#define CE_GEN_KU 2       // re-read path: pilot REs per thread whose loads are requested together in the CFO / LS / residual stages (1, 2)
#endif
#ifndef CE_GEN_NDC
#define CE_GEN_NDC 1      // re-read path: LS / residual stages specialised for 1..4 DM-RS symbols (0: run-time symbol loop, A/B builds)
#endif
#ifndef CE_WR_UNROLL
#define CE_WR_UNROLL 4    // direct writer: iterations whose LDS reads are requested together
#endif
#ifndef CE_PF1_LIMIT
#define CE_PF1_LIMIT 4    // two hops: up to this many pilot REs x symbols per thread, hop 2's pilots are prefetched with hop 1's
#endif
#ifndef CE_RELOAD_RESID
#define CE_RELOAD_RESID 0 // 1: the residual stage re-reads rx / pilots instead of keeping them in registers across smoothing
#endif

#if defined(CE_STAMPS)
// diagnostic build only (tools/stamps.py): per-stage wall-clock stamps of thread 0, written to a buffer
// nothing else reads (CeKernelArgs::stamps, set through ce_debug_set_stamps); never compiled into the shipped library
#define STAMP(i)                                                      \
  do {                                                                \
    if (threadIdx.x == 0 && a.stamps) a.stamps[item * 16 + (i)] = wall_clock64(); \
  } while (0)
// where the workgroup runs: HW_ID (wave / SIMD / CU / SH / SE fields) in the low word, XCC_ID in the high word
#define STAMP_HWID(i)                                                 \
  do {                                                                \
    if (threadIdx.x == 0 && a.stamps)                                 \
      a.stamps[item * 16 + (i)] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | \
                                  (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);                      \
  } while (0)
#else
#define STAMP(i)
#define STAMP_HWID(i)
#endif
#if defined(CE_STAMP_STARTUP)
#define STAMP_TA(i)
#else
#define STAMP_TA(i) STAMP(i)
#endif
#if defined(CE_STAMPS) && defined(CE_STAMP_STARTUP)   // slots 14 / 15 stamp the start-up instead of the TA stage (tools/stamps.py --startup)
#define STAMP_STARTUP(i)                                              \
  do {                                                                \
    if (threadIdx.x == 0 && a.stamps) a.stamps[(a.item0 + item_of(blockIdx.x, a.n_ports, a.n_local)) * 16 + (i)] = wall_clock64(); \
  } while (0)
#else
#define STAMP_STARTUP(i)
#endif

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {  // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// Measured and dropped (tools/ablate.py, same-box A/B): non-temporal stores, a per-wave cap on stores in
// flight (s_waitcnt vmcnt(N) after each store, N = 2..16), raised/lowered wave priority around the writer,
// touching a later work item's DM-RS rows ahead of time -- none faster, the last one 17 % slower.
__device__ __forceinline__ void store_f4(float4* p, float4 v) { *p = v; }
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Compiler fence on a register value (no instruction).  Used after the CFO stage so its 14 pilot
// products are not kept alive (56 VGPRs) for re-use by the LS stage across two barriers.
__device__ __forceinline__ void pin(float2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }

// sin(pi x), cos(pi x) of a float64 argument, to float32 accuracy: float32 sincospi of the rounded
// argument plus a first-order correction for the rounding residue.  (ocml's float64 sincospi parks
// ~70 VGPRs of polynomial constants, which would cost the whole kernel a wave of occupancy.)
__device__ __forceinline__ void sincospi_f64arg(double x, float* sn, float* cs) {
  const float hi = (float)x;
  const float lo = (float)(x - (double)hi) * 3.14159265358979323846f;
  float sh, ch;
  sincospif(hi, &sh, &ch);
  *sn = fmaf(lo, ch, sh);
  *cs = fmaf(-lo, sh, ch);
}

// Cross-lane moves on the DPP path (VALU) instead of ds_bpermute (LDS crossbar, ~100 cycles a hop):
// quad swaps, half-row / row mirrors, then row_bcast15 / row_bcast31 -- lane 63 ends up with the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {  // every lane returns the wave total
  v += dpp_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_f64<0x140, 0xF>(v);  // row_mirror
  v += dpp_f64<0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
  v += dpp_f64<0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
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

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xF, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long umax64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {  // every lane returns the wave max
  v = umax64(v, dpp_u64<0xB1, 0xF>(v));
  v = umax64(v, dpp_u64<0x4E, 0xF>(v));
  v = umax64(v, dpp_u64<0x141, 0xF>(v));
  v = umax64(v, dpp_u64<0x140, 0xF>(v));
  v = umax64(v, dpp_u64<0x142, 0xA>(v));
  v = umax64(v, dpp_u64<0x143, 0xC>(v));
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}

// inverse (positive-exponent) radix-4 butterfly, natural order in and out
__device__ __forceinline__ void r4inv(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
  const float2 jd = make_float2(-d13.y, d13.x);  // +j (a1 - a3)
  a0 = cadd(s02, s13);
  a1 = cadd(d02, jd);
  a2 = csub(s02, s13);
  a3 = csub(d02, jd);
}

// X[c] = sum_b v[b] exp(+j 2 pi b c / 16), in registers (radix-4 x radix-4), natural order
__device__ __forceinline__ void idft16(float2 (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
  for (int j = 0; j < 4; ++j) r4inv(v[j], v[j + 4], v[j + 8], v[j + 12]);  // v[j+4m] = A_j[m]
  // B_j[m] = A_j[m] * W16^(j m)
  v[1 + 4] = cmul(v[1 + 4], make_float2(C1, S1));      // j=1,m=1: W^1
  v[1 + 8] = cmul(v[1 + 8], make_float2(R2, R2));      // j=1,m=2: W^2
  v[1 + 12] = cmul(v[1 + 12], make_float2(S1, C1));    // j=1,m=3: W^3
  v[2 + 4] = cmul(v[2 + 4], make_float2(R2, R2));      // j=2,m=1: W^2
  v[2 + 8] = make_float2(-v[2 + 8].y, v[2 + 8].x);     // j=2,m=2: W^4 = +j
  v[2 + 12] = cmul(v[2 + 12], make_float2(-R2, R2));   // j=2,m=3: W^6
  v[3 + 4] = cmul(v[3 + 4], make_float2(S1, C1));      // j=3,m=1: W^3
  v[3 + 8] = cmul(v[3 + 8], make_float2(-R2, R2));     // j=3,m=2: W^6
  v[3 + 12] = cmul(v[3 + 12], make_float2(-C1, -S1));  // j=3,m=3: W^9
  // X[m + 4n] = sum_j B_j[m] W4^(j n): radix-4 over j for each m; B_j[m] sits at v[j + 4m]
  float2 o[16];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float2 b0 = v[4 * m], b1 = v[4 * m + 1], b2 = v[4 * m + 2], b3 = v[4 * m + 3];
    r4inv(b0, b1, b2, b3);
    o[m] = b0;
    o[m + 4] = b1;
    o[m + 8] = b2;
    o[m + 12] = b3;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = o[i];
}

// 16-lane (one DPP row) helpers for the band-edge fits: VALU cross-lane moves instead of ds_bpermute round trips
// (a __shfl on a double is two of those, ~100+ cycles each, and the fit chains about twenty of them).
template <int N>
__device__ __forceinline__ double row_shr(double v) {  // lane j of a row receives lane j - N of the same row, 0 for j < N
  return dpp_f64<0x110 + N, 0xF>(v);
}
__device__ __forceinline__ double row_sum(double v) {  // every lane of a row receives the sum over its 16 lanes
  v += dpp_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_f64<0x140, 0xF>(v);  // row_mirror
  return v;
}

// One 16-lane group per band edge: straight-line fit of modulus and unwrapped phase of the n_pils pilots
// next to the edge, extrapolated n_pils positions outwards (T:69-140, T:35-66).  `out(e)` receives the
// virtual pilot at distance e+1 from the band (e = 0 is adjacent to the first / last real pilot).
template <typename Out>
__device__ __forceinline__ void virtual_pilots(const float2* Pl, int n_re, int n_pils, bool tail, int j,
                                               double mx, double inv_n, double inv_denom, Out out) {
  const double PI = 3.14159265358979323846;
  float2 v = make_float2(0.f, 0.f);
  if (j < n_pils) v = tail ? Pl[n_re - 1 - j] : Pl[j];
  const double amp = (double)hypotf(v.x, v.y);
  const double ang = (double)atan2f(v.y, v.x);
  double outr, outi;
  if (n_pils == 1) {  // T:95-101
    float sn, cs;
    sincospi_f64arg(ang * kInvPi, &sn, &cs);
    outr = amp * (double)cs;
    outi = amp * (double)sn;
  } else {
    // unwrap: per-gap correction then inclusive prefix sum over the 16-lane group
    const double prev = row_shr<1>(ang);
    double corr = 0.0;
    if (j >= 1 && j < n_pils) {
      const double dd = ang - prev;
      const double t = dd + PI;
      double ddmod = t - 2.0 * PI * floor(t * (0.5 * kInvPi));  // remainder(dd + pi, 2 pi) in [0, 2 pi)
      ddmod -= PI;
      if (ddmod == -PI && dd > 0.0) ddmod += 2.0 * PI;
      corr = fabs(dd) < PI ? 0.0 : ddmod - dd;
    }
    corr += row_shr<1>(corr);   // inclusive prefix sum over the row (lanes below the shift receive 0)
    corr += row_shr<2>(corr);
    corr += row_shr<4>(corr);
    corr += row_shr<8>(corr);
    const double ph = ang + corr;
    const bool in = j < n_pils;
    const double x = (double)j;
    double sa = in ? amp : 0.0, sxa = in ? x * amp : 0.0, sp = in ? ph : 0.0, sxp = in ? x * ph : 0.0;
    sa = row_sum(sa);
    sxa = row_sum(sxa);
    sp = row_sum(sp);
    sxp = row_sum(sxp);
    const double n = (double)n_pils;
    const double ma = sa * inv_n, mp = sp * inv_n;
    const double a_amp = (sxa - n * mx * ma) * inv_denom, b_amp = ma - a_amp * mx;
    const double a_ph = (sxp - n * mx * mp) * inv_denom, b_ph = mp - a_ph * mx;
    const double k = (double)(j - n_pils);  // positions -nV .. -1
    const double va = a_amp * k + b_amp, vp = a_ph * k + b_ph;
    float sn, cs;
    sincospi_f64arg(vp * kInvPi, &sn, &cs);
    outr = va * (double)cs;
    outi = va * (double)sn;
  }
  if (j < n_pils) out(n_pils - 1 - j, make_float2((float)outr, (float)outi));  // lane j sits at distance n_pils - j
}

// Workgroup index -> work item.  Workgroups b, b+8, b+16, .. share an XCD (and its L2), so the Rx ports of one
// slot -- which read the same DM-RS symbols -- are dealt to indices 8 apart.  Placement only affects speed.
#ifndef CE_XCD_MAP
#define CE_XCD_MAP 1
#endif
__device__ __forceinline__ int64_t item_of(int64_t b, int n_ports, int64_t n_items) {
  if (!CE_XCD_MAP) return b;
  const int64_t per = 8 * (int64_t)n_ports, g = b / per;
  if ((g + 1) * per > n_items) return b;  // ragged tail: identity
  const int j = (int)(b - g * per);
  return (g * 8 + (j & 7)) * n_ports + (j >> 3);
}

// subcarrier of pilot k of a CDM group: computed for a contiguous allocation, looked up otherwise (T:572-576).  The few
// plan fields this takes are fetched ONCE into a PilotMap (scalar registers) and then used for every pilot RE of the
// thread: read through the plan pointer inside the per-RE code they become a chain of dependent scalar loads in front of
// every pilot load (measured: most of a 20 us start-up per workgroup under load).
struct PilotMap {
  int contig, dpp, prb_start, re_off;
  unsigned magic;
  unsigned long long pos;
};
__device__ __forceinline__ PilotMap pilot_map(const CeDevHop& hp, int c) {
  PilotMap m;
  m.contig = hp.contig; m.dpp = hp.dpp[c]; m.prb_start = hp.prb_start; m.re_off = hp.re_off[c];
  m.magic = hp.div_magic[c]; m.pos = hp.pos_packed[c];
  return m;
}
__device__ __forceinline__ int pilot_sc(const PilotMap& m, const uint16_t* __restrict__ re_idx, int k) {
  if (m.contig) {
    const int q = m.dpp == 1 ? k : (int)__umulhi((unsigned)k, m.magic);
    const int j = k - q * m.dpp;
    return 12 * (m.prb_start + q) + (int)((m.pos >> (4 * j)) & 15u);
  }
  return re_idx[m.re_off + k];
}

// ---- src/ce_dl_cnn.py's fixed-weight stencil (C:433-508), the reference's alternative to linear interpolation ----
__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// one pass of the [.25 .5 .25] stencil with reflect padding at index i, float64 (C:433-451)
__device__ __forceinline__ void lp3(const float2* x, int i, int n, double* yr, double* yi) {
  const float2 a = x[reflect_idx(i - 1, n)], b = x[i], c = x[reflect_idx(i + 1, n)];
  *yr = (0.25 * (double)a.x + 0.5 * (double)b.x) + 0.25 * (double)c.x;
  *yi = (0.25 * (double)a.y + 0.5 * (double)b.y) + 0.25 * (double)c.y;
}

// two passes (C:454-470): y2[i] = stencil(y1)[i], y1 = stencil(x), each pass with its own reflect padding
__device__ __forceinline__ float2 lp3x2(const float2* x, int i, int n) {
  double r[3], q[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) lp3(x, reflect_idx(i - 1 + d, n), n, &r[d], &q[d]);
  return make_float2((float)((0.25 * r[0] + 0.5 * r[1]) + 0.25 * r[2]), (float)((0.25 * q[0] + 0.5 * q[1]) + 0.25 * q[2]));
}

// Element (row c, column d) of a 16 x 16 residue block of the TA transform.  The first pass writes columns, the second
// reads and writes rows (a 16-element row stride would put a row access on two LDS banks).  Two layouts (ce_ta_row):
// SWZ = false: rows of 17 (one padding column), immediate offsets from one address register; SWZ = true: unpadded, column
// d ^ c -- 1 KB less per 8 residues, which is what lets a second set of blocks fit beside the first without costing a
// workgroup per CU (time_alignment), but a computed address per element: +30 spilled VGPRs where the stage runs inside the
// hop loop of the multi-layer two-hop kernels (-12 % there), so only the kernels that run it after the writer use it.
template <bool SWZ>
__device__ __forceinline__ int ta_at(int c, int d) { return SWZ ? c * 16 + (d ^ c) : c * 17 + d; }

// One run of g <= G unknown REs between two pilots (or between a band edge and a pilot), iterated in registers exactly as
// the reference iterates the whole band (C:489-505): x <- (0.25 x[i-1] + 0.5 x[i] + 0.25 x[i+1]) / (den + eps) in float64
// with a float32 round trip per iteration, den = the partial-convolution mask sum, the mask growing by one RE per side and
// iteration.  `kind`: 0 = pilots Lv / Rv on both sides; 1 = the run starts at subcarrier 0 (reflect padding: index -1 is
// index +1, i.e. the left neighbour of element 0 is its right neighbour); 2 = the run ends at the band's last subcarrier
// (index n is index n-2: the right neighbour of the last element is its left neighbour).  Stops early at the run's own
// bitwise fixed point.  num / (den + eps) is evaluated as num * rcp[code], rcp = the five possible reciprocals (<= 1 ulp in
// float64 before the float32 round trip).
template <int G>
__device__ __forceinline__ void cnn_inpaint_run(float2 (&x)[G], int g, float2 Lv, float2 Rv, int kind, int n_iters, const double* rcp) {
  unsigned m = 0u;  // bit i: mask of element i (unknown REs start at 0; the bounding pilots count as 1)
#pragma unroll 1
  for (int it = 0; it < n_iters; ++it) {
    float2 xn[G];
    unsigned mn = m;
    bool changed = false;
#pragma unroll
    for (int i = 0; i < G; ++i) {
      xn[i] = x[i];
      if (i < g) {
        float2 a = Lv, c = Rv;
        int ma = 1, mc = 1;
        if (i > 0) {
          a = x[i > 0 ? i - 1 : 0];
          ma = (int)((m >> (i > 0 ? i - 1 : 0)) & 1u);
        }
        if (i < g - 1) {
          c = x[i + 1 < G ? i + 1 : G - 1];
          mc = (int)((m >> (i + 1)) & 1u);
        }
        if (kind == 1 && i == 0) { a = c; ma = mc; }
        if (kind == 2 && i == g - 1) { c = a; mc = ma; }
        const float2 b = x[i];
        const int mb = (int)((m >> i) & 1u), code = ma + 2 * mb + mc;
        const double w = rcp[code];
        xn[i] = make_float2((float)(((0.25 * (double)a.x + 0.5 * (double)b.x) + 0.25 * (double)c.x) * w),
                            (float)(((0.25 * (double)a.y + 0.5 * (double)b.y) + 0.25 * (double)c.y) * w));
        if (code > 0) mn |= 1u << i;
        changed |= (__float_as_uint(xn[i].x) != __float_as_uint(b.x)) | (__float_as_uint(xn[i].y) != __float_as_uint(b.y));
      }
    }
    changed |= mn != m;
#pragma unroll
    for (int i = 0; i < G; ++i) x[i] = xn[i];
    m = mn;
    if (!changed) break;
  }
}

// All runs of unknown REs of one layer over a hop band of n subcarriers, one run per thread and turn: between two pilots
// the iteration never looks past them (pilots keep their value and their mask), and at the band edges the reflect padding
// folds back into the run itself, so the runs are independent and need neither LDS traffic nor barriers between
// iterations (150 barrier-separated LDS sweeps for a 100-PRB type-2 hop before).  `x` = the band with the pilots in place.
template <int G>
__device__ __forceinline__ void cnn_inpaint_runs(float2* x, int n, unsigned mask12, int dpp, int n_iters, const double* rcp, int tid) {
  const unsigned mask24 = mask12 | (mask12 << 12);
  // a run starts after pilot RE r when RE r+1 (of this or the next PRB) is not a pilot
  const unsigned gapmask = mask12 & ~(mask24 >> 1) & 0xFFFu;
  const int nne = __popc(gapmask), n_prbs = n / 12, first = __ffs(mask12) - 1;
  const int total = n_prbs * nne + (first > 0 ? 1 : 0);
  for (int w = tid; w < total; w += NT) {
    float2 v[G];
#pragma unroll
    for (int i = 0; i < G; ++i) v[i] = make_float2(0.f, 0.f);
    int start, g, kind = 0;
    float2 Lv = make_float2(0.f, 0.f), Rv = Lv;
    if (w == n_prbs * nne) {  // the run in front of the first pilot
      start = 0; g = first; kind = 1;
      Rv = x[first];
    } else {
      const int q = w / nne, jj = w - q * nne;
      unsigned mm = gapmask;
      for (int c = 0; c < jj; ++c) mm &= mm - 1u;
      const int r = __ffs(mm) - 1;                       // the jj-th pilot of the PRB that is followed by a run
      const int a = 12 * q + r;
      g = __ffs(mask24 >> (r + 1)) - 1;                  // distance to the next pilot (possibly in the next PRB)
      start = a + 1;
      Lv = x[a];
      if (start + g >= n) { g = n - start; kind = 2; }   // the last PRB's wrap-around run ends at the band edge
      else Rv = x[start + g];
    }
    if (g <= 0) continue;
    cnn_inpaint_run<G>(v, g, Lv, Rv, kind, n_iters, rcp);
#pragma unroll
    for (int i = 0; i < G; ++i)
      if (i < g) x[start + i] = v[i];
  }
}

// Partial-convolution in-painting of one layer over a hop band of n subcarriers (C:473-508, C:276-295), then the two
// low-pass passes with the pilots restored (C:507-508).  Result ends in `dst`; `pong` is a second band-sized buffer.
// `gmax` = the longest run of unknown REs (plan: cnn_gmax).
__device__ __forceinline__ void cnn_inpaint_layer(float2* dst, float2* pong, const float2* Pl, int n, unsigned mask12, int dpp,
                                                  int n_iters, int gmax, const double* rcp, int tid) {
  for (int i = tid; i < n; i += NT) {
    const int q = i / 12, r = i - 12 * q;
    const bool known = (mask12 >> r) & 1u;
    dst[i] = known ? Pl[q * dpp + __popc(mask12 & ((1u << r) - 1u))] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  if (dpp < 12) {  // known_mask.all() skips the in-painting (C:487-488)
    if (gmax <= 4) cnn_inpaint_runs<4>(dst, n, mask12, dpp, n_iters, rcp, tid);
    else cnn_inpaint_runs<11>(dst, n, mask12, dpp, n_iters, rcp, tid);
    __syncthreads();
  }
  // low-pass twice; known pilots are restored unless every RE is a pilot (C:487-488, C:507-508)
  for (int i = tid; i < n; i += NT) {
    const int q = i / 12, r = i - 12 * q;
    pong[i] = (dpp < 12 && ((mask12 >> r) & 1u)) ? dst[i] : lp3x2(dst, i, n);
  }
  __syncthreads();
  for (int i = tid; i < n; i += NT) dst[i] = pong[i];
  __syncthreads();
}

// RC FIR over one layer's pilots, in place: conv([virtual head ; P ; virtual tail], rc, "same") cropped back
// to P (T:649-664), float64 MACs (T:477-490).  PAD = len(rc) / 2 (the 31-tap instantiation also serves shorter
// odd lengths through zero taps).  Waves 0..2: each thread owns CE_CONV_C consecutive outputs and slides a
// fully unrolled window over P (one LDS read per input sample, taps held in registers).  Last wave: fits the
// virtual pilots (two 16-lane groups), then its first 2*PAD lanes compute the outputs whose window reaches
// past a band edge.  One barrier separates all reads of P from the writes.
// CV = consecutive outputs per FIR thread: CE_CONV_C (9) covers the widest bands with the 192 FIR threads; the narrow tiers
// take just enough for their band (ce_conv_c) -- fewer float64 accumulators (4 VGPRs per output: the 128 -> 96 VGPR step
// that lets a fifth workgroup onto a CU) and a shorter serial window.  Each output adds its taps in the same order
// whatever CV is: results are bit-identical.
constexpr int ce_conv_c(int nd, int kpt) { return nd == 0 ? CE_CONV_C : kpt == 1 ? 2 : kpt == 2 ? 3 : kpt == 4 ? 6 : CE_CONV_C; }
// `nl` = 1 or 2 layers in one call (rows `ls` elements apart): with two, FIR threads 0-95 / 96-191 and the two halves of the
// last wave (two DPP rows each) take one layer each -- a narrow band leaves most FIR threads idle and the band-edge fit is
// a fixed 3 us whatever the band, so two layers cost what one does.  Needs n_re <= (NCV / 2) * CV.
template <int PAD, int CV>
__device__ __forceinline__ void smooth_windowed(float2* Pl0, int ls, int nl, int n_re, int n_pils, int pad_rt, const double* rcz,
                                                float2* vpb0, int tid, double vmx, double vin, double vid) {
  constexpr int NCV = NT - 64, NTAP = 2 * PAD + 1;
  // rc_ext[j], j = 0..NTAP-1: the actual taps centred in the PAD-wide template (zeros outside); the host lays the taps
  // out behind CE_CONV_C - 1 zeros (ce_plan.h: rcz)
  const double* rc = rcz + (CE_CONV_C - 1) - (PAD - pad_rt);
  const bool two = nl == 2;
  // FIR threads: layer (tid >= NCV/2) with two layers; last wave: layer (lane >= 32)
  const int sub = tid < NCV ? (two && tid >= NCV / 2 ? 1 : 0) : (two && tid - NCV >= 32 ? 1 : 0);
  float2* Pl = Pl0 + sub * ls;
  float2* vpb = vpb0 + sub * 32;
  const int m0 = (tid - (tid < NCV ? sub * (NCV / 2) : 0)) * CV;
  double ar[CV], ai[CV];      // defined on the FIR threads only (not live across the virtual-pilot branch)
  double er = 0.0, ei = 0.0;  // edge output of this lane (last wave)
  int em = -1;
  if (tid >= NCV) {
    const int q = (tid - NCV) - sub * 32;
    if (q < 32) {
      const int e = q >> 4;
      virtual_pilots(Pl, n_re, n_pils, e != 0, q & 15, vmx, vin, vid, [&](int dist, float2 val) { vpb[e * 16 + dist] = val; });
    }
    // the lanes below read vpb entries other lanes of this wave just wrote: make the writes visible to the wave
    // (and keep the compiler from moving the reads above them); no workgroup barrier is involved
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (q < 2 * PAD) {
      em = q < PAD ? q : n_re - 2 * PAD + q;
#pragma unroll
      for (int j = 0; j < NTAP; ++j) {
        const int idx = em + PAD - j;
        float2 x = make_float2(0.f, 0.f);
        if (idx < 0) {
          if (-1 - idx < n_pils) x = vpb[-1 - idx];
        } else if (idx >= n_re) {
          if (idx - n_re < n_pils) x = vpb[16 + idx - n_re];
        } else {
          x = Pl[idx];
        }
        const double h = rc[j];
        er += h * (double)x.x;
        ei += h * (double)x.y;
      }
    }
  } else {
    // y[m0+o] = sum_j rc[j] x[m0 + o + PAD - j]; input sample w (index m0 - PAD + w) meets tap j = o + 2 PAD - w
#pragma unroll
    for (int o = 0; o < CV; ++o) ar[o] = ai[o] = 0.0;
    double h[PAD + 1];
#pragma unroll
    for (int j = 0; j <= PAD; ++j) {  // symmetric taps: PAD + 1 distinct values, wave-uniform -> scalar registers
      const double t = rc[j];
      h[j] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t)), __builtin_amdgcn_readfirstlane(__double2loint(t)));
    }
#pragma unroll
    for (int w = 0; w < CV + 2 * PAD; ++w) {
      int idx = m0 - PAD + w;
      idx = idx < 0 ? 0 : (idx >= n_re ? n_re - 1 : idx);  // clamped reads only feed outputs the last wave overwrites
      // Order this sample's LDS read after the previous samples' MACs (the address becomes known only here, the
      // accumulators pass through the same statement).  Left alone, the optimizer hoists all CV+2*PAD reads and
      // their float64 conversions (4 VGPRs a sample, 92 in all) above the first MAC.
      if ((w & 1) == 0) {
#pragma unroll
        for (int o = 0; o < CV; ++o) asm volatile("" : "+v"(idx), "+v"(ar[o]), "+v"(ai[o]));
      }
      const float2 x = Pl[idx];
      const double dx = (double)x.x, dy = (double)x.y;
#pragma unroll
      for (int o = 0; o < CV; ++o) {
        const int j = o + 2 * PAD - w;
        if (j >= 0 && j < NTAP) {
          ar[o] += h[j <= PAD ? j : 2 * PAD - j] * dx;
          ai[o] += h[j <= PAD ? j : 2 * PAD - j] * dy;
        }
      }
    }
  }
  __syncthreads();
  if (tid < NCV) {
#pragma unroll
    for (int o = 0; o < CV; ++o) {
      const int m = m0 + o;
      if (m >= PAD && m < n_re - PAD) Pl[m] = make_float2((float)ar[o], (float)ai[o]);
    }
  } else if (em >= 0) {
    Pl[em] = make_float2((float)er, (float)ei);
  }
  __syncthreads();
}

template <int SC_STEP>
__device__ constexpr bool direct_ok() { return SC_STEP % 12 == 0; }  // whole-PRB steps: L = 1, 3

// Grid writer, direct form (L = 1, 3; 14 symbols): each of 252 threads owns ONE (symbol, layer) float4 phase of
// the 7L float4 a subcarrier spans, so its two CFO phasors and hop/layer selection are thread constants and a
// workgroup iteration stores 4032 contiguous bytes.  Its subcarriers advance by whole PRBs, so its RE position
// inside the PRB -- hence its interpolation weight and anchor ordinals (T:325-337) -- is constant too:
// interpolate straight from P in LDS (left + alpha (right - left), also AT pilots, as the reference does), no
// staging buffer, no barrier.
template <int L, int NH, int WRU, int NS2 = 7>
__device__ __forceinline__ void write_grid_direct(const CeDevPlan* __restrict__ plan, const float2* P, const float2* tab,
                                                  const float2* rot_final, float4* out4, int n_re, int n_re_pad, int tid) {
  // NS2 = symbol pairs per subcarrier: 7 (14-symbol slot), or 6 (12 symbols: 6L float4 per subcarrier, 216 active threads -- the same
  // 36 / L subcarriers per workgroup iteration)
  constexpr int ROW4 = NS2 * L, ACTIVE = (NT / (36 * NS2)) * (36 * NS2), SC_STEP = ACTIVE / ROW4, QS = SC_STEP / 12;
  static_assert(SC_STEP % 12 == 0, "direct form needs whole-PRB steps");
  const int ph = tid % ROW4, sc_lane = tid / ROW4, r12 = sc_lane % 12;
  float2 rsel[2];
  const float2* Pe[2];
  float al[2];
  int ro[2], q[2], nprb[2], dro[2];
  bool tail_r[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int rem = 2 * ph + e, sym = rem / L, l = rem - sym * L;
    int h = -1;
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
      if (sym >= plan->hop[hh].sym0 && sym < plan->hop[hh].sym1) h = hh;  // a later hop overwrites (T:872-896)
    rsel[e] = h < 0 ? make_float2(0.f, 0.f) : rot_final[sym];            // rot_final == 1 when no CFO ramp applies
    h = h < 0 ? 0 : h;
    const CeDevHop& hp = plan->hop[h];
    const int c = l >> 1;
    const float2 t = tab[(h * CE_MAX_CDM + c) * 12 + r12];
    al[e] = t.x;
    Pe[e] = P + (h * L + l) * n_re_pad;
    q[e] = sc_lane / 12 - hp.prb_start;
    nprb[e] = hp.n_prbs;
    dro[e] = QS * hp.dpp[c];
    ro[e] = q[e] * hp.dpp[c] + __float_as_int(t.y);
    tail_r[e] = 12 * (hp.n_prbs - 1) + r12 >= hp.last_idx[c];
  }
  // One hop: the iterations in which any lane of the wave is inside the hop's band form one interval [it_lo, it_hi) -- a
  // wave's lanes span at most two PRBs and move up by QS PRBs per iteration.  Outside it the wave stores zeros and does
  // nothing else; inside, the body is branch-free (per-lane selects), so that the unrolled iterations' LDS reads are
  // requested together instead of one exec-masked region after the other.  (Two hops with different bands: the union of
  // the lanes' intervals is nearly everything, and the per-element branches measured faster under load.)  WRU = unrolled
  // iterations of the branch-free body, 0 = per-element branches (profiles/round2_writer_ab.txt).
  int it_lo = 0, it_hi = 0;
  if constexpr (NH == 1 && WRU > 0) {
    // lane-wise: first iteration with q >= 0, one past the last with q < n_prbs (q = q[0] + it * QS); the wave's interval
    // starts with its LAST lane's (largest subcarrier) and ends with its FIRST lane's
    const int q0 = q[0], np = nprb[0];
    const int first = q0 >= 0 ? 0 : (-q0 + QS - 1) / QS;
    const int last = q0 >= np ? 0 : (np - q0 + QS - 1) / QS;
    it_lo = __builtin_amdgcn_readlane(first, 63);
    it_hi = __builtin_amdgcn_readfirstlane(last);
  }
  if (tid < ACTIVE) {
    float4* o = out4 + tid;
    const int n_iter = (plan->n_sc - sc_lane + SC_STEP - 1) / SC_STEP;
    auto body = [&](auto branchy) __attribute__((always_inline)) {
      constexpr bool BR = decltype(branchy)::value;
      float2 y[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e == 1 && L == 1 && NH == 1) {  // same subcarrier, same layer, same hop: reuse the interpolation
          y[1] = y[0];
        } else {
          const bool valid = (unsigned)q[e] < (unsigned)nprb[e];
          int hi = ro[e], lo = ro[e] - 1;
          if (q[e] == nprb[e] - 1 && tail_r[e]) lo = hi = n_re - 1;  // at/after the last pilot: hold (T:316,321)
          lo = lo < 0 ? 0 : lo;                                       // at/before the first pilot: hold (T:315,320)
          if (!valid) lo = hi = 0;
          const float2 u = (CE_ABLATE & 64) ? make_float2(1.f, 2.f) : Pe[e][lo], v = (CE_ABLATE & 64) ? u : Pe[e][hi];  // 64: timing experiment, no LDS reads in the writer
          if constexpr (BR) {
            y[e] = valid ? make_float2(u.x + al[e] * (v.x - u.x), u.y + al[e] * (v.y - u.y)) : make_float2(0.f, 0.f);
          } else {
            const float2 w = make_float2(u.x + al[e] * (v.x - u.x), u.y + al[e] * (v.y - u.y));
            y[e] = make_float2(__builtin_unpredictable(valid) ? w.x : 0.f, __builtin_unpredictable(valid) ? w.y : 0.f);
          }
          q[e] += QS;
          ro[e] += dro[e];
        }
      }
      const float2 ya = cmul(y[0], rsel[0]), yb = cmul(y[1], rsel[1]);
      store_f4(o, make_float4(ya.x, ya.y, yb.x, yb.y));
      o += ACTIVE;
    };
    if constexpr (NH == 1 && WRU > 0) {
      const int lo_it = it_lo < n_iter ? it_lo : n_iter, hi_it = it_hi < n_iter ? (it_hi > lo_it ? it_hi : lo_it) : n_iter;
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int it = 0; it < lo_it; ++it) { store_f4(o, z4); o += ACTIVE; }
#pragma unroll
      for (int e = 0; e < 2; ++e) { q[e] += lo_it * QS; ro[e] += lo_it * dro[e]; }
#pragma unroll WRU
      for (int it = lo_it; it < hi_it; ++it) body(std::false_type{});
      for (int it = hi_it; it < n_iter; ++it) { store_f4(o, z4); o += ACTIVE; }
    } else {
#pragma unroll CE_WR_BR_UNROLL
      for (int it = 0; it < n_iter; ++it) body(std::true_type{});
    }
  }
}

// The same writer for two hops whose fill rectangles SHARE symbols (the reference harness describes both hops of a hopping
// allocation with the slot's whole symbol range, scripts/validation/validate_case4.py:85-103; hop 2 overwrites hop 1 where
// they also share PRBs, T:872-896).  The hop of an element is then "the last hop whose symbol AND PRB range cover it": the
// symbol half is a constant of the phase-owning thread (`cand`: bit h = hop h's symbols cover this thread's symbol), the PRB
// half is one range compare per hop on the PRB index the thread tracks anyway.  Both hops carry the same DM-RS RE mask (the
// host refuses anything else, T:869), so the interpolation weight, the anchor ordinal inside the PRB and the "at / after the
// last pilot of the PRB" test are hop-independent thread constants; only the P row, the PRB origin and the PRB count are
// selected per element.  Branch-free body (selects), so unrolled iterations keep their LDS reads in flight together.
template <int L, int NS2 = 7>
__device__ __forceinline__ void write_grid_direct_ovl(const CeDevPlan* __restrict__ plan, const float2* P, const float2* tab,
                                                      const float2* rot_final, float4* out4, int n_re, int n_re_pad, int tid) {
  constexpr int ROW4 = NS2 * L, ACTIVE = (NT / (36 * NS2)) * (36 * NS2), SC_STEP = ACTIVE / ROW4, QS = SC_STEP / 12;
  static_assert(SC_STEP % 12 == 0, "direct form needs whole-PRB steps");
  const int ph = tid % ROW4, sc_lane = tid / ROW4, r12 = sc_lane % 12;
  const CeDevHop& h0 = plan->hop[0];
  const CeDevHop& h1 = plan->hop[1];
  const int p0a = h0.prb_start, n0 = h0.n_prbs, p1a = h1.prb_start, n1 = h1.n_prbs;
  float2 rsel[2];
  float al[2];
  int ord[2], dppe[2], lrow[2];
  unsigned cand[2];
  bool tail_r[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int rem = 2 * ph + e, sym = rem / L, l = rem - sym * L, c = l >> 1;
    cand[e] = (sym >= h0.sym0 && sym < h0.sym1 ? 1u : 0u) | (sym >= h1.sym0 && sym < h1.sym1 ? 2u : 0u);
    rsel[e] = cand[e] ? rot_final[sym] : make_float2(0.f, 0.f);   // rot_final == 1 when no CFO ramp applies
    const float2 t = tab[c * 12 + r12];                           // hop 0's anchors = hop 1's (same RE mask)
    al[e] = t.x;
    ord[e] = __float_as_int(t.y);
    dppe[e] = h0.dpp[c];
    lrow[e] = l * n_re_pad;
    tail_r[e] = 12 * (n0 - 1) + r12 >= h0.last_idx[c];
  }
  const bool same = L == 1 && cand[0] == cand[1];                 // one layer: both elements are the same subcarrier of the same row
  if (tid < ACTIVE) {
    float4* o = out4 + tid;
    const int n_iter = (plan->n_sc - sc_lane + SC_STEP - 1) / SC_STEP;
    int qabs = sc_lane / 12;
#pragma unroll 2
    for (int it = 0; it < n_iter; ++it) {
      const bool in0 = (unsigned)(qabs - p0a) < (unsigned)n0, in1 = (unsigned)(qabs - p1a) < (unsigned)n1;
      float2 y[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (e == 1 && same) {
          y[1] = y[0];
        } else {
          const bool use1 = in1 && (cand[e] & 2u), use0 = !use1 && in0 && (cand[e] & 1u), valid = use1 || use0;
          const int q = qabs - (use1 ? p1a : p0a), np = use1 ? n1 : n0;
          int hi = q * dppe[e] + ord[e], lo = hi - 1;
          if (q == np - 1 && tail_r[e]) lo = hi = n_re - 1;       // at/after the last pilot: hold (T:316,321)
          lo = lo < 0 ? 0 : lo;                                    // at/before the first pilot: hold (T:315,320)
          if (!valid) lo = hi = 0;
          const float2* Pe = P + (use1 ? L * n_re_pad : 0) + lrow[e];
          const float2 u = Pe[lo], v = Pe[hi];
          const float2 w = make_float2(u.x + al[e] * (v.x - u.x), u.y + al[e] * (v.y - u.y));
          y[e] = make_float2(valid ? w.x : 0.f, valid ? w.y : 0.f);
        }
      }
      const float2 ya = cmul(y[0], rsel[0]), yb = cmul(y[1], rsel[1]);
      store_f4(o, make_float4(ya.x, ya.y, yb.x, yb.y));
      o += ACTIVE;
      qabs += QS;
    }
  }
}

// L layers, NH hops; ND = DM-RS symbols per hop whose pilot REs (and pilots) stay in registers between the CFO,
// LS and residual stages (one layer, n_re <= KPT*NT); ND = 0 re-reads them from global memory
// (L2) in each of the three stages and works for any geometry.
// KPT = pilot REs per thread on the register path: CE_KPT for wide bands, 1 / 2 for bands of <= NT / 2 NT pilots
// (<= 42 / 85 PRB at comb 2), whose kernels then need 40-50 fewer VGPRs and run four workgroups per CU -- narrow
// allocations are latency-bound, so residency is what they are short of.
// FEAT = the smoothing / in-painting code compiled in (CE_FEAT_*).
template <int L, int NH, int ND, int KPT, int FEAT>
__global__ __launch_bounds__(NT, ce_min_waves(NH, ND, KPT, FEAT, L)) void ce_estimate_kernel(const CeDevPlan* __restrict__ plan,
                                                         const uint16_t* __restrict__ re_idx,
                                                         const uint16_t* __restrict__ ta_inv,
                                                         const float2* __restrict__ tw, CeKernelArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool REG = ND > 0;
  constexpr bool TA_LATE = ce_ta_late(L, NH, ND, KPT, FEAT);
  constexpr int TA_ROW = ce_ta_row(TA_LATE);   // complex elements per residue block of the TA transform (ta_at)
  constexpr int NC = (L + 1) / 2;
  const int tid0 = threadIdx.x;
  int tid = tid0;
#if defined(CE_STAMPS)
  const unsigned long long t_entry = wall_clock64();   // first instruction of the workgroup (diagnostic builds)
#endif

  const int n_re = plan->n_re, n_re_pad = plan->n_re_pad;
  const CeLdsLayout lay = ce_lds_layout(NH, L, n_re_pad, plan->scratch_bytes);
  float2* P = reinterpret_cast<float2*>(smem + lay.off_p);              // [NH][L][n_re_pad]
  float2* scratch = reinterpret_cast<float2*>(smem + lay.off_scratch);   // scratch_bytes
  double* red = reinterpret_cast<double*>(smem + lay.off_red);
  float2* rot_final = reinterpret_cast<float2*>(smem + lay.off_rot);     // [16] exp(+j ph) of the final CFO
  float2* rot_tab = rot_final + 16;                                      // per hop: [16] exp(-j ph) at the hop's DM-RS symbols, [16] exp(+j ph)
  float2* tab = reinterpret_cast<float2*>(smem + lay.off_tab);           // [NH][CDM][12] {alpha, bits(r_ord)}
  double* misc = reinterpret_cast<double*>(smem + lay.off_misc);         // [0..1] cfo_hop
  float2* tw256 = reinterpret_cast<float2*>(smem + lay.off_tw);          // [256] W256^j = exp(+j 2 pi j / 256)
  float2* tw16 = tw256 + 256;                                            // [16]  W4096^i
  // LDS copy of the plan: every stage after the first barrier reads its parameters from here (also the zero-padded RC
  // taps, the symbol start times and the interpolation anchors: they are fields of the plan).  Scalar loads from global
  // memory queue behind the chip-wide store stream (microseconds each under load); LDS reads do not.
  //
  // Everything a workgroup fetches before it can start -- this copy, the TA twiddles, the item's pilots -- is REQUESTED
  // here, back to back, and only then waited for: one memory round trip.  (Copy loops that load, wait and store one
  // after the other cost a round trip each; under full load that preamble measured 20 us of a 58 us workgroup.)
  const CeDevPlan* lp = reinterpret_cast<const CeDevPlan*>(smem + lay.off_plan);
  constexpr int PLAN4 = (int)(sizeof(CeDevPlan) / 16), TW4 = (256 + 16) / 2;
  static_assert(sizeof(CeDevPlan) % 16 == 0 && PLAN4 <= NT && TW4 <= NT, "one float4 per thread covers the plan and the twiddles");
  float4 plan_v = make_float4(0.f, 0.f, 0.f, 0.f), tw_v = plan_v;
  if (tid < PLAN4) plan_v = reinterpret_cast<const float4*>(plan)[tid];
  if (tid < TW4) tw_v = reinterpret_cast<const float4*>(tw + CE_TWC_OFF)[tid];   // W256^j (j < 256) then W4096^i (i < 16), contiguous
  const double* rcz = lp->rcz;                                           // zero-padded RC taps
  const double* sst_l = lp->sst;                                         // [14] symbolStartTime
  const double* sst_dm = &lp->sst_dmrs[0][0];                            // [2][14] ... at the hops' DM-RS symbols

  const float beta_f = plan->beta_f;
  const bool cfo_comp = plan->cfo_comp != 0;

  // Register path: received pilot REs and DM-RS symbols of (item, hop), KPT per thread.
  float2 xr[REG ? KPT * ND : 1];
  // PREG: the DM-RS symbols stay in registers next to the received pilots; otherwise (3 symbols x CE_KPT REs, or two
  // hops with more than 8 per thread) the three stages that use them re-read them -- the Rx ports of a slot share
  // them, so they come from L2
  constexpr bool PREG = ce_pilots_in_regs(NH, ND, KPT);
  float2 pr[PREG ? KPT * ND * L : 1];
  // !PREG (two hops, 9-14 pilot REs x symbols per thread): every stage that needs the DM-RS symbols would fetch them again,
  // and under load a fetch queues behind the chip-wide store stream for microseconds.  They are fetched ONCE per hop: the
  // products rx * conj(pilot) that both the CFO and the LS stage start from are kept from the one to the other, and the
  // symbols themselves wait for the residual stage in the LDS scratch (plan: pil_stash; each thread reads back what it wrote).
  constexpr bool YK = REG && !PREG && ND >= 2;
  float2 yk[YK ? KPT * ND * L : 1];
  const int pil_stash_f = YK ? plan->pil_stash : 0;
  const int pil_stash = pil_stash_f & 0xFFFFFF, stash_nd = pil_stash_f >> 24;   // offset in the scratch; symbols parked (the rest is re-read)
  // Two hops on the narrow tiers: the second hop's pilots are requested together with the first's (a few registers
  // more) instead of after the first hop's stages -- one memory round trip less in a latency-bound item.
  constexpr bool PF1 = REG && NH == 2 && KPT * ND <= CE_PF1_LIMIT;
  float2 xr1[PF1 ? KPT * ND : 1];
  float2 pr1[PF1 && PREG ? KPT * ND * L : 1];
  auto load_hop = [&](int64_t it, int h, auto& xr, auto& pr) __attribute__((always_inline)) {
    if constexpr (REG) {
      const CeDevHop& hp = plan->hop[h];
      const int64_t sl = it / a.n_ports;
      const float2* rx = a.rx + sl * a.rs_b + (it - sl * a.n_ports) * a.rs_r;
      const float2* pil = a.pil + sl * a.ps_b;
      // every plan field the loads need, fetched once up front (one scalar-load round trip for the whole hop)
      PilotMap pm = pilot_map(hp, 0);
      int dsym[ND], psym0 = hp.pil_sym0;
#pragma unroll
      for (int s = 0; s < ND; ++s) dsym[s] = hp.dmrs_sym[s];
      // one point where all of them must be present: the compiler then requests them together and waits once (left to
      // itself it sinks each scalar load to its use, behind a branch, and pays a round trip per field)
      asm volatile("" : "+s"(pm.contig), "+s"(pm.dpp), "+s"(pm.prb_start), "+s"(pm.re_off), "+s"(pm.magic), "+s"(pm.pos), "+s"(psym0));
#pragma unroll
      for (int s = 0; s < ND; ++s) asm volatile("" : "+s"(dsym[s]));
      const float2* rx_s[ND];
      const float2* pil_s[ND];
#pragma unroll
      for (int s = 0; s < ND; ++s) {
        rx_s[s] = rx + dsym[s] * a.rs_sym;
        pil_s[s] = pil + (psym0 + s) * a.ps_sym;
      }
      // Subcarrier offsets of this thread's pilot REs first -- for a scattered PRB mask that is a batch of table reads
      // with ONE wait, under a branch that is uniform for the launch (left inside the per-RE code the lookup is
      // speculated and its wait serialises the pilot loads of consecutive REs) -- then all pilot loads back to back and
      // unconditional: a thread past the band's end reads pilot 0 again and clears the value afterwards (`finish_hop`),
      // because an if / else around each load makes the compiler wait for the load before the else side's zero fill.
      unsigned xoff[KPT];
      if (pm.contig) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
          const int k = tid + i * NT, kk = k < n_re ? k : 0;
          const int q = pm.dpp == 1 ? kk : (int)__umulhi((unsigned)kk, pm.magic);
          const int j = kk - q * pm.dpp;
          xoff[i] = (unsigned)(12 * (pm.prb_start + q) + (int)((pm.pos >> (4 * j)) & 15u)) * (unsigned)a.rs_sc;
        }
      } else {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
          const int k = tid + i * NT;
          xoff[i] = (unsigned)re_idx[pm.re_off + (k < n_re ? k : 0)] * (unsigned)a.rs_sc;
        }
      }
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const int k = tid + i * NT;
        if (CE_ABLATE & 32) {  // timing experiment: no global reads at all
#pragma unroll
          for (int s = 0; s < ND; ++s) {
            xr[i * ND + s] = make_float2(1.f + k * 1e-3f, 0.5f);
#pragma unroll
            for (int l = 0; l < L; ++l)
              if constexpr (PREG) pr[(i * ND + s) * L + l] = make_float2(0.7071f, -0.7071f);
          }
        } else {
          // uniform 64-bit base (SGPR pair) + 32-bit per-thread offset: one address VGPR per pilot RE
          // instead of two per load (the host checks the offsets fit 32 bits)
          const unsigned xo = xoff[i];
          const unsigned po = (unsigned)(k < n_re ? k : 0) * (unsigned)a.ps_re;
#pragma unroll
          for (int s = 0; s < ND; ++s) {
            xr[i * ND + s] = rx_s[s][xo];
#pragma unroll
            for (int l = 0; l < L; ++l)
              if constexpr (PREG) pr[(i * ND + s) * L + l] = (pil_s[s] + l * a.ps_l)[po];
          }
        }
      }
    }
  };
  // second half of load_hop, placed where the data is first needed: pilot REs past the end of the band count as zero
  auto finish_hop = [&]() __attribute__((always_inline)) {
    if constexpr (REG) {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        if (tid + i * NT >= n_re) {
#pragma unroll
          for (int s = 0; s < ND; ++s) {
            xr[i * ND + s] = make_float2(0.f, 0.f);
#pragma unroll
            for (int l = 0; l < L; ++l)
              if constexpr (PREG) pr[(i * ND + s) * L + l] = make_float2(0.f, 0.f);
          }
        }
      }
    }
  };
  if (blockIdx.x >= a.n_local) return;  // the grid is exactly n_local workgroups
#if CE_PRIO == 2
  __builtin_amdgcn_s_setprio(3);   // experiment: the estimation stages win, so that a workgroup reaches its stores sooner
#endif
  const int64_t item = a.item0 + item_of(blockIdx.x, a.n_ports, a.n_local);
  load_hop(item, 0, xr, pr);
  if constexpr (PF1) load_hop(item, 1, xr1, pr1);  // hop 2's pilots ride the same round trip (narrow tiers: few registers)
  STAMP_STARTUP(14);  // plan fields arrived in scalar registers, every pilot load issued
  // the copies' LDS stores come after the pilot requests, so waiting for their data does not delay those
  if (tid < PLAN4) reinterpret_cast<float4*>(smem + lay.off_plan)[tid] = plan_v;
  if (tid < TW4) reinterpret_cast<float4*>(tw256)[tid] = tw_v;
  STAMP_STARTUP(15);  // plan and twiddle copies arrived
  const int64_t slot = item / a.n_ports;
  const int port = (int)(item - slot * a.n_ports);
  const float2* rx = a.rx + slot * a.rs_b + port * a.rs_r;
  const float2* pil = a.pil + slot * a.ps_b;
  int pil_sym0_h = 0;  // first column of the current hop along the pilots' symbol axis (set once per hop)
  auto pilot_of = [&](const CeDevHop&, int i, int s, int l) __attribute__((always_inline)) -> float2 {  // DM-RS symbol of pilot RE tid + i*NT (register path)
    if constexpr (PREG) {
      return pr[(i * ND + s) * L + l];
    } else {
      const int k = tid + i * NT;
      if (k >= n_re) return make_float2(0.f, 0.f);
      const float2* pil_sl = pil + (pil_sym0_h + s) * a.ps_sym + l * a.ps_l;
      return pil_sl[(unsigned)k * (unsigned)a.ps_re];
    }
  };
  // Diagnostics (ce_estimate_batch_stages; a.stage_p is null in every ordinary launch): the pilot-RE channel estimate of
  // hop h after stage `st` (0: LS + DM-RS average + de-spread, S5/S6; 1: after frequency smoothing, S7) as
  // [item][stage][hop][layer][n_re]; a.stage_s [item][hop][2] = the hop's CFO (normalised to the SCS) and its TA bin.
  auto dump_stage = [&](int st, int h) __attribute__((always_inline)) {
    if (a.stage_p) {
      float2* sp = a.stage_p + ((item * 2 + st) * NH + h) * (int64_t)(L * n_re);
      const float2* Ph = P + h * L * n_re_pad;
      for (int i = tid; i < L * n_re; i += NT) {
        const int l = i / n_re;
        sp[i] = Ph[l * n_re_pad + (i - l * n_re)];
      }
    }
  };

  // ------------------------------------------------------------ time alignment of one hop (S8)
  double tot_ta = 0.0;
  // `npar` == 2 (one layer, two hops, plan: ta_lp == 2): hops h0 and h0 + 1 side by side -- threads 0-127 run hop h0's
  // radix-16 passes, threads 128-255 hop h0 + 1's, into separate residue blocks -- then each hop's bins and arg-max in turn.
  auto time_alignment = [&](int h0, int npar) {
    STAMP(9);
    // x[n] = P[k] at the pilot subcarriers of the LAST CDM group (for every layer, T:672-675), else 0;
    // X[k] = sum_n x[n] W^(nk), W = exp(+j 2 pi / 4096), wanted only for k in [0,144) U [3952,4096).
    // n = r + 16 n':  X[k] = sum_r W^(rk) Y_r[k mod 256],  Y_r = 256-point IDFT of x[r + 16 n'] done as
    // two radix-16 passes in LDS; residues r without pilots (half of them for a comb-2 DM-RS) are skipped.
    if (!(CE_ABLATE & 1)) {
      const int b0 = tid, b1 = tid + NT;  // bins tid and tid + NT of the 288 examined (b < 144: delay side, else advance side)
      constexpr int NB = 2 * CE_TA_HALF;
      // Two layers at a time (plan: ta_lp == 2; 2-4 layers, at most 8 pilot-carrying residues, LDS to spare without
      // costing a workgroup per CU): threads 0-127 transform layer l0, threads 128-255 layer l0 + 1, each into its own
      // residue blocks; the bin sums then add the layers' powers in layer order, as the one-at-a-time form does.
      constexpr bool HPAR_OK = NH == 2 && L == 1 && TA_LATE;  // elsewhere the hop of the passes stays wave-uniform (scalar registers)
      const bool hpar = HPAR_OK && npar == 2;
      // (plan: ta_over_p) no room for a second set of residue blocks next to P: the last hop still runs two layers at a time, its
      // second set laid over the FIRST hop's rows of P -- dead once that hop's transforms are done (this stage runs after the writer)
      constexpr bool OVERP_OK = TA_LATE && NH == 2 && L >= 2;
      const bool over_p = OVERP_OK && lp->ta_over_p && h0 == NH - 1;
      const int lpn = (L >= 2) ? (over_p ? 2 : lp->ta_lp) : 1;
      const int sub = (lpn == 2 || hpar) ? (tid >> 7) : 0;
      const int ri = (lpn == 2 || hpar) ? ((tid >> 4) & 7) : (tid >> 4), a4 = tid & 15;
      float2* const set2 = over_p ? P : scratch + 8 * TA_ROW;   // the second set of residue blocks
      float2* scr = sub ? set2 : scratch;
      // the hop whose passes this thread runs
      const CeDevHop& lh = lp->hop[h0 + (hpar ? sub : 0)];
      const float2* Ph = P + (h0 + (hpar ? sub : 0)) * L * n_re_pad;
      const int nres = lh.ta_nres;
      const uint16_t* inv = ta_inv + lh.ta_inv_off;
      const unsigned long long res_packed = lh.ta_res_packed;
      auto bin_power = [&](int k, const float2* blocks, int nr, unsigned long long resp) -> float {
        const int q = k & 255, off = ta_at<TA_LATE>(q & 15, q >> 4);
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll 4
        for (int i = 0; i < nr; ++i) {  // (same order of additions whatever the unrolling: bit-identical)
          const int m = ((int)((resp >> (4 * i)) & 15u) * k) & (CE_FFT_SIZE - 1);  // W4096^(r k)
          acc = cadd(acc, cmul(cmul(tw256[m >> 4], tw16[m & 15]), blocks[i * TA_ROW + off]));
        }
        return acc.x * acc.x + acc.y * acc.y;
      };
      // subcarrier n -> ordinal of the pilot it carries (last CDM group), or -1
      const int contig = lh.contig, dpp_last = lh.dpp[NC - 1], prb0 = lh.prb_start, nprb = lh.n_prbs;
      const unsigned long long ord_packed = lh.ord_packed;
      auto pilot_at = [&](int n) -> int {
        if (contig) {
          const int q = n / 12, rem = n - 12 * q, pq = q - prb0;
          const int o = (int)((ord_packed >> (4 * rem)) & 15u);
          return (pq >= 0 && pq < nprb && o != 15) ? pq * dpp_last + o : -1;
        }
        const unsigned idx = inv[n];
        return idx == 0xFFFFu ? -1 : (int)idx;
      };
      // arg-max: T:683-696 takes the first maximum of the delay side (bins 0..143), the first maximum of the advance side
      // (bins 3952..4095) and prefers the delay side when the two are equal.  One key orders all 288 bins that way --
      // (power bits, delay side before advance side, lower index first) -- so one wave reduction per hop finds the winner.
      // NHP hops at once (their keys reduced side by side: one barrier, and the independent DPP chains overlap).
      // Adds the hops' seconds to tot_ta.
      auto arg_max = [&](auto nhp, int h, const float (&pw0)[2], const float (&pw1)[2]) {
        constexpr int NHP = decltype(nhp)::value;
        unsigned long long key[NHP];
#pragma unroll
        for (int j = 0; j < NHP; ++j) {
          key[j] = 0ull;
          auto offer = [&](int b, float pw) {
            if (b < NB) {
              const unsigned low = b < CE_TA_HALF ? 0xFFFFFFFFu - (unsigned)b : 0x7FFFFFFFu - (unsigned)(b - CE_TA_HALF);
              const unsigned long long k = ((unsigned long long)__float_as_uint(pw) << 32) | low;
              key[j] = k > key[j] ? k : key[j];
            }
          };
          offer(b0, pw0[j]);
          offer(b1, pw1[j]);
        }
#pragma unroll
        for (int j = 0; j < NHP; ++j) key[j] = wave_max_u64(key[j]);
        unsigned long long* ared = reinterpret_cast<unsigned long long*>(misc + 32);  // own slot: [hop][wave]
        if ((tid & 63) == 0) {
#pragma unroll
          for (int j = 0; j < NHP; ++j) ared[j * NW + (tid >> 6)] = key[j];
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
          for (int j = 0; j < NHP; ++j) {
            unsigned long long m = 0ull;
#pragma unroll
            for (int w = 0; w < NW; ++w) m = ared[j * NW + w] > m ? ared[j * NW + w] : m;
            const unsigned low = (unsigned)(m & 0xFFFFFFFFull);
            const int i_max = (low & 0x80000000u) ? (int)(0xFFFFFFFFu - low) : -(CE_TA_HALF - (int)(0x7FFFFFFFu - low));
            tot_ta += (double)i_max / (double)CE_FFT_SIZE / lp->scs;  // T:698, the reference's two float64 divisions
            if (a.stage_s) a.stage_s[(item * NH + h + j) * 2 + 1] = (double)i_max;
          }
        }
      };
      float pw0 = 0.f, pw1 = 0.f;
#pragma unroll 1
      for (int l0 = 0; l0 < L; l0 += lpn) {
        const int l = l0 + (hpar ? 0 : sub);
        const float2* Pl = Ph + l * n_re_pad;
        const bool unit = ri < nres && l < L;
        if (unit) {  // pass 1: DFT16 over b of x[r + 16 a + 256 b], times W256^(a c)
          const int r = (int)((res_packed >> (4 * ri)) & 15u);
          float2* blk = scr + ri * TA_ROW;
          // (not in the 2-4-layer x 2-hop kernels: the stage sits inside their hop loop and a second form of it costs 10 more spilled registers)
          const unsigned win = (L >= 2 && NH == 2) ? 0u : lh.ta_win;
          if (win) {
            // narrow band, moved down by `shift` subcarriers (plan): only b = 0 (and 1) carry pilots, so the 16-point
            // transform over b is x0 + x1 W16^c -- one or two LDS reads and 16 multiplies instead of 16 reads + a DFT16
            const int n0 = (int)(win & 0xFFFFu) + r + 16 * a4;
            const int i0 = pilot_at(n0), i1 = (win >> 16) == 2u ? pilot_at(n0 + 256) : -1;
            const float2 x0 = i0 >= 0 ? Pl[i0] : make_float2(0.f, 0.f), x1 = i1 >= 0 ? Pl[i1] : make_float2(0.f, 0.f);
            blk[ta_at<TA_LATE>(0, a4)] = cadd(x0, x1);
#pragma unroll
            for (int c = 1; c < 16; ++c) blk[ta_at<TA_LATE>(c, a4)] = cmul(cadd(x0, cmul(x1, tw256[16 * c])), tw256[a4 * c]);
          } else {
            float2 v[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) {
              const int idx = pilot_at(r + 16 * a4 + 256 * b);
              v[b] = idx >= 0 ? Pl[idx] : make_float2(0.f, 0.f);
            }
            idft16(v);
            blk[ta_at<TA_LATE>(0, a4)] = v[0];
#pragma unroll
            for (int c = 1; c < 16; ++c) blk[ta_at<TA_LATE>(c, a4)] = cmul(v[c], tw256[a4 * c]);
          }
        }
        __syncthreads();
        if (unit) {  // pass 2 (in place): DFT16 over a for fixed c = a4 -> Y_r[c + 16 d] at element (c, d)
          float2* blk = scr + ri * TA_ROW;
          float2 v[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = blk[ta_at<TA_LATE>(a4, i)];
          idft16(v);
#pragma unroll
          for (int i = 0; i < 16; ++i) blk[ta_at<TA_LATE>(a4, i)] = v[i];
        }
        __syncthreads();
        STAMP_TA(14);
        if (!hpar) {
          const CeDevHop& bh = lp->hop[h0];
#pragma unroll 1
          for (int s2 = 0; s2 < lpn && l0 + s2 < L; ++s2) {
            const float2* blocks = s2 ? set2 : scratch;
            if (b0 < NB) pw0 += bin_power(b0 < CE_TA_HALF ? b0 : CE_FFT_SIZE - NB + b0, blocks, bh.ta_nres, bh.ta_res_packed);
            if (b1 < NB) pw1 += bin_power(b1 < CE_TA_HALF ? b1 : CE_FFT_SIZE - NB + b1, blocks, bh.ta_nres, bh.ta_res_packed);
          }
          __syncthreads();
        }
        STAMP_TA(15);
      }
      if (!hpar) {
        const float p0[2] = {pw0, 0.f}, p1[2] = {pw1, 0.f};
        arg_max(std::integral_constant<int, 1>{}, h0, p0, p1);
      } else {
        // one layer: each hop's bins straight from its blocks, then both hops' arg-max in one round
        float p0[2], p1[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const CeDevHop& bh = lp->hop[h0 + s2];
          const float2* blocks = scratch + s2 * (8 * TA_ROW);
          p0[s2] = b0 < NB ? bin_power(b0 < CE_TA_HALF ? b0 : CE_FFT_SIZE - NB + b0, blocks, bh.ta_nres, bh.ta_res_packed) : 0.f;
          p1[s2] = b1 < NB ? bin_power(b1 < CE_TA_HALF ? b1 : CE_FFT_SIZE - NB + b1, blocks, bh.ta_nres, bh.ta_res_packed) : 0.f;
        }
        arg_max(std::integral_constant<int, 2>{}, h0, p0, p1);
      }
    }
  };

  double tot_epre = 0.0, tot_noise = 0.0, tot_rsrp = 0.0;
  STAMP(0);
  STAMP_HWID(12);
#if defined(CE_STAMPS)
  if (threadIdx.x == 0 && a.stamps) a.stamps[item * 16 + 13] = t_entry;
#endif

  // Narrow two-hop tiers (both hops' pilots in registers from the start): the hop loop runs twice -- pass 0: CFO, LS for
  // hop 1 then hop 2; ONE smoothing call for both hops (their rows of P are adjacent: the two-layer form of the windowed
  // FIR, whose band-edge stage costs the same for one row or two); pass 1: the residuals, hop 2 first (its pilots are the
  // ones in `xr` after pass 0), then hop 1.  Same arithmetic per hop, two of the item's longest serial stages side by side.
  // (Both hops' pilots then stay live across the FIR: beyond KPT x ND = 2 that costs the fourth workgroup per CU.)
  constexpr bool FUSED2 = PF1 && KPT * ND <= 2;
  constexpr int NPASS = FUSED2 ? 2 : 1;
  constexpr int LSM = FUSED2 ? 2 : L;   // rows of P one smoothing invocation covers
  float epre_h0 = 0.f, epre_h1 = 0.f;   // FUSED2: each hop's EPRE partial sum, carried from pass 0 to pass 1
  auto swap_hops = [&]() __attribute__((always_inline)) {
    if constexpr (PF1) {
#pragma unroll
      for (int i = 0; i < (REG ? KPT * ND : 1); ++i) { const float2 t = xr[i]; xr[i] = xr1[i]; xr1[i] = t; }
#pragma unroll
      for (int i = 0; i < (PREG ? KPT * ND * L : 1); ++i) { const float2 t = pr[i]; pr[i] = pr1[i]; pr1[i] = t; }
    }
  };
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll 1
  for (int hi = 0; hi < NH; ++hi) {
    const int h = (FUSED2 && pass == 1) ? NH - 1 - hi : hi;
    const bool front = !FUSED2 || pass == 0, back = !FUSED2 || pass == 1;
    const CeDevHop& hp = plan->hop[h];  // global copy: load + CFO stages (before the first barrier)
    const CeDevHop& lh = lp->hop[h];    // LDS copy: everything after
    float2* Ph = P + h * L * n_re_pad;
    float2* rot_neg = rot_tab + h * 32;
    float2* rot_pos = rot_neg + 16;
    const int n_dmrs = REG ? ND : hp.n_dmrs;
    const float n_dmrs_f = (float)n_dmrs;
    const bool has_cfo = REG ? (ND >= 2) : (hp.has_cfo != 0);
    float epre_part = 0.f;
    if (front) {
    if (h > 0) {
      if constexpr (PF1) {
        swap_hops();  // hop 2's pilots were requested together with hop 1's
      } else {
        load_hop(item, h, xr, pr);
      }
    }
    finish_hop();
    pil_sym0_h = hp.pil_sym0;
    if (NH > 1) {
      // Everything a stage derives from the thread index and the plan is loop-invariant; left alone, the compiler
      // hoists all of it out of the hop loop and keeps it live across every stage.  An opaque copy of the thread index
      // (and a memory barrier for the plan reads) keeps each stage's temporaries local to the stage.
      tid = tid0;
      asm volatile("" : "+v"(tid) : : "memory");
    }

    STAMP(1);
    // ------------------------------------------------------------ CFO of the hop (S4)
    double cfo_hop = 0.0;
    if constexpr (YK) {
      float2 pt[KPT * ND * L];
#pragma unroll
      for (int i = 0; i < KPT; ++i)
#pragma unroll
        for (int s = 0; s < ND; ++s)
#pragma unroll
          for (int l = 0; l < L; ++l) pt[(i * ND + s) * L + l] = pilot_of(hp, i, s, l);
      if (pil_stash) {
        float2* st = scratch + pil_stash;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
          const int k = tid + i * NT;
          if (k < n_re) {
#pragma unroll
            for (int s = 0; s < ND; ++s)
              if (s < stash_nd) {
#pragma unroll
                for (int l = 0; l < L; ++l) st[(s * L + l) * n_re_pad + k] = pt[(i * ND + s) * L + l];
              }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < KPT; ++i)
#pragma unroll
        for (int s = 0; s < ND; ++s)
#pragma unroll
          for (int l = 0; l < L; ++l) yk[(i * ND + s) * L + l] = cmul_conj(xr[i * ND + s], pt[(i * ND + s) * L + l]);
    }
    if (has_cfo && !(CE_ABLATE & 16)) {
      double acc[2 * L];
#pragma unroll
      for (int i = 0; i < 2 * L; ++i) acc[i] = 0.0;
      if constexpr (REG && ND >= 2) {
        float part[2 * L];
#pragma unroll
        for (int i = 0; i < 2 * L; ++i) part[i] = 0.f;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
#pragma unroll
          for (int l = 0; l < L; ++l) {
            const float2 r0 = YK ? yk[(i * ND) * L + l] : cmul_conj(xr[i * ND], pilot_of(hp, i, 0, l));
            const float2 r1 = YK ? yk[(i * ND + 1) * L + l] : cmul_conj(xr[i * ND + 1], pilot_of(hp, i, 1, l));
            const float2 in = cmul_conj(r1, r0);  // conj(r0) * r1
            part[2 * l] += in.x;
            part[2 * l + 1] += in.y;
          }
          // keep the unrolled iterations sequential: otherwise all 14 products are formed first and the
          // stage peaks at 2x the registers of the pilots it reads
          if (i + 1 < KPT) asm volatile("" : "+v"(part[0]), "+v"(part[1]), "+v"(xr[(i + 1) * ND].x));
        }
#pragma unroll
        for (int i = 0; i < 2 * L; ++i) acc[i] = (double)part[i];
      } else if constexpr (!REG) {
        const int64_t o0 = hp.dmrs_sym[0] * a.rs_sym, o1 = hp.dmrs_sym[1] * a.rs_sym;
        const int64_t p0 = hp.pil_sym0 * a.ps_sym, p1 = (hp.pil_sym0 + 1) * a.ps_sym;
        PilotMap pmc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) pmc[c] = pilot_map(hp, c);
        // KU pilot REs per thread and iteration, every load of theirs (2 received REs + 2 x layers DM-RS values per CDM
        // group) requested before the first is used: the stage is a chain of L2 round trips, one per iteration
        constexpr int KU = L <= 2 ? CE_GEN_KU : 1;
        for (int k0 = tid; k0 < n_re; k0 += KU * NT) {
          float2 x0[KU][NC], x1[KU][NC], q0[KU][NC][2], q1[KU][NC][2];
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            const int k = (u == 0 || k0 + u * NT < n_re) ? k0 + u * NT : k0;  // past the band: a valid address, the values are dropped
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int64_t sc = pilot_sc(pmc[c], re_idx, k);
              x0[u][c] = rx[sc * a.rs_sc + o0];
              x1[u][c] = rx[sc * a.rs_sc + o1];
#pragma unroll
              for (int j = 0; j < 2; ++j)
                if (2 * c + j < L) {
                  q0[u][c][j] = pil[k * a.ps_re + p0 + (2 * c + j) * a.ps_l];
                  q1[u][c][j] = pil[k * a.ps_re + p1 + (2 * c + j) * a.ps_l];
                }
            }
          }
#pragma unroll
          for (int u = 0; u < KU; ++u)
            if (u == 0 || k0 + u * NT < n_re) {
#pragma unroll
              for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                  if (2 * c + j < L) {
                    const int l = 2 * c + j;
                    const float2 r0 = cmul_conj(x0[u][c], q0[u][c][j]), r1 = cmul_conj(x1[u][c], q1[u][c][j]);
                    const float2 in = cmul_conj(r1, r0);
                    acc[2 * l] += (double)in.x;
                    acc[2 * l + 1] += (double)in.y;
                  }
            }
        }
      }
      block_sum<2 * L>(acc, red);
      if (tid < 16) {  // every thread holds the totals: lanes 0..15 turn them into the hop's CFO and phasors
        double ang = 0.0;
#pragma unroll
        for (int l = 0; l + 1 < L; l += 2)  // CDM pairs are summed before the angle (T:410-413)
          ang += (double)atan2f((float)(acc[2 * l + 1] + acc[2 * l + 3]), (float)(acc[2 * l] + acc[2 * l + 2]));
        if (L & 1) ang += (double)atan2f((float)acc[2 * L - 1], (float)acc[2 * L - 2]);
        cfo_hop = ang * hp.inv_two_pi_nsamples * plan->inv_denom_cdm;  // T:426, reciprocals from the plan (<= 1 ulp)
        if (tid == 0) misc[h] = cfo_hop;
        if (tid == 0 && a.stage_s) a.stage_s[(item * NH + h) * 2] = cfo_hop;
      }
    }
    STAMP(2);
    if constexpr (REG) {
#pragma unroll
      for (int i = 0; i < KPT * ND; ++i) pin(xr[i]);
    }
    if (tid < 16) {  // de-rotation / re-rotation phasors of the hop's DM-RS symbols (T:439-447, T:713-718)
      float2 rn = make_float2(1.f, 0.f), rp = make_float2(1.f, 0.f);
      if (cfo_comp && has_cfo && tid < n_dmrs && !(CE_ABLATE & 16)) {
        float sn, cs;
        sincospi_f64arg(2.0 * sst_dm[h * CE_MAX_SYMBOLS + tid] * cfo_hop, &sn, &cs);
        rn = make_float2(cs, -sn);
        rp = make_float2(cs, sn);
      }
      rot_neg[tid] = rn;
      rot_pos[tid] = rp;
    }
    __syncthreads();

    STAMP(3);
    // ------------------------------------------------------------ EPRE, LS, DM-RS average (S2, S3, S5)
    if constexpr (REG) {
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const int k = tid + i * NT;
        float2 acc[L];
#pragma unroll
        for (int l = 0; l < L; ++l) acc[l] = make_float2(0.f, 0.f);
#pragma unroll
        for (int s = 0; s < ND; ++s) {
          const float2 x = xr[i * ND + s];
          epre_part += x.x * x.x + x.y * x.y;
          const float2 rn = rot_neg[s];
#pragma unroll
          for (int l = 0; l < L; ++l) acc[l] = cadd(acc[l], cmul(YK ? yk[(i * ND + s) * L + l] : cmul_conj(x, pilot_of(lh, i, s, l)), rn));
        }
        if (k < n_re) {
#pragma unroll
          for (int l = 0; l < L; ++l)
            Ph[l * n_re_pad + k] = make_float2(acc[l].x / beta_f / n_dmrs_f, acc[l].y / beta_f / n_dmrs_f);
        }
      }
    } else {
      PilotMap pmc[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) pmc[c] = pilot_map(lh, c);
      // The hop's DM-RS symbol count as a compile-time constant (1..4), KU pilot REs per thread and iteration: all their
      // loads -- n_dmrs received REs and n_dmrs x layers DM-RS values per CDM group -- are requested before the first is
      // used.  With the symbol loop's trip count a run-time value the stage was one L2 round trip per (RE, CDM group, symbol).
      // Same operations in the same order either way.
      auto ls_pass = [&](auto ndc) __attribute__((always_inline)) {
        constexpr int NDc = decltype(ndc)::value;
        constexpr int KU = NDc * L <= 4 ? CE_GEN_KU : 1;   // (4 layers x 2 symbols x 2 REs: 166 VGPRs and 1 % slower than one RE per iteration)
        int64_t osym[NDc], psym[NDc];
#pragma unroll
        for (int s = 0; s < NDc; ++s) {
          osym[s] = lh.dmrs_sym[s] * a.rs_sym;
          psym[s] = (lh.pil_sym0 + s) * a.ps_sym;
        }
        for (int k0 = tid; k0 < n_re; k0 += KU * NT) {
          float2 x[KU][NC][NDc], q[KU][NC][NDc][2];
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            const int k = (u == 0 || k0 + u * NT < n_re) ? k0 + u * NT : k0;  // past the band: a valid address, the values are dropped
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int64_t sc = pilot_sc(pmc[c], re_idx, k);
#pragma unroll
              for (int s = 0; s < NDc; ++s) {
                x[u][c][s] = rx[sc * a.rs_sc + osym[s]];
                q[u][c][s][0] = pil[k * a.ps_re + psym[s] + (2 * c) * a.ps_l];
                if (2 * c + 1 < L) q[u][c][s][1] = pil[k * a.ps_re + psym[s] + (2 * c + 1) * a.ps_l];
              }
            }
          }
#pragma unroll
          for (int u = 0; u < KU; ++u) {
            const int k = k0 + u * NT;
            if (u == 0 || k < n_re) {
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
#pragma unroll
                for (int s = 0; s < NDc; ++s) {
                  const float2 xv = x[u][c][s];
                  epre_part += xv.x * xv.x + xv.y * xv.y;
                  const float2 rn = rot_neg[s];
                  acc0 = cadd(acc0, cmul(cmul_conj(xv, q[u][c][s][0]), rn));
                  if (2 * c + 1 < L) acc1 = cadd(acc1, cmul(cmul_conj(xv, q[u][c][s][1]), rn));
                }
                Ph[(2 * c) * n_re_pad + k] = make_float2(acc0.x / beta_f / n_dmrs_f, acc0.y / beta_f / n_dmrs_f);
                if (2 * c + 1 < L)
                  Ph[(2 * c + 1) * n_re_pad + k] = make_float2(acc1.x / beta_f / n_dmrs_f, acc1.y / beta_f / n_dmrs_f);
              }
            }
          }
        }
      };
      switch (CE_GEN_NDC ? n_dmrs : 0) {
        case 1: ls_pass(std::integral_constant<int, 1>{}); break;
        case 2: ls_pass(std::integral_constant<int, 2>{}); break;
        case 3: ls_pass(std::integral_constant<int, 3>{}); break;
        case 4: ls_pass(std::integral_constant<int, 4>{}); break;
        default:   // any other count (the reference takes any DMRSsymbols mask, T:564-568): one symbol after the other
          for (int k = tid; k < n_re; k += NT) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const int64_t sc = pilot_sc(pmc[c], re_idx, k);
              float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
              for (int s = 0; s < n_dmrs; ++s) {
                const float2 x = rx[sc * a.rs_sc + lh.dmrs_sym[s] * a.rs_sym];
                epre_part += x.x * x.x + x.y * x.y;
                const float2 rn = rot_neg[s];
                const int64_t pb = k * a.ps_re + (lh.pil_sym0 + s) * a.ps_sym;
                acc0 = cadd(acc0, cmul(cmul_conj(x, pil[pb + (2 * c) * a.ps_l]), rn));
                if (2 * c + 1 < L) acc1 = cadd(acc1, cmul(cmul_conj(x, pil[pb + (2 * c + 1) * a.ps_l]), rn));
              }
              Ph[(2 * c) * n_re_pad + k] = make_float2(acc0.x / beta_f / n_dmrs_f, acc0.y / beta_f / n_dmrs_f);
              if (2 * c + 1 < L)
                Ph[(2 * c + 1) * n_re_pad + k] = make_float2(acc1.x / beta_f / n_dmrs_f, acc1.y / beta_f / n_dmrs_f);
            }
          }
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
    dump_stage(0, h);
    if (FUSED2) { if (h == 0) epre_h0 = epre_part; else epre_h1 = epre_part; }
    }  // front
    if (FUSED2 && back) {
      if (h == 0) swap_hops();  // hop 1's pilots back into `xr` (pass 0 left hop 2's there)
      pil_sym0_h = lh.pil_sym0;
      epre_part = h == 0 ? epre_h0 : epre_h1;
      tid = tid0;
      asm volatile("" : "+v"(tid) : : "memory");
    }

    STAMP(4);
    // ------------------------------------------------------------ frequency smoothing (S7)
    // (FUSED2: once, after the second hop's LS, over both hops' rows)
    float2* Psm = FUSED2 ? P : Ph;
    if ((CE_ABLATE & 2) || (FUSED2 && !(pass == 0 && hi == NH - 1))) {
    } else if (lp->smoothing == CE_SMOOTH_MEAN) {
      double m[2 * LSM];
#pragma unroll
      for (int i = 0; i < 2 * LSM; ++i) m[i] = 0.0;
      for (int k = tid; k < n_re; k += NT) {
#pragma unroll
        for (int l = 0; l < LSM; ++l) {
          const float2 v = Psm[l * n_re_pad + k];
          m[2 * l] += (double)v.x;
          m[2 * l + 1] += (double)v.y;
        }
      }
      block_sum<2 * LSM>(m, red);
      for (int k = tid; k < n_re; k += NT) {
#pragma unroll
        for (int l = 0; l < LSM; ++l)
          Psm[l * n_re_pad + k] = make_float2((float)(m[2 * l] / (double)n_re), (float)(m[2 * l + 1] / (double)n_re));
      }
      __syncthreads();
    } else if ((FEAT & CE_FEAT_EXT) && lp->smoothing == CE_SMOOTH_MMSE) {
      // EXTENSION (not in the reference): block LMMSE smoothing.  Y = W X for every block of 32 pilots of the layer
      // at once: X^T (blocks as columns) and W^T are staged in the scratch, the complex product is four real
      // v_mfma_f32_16x16x4_f32 chains per 16x16 output tile (A = W: lane -> [m = lane&15][k = lane>>4],
      // B = X: [k = lane>>4][n = lane&15], C/D: col = lane&15, row = 4*(lane>>4) + reg).
      constexpr int MB = CE_MMSE_BLOCK;
      const int nb = lp->mmse_nb, nbp = lp->mmse_nbp;
      float* Wr = reinterpret_cast<float*>(scratch);
      float* Wi = Wr + MB * MB;
      float* Xr = Wi + MB * MB;
      float* Xi = Xr + MB * nbp;
      const float* wsrc = reinterpret_cast<const float*>(tw + CE_FFT_SIZE);
      for (int i = tid; i < 2 * MB * MB; i += NT) Wr[i] = wsrc[i];
      const int m_eff = n_re < MB ? n_re : MB;
      const int lane = tid & 63, wave = tid >> 6;
#pragma unroll 1
      for (int l = 0; l < L; ++l) {
        float2* Pl = Ph + l * n_re_pad;
        for (int i = tid; i < MB * nbp; i += NT) {
          const int k = i / nbp, n = i - k * nbp;
          float2 v = make_float2(0.f, 0.f);
          if (n < nb && k < m_eff) {
            const int s0 = n * m_eff < n_re - m_eff ? n * m_eff : n_re - m_eff;
            v = Pl[s0 + k];
          }
          Xr[i] = v.x;
          Xi[i] = v.y;
        }
        __syncthreads();
        for (int nt = wave; nt < nbp / 16; nt += NW) {
          f32x4 yr[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, yi[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int ks = 0; ks < MB / 4; ++ks) {
            const int k = ks * 4 + (lane >> 4);
            const float xr = Xr[k * nbp + nt * 16 + (lane & 15)], xi = Xi[k * nbp + nt * 16 + (lane & 15)];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const float wr = Wr[k * MB + mt * 16 + (lane & 15)], wi = Wi[k * MB + mt * 16 + (lane & 15)];
              yr[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr, xr, yr[mt], 0, 0, 0);
              yr[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi, -xi, yr[mt], 0, 0, 0);
              yi[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wi, xr, yi[mt], 0, 0, 0);
              yi[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr, xi, yi[mt], 0, 0, 0);
            }
          }
          const int n = nt * 16 + (lane & 15);
          if (n < nb) {
            const int s0 = n * m_eff < n_re - m_eff ? n * m_eff : n_re - m_eff;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + (lane >> 4) * 4 + r, idx = s0 + m;
                if (m < m_eff && idx >= n * m_eff) Pl[idx] = make_float2(yr[mt][r], yi[mt][r]);  // the anchored last block only fills what is left
              }
          }
        }
        __syncthreads();
      }
    } else if ((FEAT & CE_FEAT_FIR) && lp->smoothing == CE_SMOOTH_FILTER) {
      const int n_pils = lp->n_pils, rc_len = lp->rc_len;
      const int pad = rc_len / 2;
      const double vmx = lp->vp_mx, vin = lp->vp_inv_n, vid = lp->vp_inv_denom;
      if (lp->filt_windowed) {  // host sets it only for the 15-tap filter; other lengths take the generic form
        float2* vpb = scratch;  // [layer of the call][2][16]: virtual pilot at distance e+1 beyond the head / tail edge
        constexpr int CVK = FUSED2 ? (KPT == 1 ? 3 : 6) : ce_conv_c(ND, KPT);  // FUSED2: 96 FIR threads per row must cover KPT * 256 pilots
        const int step = (LSM >= 2 && n_re <= ((NT - 64) / 2) * CVK) ? 2 : 1;  // two layers per call where the band leaves room
#pragma unroll 1
        for (int l = 0; l < LSM; l += step) {
          smooth_windowed<7, CVK>(Psm + l * n_re_pad, n_re_pad, min(step, LSM - l), n_re, n_pils, pad, rcz, vpb, tid, vmx, vin, vid);  // 15 taps: >= 3 PRB, comb 2
        }
      } else {
        // generic form (very wide bands): copy [virtual ; P ; virtual] to the scratch, one output per thread
        const int ext_len = lp->ext_len, lpp = lp->filt_lpp;
#pragma unroll 1
        for (int l0 = 0; l0 < LSM; l0 += lpp) {
          const int nl = min(lpp, LSM - l0);
          if (tid < nl * 32) {
            const int g = tid >> 4;
            float2* ext = scratch + (g >> 1) * ext_len;
            const bool tail = (g & 1) != 0;
            virtual_pilots(Psm + (l0 + (g >> 1)) * n_re_pad, n_re, n_pils, tail, tid & 15, vmx, vin, vid,
                           [&](int dist, float2 val) { ext[tail ? n_pils + n_re + dist : n_pils - 1 - dist] = val; });
          }
          for (int i = tid; i < nl * n_re; i += NT) {
            const int ll = i / n_re, k = i - ll * n_re;
            scratch[ll * ext_len + n_pils + k] = Psm[(l0 + ll) * n_re_pad + k];
          }
          __syncthreads();
          for (int i = tid; i < nl * n_re; i += NT) {
            const int ll = i / n_re, m = i - ll * n_re;
            const float2* x = scratch + ll * ext_len;
            double ar = 0.0, ai = 0.0;
            for (int j = 0; j < rc_len; ++j) {
              const int xi = m + n_pils + pad - j;
              if (xi >= 0 && xi < ext_len) {
                const float2 v = x[xi];
                const double w = rcz[j + CE_CONV_C - 1];
                ar += w * (double)v.x;
                ai += w * (double)v.y;
              }
            }
            Psm[(l0 + ll) * n_re_pad + m] = make_float2((float)ar, (float)ai);
          }
          __syncthreads();
        }
      }
      if (lp->interp == CE_INTERP_CNN && lp->cnn_alpha > 0.f) {
        // optional blend with one low-pass pass over the smoothed pilots (src/ce_dl_cnn.py:712-715)
        const float al = lp->cnn_alpha;
#pragma unroll 1
        for (int l = 0; l < LSM; ++l) {
          float2* Pl = Psm + l * n_re_pad;
          for (int k = tid; k < n_re; k += NT) {
            const float2 rcv = Pl[k];
            float2 sm = rcv;
            if (n_re > 2) {
              double yr, yi;
              lp3(Pl, k, n_re, &yr, &yi);
              sm = make_float2((float)yr, (float)yi);
            }
            scratch[k] = make_float2(rcv.x + al * (sm.x - rcv.x), rcv.y + al * (sm.y - rcv.y));
          }
          __syncthreads();
          for (int k = tid; k < n_re; k += NT) Pl[k] = scratch[k];
          __syncthreads();
        }
      }
    }

    if (back) {
    dump_stage(1, h);
    STAMP(5);
    // ------------------------------------------------------------ residual noise, RSRP (S9, S11)
    {
      float noise_part = 0.f, rsrp_part = 0.f;
      if constexpr (REG && !CE_RELOAD_RESID) {
        const float2* stash_p = scratch + pil_stash;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
          const int k = tid + i * NT;
          if (k < n_re && !(CE_ABLATE & 4)) {
            float2 hl[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
              hl[l] = Ph[l * n_re_pad + k];
              rsrp_part += hl[l].x * hl[l].x + hl[l].y * hl[l].y;
            }
#pragma unroll
            for (int s = 0; s < ND; ++s) {
              const float2 rp = rot_pos[s];
              float2 est = make_float2(0.f, 0.f);
#pragma unroll
              for (int l = 0; l < L; ++l) {
                float2 pv;
                if (YK && s < stash_nd) {   // (two separate loads: one select over an LDS and a global address would be a flat load)
                  pv = stash_p[(s * L + l) * n_re_pad + k];
                  pin(pv);
                } else {
                  pv = pilot_of(lh, i, s, l);
                }
                est = cadd(est, cmul(pv, cmul(hl[l], rp)));
              }
              const float dr = xr[i * ND + s].x - beta_f * est.x, di = xr[i * ND + s].y - beta_f * est.y;
              noise_part += dr * dr + di * di;
            }
          }
        }
      } else {
        PilotMap pmc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) pmc[c] = pilot_map(lh, c);
        // as in the LS stage: compile-time symbol count, KU pilot REs per iteration, every load requested up front
        auto resid_pass = [&](auto ndc) __attribute__((always_inline)) {
          constexpr int NDc = decltype(ndc)::value;
          constexpr int KU = NDc * L <= 4 ? CE_GEN_KU : 1;   // (4 layers x 2 symbols x 2 REs: 166 VGPRs and 1 % slower than one RE per iteration)
          int64_t osym[NDc], psym[NDc];
#pragma unroll
          for (int s = 0; s < NDc; ++s) {
            osym[s] = lh.dmrs_sym[s] * a.rs_sym;
            psym[s] = (lh.pil_sym0 + s) * a.ps_sym;
          }
          for (int k0 = tid; k0 < n_re; k0 += KU * NT) {
            float2 x[KU][NC][NDc], q[KU][NC][NDc][2];
#pragma unroll
            for (int u = 0; u < KU; ++u) {
              const int k = (u == 0 || k0 + u * NT < n_re) ? k0 + u * NT : k0;  // past the band: a valid address, the values are dropped
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                const int64_t sc = pilot_sc(pmc[c], re_idx, k);
#pragma unroll
                for (int s = 0; s < NDc; ++s) {
                  x[u][c][s] = rx[sc * a.rs_sc + osym[s]];
                  q[u][c][s][0] = pil[k * a.ps_re + psym[s] + (2 * c) * a.ps_l];
                  if (2 * c + 1 < L) q[u][c][s][1] = pil[k * a.ps_re + psym[s] + (2 * c + 1) * a.ps_l];
                }
              }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
              const int k = k0 + u * NT;
              if (u == 0 || k < n_re) {
#pragma unroll
                for (int l = 0; l < L; ++l) {
                  const float2 v = Ph[l * n_re_pad + k];
                  rsrp_part += v.x * v.x + v.y * v.y;
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                  const float2 h0 = Ph[(2 * c) * n_re_pad + k];
                  const float2 h1 = (2 * c + 1 < L) ? Ph[(2 * c + 1) * n_re_pad + k] : make_float2(0.f, 0.f);
#pragma unroll
                  for (int s = 0; s < NDc; ++s) {
                    const float2 xv = x[u][c][s];
                    const float2 rp = rot_pos[s];
                    float2 est = cmul(q[u][c][s][0], cmul(h0, rp));
                    if (2 * c + 1 < L) est = cadd(est, cmul(q[u][c][s][1], cmul(h1, rp)));
                    const float dr = xv.x - beta_f * est.x, di = xv.y - beta_f * est.y;
                    noise_part += dr * dr + di * di;
                  }
                }
              }
            }
          }
        };
        switch ((CE_ABLATE & 4) ? -1 : CE_GEN_NDC ? n_dmrs : 0) {
          case -1: break;
          case 1: resid_pass(std::integral_constant<int, 1>{}); break;
          case 2: resid_pass(std::integral_constant<int, 2>{}); break;
          case 3: resid_pass(std::integral_constant<int, 3>{}); break;
          case 4: resid_pass(std::integral_constant<int, 4>{}); break;
          default:   // any other count: one symbol after the other
            for (int k = tid; k < n_re; k += NT) {
#pragma unroll
              for (int l = 0; l < L; ++l) {
                const float2 v = Ph[l * n_re_pad + k];
                rsrp_part += v.x * v.x + v.y * v.y;
              }
#pragma unroll
              for (int c = 0; c < NC; ++c) {
                const int64_t sc = pilot_sc(pmc[c], re_idx, k);
                const float2 h0 = Ph[(2 * c) * n_re_pad + k];
                const float2 h1 = (2 * c + 1 < L) ? Ph[(2 * c + 1) * n_re_pad + k] : make_float2(0.f, 0.f);
                for (int s = 0; s < n_dmrs; ++s) {
                  const float2 x = rx[sc * a.rs_sc + lh.dmrs_sym[s] * a.rs_sym];
                  const float2 rp = rot_pos[s];
                  const int64_t pb = k * a.ps_re + (lh.pil_sym0 + s) * a.ps_sym;
                  float2 est = cmul(pil[pb + (2 * c) * a.ps_l], cmul(h0, rp));
                  if (2 * c + 1 < L) est = cadd(est, cmul(pil[pb + (2 * c + 1) * a.ps_l], cmul(h1, rp)));
                  const float dr = x.x - beta_f * est.x, di = x.y - beta_f * est.y;
                  noise_part += dr * dr + di * di;
                }
              }
            }
        }
      }
      double v[3] = {(double)epre_part, (double)noise_part, (double)rsrp_part};
      block_sum<3>(v, red);
      tot_epre += v[0];
      tot_noise += v[1];
      tot_rsrp += lp->beta * lp->beta * v[2] * (double)n_dmrs;
    }
    if constexpr (!TA_LATE) time_alignment(h, 1);
    }  // back
  }
  }  // passes

  STAMP(6);
  STAMP(7);
  // ---------------------------------------------------------------- slot-level epilogue (T:898-937)
  const bool apply_rot = cfo_comp && lp->cfo_estimated && !(CE_ABLATE & 16);
  // interpolation anchors {alpha, bits(r_ord)} packed for one 8-byte LDS read in the writers (T:325-337)
  for (int i = tid; i < NH * CE_MAX_CDM * 12; i += NT) {
    const int h = i / (CE_MAX_CDM * 12), c = (i / 12) % CE_MAX_CDM, r = i % 12;
    tab[i] = make_float2(lp->hop[h].alpha[c][r], __int_as_float(lp->hop[h].r_ord[c][r]));
  }
  if (tid < 16) {
    double cfo = 0.0;  // running mean over the hops that estimated one (T:605-609)
    bool have = false;
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (lp->hop[h].has_cfo && !(CE_ABLATE & 16)) {
        cfo = have ? (cfo + misc[h]) / 2 : misc[h];
        have = true;
      }
    float2 r = make_float2(1.f, 0.f);
    if (apply_rot && tid < CE_MAX_SYMBOLS) {
      float sn, cs;
      sincospi_f64arg(2.0 * sst_l[tid] * cfo, &sn, &cs);
      r = make_float2(cs, sn);
    }
    rot_final[tid] = r;
    if (tid == 0 && !(CE_ABLATE & 128)) {  // 128: timing experiment, no scalar outputs
      const double np = lp->inv_n_pilots;
      a.rsrp[item] = tot_rsrp * np * lp->inv_layers;
      a.epre[item] = tot_epre * np;
      a.noise[item] = tot_noise * lp->inv_noise_den;
      a.cfo[item] = lp->cfo_estimated ? cfo * lp->scs : __longlong_as_double(0x7FF8000000000000ll);
    }
  }
  __syncthreads();

  STAMP(8);
#if CE_PRIO == 1
  __builtin_amdgcn_s_setprio(3);   // experiment: the waves that feed the store stream win the issue arbitration
#elif CE_PRIO == 2
  __builtin_amdgcn_s_setprio(0);
#endif
  // ---------------------------------------------------------------- interpolate + replicate + CFO ramp (S10)
  const int n_sym = lp->n_sym;
  const int row = n_sym * L;  // complex values per subcarrier
  const int64_t total = (int64_t)lp->n_sc * row;
  float2* out = a.out + item * total;

  // linear interpolation at hop-relative subcarrier p of layer l (T:311-338); exact pilot value outside
  // the first/last pilot, left + alpha*(right-left) elsewhere (also AT pilots, as the reference does)
  auto interp_at = [&](int h, int l, int p) -> float2 {
    const CeDevHop& lh = lp->hop[h];
    const int c = l >> 1;
    const int q = p / 12, r = p - 12 * q;
    const float2 t = tab[(h * CE_MAX_CDM + c) * 12 + r];
    int ro = q * lh.dpp[c] + __float_as_int(t.y);
    int lo = ro - 1;
    if (p >= lh.last_idx[c]) lo = ro = n_re - 1;
    lo = lo < 0 ? 0 : lo;
    const float2* Pl = P + (h * L + l) * n_re_pad;
    const float2 u = Pl[lo], v = Pl[ro];
    return make_float2(u.x + t.x * (v.x - u.x), u.y + t.x * (v.y - u.y));
  };

  // ce_dl_cnn.py's in-painting + two low-pass passes for a comb-2 DM-RS, closed form (ce_api.hip: cnn_comb2): pilots
  // keep their value (C:507-508), an RE between pilots k and k+1 gets (P[k-1] + 15 P[k] + 15 P[k+1] + P[k+2]) / 32, pilot
  // indices outside the band reflected the way the RE-domain reflect padding (C:433-451) maps them.
  auto cnn2_at = [&](int h, int l, int p) -> float2 {
    const CeDevHop& lh = lp->hop[h];
    const int off = ((lh.mask12 >> (16 * (l >> 1))) & 1u) ? 0 : 1;  // pilots on even or on odd REs
    const float2* Pl = P + (h * L + l) * n_re_pad;
    const int q = p - off, K = n_re - 1;
    if (q >= 0 && !(q & 1)) return Pl[q >> 1];
    const int kl = (q - 1) >> 1;  // pilot on the left (-1: the RE in front of the first pilot)
    auto refl = [&](int k) { return k < 0 ? -k - off : (k > K ? 2 * K + 1 - off - k : k); };
    const float2 a = Pl[refl(kl - 1)], b = Pl[refl(kl)], c = Pl[refl(kl + 1)], d2 = Pl[refl(kl + 2)];
    return make_float2(((a.x + d2.x) + 15.f * (b.x + c.x)) * (1.f / 32.f), ((a.y + d2.y) + 15.f * (b.y + c.y)) * (1.f / 32.f));
  };

  // the same for any mask whose in-painting reaches its fixed point within the reference's iteration count (ce_api.hip:
  // cnn_comb2 == 2): 5-tap binomial (two [1 2 1] / 4 passes, reflect padding C:433-451) over the linear fill, pilots restored
  auto cnnfp_at = [&](int h, int l, int p) -> float2 {
    const CeDevHop& lh = lp->hop[h];
    const int n = lh.n_sc_hop;
    const int q12 = p / 12, r12 = p - 12 * q12;
    if ((lh.mask12 >> (16 * (l >> 1) + r12)) & 1u) return interp_at(h, l, p);   // a pilot RE keeps its value (C:507-508)
    auto rf = [&](int i) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); };
    // y = lp(lp(x)), each pass with its own reflect padding: y[p] = sum_d w_d lp1(rf(p + d)), lp1(i) = sum_e w_e x(rf(i + e))
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int d = -1; d <= 1; ++d) {
      const int i = rf(p + d);
      const float wd = d == 0 ? 0.5f : 0.25f;
#pragma unroll
      for (int e = -1; e <= 1; ++e) {
        const float2 x = interp_at(h, l, rf(i + e));
        const float w = wd * (e == 0 ? 0.5f : 0.25f);
        acc.x += w * x.x;
        acc.y += w * x.y;
      }
    }
    return acc;
  };

  // ce_dl_cnn.py's in-painting iterated as the reference does (masks without a closed form): the response of every
  // (hop, layer) over the hop's band -> scratch rows of cnn_h_stride elements (band-relative: element 0 is subcarrier
  // sc0), all before the writers run.  When the rows of all (hop, layer) pairs do not fit the LDS together (many layers
  // of wide hops, plan flag cnn_rowwise) there is ONE row: each pair is in-painted and stored in turn, every element
  // exactly once -- by the pass of the LAST hop whose rectangle covers it (hop 2 overwrites, src/ce_dl_cnn.py:233-352), or
  // by the zero pass when none does -- so no ordering between passes is needed.  (One call site of the in-painting, and
  // it is inlined: an out-of-line copy would receive its LDS pointers as flat addresses.)
  const bool cnn_iterated = (FEAT & CE_FEAT_EXT) && lp->interp == CE_INTERP_CNN && !lp->cnn_comb2;
  const bool rowwise = cnn_iterated && lp->cnn_rowwise;
  if constexpr ((FEAT & CE_FEAT_EXT) != 0) {
    if (cnn_iterated) {
      float2* pong = reinterpret_cast<float2*>(reinterpret_cast<unsigned char*>(scratch) + lp->cnn_pong_off);
      auto owner = [&](int sc, int sym) __attribute__((always_inline)) -> int {
        for (int h = NH - 1; h >= 0; --h) {
          const CeDevHop& lh = lp->hop[h];
          if (sym >= lh.sym0 && sym < lh.sym1 && sc >= lh.sc0 && sc < lh.sc0 + lh.n_sc_hop) return h;
        }
        return -1;
      };
      if (rowwise) {
        for (int64_t e = tid; e < total; e += NT) {
          const int sc = (int)(e / row), sym = (int)(e - (int64_t)sc * row) / L;
          if (owner(sc, sym) < 0) out[e] = make_float2(0.f, 0.f);
        }
      }
#pragma unroll 1
      for (int hl = 0; hl < NH * L; ++hl) {
        const int h = hl / L, l = hl - h * L;
        const CeDevHop& lh = lp->hop[h];
        const int c = l >> 1;
        const unsigned mask12 = (unsigned)((lh.mask12 >> (16 * c)) & 0xFFFu);
        const int n_it = lh.n_sc_hop / 8 > 6 ? lh.n_sc_hop / 8 : 6;  // C:293
        float2* rowp = scratch + (rowwise ? 0 : hl * lp->cnn_h_stride);
        cnn_inpaint_layer(rowp, pong, P + (h * L + l) * n_re_pad, lh.n_sc_hop, mask12, lh.dpp[c], n_it, lp->cnn_gmax, lp->cnn_rcp, tid);
        if (rowwise) {
          const int ns = lh.sym1 - lh.sym0, cnt = lh.n_sc_hop * ns;
          for (int i = tid; i < cnt; i += NT) {
            const int p = i / ns, sym = lh.sym0 + (i - p * ns), sc = lh.sc0 + p;
            if (owner(sc, sym) == h) {
              float2 val = rowp[p];
              if (apply_rot) val = cmul(val, rot_final[sym]);
              out[((int64_t)sc * n_sym + sym) * L + l] = val;
            }
          }
          __syncthreads();
        }
      }
      __syncthreads();
    }
  }

  if ((CE_ABLATE & 8) || rowwise) {
  } else if (n_sym == CE_MAX_SYMBOLS || n_sym == 12) {
    auto fast_writer = [&](auto ns2c) __attribute__((always_inline)) {
    // Fast writer.  The hop of an element is decided by its symbol alone -- or, for two hops whose fill rectangles share
    // symbols (plan: sym_overlap; the harness's own two-hop convention), by its symbol AND its subcarrier: the last hop
    // whose symbol range (thread constant `cand`) and band cover it (T:872-896).  A subcarrier's (14 symbols x L layers) is 7L float4 and 7L divides 252 for L = 1..4, so
    // each of ACTIVE (a multiple of 252) threads owns ONE (symbol, layer) float4 phase for the whole item:
    // its two rotation phasors and hop/layer selection live in registers, and a workgroup iteration
    // stores ACTIVE*16 contiguous bytes.  The interpolated, un-rotated response H[hop][layer][sc] is staged in the LDS
    // scratch, a chunk of subcarriers at a time.
    constexpr int NS2 = decltype(ns2c)::value;   // symbol pairs per subcarrier: 7, or 6 for 12-symbol grids (extended CP)
    constexpr int ROW4 = NS2 * L;           // float4 per subcarrier
    constexpr int ACTIVE = (NT / (36 * NS2)) * (36 * NS2);   // 252 / 216 threads: 36 / L subcarriers per iteration either way
    constexpr int SC_STEP = ACTIVE / ROW4;  // subcarriers per workgroup iteration
    const int ch_log2 = lp->wr_ch_log2, CH = 1 << ch_log2;
    const int ph = tid % ROW4, sc_lane = tid / ROW4;
    int hsel[2], lsel[2];
    float2 rsel[2];
    unsigned cand[2];   // bit h: hop h's symbols cover this thread's symbol
    const bool ovl = NH == 2 && lp->sym_overlap;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int rem = 2 * ph + e, sym = rem / L;
      lsel[e] = rem - sym * L;
      int h = -1;
      cand[e] = 0u;
#pragma unroll
      for (int hh = 0; hh < NH; ++hh)
        if (sym >= lp->hop[hh].sym0 && sym < lp->hop[hh].sym1) { h = hh; cand[e] |= 1u << hh; }  // a later hop overwrites (T:872-896)
      hsel[e] = h < 0 ? 0 : h;
      rsel[e] = h < 0 ? make_float2(0.f, 0.f) : rot_final[sym];  // rot_final == 1 when no CFO ramp applies
    }
    float4* out4 = reinterpret_cast<float4*>(out);
    // sym_overlap: subcarrier ranges of the two hops; element e at subcarrier s takes hop 1 where (cand[e] & 2) and s is in
    // hop 1's band, else hop 0 where (cand[e] & 1) and s is in hop 0's band, else zero
    const int ob0 = lp->hop[0].sc0, on0 = lp->hop[0].n_sc_hop, ob1 = lp->hop[NH - 1].sc0, on1 = lp->hop[NH - 1].n_sc_hop;
    if (cnn_iterated) {
      // the in-painted rows (staged above) through the same phase-owning store loop; zeros outside a hop's band
      const int hs = lp->cnn_h_stride, n_sc = lp->n_sc;
      const float2* HA = scratch + (hsel[0] * L + lsel[0]) * hs;
      const float2* HB = scratch + (hsel[1] * L + lsel[1]) * hs;
      const int a0 = lp->hop[hsel[0]].sc0, an = lp->hop[hsel[0]].n_sc_hop;
      const int b0 = lp->hop[hsel[1]].sc0, bn = lp->hop[hsel[1]].n_sc_hop;
      if (ovl) {
        if (tid < ACTIVE) {
          float4* o = out4 + tid;
          const float2* R0a = scratch + lsel[0] * hs;
          const float2* R1a = scratch + (L + lsel[0]) * hs;
          const float2* R0b = scratch + lsel[1] * hs;
          const float2* R1b = scratch + (L + lsel[1]) * hs;
#pragma unroll 2
          for (int s = sc_lane; s < n_sc; s += SC_STEP) {
            const bool in0 = (unsigned)(s - ob0) < (unsigned)on0, in1 = (unsigned)(s - ob1) < (unsigned)on1;
            const float2 a0v = R0a[in0 ? s - ob0 : 0], a1v = R1a[in1 ? s - ob1 : 0], b0v = R0b[in0 ? s - ob0 : 0], b1v = R1b[in1 ? s - ob1 : 0];
            const bool ua1 = in1 && (cand[0] & 2u), ua0 = !ua1 && in0 && (cand[0] & 1u);
            const bool ub1 = in1 && (cand[1] & 2u), ub0 = !ub1 && in0 && (cand[1] & 1u);
            const float2 za = ua1 ? a1v : ua0 ? a0v : make_float2(0.f, 0.f), zb = ub1 ? b1v : ub0 ? b0v : make_float2(0.f, 0.f);
            const float2 ya = cmul(za, rsel[0]), yb = cmul(zb, rsel[1]);
            store_f4(o, make_float4(ya.x, ya.y, yb.x, yb.y));
            o += ACTIVE;
          }
        }
      } else if (tid < ACTIVE) {
        float4* o = out4 + tid;
#pragma unroll 4
        for (int s = sc_lane; s < n_sc; s += SC_STEP) {
          const bool ina = (unsigned)(s - a0) < (unsigned)an, inb = (unsigned)(s - b0) < (unsigned)bn;
          const float2 ua = HA[ina ? s - a0 : 0], ub = HB[inb ? s - b0 : 0];   // always an in-range LDS address
          const float2 za = ina ? ua : make_float2(0.f, 0.f), zb = inb ? ub : make_float2(0.f, 0.f);
          const float2 ya = cmul(za, rsel[0]), yb = cmul(zb, rsel[1]);
          store_f4(o, make_float4(ya.x, ya.y, yb.x, yb.y));
          o += ACTIVE;
        }
      }
    } else if (direct_ok<SC_STEP>() && lp->interp != CE_INTERP_CNN && ovl) {
      if constexpr (direct_ok<SC_STEP>() && NH == 2) write_grid_direct_ovl<L, NS2>(lp, P, tab, rot_final, out4, n_re, n_re_pad, tid);
    } else if (direct_ok<SC_STEP>() && lp->interp != CE_INTERP_CNN) {
      if constexpr (direct_ok<SC_STEP>()) write_grid_direct<L, NH, (ce_min_waves(NH, ND, KPT, FEAT, L) >= 5 ? 2 : ce_min_waves(NH, ND, KPT, FEAT, L) == 2 ? CE_WR_WIDE : CE_WR_UNROLL), NS2>(lp, P, tab, rot_final, out4, n_re, n_re_pad, tid);  // (the 96-VGPR tiers have no room for four iterations' operands; 0 = per-element branches: measured 2-4 % faster where only two workgroups share a CU)
    } else if (!(CE_LEAN == 1 && direct_ok<SC_STEP>())) {
      const float2* HA = scratch + ((hsel[0] * L + lsel[0]) << ch_log2);
      const float2* HB = scratch + ((hsel[1] * L + lsel[1]) << ch_log2);
#pragma unroll 1
      for (int c0 = 0; c0 < lp->n_sc; c0 += CH) {
        const int cn = min(CH, lp->n_sc - c0);
        if (CE_CNNFP_STAGED && lp->interp == CE_INTERP_CNN && lp->cnn_comb2 == 2) {
          // ce_dl_cnn's binomial closed form (cnnfp_at) in two steps: the linear fill of the chunk and two subcarriers either
          // side -> a second LDS row per (hop, layer), then the nine taps from there: one interpolation per RE instead of
          // nine.  Same terms in the same order (bit-identical); the two REs at either band edge, where the reflect padding
          // of the two passes matters, keep the direct form.
          float2* xb = scratch + NH * L * CH;   // [hop x layer][CH + 4] (plan: scratch sized for it)
#pragma unroll 1
          for (int hl = 0; hl < NH * L; ++hl) {
            const int h = hl / L, l = hl - h * L;
            const int n = lp->hop[h].n_sc_hop, base = c0 - lp->hop[h].sc0;
            for (int s = tid - 2; s < cn + 2; s += NT) {
              const int p = base + s;
              xb[hl * (CH + 4) + s + 2] = (p >= 0 && p < n) ? interp_at(h, l, p) : make_float2(0.f, 0.f);
            }
          }
          __syncthreads();
          for (int i = tid; i < NH * L * CH; i += NT) {
            const int s = i & (CH - 1), hl = i >> ch_log2;
            if (s < cn) {
              const int h = hl / L, l = hl - h * L;
              const CeDevHop& lh = lp->hop[h];
              const int n = lh.n_sc_hop, p = c0 + s - lh.sc0;
              float2 v = make_float2(0.f, 0.f);
              if (p >= 0 && p < n) {
                const float2* x = xb + hl * (CH + 4) + s + 2;
                const int q12 = p / 12, r12 = p - 12 * q12;
                if ((lh.mask12 >> (16 * (l >> 1) + r12)) & 1u) {
                  v = x[0];                                   // a pilot RE keeps its value (C:507-508)
                } else if (p < 2 || p > n - 3) {
                  v = cnnfp_at(h, l, p);
                } else {
#pragma unroll
                  for (int d = -1; d <= 1; ++d) {
                    const float wd = d == 0 ? 0.5f : 0.25f;
#pragma unroll
                    for (int e = -1; e <= 1; ++e) {
                      const float2 xv = x[d + e];
                      const float w = wd * (e == 0 ? 0.5f : 0.25f);
                      v.x += w * xv.x;
                      v.y += w * xv.y;
                    }
                  }
                }
              }
              scratch[i] = v;
            }
          }
        } else {
        for (int i = tid; i < NH * L * CH; i += NT) {
          const int s = i & (CH - 1), hl = i >> ch_log2;
          if (s < cn) {
            const int h = hl / L, l = hl - h * L;
            const int p = c0 + s - lp->hop[h].sc0;
            float2 v = make_float2(0.f, 0.f);
            if (p >= 0 && p < lp->hop[h].n_sc_hop)
              v = lp->interp != CE_INTERP_CNN ? interp_at(h, l, p) : lp->cnn_comb2 == 1 ? cnn2_at(h, l, p) : cnnfp_at(h, l, p);
            scratch[i] = v;
          }
        }
        }
        __syncthreads();
        if (ovl) {
          // rows of both hops are staged (zeros outside a hop's own band): the later hop wins where its band covers the subcarrier
          if (tid < ACTIVE) {
            float4* o = out4 + (int64_t)c0 * ROW4 + tid;
            const float2* R0a = scratch + (lsel[0] << ch_log2);
            const float2* R1a = scratch + ((L + lsel[0]) << ch_log2);
            const float2* R0b = scratch + (lsel[1] << ch_log2);
            const float2* R1b = scratch + ((L + lsel[1]) << ch_log2);
#pragma unroll 2
            for (int s = sc_lane; s < cn; s += SC_STEP) {
              const bool in1 = (unsigned)(c0 + s - ob1) < (unsigned)on1;
              const float2 a0v = R0a[s], a1v = R1a[s], b0v = R0b[s], b1v = R1b[s];
              const float2 za = (in1 && (cand[0] & 2u)) ? a1v : (cand[0] & 1u) ? a0v : make_float2(0.f, 0.f);
              const float2 zb = (in1 && (cand[1] & 2u)) ? b1v : (cand[1] & 1u) ? b0v : make_float2(0.f, 0.f);
              const float2 ya = cmul(za, rsel[0]), yb = cmul(zb, rsel[1]);
              store_f4(o, make_float4(ya.x, ya.y, yb.x, yb.y));
              o += ACTIVE;
            }
          }
        } else if (tid < ACTIVE) {
          float4* o = out4 + (int64_t)c0 * ROW4 + tid;
#pragma unroll 4
          for (int s = sc_lane; s < cn; s += SC_STEP) {
            const float2 va = HA[s], vb = HB[s];
            const float2 ya = cmul(va, rsel[0]), yb = cmul(vb, rsel[1]);
            store_f4(o, make_float4(ya.x, ya.y, yb.x, yb.y));
            o += ACTIVE;
          }
        }
        __syncthreads();
      }
    }
    };   // fast_writer
    if (n_sym == CE_MAX_SYMBOLS) fast_writer(std::integral_constant<int, 7>{});
    else fast_writer(std::integral_constant<int, 6>{});
  } else if (!CE_LEAN) {
    // generic writer (grids of neither 14 nor 12 symbols, for either interpolator): decode (subcarrier, symbol, layer) per
    // element; where the hops' rectangles overlap the later hop wins (T:872-896, src/ce_dl_cnn.py:233-352)
    auto elem = [&](int sc, int rem) -> float2 {
      const int sym = rem / L, l = rem - sym * L;
      float2 val = make_float2(0.f, 0.f);
#pragma unroll
      for (int h = NH - 1; h >= 0; --h) {
        const CeDevHop& lh = lp->hop[h];
        const int p = sc - lh.sc0;
        if (sym >= lh.sym0 && sym < lh.sym1 && p >= 0 && p < lh.n_sc_hop) {
          if (lp->interp != CE_INTERP_CNN) val = interp_at(h, l, p);
          else if (lp->cnn_comb2 == 1) val = cnn2_at(h, l, p);
          else if (lp->cnn_comb2 == 2) val = cnnfp_at(h, l, p);
          else if constexpr ((FEAT & CE_FEAT_EXT) != 0) val = scratch[(h * L + l) * lp->cnn_h_stride + p];
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
  STAMP(10);
#if CE_PRIO == 1
  __builtin_amdgcn_s_setprio(0);
#endif
  if constexpr (TA_LATE) {
    // ---------------------------------------------------------------- time alignment of each hop (S8)
    // Nothing the grid needs depends on it, so it runs here, while this workgroup's stores drain: the read ->
    // estimate -> write chain of an item is shorter by this stage, the longest of the estimation.
    __syncthreads();  // the writers are done with the scratch
    if (NH == 2 && L == 1 && lp->ta_lp == 2) {
      time_alignment(0, 2);  // both hops' transforms side by side
    } else {
#pragma unroll 1
      for (int h = 0; h < NH; ++h) {
        if (NH > 1) {
          tid = tid0;
          asm volatile("" : "+v"(tid) : : "memory");
        }
        time_alignment(h, 1);
      }
    }
  }
  if (tid == 0 && !(CE_ABLATE & 128)) a.ta[item] = (NH == 2) ? tot_ta / 2.0 : tot_ta;  // T:918-919
  STAMP(11);
}

template <int L, int NH, int ND, int KPT, int FEAT>
int launch_t(const CeLaunchCtx& c) {
  hipLaunchKernelGGL((ce_estimate_kernel<L, NH, ND, KPT, FEAT>), dim3((unsigned)c.args->n_local), dim3(NT), c.lds, c.stream,
                     c.dplan, c.re_idx, c.ta_inv, c.tw, *c.args);
  return (int)hipGetLastError();
}

// Raises the kernel's dynamic-LDS limit to what this plan needs and reports how many workgroups fit a CU.  Plans of
// different sizes share an instantiation, so the limit only ever grows (per device): a small plan created after a
// large one must not lower it under the large plan's launches.  Read, hipFuncSetAttribute and record happen under one
// mutex per instantiation: ctypes releases the GIL, so plans may be created from several host threads at once, and two
// creators interleaving "set 80 KB" / "set 60 KB" would otherwise leave the attribute below the recorded limit.
template <int L, int NH, int ND, int KPT, int FEAT>
int prepare_t(const CeLaunchCtx& c) {
  const void* fn = reinterpret_cast<const void*>(&ce_estimate_kernel<L, NH, ND, KPT, FEAT>);
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

template <int L, int NH, int ND, int KPT, int FEAT>
int run_t(int op, const CeLaunchCtx& c) {
  return op == CE_OP_LAUNCH ? launch_t<L, NH, ND, KPT, FEAT>(c) : prepare_t<L, NH, ND, KPT, FEAT>(c);
}

}  // namespace

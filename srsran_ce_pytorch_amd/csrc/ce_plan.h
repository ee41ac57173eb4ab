// Device-resident plan: everything process_hop() re-derives per call in the reference
// (src/ce_rule_tensorized.py:563-576, 638-647, 809-820) resolved once on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ce_hip.h"

#define CE_MAX_RC_TAPS 31   // stride 1, 3 RB (T:184-234)
#define CE_TA_HALF 144      // floor(72 * 4096 / 2048) bins each side (T:682)
#ifndef CE_THREADS
#define CE_THREADS 256      // workgroup size (4 waves); 512 also builds
#endif
#define CE_KPT (CE_THREADS == 256 ? 7 : 4)  // pilot REs per thread on the register path (n_re <= CE_KPT * CE_THREADS)
#define CE_CONV_C 9         // consecutive RC-FIR outputs per thread (sliding window), conv threads = CE_THREADS - 64
#define CE_RCZ_LEN (CE_MAX_RC_TAPS + 2 * (CE_CONV_C - 1))
#define CE_TA_ROW 272       // largest residue block of the TA transform, complex elements (ce_ta_row)
// the twiddle buffer (complex64 elements): exp(+j 2 pi m / 4096), m < 4096 | mmse W^T (extension) | the 272 twiddles the
// TA transform uses, contiguous: W256^j = tw[16 j] (j < 256), then W4096^i = tw[i] (i < 16)
#define CE_TWC_OFF (CE_FFT_SIZE + CE_MMSE_BLOCK * CE_MMSE_BLOCK)
#define CE_TW_TOTAL (CE_TWC_OFF + 256 + 16)

// ---- per-shape policies shared by the kernel template (launch bounds, stage order) and the host (LDS sizing) ----
#ifndef CE_MIN_WAVES
#define CE_MIN_WAVES 3   // waves per SIMD the register allocator must leave room for (3 workgroups per CU; 4 would spill)
#endif
#ifndef CE_MIN_WAVES_LIGHT
#define CE_MIN_WAVES_LIGHT 3   // the same for single-hop register-path kernels built without the FIR ("none" / "mean" smoothing)
#endif
#ifndef CE_MIN_WAVES_L2H2
#define CE_MIN_WAVES_L2H2 3   // re-read path, 2-4 layers x 2 hops (3: 4-11 spilled VGPRs; 2: none, but one workgroup less per CU on narrow bands)
#endif
#ifndef CE_MW3_LIMIT
#define CE_MW3_LIMIT 7   // single hop: up to this many pilot REs x symbols per thread the kernel is built for 3 workgroups per CU (168
                         // VGPRs); beyond it for CE_MW_WIDE = 2 -- the wide kernels still NEED only 114 / 161 VGPRs and run 4 / 3 per CU, but
                         // built with the looser bound (and with the TA stage inside the hop loop, ce_ta_late) they measured 2-4 %
                         // faster at every batch size (profiles/round2_budget_policy_ab.txt: one process, same buffers, +-0.3 %)
#endif
#ifndef CE_MW_WIDE
#define CE_MW_WIDE 2     // single hop beyond CE_MW3_LIMIT: waves per SIMD the allocator leaves room for
#endif
#ifndef CE_MW5_LIMIT
#define CE_MW5_LIMIT 2   // single hop: up to this many pilot REs x symbols per thread, 5 workgroups per CU (<= 96 VGPRs: the FIR shapes spill 1-2
                         // registers for it and gain 10-14 % on <= 25-PRB hops; at 4 per thread the 6-8 spilled registers cost more than the fifth workgroup gives)
#endif
#ifndef CE_MW4_LIMIT
#define CE_MW4_LIMIT 4   // single-hop register-path kernels holding <= this many pilot REs x symbols per thread are built for 4 workgroups per CU (128 VGPRs)
#endif
#ifndef CE_NH2_MW4_LIMIT
#define CE_NH2_MW4_LIMIT 2   // two hops: up to this many pilot REs x symbols per thread, 4 workgroups per CU (4-8 spilled VGPRs; measured +6..12 % on narrow hops, nothing at 4)
#endif
#ifndef CE_NH2_BOUND2_FROM
#define CE_NH2_BOUND2_FROM 99  // two hops: from this many pilot REs x symbols per thread on, the 2-wave launch bound (DM-RS symbols still parked in the LDS)
#endif
#ifndef CE_NH1_PREG_LIMIT
#define CE_NH1_PREG_LIMIT 99  // single hop: up to this many pilot REs x symbols per thread the DM-RS symbols stay in registers too
#endif
#ifndef CE_NH2_MW2_FROM
#define CE_NH2_MW2_FROM 22  // two hops: from this many pilot REs x symbols per thread on, 2 workgroups per CU with everything in registers
                            // (no shape reaches it: with the DM-RS symbols fetched once per hop and parked in the LDS, three
                            // workgroups per CU beat two with everything in registers -- 3 symbols x 200 PRB: 4.09 -> 3.37 ms)
#endif
#ifndef CE_TA_OVER_P
#define CE_TA_OVER_P 1      // plan: ta_over_p (the last hop's second set of TA residue blocks over the first hop's P); 0: A/B builds
#endif
#ifndef CE_CNNFP_STAGED
#define CE_CNNFP_STAGED 1   // ce_dl_cnn's binomial closed form (cnn_comb2 == 2) in the staged writer: the linear fill is staged first, then the
                            // nine taps from LDS (0: nine interpolations per RE; A/B builds).  Kernel and host (scratch sizing) must agree.
#endif
// Feature set compiled into an instantiation (template parameter FEAT): a register-path kernel only carries the
// smoothing code its plans run, so e.g. the headline kernel's register allocation is not shaped by the MFMA block
// of the mmse extension or the iterated in-painting it never executes.
//   CE_FEAT_FIR  the raised-cosine FIR with virtual pilots (Smoothing="filter", T:637-664) and the CNNSmoothingAlpha blend
//   CE_FEAT_EXT  the unpinned mmse extension (MFMA) and ce_dl_cnn's iterated in-painting (masks without a closed form)
// "none" / "mean" smoothing, both interpolation closed forms and every writer are in all kernels.
constexpr int CE_FEAT_FIR = 1, CE_FEAT_EXT = 2;

// Register budget of the register-path kernels by pilot REs x DM-RS symbols per thread (KPT * ND): workgroups per CU
// the allocator must leave room for, and whether the DM-RS symbols stay in registers next to the received pilots
// (otherwise they are fetched once per hop, the CFO and LS stages share the products rx * conj(pilot), and the residual
// stage reads them back from the LDS scratch: plan pil_stash).  The single-hop 3-symbol wide kernel measured 2.80 ms at
// 2 workgroups per CU with everything in registers vs 2.93 ms at 3 with the symbols parked; two hops the other way round.
constexpr int ce_min_waves(int nh, int nd, int kpt, int feat, int layers = 1) {
  const int n = nd * kpt;
  if (nd == 0) return (layers >= 2 && nh == 2) ? CE_MIN_WAVES_L2H2 : CE_MIN_WAVES;
  if (nh == 1) return n <= CE_MW5_LIMIT ? 5 : n <= CE_MW4_LIMIT ? 4 : n <= CE_MW3_LIMIT ? ((feat & CE_FEAT_FIR) ? CE_MIN_WAVES : CE_MIN_WAVES_LIGHT) : CE_MW_WIDE;
  return n <= CE_NH2_MW4_LIMIT ? 4 : n < CE_NH2_MW2_FROM ? (n >= CE_NH2_BOUND2_FROM ? 2 : CE_MIN_WAVES) : 2;
}
constexpr bool ce_pilots_in_regs(int nh, int nd, int kpt) {
  const int n = nd * kpt;
  return nd > 0 && (nh == 1 ? n <= CE_NH1_PREG_LIMIT : (n <= 8 || n >= CE_NH2_MW2_FROM));
}
// Where the time-alignment stage runs: after the grid writer (the read -> estimate -> write chain of an item is shorter by
// its longest stage, which then overlaps the draining stores), or inside the hop loop before it.  Same arithmetic either way.
#ifndef CE_TA_LATE
#define CE_TA_LATE -1   // -1: per-shape policy below; 0 / 1: force (A/B builds)
#endif
// Policy from A/Bs of both placements over tools/perf_cases.py -- first between child processes
// (profiles/round2_ta_placement_ab.txt), then inside one process on the same buffers (tools/ab_inproc.py, +-0.3 %;
// profiles/round2_budget_policy_ab.txt): late wins 8-17 % on the narrow shapes (one or two hops), 6-7 % on two-hop mid
// bands and 4 layers, 2-4 % for 2-4 layers x 2 hops; early wins 4 % on the 3-symbol wide kernel and 1-2 % on the wide
// single-hop kernels built with the 2-wave bound; the rest is within the resolution.
constexpr bool ce_ta_late(int layers, int nh, int nd, int kpt, int feat) {
  if (CE_TA_LATE >= 0) return CE_TA_LATE != 0;
  if (nd > 0 && nh == 1 && ce_min_waves(nh, nd, kpt, feat) == 2) return false;
  // the widest shapes of the 4-workgroup tier (128 VGPRs) with the FIR compiled in: late placement would spill 1-4 registers
  if (nh == 1 && nd * kpt == CE_MW4_LIMIT && (feat & CE_FEAT_FIR)) return false;
  return true;
}
// complex elements per 16 x 16 residue block of the TA transform: unpadded + swizzled where the stage runs after the writer,
// 16 x 17 where it runs inside the hop loop (ce_estimate_kernel.h: ta_at)
constexpr int ce_ta_row(bool late) { return late ? 256 : 272; }

struct CeDevHop {
  int32_t n_dmrs;                     // DM-RS symbols in the hop
  int32_t dmrs_sym[CE_MAX_SYMBOLS];   // their OFDM symbol indices (ascending)
  int32_t pil_sym0;                   // first column of this hop along the pilots' symbol axis (T:823, T:871)
  int32_t sc0, n_sc_hop;              // fill window 12*PRBstart .. +12*nPRBs (T:301-304)
  int32_t sym0, sym1;                 // fill symbols [startSymbol, startSymbol+nAllocatedSymbols) (T:306-309)
  int32_t has_cfo;                    // n_dmrs >= 2 (T:388)
  int32_t re_off[CE_MAX_CDM];         // offset of the CDM group's pilot subcarrier list in the re_idx table
  int32_t last_idx[CE_MAX_CDM];       // hop-relative position of the last pilot RE (T:314)
  int32_t dpp[CE_MAX_CDM];            // pilots per PRB of the CDM group
  int32_t r_ord[CE_MAX_CDM][12];      // right-anchor ordinal inside the PRB for RE r (T:325)
  float alpha[CE_MAX_CDM][12];        // (pos-left)/(right-left) in float32 (T:333-337)
  double two_pi_nsamples;             // 2*pi*nSamples (T:418-426)
  double inv_two_pi_nsamples;
  int32_t ta_nres;                    // residues mod 16 of the subcarriers the TA scatter touches (T:672-675)
  int32_t ta_res[16];
  int32_t ta_inv_off;                 // offset of this hop's subcarrier -> pilot-ordinal table (0xFFFF = no pilot)
  int32_t contig;                     // maskPRBs == [PRBstart, PRBstart+nPRBs): pilot positions are computed, not looked up
  int32_t prb_start, n_prbs;
  uint32_t div_magic[CE_MAX_CDM];     // floor(2^32 / dpp) + 1: k / dpp == umulhi(k, magic) for k < 2^16 (dpp >= 2; unused for dpp == 1)
  uint64_t pos_packed[CE_MAX_CDM];    // 4 bits per pilot j of a PRB: its RE position
  uint64_t ord_packed;                // last CDM group: 4 bits per RE r: pilot ordinal inside the PRB, 15 = not a pilot
  uint64_t ta_res_packed;             // 4 bits per entry of ta_res
  uint32_t mask12;                    // DMRSREmask columns: bits 0-11 CDM group 0, bits 16-27 CDM group 1
  uint32_t ta_win;                    // narrow band: (blocks of 256 subcarriers the shifted band spans, 1 or 2) << 16 | shift (multiple of 16); 0: full first pass
};

struct alignas(16) CeDevPlan {
  int32_t n_sc, n_sym, n_layers, n_cdm, n_hops, smoothing, cfo_comp, interp;
  int32_t n_re, n_re_pad, n_pils, rc_len, ext_len, filt_lpp;
  int32_t cfo_estimated, reg_nd;      // reg_nd: DM-RS symbols per hop held in registers (0 = re-read path)
  int32_t feat, cnn_rowwise;          // cnn_rowwise: the in-painted rows of all (hop, layer) pairs exceed the LDS: one at a time; CE_FEAT_* bits the plan's kernel must carry (ce_estimate_kernel.h)
  int32_t reg_kpt, sym_overlap;              // pilot REs per thread on the register path: smallest of 1, 2, 4, CE_KPT covering n_re; sym_overlap: the hops' fill rectangles share symbols
  int32_t scratch_bytes, wr_ch_log2;  // LDS scratch size; log2 of the writer's subcarrier chunk
  float beta_f;
  double beta, scs, denom_cdm, n_pilots, noise_den;
  double inv_n_pilots, inv_layers, inv_noise_den, inv_denom_cdm;  // reciprocals: one multiply instead of a float64 divide
  double sst[CE_MAX_SYMBOLS];         // symbolStartTime (T:809-820)
  double sst_dmrs[CE_MAX_HOPS][CE_MAX_SYMBOLS];  // symbolStartTime at each hop's DM-RS symbols
  double rc[CE_MAX_RC_TAPS];          // RC taps, unit sum (T:184-234)
  double rcz[CE_RCZ_LEN];             // the same taps with CE_CONV_C-1 zeros on both sides (windowed FIR)
  double vp_mx, vp_inv_n, vp_inv_denom;  // regression constants of the n_pils-point straight-line fit (T:105-117)
  int32_t filt_windowed, pad1;        // 1: n_re <= (CE_THREADS-64)*CE_CONV_C -> sliding-window FIR
  // ce_dl_cnn.py in-painting (interp == CE_INTERP_CNN): whole-band H per (hop, layer) in the scratch
  int32_t cnn_h_stride;               // complex elements between consecutive (hop, layer) H rows (band-relative: longest hop band, even)
  int32_t ta_lp, pil_stash;           // layers the TA transform handles at a time (1, or 2: ce_estimate_kernel.h time_alignment);
                                      // pil_stash > 0: register-path kernels that cannot keep the DM-RS symbols in registers park the
                                      // current hop's at this complex-element offset of the scratch instead of re-reading them per stage
  int32_t cnn_pong_off, cnn_gmax;     // byte offset of the second band buffer inside the scratch; longest run of unknown REs (iterated in-painting)
  int32_t cnn_n_max;                  // longest hop band (subcarriers)
  float cnn_alpha;                    // clamp(CNNSmoothingAlpha, 0, 1) (src/ce_dl_cnn.py:712-715)
  int32_t cnn_comb2;                  // in-painting in closed form: 1 comb-2 masks (1010.. / 0101..), 2 any mask whose gaps converge within n_iters; 0 iterate
  // extension: block LMMSE smoothing (CE_SMOOTH_MMSE)
  int32_t mmse_nb, mmse_nbp;          // blocks of CE_MMSE_BLOCK pilots, padded to a multiple of 16 (MFMA N tiles)
  double cnn_rcp[5];                  // 1 / (code/4 + 1e-12), code = m[i-1] + 2 m[i] + m[i+1] (src/ce_dl_cnn.py:498-501)
  // wave-per-item kernel (ce_narrow_kernel.h) for narrow allocations: narrow = 1 when the plan runs on it
  int32_t narrow, nrw_nd_max;         // nrw_nd_max: most DM-RS symbols in a hop (sizes the per-wave staging rows)
  uint32_t nrw_magic_nre;             // floor(2^32 / n_re) + 1: idx / n_re == umulhi(idx, magic) for idx < 2^16
  int32_t nrw_h_stride;               // complex elements per (hop, layer) row of the interpolated response: widest hop band, even
  int32_t nrw_halo;                   // slots on either side of a P row: the RC filter's reach (virtual pilots + zeros), 0 without the filter
  int32_t ta_over_p;                  // 2-4 layers x 2 hops whose LDS has no room for a second set of TA residue blocks (ta_lp == 1): the LAST hop's transforms still
                                      // run two layers at a time, the second set laid over the first hop's rows of P, which nobody reads once that hop's TA is done
  int32_t nrw_pad[2];
  CeDevHop hop[CE_MAX_HOPS];
};

struct CeKernelArgs {
  const float2* rx;
  int64_t rs_b, rs_r, rs_sc, rs_sym;
  const float2* pil;
  int64_t ps_b, ps_re, ps_sym, ps_l;
  float2* out;
  double *noise, *rsrp, *epre, *ta, *cfo;
  int64_t n_items;
  int32_t n_ports;
  int64_t item0, n_local;  // a launch covers work items [item0, item0 + n_local) of the n_items batch
  float2* stage_p;             // ce_estimate_batch_stages only (else null): per-stage pilot-RE estimates, see dump_stage
  double* stage_s;             //   ... and per-hop CFO / TA bin
  unsigned long long* stamps;  // diagnostic builds (-DCE_STAMPS) only: 16 wall-clock stamps per item; otherwise null
};

// LDS carve-up shared by host (sizing) and device (offsets); all offsets multiples of 16 B.
struct CeLdsLayout {
  int32_t off_p, off_scratch, off_red, off_rot, off_tab, off_misc, off_tw, off_rcz, off_plan, total;
};

static inline __host__ __device__ CeLdsLayout ce_lds_layout(int n_hops, int n_layers, int n_re_pad, int scratch_bytes) {
  CeLdsLayout l;
  int o = 0;
  l.off_p = o;        o += n_hops * n_layers * n_re_pad * 8;
  l.off_scratch = o;  o += (scratch_bytes + 15) & ~15;
  l.off_red = o;      o += (CE_THREADS / 64) * 16 * 8;           // 16 doubles per wave
  l.off_rot = o;      o += (1 + 2 * CE_MAX_HOPS) * 16 * 8;       // final, per-hop -/+ phasors, 16 float2 each
  l.off_tab = o;      o += CE_MAX_HOPS * CE_MAX_CDM * 12 * 8;    // {alpha, r_ord} pairs
  l.off_misc = o;     o += 56 * 8;                               // doubles: cfo_hop[2], ..., TA arg-max keys [2 hops][4 waves][2] at [32]
  l.off_tw = o;       o += (256 + 16) * 8;                       // W256^j, W4096^i for the TA transform
  l.off_rcz = o;                                                 // (the RC taps are read from the LDS plan copy)
  l.off_plan = o;     o += (int)((sizeof(CeDevPlan) + 15) & ~15); // LDS copy of the plan (no scalar loads from global later)
  l.total = (o + 15) & ~15;
  return l;
}

// ---- wave-per-item kernel for narrow allocations (ce_narrow_kernel.h) ----
// A 256-thread workgroup carries CE_THREADS / 64 work items, one per wave; after one workgroup barrier (plan + twiddle copy)
// the waves never synchronise with each other again.  Shared: LDS copy of the plan, the TA twiddles.  Per wave: the hop's
// received pilots and DM-RS symbols (staged once per hop: rows of n_re_pad), P, virtual pilots, phasor tables.
#define CE_NARROW_MAX_RE 192          // pilots per DM-RS symbol and CDM group (<= 3 per lane): 32 PRB at comb 2, 16 PRB with every RE a pilot
#define CE_NARROW_HALO 16             // most slots on either side of a P row: virtual pilots (<= 12) and zeros up to the FIR's reach (<= 15);
                                      // a plan allocates what its filter reaches (nrw_halo: 7 for the usual 15 taps, 0 without the filter)
#ifndef CE_NARROW_1H1L_MAX_RE
#define CE_NARROW_1H1L_MAX_RE 24      // one hop, one layer: pilots per symbol up to which the plan takes the kernel (4 PRB at comb 2)
#endif
#ifndef CE_NARROW_LDS_LIMIT
#define CE_NARROW_LDS_LIMIT (52 * 1024)   // dynamic LDS per workgroup up to which a plan takes the kernel (3 workgroups = 12 items per CU)
#endif
#ifndef CE_NARROW_MIN_WAVES
#define CE_NARROW_MIN_WAVES 3         // waves per SIMD the register allocator leaves room for (168 VGPRs: no spills; with 4 -- 128 VGPRs, 8-50
                                      // spilled registers -- every narrow shape measured 11-22 % slower, profiles/round3_narrow_kernel_ab.txt)
#endif
struct CeNarrowLayout {
  int32_t off_plan, off_tw, off_wave0, wave_stride;   // bytes
  int32_t stage_off, p_off, vp_off, rot_off;          // inside a wave's region, bytes
  int32_t total;
};
static inline __host__ __device__ CeNarrowLayout ce_narrow_layout(int n_hops, int n_layers, int nd_max, int n_re_pad, int h_stride, int halo) {
  CeNarrowLayout l;
  const int n_cdm = (n_layers + 1) / 2;
  int o = 0;
  l.off_plan = o;   o += (int)((sizeof(CeDevPlan) + 15) & ~15);
  l.off_tw = o;     o += (256 + 16) * 8;
  l.off_wave0 = o;
  int w = 0;
  // staged hop [rx: cdm][symbol][n_re_pad] then [pilots: layer][symbol][n_re_pad]; after the hops the same bytes hold the
  // interpolated response H [hop][layer][h_stride] the writer streams out
  const int s_bytes = (n_cdm + n_layers) * nd_max * n_re_pad * 8, h_bytes = n_hops * n_layers * h_stride * 8;
  // ... and after the writer the time alignment's [4 rows][x0 | x1][16] pilot buffer (1 KB)
  const int sh_bytes = s_bytes > h_bytes ? s_bytes : h_bytes;
  l.stage_off = w;  w += sh_bytes > 128 * 8 ? sh_bytes : 128 * 8;
  l.p_off = w;      w += n_hops * n_layers * (n_re_pad + halo * 2) * 8;   // rows with a halo for the virtual pilots on both sides
  l.vp_off = l.stage_off;                                              // (the TA's pilot buffer: over the staging bytes)
  l.rot_off = w;    w += (16 + CE_MAX_HOPS * 8) * 8;                   // final[16] | per hop: -phasors[4], +phasors[4]
  l.wave_stride = (w + 15) & ~15;
  l.total = l.off_wave0 + (CE_THREADS / 64) * l.wave_stride;
  return l;
}

// Makes `device` current for the scope and restores the caller's device afterwards (no HIP call when it already is).
struct CeDeviceScope {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit CeDeviceScope(int device) {
    int cur = -1;
    err = hipGetDevice(&cur);
    if (err == hipSuccess && cur != device) {
      err = hipSetDevice(device);
      if (err == hipSuccess) prev = cur;
    }
  }
  ~CeDeviceScope() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  CeDeviceScope(const CeDeviceScope&) = delete;
  CeDeviceScope& operator=(const CeDeviceScope&) = delete;
};

// sets the calling thread's ce_last_error() text and returns `code` (ce_api.hip)
int ce_fail(int code, const char* fmt, ...);

// One call into the translation unit that holds the instantiation a plan selected (ce_inst_*.hip).
enum { CE_OP_LAUNCH = 0, CE_OP_PREPARE = 1 };
#define CE_MAX_DEVICES 64
struct CeLaunchCtx {
  const CeDevPlan* dplan;
  const uint16_t* re_idx;
  const uint16_t* ta_inv;
  const float2* tw;
  const CeKernelArgs* args;  // CE_OP_LAUNCH
  int lds;
  hipStream_t stream;
  int* blocks_per_cu;        // CE_OP_PREPARE (the plan's device is current)
};
// key = feat * 10000 + layers * 1000 + reg_nd * 10 + (narrow KPT, or 0 for the wide / generic kernels); -1: not built
#define CE_KERNEL_KEY(feat, layers, reg_nd, reg_kpt) \
  ((feat) * 10000 + (layers) * 1000 + (reg_nd) * 10 + ((reg_nd) && (reg_kpt) < CE_KPT ? (reg_kpt) : 0))
int ce_tu_reg_h1_f0(int op, int key, const CeLaunchCtx& c);   // register path (one layer), one hop, none / mean smoothing
int ce_tu_reg_h1_f1(int op, int key, const CeLaunchCtx& c);   //                               ... + RC filter, band tiers KPT 1 / 2
int ce_tu_reg_h1_f1w(int op, int key, const CeLaunchCtx& c);  //                               ... + RC filter, band tiers KPT 4 / CE_KPT
int ce_tu_reg_h2_f0(int op, int key, const CeLaunchCtx& c);   // two hops
int ce_tu_reg_h2_f1(int op, int key, const CeLaunchCtx& c);
int ce_tu_gen_h1(int op, int key, const CeLaunchCtx& c);      // re-read path (1-4 layers), every feature set; + the wide headline kernel with extensions
int ce_tu_gen_h2(int op, int key, const CeLaunchCtx& c);
int ce_tu_narrow(int op, int key, const CeLaunchCtx& c);      // wave-per-item kernel: key = layers * 10 + hops
// whether the register path (reg_nd > 0) has an instantiation carrying CE_FEAT_EXT for this shape
static inline bool ce_reg_has_ext(int n_hops, int reg_nd, int reg_kpt) { return n_hops == 1 && reg_nd == 2 && reg_kpt >= 2; }

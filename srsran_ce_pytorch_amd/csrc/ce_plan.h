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
#define CE_TA_ROW 272       // 16 x 17 complex per residue block (padded against LDS bank conflicts)
// the twiddle buffer (complex64 elements): exp(+j 2 pi m / 4096), m < 4096 | mmse W^T (extension) | the 272 twiddles the
// TA transform uses, contiguous: W256^j = tw[16 j] (j < 256), then W4096^i = tw[i] (i < 16)
#define CE_TWC_OFF (CE_FFT_SIZE + CE_MMSE_BLOCK * CE_MMSE_BLOCK)
#define CE_TW_TOTAL (CE_TWC_OFF + 256 + 16)

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
  uint32_t pad1;
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
  int32_t ta_lp, pad_ta;              // layers the TA transform handles at a time (1, or 2: ce_estimate_kernel.h time_alignment)
  int32_t cnn_pong_off, cnn_gmax;     // byte offset of the second band buffer inside the scratch; longest run of unknown REs (iterated in-painting)
  int32_t cnn_n_max;                  // longest hop band (subcarriers)
  float cnn_alpha;                    // clamp(CNNSmoothingAlpha, 0, 1) (src/ce_dl_cnn.py:712-715)
  int32_t cnn_comb2;                  // in-painting in closed form: 1 comb-2 masks (1010.. / 0101..), 2 any mask whose gaps converge within n_iters; 0 iterate
  // extension: block LMMSE smoothing (CE_SMOOTH_MMSE)
  int32_t mmse_nb, mmse_nbp;          // blocks of CE_MMSE_BLOCK pilots, padded to a multiple of 16 (MFMA N tiles)
  double cnn_rcp[5];                  // 1 / (code/4 + 1e-12), code = m[i-1] + 2 m[i] + m[i+1] (src/ce_dl_cnn.py:498-501)
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
  l.off_misc = o;     o += 56 * 8;                               // doubles: cfo_hop[2], ..., TA arg-max keys[8] at [48]
  l.off_tw = o;       o += (256 + 16) * 8;                       // W256^j, W4096^i for the TA transform
  l.off_rcz = o;                                                 // (the RC taps are read from the LDS plan copy)
  l.off_plan = o;     o += (int)((sizeof(CeDevPlan) + 15) & ~15); // LDS copy of the plan (no scalar loads from global later)
  l.total = (o + 15) & ~15;
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
int ce_tu_reg_h1_f1(int op, int key, const CeLaunchCtx& c);   //                               ... + RC filter
int ce_tu_reg_h2_f0(int op, int key, const CeLaunchCtx& c);   // two hops
int ce_tu_reg_h2_f1(int op, int key, const CeLaunchCtx& c);
int ce_tu_gen_h1(int op, int key, const CeLaunchCtx& c);      // re-read path (1-4 layers), every feature set; + the wide headline kernel with extensions
int ce_tu_gen_h2(int op, int key, const CeLaunchCtx& c);
// whether the register path (reg_nd > 0) has an instantiation carrying CE_FEAT_EXT for this shape
static inline bool ce_reg_has_ext(int n_hops, int reg_nd, int reg_kpt) { return n_hops == 1 && reg_nd == 2 && reg_kpt == CE_KPT; }

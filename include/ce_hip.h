/*
 * ce_hip.h -- C ABI of the MI355X (gfx950) PUSCH DM-RS channel-estimation library (libce_hip.so).
 *
 * This is the drop-in boundary for ONE path of the reference (pjookim/srsran-ce-pytorch):
 *     src/ce_rule_tensorized.py:745  srs_channel_estimator(received_rg, pilots, beta_dmrs, hop1, hop2, config)
 * i.e. process_hop (src/ce_rule_tensorized.py:495-739) for 1-2 hops plus the slot-level epilogue
 * (src/ce_rule_tensorized.py:898-937), batched over slots x Rx ports.  The reference has no FFI of
 * its own (it is eager PyTorch); the host side that binds these symbols is
 * srsran_ce_pytorch_amd/_lib.py (ctypes), see INTEGRATION.md.
 *
 * Plain pointers and sizes only -- no torch / C++ types cross this boundary.  All device
 * pointers must belong to the plan's device.  Every entry point returns 0 on success or a
 * negative CE_ERR_* code; ce_last_error() gives the message for the calling thread.
 */
#ifndef CE_HIP_H
#define CE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CE_ABI_VERSION 2

#define CE_MAX_LAYERS 4   /* Tx layers (pilots.shape[2]); 2 per CDM group (T:551-552) */
#define CE_MAX_CDM 2
#define CE_MAX_HOPS 2
#define CE_MAX_SYMBOLS 14 /* the reference's CFO ramp assumes a 14-symbol slot (T:928-929) */
#define CE_FFT_SIZE 4096  /* time-alignment IFFT length (T:677) */

/* config.Smoothing (T:633-668).  CE_SMOOTH_MMSE is an EXTENSION with no counterpart in the reference ("parity
 * unpinned", its only oracle is oracle/ce_oracle.py::smooth_mmse): block-wise LMMSE / Wiener smoothing over
 * CE_MMSE_BLOCK consecutive pilots, W = R (R + nsr I)^-1, R from a uniform power-delay profile on [0, tau];
 * the complex 32x32 filter is applied to all blocks of a work item as f32 MFMA (v_mfma_f32_16x16x4_f32) tiles. */
enum { CE_SMOOTH_NONE = 0, CE_SMOOTH_MEAN = 1, CE_SMOOTH_FILTER = 2, CE_SMOOTH_MMSE = 3 };
#define CE_MMSE_BLOCK 32
/* frequency interpolation: T:311-340 (linear) or src/ce_dl_cnn.py:276-295, 473-508 (fixed 3-tap partial-convolution
 * in-painting + two low-pass passes; also enables the CNNSmoothingAlpha blend of src/ce_dl_cnn.py:712-715) */
enum { CE_INTERP_LINEAR = 0, CE_INTERP_CNN = 1 };

enum {
  CE_OK = 0,
  CE_ERR_INVALID = -1,     /* malformed descriptor / arguments (maps to ValueError on the host) */
  CE_ERR_UNSUPPORTED = -2, /* valid for the reference but outside this build's limits */
  CE_ERR_HIP = -3,         /* a HIP runtime call failed */
  CE_ERR_NOMEM = -4
};

/* One frequency hop: mirrors HopConfig (src/ce_rule_tensorized.py:13-21). */
typedef struct ce_hop_desc {
  uint8_t dmrs_symbols[CE_MAX_SYMBOLS]; /* DMRSsymbols: 1 where the OFDM symbol carries DM-RS */
  uint16_t re_mask[CE_MAX_CDM];         /* DMRSREmask column c: bit r set <=> RE r of each PRB is a pilot */
  int32_t prb_start;                    /* PRBstart  */
  int32_t n_prbs;                       /* nPRBs     */
  const uint8_t* mask_prbs;             /* maskPRBs: n_prb_grid bytes (0/1); host memory, read during create */
  int32_t start_symbol;                 /* startSymbol */
  int32_t n_alloc_symbols;              /* nAllocatedSymbols */
} ce_hop_desc;

/* Slot geometry + EstimatorConfig (src/ce_rule_tensorized.py:24-29), resolved once per plan. */
typedef struct ce_plan_desc {
  int32_t abi_version;   /* = CE_ABI_VERSION */
  int32_t device;        /* HIP device ordinal the plan (tables, launches) lives on */
  int32_t n_prb_grid;    /* grid has 12*n_prb_grid subcarriers (<= 4096 for the TA IFFT, T:679) */
  int32_t n_sym;         /* OFDM symbols in the grid (14 when the CFO ramp is applied) */
  int32_t n_layers;      /* 1..CE_MAX_LAYERS */
  int32_t n_hops;        /* 1 or 2 (hop2 "empty" => 1) */
  int32_t smoothing;     /* CE_SMOOTH_* */
  int32_t cfo_compensate;/* config.CFOCompensate */
  int32_t interp;        /* CE_INTERP_* */
  int32_t reserved0;
  double scs_hz;         /* config.scs */
  double beta_dmrs;      /* betaDMRS */
  double cp_ms[CE_MAX_SYMBOLS]; /* config.CyclicPrefixDurations[0:14], milliseconds */
  double cnn_smoothing_alpha;   /* config.CNNSmoothingAlpha (src/ce_dl_cnn.py:864); 0 = off */
  double mmse_delay_spread_s;   /* CE_SMOOTH_MMSE: tau, seconds (extension) */
  double mmse_noise_to_signal;  /* CE_SMOOTH_MMSE: nsr (extension) */
  ce_hop_desc hop[CE_MAX_HOPS];
} ce_plan_desc;

typedef struct ce_plan ce_plan; /* opaque */

/* Derived facts a host needs to size buffers / shape results. */
typedef struct ce_plan_info {
  int32_t n_sc;            /* 12*n_prb_grid */
  int32_t n_re;            /* pilots per DM-RS symbol per CDM group (= pilots.shape[0]) */
  int32_t n_dmrs_total;    /* DM-RS symbols over both hops (= pilots.shape[1]) */
  int32_t cfo_estimated;   /* 1 if some hop has >= 2 DM-RS symbols, else cfo outputs are NaN (T:388-391, T:931-933) */
  int32_t lds_bytes;       /* dynamic LDS per workgroup */
  int32_t threads;         /* workgroup size */
  int64_t alg_bytes_per_item; /* n_re*n_dmrs*8*nCDM + n_sc*n_sym*L*8 (per slot x port) */
  int64_t pilot_bytes_per_slot; /* n_re*n_dmrs*L*8 */
} ce_plan_info;

/* What ce_plan_create derives on the host, for inspection and CPU-only tests. */
typedef struct ce_plan_host_view {
  int32_t n_re, n_dmrs_total, n_pils, rc_len;   /* n_pils, len(rcFilter): T:638-647 */
  int32_t reg_nd, lds_bytes, scratch_bytes, filt_windowed, cfo_estimated;
  int32_t narrow;                                /* 1: the plan runs on the wave-per-item kernel for narrow allocations */
  int32_t ta_nres[CE_MAX_HOPS], contig[CE_MAX_HOPS];
  int32_t last_idx[CE_MAX_HOPS][CE_MAX_CDM];     /* last pilot RE of the hop band (T:314) */
  int32_t r_ord[CE_MAX_HOPS][CE_MAX_CDM][12];    /* right-anchor ordinal inside the PRB per RE (T:325) */
  float alpha[CE_MAX_HOPS][CE_MAX_CDM][12];      /* interpolation weight per RE (T:333-337) */
  double rc[31];                                  /* raised-cosine taps, unit sum (T:184-234) */
  double sst[CE_MAX_SYMBOLS];                     /* symbolStartTime (T:809-820) */
  double two_pi_nsamples[CE_MAX_HOPS];            /* 2*pi*nSamples (T:418-426) */
  double n_pilots, noise_den;                     /* T:901-915 */
  float mmse_w[2][CE_MMSE_BLOCK][CE_MMSE_BLOCK];  /* CE_SMOOTH_MMSE: Re / Im of W[m][k] */
} ce_plan_host_view;

/* Same validation and float64 derivation as ce_plan_create but touches no GPU: usable on a CPU-only host. */
int ce_plan_derive_host(const ce_plan_desc* desc, ce_plan_host_view* view);

/* Validates the descriptor, derives every per-plan table on the host in float64 (pilot RE index
 * lists T:571-576, symbol start times T:809-820, CFO sample span T:418-426, RC taps T:184-234,
 * interpolation anchors T:311-338, IFFT twiddles) and uploads them.  Synchronous; not for the
 * per-slot hot loop.  Replaces the per-call re-derivation the reference does inside process_hop. */
int ce_plan_create(const ce_plan_desc* desc, ce_plan** out);
void ce_plan_destroy(ce_plan* plan);
int ce_plan_get_info(const ce_plan* plan, ce_plan_info* info);

/*
 * Estimate n_slots x n_ports work items in ONE kernel launch on `stream` (a hipStream_t; NULL =
 * the default stream).  Asynchronous: returns after the launch is enqueued.
 *
 *  rx            complex64 received grids, element (slot b, port r, subcarrier k, symbol s) at
 *                rx[b*rx_strides[0] + r*rx_strides[1] + k*rx_strides[2] + s*rx_strides[3]] (strides in
 *                complex elements).  Any layout works; [slot][port][symbol][subcarrier] (rx_strides[2]==1)
 *                gives coalesced pilot loads.  Replaces received_rg of T:745 (one (n_sc,n_sym) grid per item).
 *  pilots        complex64 DM-RS symbols, element (slot b, re k, dmrs symbol s, layer l) at
 *                pilots[b*pil_strides[0] + k*pil_strides[1] + s*pil_strides[2] + l*pil_strides[3]];
 *                pil_strides[0]==0 shares one pilot set across the batch.  Axis order of T:760.
 *                The Rx ports of a slot always share pilots.
 *  ch_est        out, complex64, dense [slot][port][subcarrier][symbol][layer] (the reference's
 *                (n_sc,n_sym,L) layout per item, T:766); EVERY element is written (zeros outside hops).
 *  noise,rsrp,epre,ta,cfo_hz   out, float64 [slot][port] each (T:767-771); cfo_hz = NaN when not estimated.
 */
int ce_estimate_batch(const ce_plan* plan, const void* rx, const int64_t rx_strides[4], const void* pilots,
                      const int64_t pil_strides[4], int64_t n_slots, int32_t n_ports, void* ch_est,
                      double* noise, double* rsrp, double* epre, double* ta, double* cfo_hz, void* stream);

/* Diagnostic form of ce_estimate_batch for per-stage tests: same launch, same results, and in
 * addition the intermediates of process_hop the fused kernel otherwise never materialises:
 *  stage_estimates  out, complex64 [slot][port][2][n_hops][n_layers][n_re]: the pilot-RE channel estimate of each hop
 *                   after S5/S6 (LS, DM-RS average, CDM de-spread; T:593-628) and after S7 (frequency smoothing, T:633-668)
 *  stage_scalars    out, float64 [slot][port][n_hops][2]: the hop's CFO normalised to the SCS (T:426; not written when
 *                   the hop has one DM-RS symbol) and the signed arg-max bin of its time alignment (T:686-696)
 * n_re / n_hops as in ce_plan_info / the descriptor.  Not for the hot loop (extra ~16 B per pilot RE of stores). */
int ce_estimate_batch_stages(const ce_plan* plan, const void* rx, const int64_t rx_strides[4], const void* pilots,
                             const int64_t pil_strides[4], int64_t n_slots, int32_t n_ports, void* ch_est,
                             double* noise, double* rsrp, double* epre, double* ta, double* cfo_hz,
                             void* stage_estimates, double* stage_scalars, void* stream);

/* Times `iters` back-to-back launches of ce_estimate_batch with HIP events recorded on `stream`
 * (after `warmup` untimed ones); *avg_ms = mean kernel-launch duration.  Used by bench.py for the
 * roofline line.  Synchronous. */
int ce_time_batch(const ce_plan* plan, const void* rx, const int64_t rx_strides[4], const void* pilots,
                  const int64_t pil_strides[4], int64_t n_slots, int32_t n_ports, void* ch_est, double* noise,
                  double* rsrp, double* epre, double* ta, double* cfo_hz, void* stream, int32_t warmup,
                  int32_t iters, double* avg_ms);

/*
 * Tuning knobs.  libce_hip.so reads NO environment variable.  The diagnostic build of the same sources
 * (csrc/libce_hip_knobs.so, compiled with -DCE_TUNING_KNOBS; loaded by tests/test_hip_tiers.py and the dev tools through
 * CE_HIP_LIB=<path>, which srsran_ce_pytorch_amd/_lib.py honours) reads these when a plan is created -- each selects an
 * alternative kernel path of the same arithmetic, for A/B timing and for testing the tiers against each other:
 *   CE_FORCE_GENERIC   re-read path (ND = 0) instead of the register path
 *   CE_FORCE_WIDE      widest register tier (KPT = 7) instead of the band's own
 *   CE_TA_FULL         full first radix-16 pass of the time-alignment transform instead of the collapsed narrow-band one
 *   CE_TA_LP1          one time-alignment transform at a time (no layer- / hop-parallel form)
 *   CE_NO_PIL_STASH    DM-RS symbols re-read per stage instead of parked in the LDS
 *   CE_CNN_GENERAL     ce_dl_cnn in-painting always iterated (no closed forms)
 *   CE_LDS_PAD_BYTES   extra dynamic LDS per workgroup (lowers the workgroups resident per CU)
 *   CE_NO_LDS_BIG      no 80 KB LDS request for large launches of the wide none / mean kernel
 *   CE_NO_NARROW       narrow allocations on the workgroup-per-item kernels instead of the wave-per-item kernel
 *   CE_FORCE_NARROW    the wave-per-item kernel for every plan it covers, also where the policy prefers the other kernels
 * Knobs are not part of the host-side plan-cache key: set them before the process creates its first plan.
 */
const char* ce_last_error(void);
int ce_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CE_HIP_H */

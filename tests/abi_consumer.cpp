// A host with no Python and no torch: reads one case (descriptor + received grid + pilots) from a file, calls the C ABI of
// libce_hip.so (include/ce_hip.h) with buffers from hipMalloc on the null stream, writes the six outputs to a file.
// tests/test_abi_consumer.py (GPU box) builds it with hipcc, feeds it fixtures and compares with the Python path bit for bit.
//   hipcc -O1 -Iinclude tests/abi_consumer.cpp -Lsrsran_ce_pytorch_amd/csrc -lce_hip -Wl,-rpath,$PWD/srsran_ce_pytorch_amd/csrc -o /tmp/abi_consumer
//   /tmp/abi_consumer in.bin out.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ce_hip.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: abi_consumer in.bin out.bin\n"); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  // header: 12 x int32 {n_prb_grid, n_sym, n_layers, n_hops, smoothing, cfo_compensate, interp, n_slots, n_ports, n_re, n_dmrs_total, rx_sym_major},
  // 17 x float64 {scs_hz, beta, cnn_alpha, cp_ms[14]}, then per hop {dmrs_symbols[14] u8, re_mask[2] u16, prb_start, n_prbs, start_symbol,
  // n_alloc (int32), mask_prbs[n_prb_grid] u8}, then rx and pilots as complex64
  int32_t h[12];
  double d[17];
  if (!rd(f, h, sizeof h) || !rd(f, d, sizeof d)) { fprintf(stderr, "short header\n"); return 1; }
  ce_plan_desc desc;
  memset(&desc, 0, sizeof desc);
  desc.abi_version = CE_ABI_VERSION;
  desc.device = 0;
  desc.n_prb_grid = h[0]; desc.n_sym = h[1]; desc.n_layers = h[2]; desc.n_hops = h[3];
  desc.smoothing = h[4]; desc.cfo_compensate = h[5]; desc.interp = h[6];
  const int64_t B = h[7], R = h[8], n_re = h[9], n_dm = h[10];
  const bool sym_major = h[11] != 0;
  desc.scs_hz = d[0]; desc.beta_dmrs = d[1]; desc.cnn_smoothing_alpha = d[2];
  for (int i = 0; i < CE_MAX_SYMBOLS; ++i) desc.cp_ms[i] = d[3 + i];
  std::vector<std::vector<uint8_t>> masks(desc.n_hops, std::vector<uint8_t>(desc.n_prb_grid));
  for (int k = 0; k < desc.n_hops; ++k) {
    int32_t q[4];
    if (!rd(f, desc.hop[k].dmrs_symbols, CE_MAX_SYMBOLS) || !rd(f, desc.hop[k].re_mask, sizeof desc.hop[k].re_mask) || !rd(f, q, sizeof q) ||
        !rd(f, masks[k].data(), masks[k].size())) { fprintf(stderr, "short hop %d\n", k); return 1; }
    desc.hop[k].prb_start = q[0]; desc.hop[k].n_prbs = q[1]; desc.hop[k].start_symbol = q[2]; desc.hop[k].n_alloc_symbols = q[3];
    desc.hop[k].mask_prbs = masks[k].data();
  }
  const int64_t n_sc = 12 * (int64_t)desc.n_prb_grid, n_sym = desc.n_sym, L = desc.n_layers;
  const size_t rx_n = (size_t)(B * R * n_sc * n_sym), pil_n = (size_t)(n_re * n_dm * L), ch_n = rx_n * (size_t)L, items = (size_t)(B * R);
  std::vector<float> rx(2 * rx_n), pil(2 * pil_n);
  if (!rd(f, rx.data(), rx.size() * 4) || !rd(f, pil.data(), pil.size() * 4)) { fprintf(stderr, "short payload\n"); return 1; }
  fclose(f);

  ce_plan* plan = nullptr;
  int rc = ce_plan_create(&desc, &plan);
  if (rc != CE_OK) { fprintf(stderr, "ce_plan_create: %d: %s\n", rc, ce_last_error()); return 3; }
  ce_plan_info info;
  ce_plan_get_info(plan, &info);
  if (info.n_re != n_re || info.n_dmrs_total != n_dm) { fprintf(stderr, "plan says n_re %d n_dmrs %d, file %ld %ld\n", info.n_re, info.n_dmrs_total, (long)n_re, (long)n_dm); return 3; }

  void *d_rx, *d_pil, *d_ch;
  double* d_sc;
  HIPCHECK(hipMalloc(&d_rx, rx.size() * 4));
  HIPCHECK(hipMalloc(&d_pil, pil.size() * 4));
  HIPCHECK(hipMalloc(&d_ch, ch_n * 8));
  HIPCHECK(hipMalloc((void**)&d_sc, 5 * items * sizeof(double)));
  HIPCHECK(hipMemcpy(d_rx, rx.data(), rx.size() * 4, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(d_pil, pil.data(), pil.size() * 4, hipMemcpyHostToDevice));
  // element strides of the logical [slot][port][subcarrier][symbol] grid and [slot][re][symbol][layer] pilots (one pilot set for all slots: stride 0)
  const int64_t rs[4] = {R * n_sc * n_sym, n_sc * n_sym, sym_major ? 1 : n_sym, sym_major ? n_sc : 1};
  const int64_t ps[4] = {0, n_dm * L, L, 1};
  rc = ce_estimate_batch(plan, d_rx, rs, d_pil, ps, B, (int32_t)R, d_ch, d_sc, d_sc + items, d_sc + 2 * items, d_sc + 3 * items, d_sc + 4 * items, nullptr);
  if (rc != CE_OK) { fprintf(stderr, "ce_estimate_batch: %d: %s\n", rc, ce_last_error()); return 4; }
  HIPCHECK(hipDeviceSynchronize());
  std::vector<float> ch(2 * ch_n);
  std::vector<double> sc(5 * items);
  HIPCHECK(hipMemcpy(ch.data(), d_ch, ch.size() * 4, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(sc.data(), d_sc, sc.size() * 8, hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 1; }
  const int32_t cfo_estimated = info.cfo_estimated;
  fwrite(&cfo_estimated, 4, 1, o);
  fwrite(ch.data(), 4, ch.size(), o);
  fwrite(sc.data(), 8, sc.size(), o);
  fclose(o);
  ce_plan_destroy(plan);
  (void)hipFree(d_rx); (void)hipFree(d_pil); (void)hipFree(d_ch); (void)hipFree(d_sc);
  printf("abi_consumer: %ld slots x %ld ports, %ld subcarriers x %ld symbols x %ld layers, LDS %d B per workgroup: ok\n", (long)B, (long)R, (long)n_sc, (long)n_sym, (long)L, info.lds_bytes);
  return 0;
}

// Instantiation unit: the wave-per-item kernel for narrow allocations (ce_narrow_kernel.h), 1-4 layers x 1-2 hops.
#include "ce_narrow_kernel.h"

int ce_tu_narrow(int op, int key, const CeLaunchCtx& c) {
  switch (key) {
#define CE_NRW(L, NH) case L * 10 + NH: return narrow_run_t<L, NH>(op, c);
    CE_NRW(1, 1) CE_NRW(2, 1) CE_NRW(3, 1) CE_NRW(4, 1)
    CE_NRW(1, 2) CE_NRW(2, 2) CE_NRW(3, 2) CE_NRW(4, 2)
    default: return -1;
  }
}

// Instantiation unit: register-path kernels, 1 hop(s), feature set 1, the two narrow band tiers (ce_inst.inc).
#define CE_TU_NAME ce_tu_reg_h1_f1
#define CE_TU_NH 1
#define CE_TU_FEAT 1
#define CE_TU_KSEL 2
#include "ce_inst.inc"

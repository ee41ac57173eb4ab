// Instantiation unit: re-read (generic) kernels, 2 hop(s), every feature set (ce_inst.inc).
#define CE_TU_NAME ce_tu_gen_h2
#define CE_TU_NH 2
#define CE_TU_FEAT -1
#include "ce_inst.inc"

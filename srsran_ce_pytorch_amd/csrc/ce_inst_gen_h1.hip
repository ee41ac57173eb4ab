// Instantiation unit: re-read (generic) kernels, 1 hop(s), every feature set (ce_inst.inc).
#define CE_TU_NAME ce_tu_gen_h1
#define CE_TU_NH 1
#define CE_TU_FEAT -1
#include "ce_inst.inc"

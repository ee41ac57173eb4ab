// Instantiation unit: register-path kernels, 1 hop(s), feature set 1, the two wide band tiers (ce_inst.inc) -- compiled with
// the max-ILP scheduling strategy (_lib.py: EXTRA_FLAGS): measured in process 2-3 % faster on these shapes, the headline's among
// them, and up to 45 % slower on the narrow tiers, which is why the unit is split (profiles/round2_budget_policy_ab.txt, 9).
#define CE_TU_NAME ce_tu_reg_h1_f1w
#define CE_TU_NH 1
#define CE_TU_FEAT 1
#define CE_TU_KSEL 1
#include "ce_inst.inc"

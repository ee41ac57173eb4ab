#!/bin/bash
# Host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (build container, no GPU: GPU sanitizers are not available on the
# pool).  ce_api.hip's host code -- plan validation, the derived tables (pilot index lists, TA scatter maps, RC taps, LDS layouts) -- is rebuilt with
# -fsanitize=address,undefined -fno-gpu-sanitize and linked with the shipped kernel objects; then tests/test_host_and_abi.py and ce_plan_derive_host
# over 30 000 random plans (15 000 from the fuzzer's --wide draw, 15 000 from the suite's draw) run against it.
#     tools/host_sanitizers.sh        (needs a built tree: python -c "import __graft_entry__ as g; g.build()")
set -e
cd "$(dirname "$0")/.."
C=srsran_ce_pytorch_amd/csrc; B=$C/.build
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Iinclude -I$C -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -c $C/ce_api.hip -o /tmp/ce_api_asan.o
hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan -o /tmp/libce_asan.so /tmp/ce_api_asan.o $(ls $B/*.o | grep -v "ce_api")
export LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 CE_HIP_LIB=/tmp/libce_asan.so
python -m pytest tests/test_host_and_abi.py -q -x
cd tests && python - <<'PY'
import sys, collections
sys.path[:0] = ['..', '../oracle']
import numpy as np
import fuzz_cases as F
from srsran_ce_pytorch_amd import estimator as E, _lib
print("library:", _lib.load()._name)
cnt = collections.Counter()
for wide in (True, False):
    for i in range(15000):
        rng = np.random.default_rng([4242 + wide, i])
        case, extras = F.draw(rng, 275 if wide else 273, wide)
        b = F.realize(case, extras, 1)
        try:
            v = E.derive_host(b.hop1, b.hop2, b.config, b.beta, case['n_layers'], case['n_prb_grid'], case['n_sym'], interp=extras['interp'])
            cnt['plan for the wave-per-item kernel' if v.narrow else 'plan for a workgroup kernel'] += 1
        except Exception as e:      # the reference's own refusals (AssertionError) and the mmse extension's limits (NotImplementedError)
            cnt[type(e).__name__] += 1
print(dict(cnt), "-- no sanitizer report")
PY

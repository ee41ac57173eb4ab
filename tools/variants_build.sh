#!/bin/bash
# usage: tools/variants_build.sh <out.so> <extra flags...>   (dev helper for same-box A/B of whole libraries)
out=$1; shift
python - "$out" "$@" <<'PY'
import sys
sys.path.insert(0, ".")
from srsran_ce_pytorch_amd import _lib
_lib.build(force=True, extra_flags=sys.argv[2:], out=sys.argv[1])
PY

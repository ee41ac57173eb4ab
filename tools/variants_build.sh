#!/bin/bash
# usage: tools/variants_build.sh <out.so> <extra flags...>   (dev helper for same-box A/B of whole libraries)
out=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Isrsran_ce_pytorch_amd/csrc "$@" -o "$out" srsran_ce_pytorch_amd/csrc/ce_api.hip srsran_ce_pytorch_amd/csrc/ce_denoise.hip srsran_ce_pytorch_amd/csrc/ce_kernels.hip 2>/dev/null

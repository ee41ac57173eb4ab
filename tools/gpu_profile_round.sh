#!/bin/bash
# One gpurun lease: the headline bench, the access-pattern micro-benchmark on the SAME box, rocprofv3 kernel trace of the
# bench, PMC passes of the bench (FETCH / WRITE) and of the geometry set (trace, FETCH, WRITE, SQ).  Everything under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/rwmix tools/micro/rwmix.hip 2>/dev/null && /tmp/rwmix > $O/r2_rwmix.log 2>&1
python3 bench.py > $O/r2_bench_same_lease.log 2> $O/r2_bench_same_lease.err || exit 1
rocprofv3 --kernel-trace --stats --kernel-include-regex ce_estimate --output-format csv -d $O/r2_bench_trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/r2_bench_trace.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex ce_estimate --pmc FETCH_SIZE --output-format csv -d $O/r2_bench_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/r2_bench_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex ce_estimate --pmc WRITE_SIZE --output-format csv -d $O/r2_bench_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/r2_bench_write.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex ce_estimate --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r2_bench_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/r2_bench_sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --kernel-include-regex ce_estimate --output-format csv -d $O/r2_geo_trace -- python3 tools/prof_geometries.py > $O/r2_geo_trace.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex ce_estimate --pmc FETCH_SIZE --output-format csv -d $O/r2_geo_fetch -- python3 tools/prof_geometries.py > $O/r2_geo_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex ce_estimate --pmc WRITE_SIZE --output-format csv -d $O/r2_geo_write -- python3 tools/prof_geometries.py > $O/r2_geo_write.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex ce_estimate --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r2_geo_sq -- python3 tools/prof_geometries.py > $O/r2_geo_sq.log 2>&1 || exit 1
python3 tools/distill_round2.py > $O/r2_distill.log 2>&1
python3 tools/distill_geometry_counters.py round2 > $O/r2_distill_geo.log 2>&1
du -sh $O | tail -1
echo profile round done

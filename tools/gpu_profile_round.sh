#!/bin/bash
# One gpurun lease: the headline bench, the access-pattern micro-benchmark on the SAME box, rocprofv3 kernel trace of the
# bench, PMC passes of the bench (FETCH / WRITE / SQ) and of the geometry set (trace, FETCH, WRITE, SQ).  Everything under
# gpurun_out/; the distillers write gpurun_out/distilled/ (copy into profiles/).
#     tools/gpu_profile_round.sh <tag, e.g. round3> <prefix, e.g. r3> <git head> [bench|geo|all]
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-round3}; P=${2:-r3}; HEAD=${3:-unknown}; WHAT=${4:-all}
O=gpurun_out
mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
SQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
if [ "$WHAT" != geo ]; then
hipcc --offload-arch=gfx950 -O3 -o /tmp/rwmix tools/micro/rwmix.hip 2>/dev/null && /tmp/rwmix > $O/${P}_rwmix.log 2>&1
python3 bench.py > $O/${P}_bench_same_lease.log 2> $O/${P}_bench_same_lease.err || exit 1
rocprofv3 --kernel-trace --stats --kernel-include-regex "ce_estimate|ce_narrow" --output-format csv -d $O/${P}_bench_trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/${P}_bench_trace.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex "ce_estimate|ce_narrow" --pmc FETCH_SIZE --output-format csv -d $O/${P}_bench_fetch -- $B > $O/${P}_bench_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex "ce_estimate|ce_narrow" --pmc WRITE_SIZE --output-format csv -d $O/${P}_bench_write -- $B > $O/${P}_bench_write.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex "ce_estimate|ce_narrow" --pmc $SQ --output-format csv -d $O/${P}_bench_sq -- $B > $O/${P}_bench_sq.log 2>&1 || exit 1
python3 tools/distill_round.py $TAG $P $HEAD > $O/${P}_distill.log 2>&1
fi
if [ "$WHAT" != bench ]; then
rocprofv3 --kernel-trace --kernel-include-regex "ce_estimate|ce_narrow" --output-format csv -d $O/${P}_geo_trace -- python3 tools/prof_geometries.py > $O/${P}_geo_trace.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex "ce_estimate|ce_narrow" --pmc FETCH_SIZE --output-format csv -d $O/${P}_geo_fetch -- python3 tools/prof_geometries.py > $O/${P}_geo_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex "ce_estimate|ce_narrow" --pmc WRITE_SIZE --output-format csv -d $O/${P}_geo_write -- python3 tools/prof_geometries.py > $O/${P}_geo_write.log 2>&1 || exit 1
rocprofv3 --kernel-include-regex "ce_estimate|ce_narrow" --pmc $SQ --output-format csv -d $O/${P}_geo_sq -- python3 tools/prof_geometries.py > $O/${P}_geo_sq.log 2>&1 || exit 1
python3 tools/distill_geometry_counters.py $TAG $P > $O/${P}_distill_geo.log 2>&1
fi
du -sh $O | tail -1
echo profile round done

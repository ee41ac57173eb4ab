#!/bin/bash
# instruction-cache and memory-latency counters of the headline kernel and of the access-pattern
# micro-benchmark (rwmix) on ONE lease.  Each --pmc pass is its own rocprofv3 run (never combined with tracing).
#     tools/headline_counters.sh <prefix, e.g. r3>
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
P=${1:-r3}; O=gpurun_out; mkdir -p $O
rocprofv3 -L > $O/${P}_list_avail.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/rwmix tools/micro/rwmix.hip 2>/dev/null
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
i=0
while read -r set; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-include-regex ce_estimate --pmc $set --output-format csv -d $O/${P}_hc_bench_$i -- $B > $O/${P}_hc_bench_$i.log 2>&1; echo "bench set $i ($set): rc=$?"
  timeout -k 10 300 rocprofv3 --kernel-include-regex rwmix --pmc $set --output-format csv -d $O/${P}_hc_rwmix_$i -- /tmp/rwmix > $O/${P}_hc_rwmix_$i.log 2>&1; echo "rwmix set $i: rc=$?"
done <<SETS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
SETS
echo counters done

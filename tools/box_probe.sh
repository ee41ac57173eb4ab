#!/bin/bash
# Why do the boxes of the pool differ by up to 10 %?  One gpurun call: rocm-smi's clocks / power / temperature sampled twice a second
# (sysfs reads, no HIP) while the hardware-only access-pattern micro-benchmark (tools/micro/rwmix.hip) and the headline bench run.
#     tools/box_probe.sh <tag>        -> gpurun_out/<tag>_box_probe.txt
cd "$(dirname "$0")/.."
O=gpurun_out; T=${1:-box}
mkdir -p $O
{
echo "## idle"; rocm-smi --showclocks --showpower --showmaxpower --showtemp --showperflevel --showvbios --showmemvendor 2>&1 | grep -v "^=\|^$" | head -60
hipcc --offload-arch=gfx950 -O3 -o /tmp/rwmix tools/micro/rwmix.hip 2>/dev/null
( while true; do rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|power\|junction\|memory (c)\|hbm" | tr '\n' ';'; echo; sleep 0.2; done ) > $O/${T}_smi_samples.txt 2>&1 &
SAMPLER=$!
echo "## rwmix"; /tmp/rwmix 2>&1 | head -12
echo "## bench"; python3 bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline ms_per_step', l['ms_per_step'], 'kernel_ms', l['roofline']['kernel_ms'])"
kill $SAMPLER
echo "## under load (samples while rwmix + bench ran)"
for k in "sclk clock level" "mclk clock level" "fclk clock level" "Power (W)" "junction) (C)" "(Sensor HBM"; do echo "$k:" $(grep -o "$k[^;]*" $O/${T}_smi_samples.txt | sed "s/.*: //" | sort | uniq -c | sort -k2 -n | awk '{printf "%s x%s  ", $2 $3, $1}'); done
} > $O/${T}_box_probe.txt 2>&1
tail -30 $O/${T}_box_probe.txt

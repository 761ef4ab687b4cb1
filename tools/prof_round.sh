#!/bin/bash
# Everything that goes into profiles/ for a round, in one call on the GPU box: tools/prof_round.sh <tag>
#   headline: rocprofv3 kernel stats + PMC passes (tools/prof_pmc.sh)       -> gpurun_out/prof_<tag>
#   FIR / Costas / estimate chain: kernel stats (+ PMC for FIR)             -> gpurun_out/prof_<tag>_fir, _costas, est_<tag>
#   deviation / dtype sweeps, VALU microbenchmark                           -> gpurun_out/<tag>_*.txt
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
tools/prof_pmc.sh $TAG > /dev/null 2>&1
tools/prof_cmd.sh ${TAG}_fir python $R/tools/fir_only.py > /dev/null 2>&1
tools/prof_cmd.sh ${TAG}_costas python $R/tools/costas_only.py > /dev/null 2>&1
tools/prof_estimate.sh $TAG > gpurun_out/${TAG}_estimate.txt 2>&1
python tools/deviation_probe.py 2>/dev/null | grep deviation > gpurun_out/${TAG}_deviation.txt
python tools/dtype_probe.py 2>/dev/null | grep "ms/step" > gpurun_out/${TAG}_dtypes.txt
(cd tools/kbench && ([ -x vbench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 vbench.hip -o vbench) && timeout 120 ./vbench) > gpurun_out/${TAG}_vbench.txt 2>&1
python bench.py --steps 50 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
ls gpurun_out | head -40
# the default command (pipelined passes): kernel stats of what the driver's bench line times
OUTP=$R/gpurun_out/stats_${TAG}_pipelined; mkdir -p $OUTP
(cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUTP -o b -- python $R/bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-extra --no-d2h --no-reference-loop > $OUTP/log.txt 2>&1)

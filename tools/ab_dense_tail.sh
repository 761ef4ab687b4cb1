#!/bin/bash
# tools/ab_dense_tail.sh <tag>: the product library against an A/B build of it (python -m urh_amd.build -DURH_<KNOB>=0 --tag <tag> ->
# urh_amd/liburhgpu_<tag>.so), alternating on one box: pipelined steps over the 10-samples-per-symbol capture (with and without the kernel
# the knob touches: tools/sps10_skips.py), over the headline capture, and the bits-only / int16 / int8 steps.  Knobs of round 6's end:
# URH_EXPAND_PREFETCH, URH_PACK_WORDS, URH_EXPAND_EARLY_EXIT (pulse_table.hip); records: profiles/r06fin_*_ab.txt
TAG=${1:?tag of the A/B library}
cd ${GRAFT_REPO_ROOT:-$(pwd)}
R=$(pwd)
for rep in 1 2; do
for lib in liburhgpu.so liburhgpu_$TAG.so; do
  echo "## $lib sps10"; URHGPU_LIB=$R/urh_amd/$lib timeout 200 python tools/sps10_skips.py 2>&1 | grep -E "product|expand|pack"
  echo "## $lib sps100"; URH_SPS=100 URHGPU_LIB=$R/urh_amd/$lib timeout 200 python tools/sps10_skips.py 2>&1 | grep -E "product"
done; done
for lib in liburhgpu.so liburhgpu_$TAG.so liburhgpu.so liburhgpu_$TAG.so; do echo "## $lib variants"; URHGPU_LIB=$R/urh_amd/$lib timeout 200 python tools/dtype_mask_sweep.py 4 2>&1 | grep -v "amdgpu.ids\|^#"; done

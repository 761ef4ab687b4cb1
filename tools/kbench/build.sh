#!/bin/bash
# Build kbench variants: kbench_b<KBATCH>w<MINWAVES>
set -e
cd "$(dirname "$0")"
FLAGS="$XFLAGS --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I../../urh_amd/csrc -Wno-unused-function"
for v in "$@"; do
  b=${v%%w*}; b=${b#b}; w=${v##*w}; w=${w%%_*}
  /opt/rocm/bin/hipcc $FLAGS -DURH_KBATCH=$b -DURH_MINWAVES=$w kbench.hip -o kbench_$v$SUFFIX &
done
wait
ls -la kbench_*

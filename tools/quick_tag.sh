#!/bin/bash
# tools/quick_tag.sh TAG file.hip [-DURH_X=1 ...]: an A/B library that differs from the default build in ONE source file's flags --
# compiles only that file and links it with the default build's objects (python -m urh_amd.build first) -> urh_amd/liburhgpu_TAG.so
TAG=$1; SRC=$2; shift; shift
R=$(cd $(dirname $0)/.. && pwd); C=$R/urh_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -Wall -Wno-unused-function "$@" -c $C/$SRC -o $C/${SRC%.hip}_$TAG.o || exit 1
OBJS=""
for f in $(python3 -c "import sys; sys.path.insert(0,'$R'); from urh_amd.build import SOURCES; print(' '.join(SOURCES))"); do
  if [ $f = $SRC ]; then OBJS="$OBJS $C/${SRC%.hip}_$TAG.o"; else OBJS="$OBJS $C/${f%.hip}.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $R/urh_amd/liburhgpu_$TAG.so && echo urh_amd/liburhgpu_$TAG.so

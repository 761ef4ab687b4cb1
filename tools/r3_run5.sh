#!/bin/bash
mkdir -p gpurun_out/r3e
bash tools/r3_ab.sh gpurun_out/r3e/ab.txt 2 default cap80 default:URH_HOT_LDS_KB=24 cap80:URH_HOT_LDS_KB=24 default:URH_HOT_LDS_KB=28
URHGPU_LIB=$(pwd)/urh_amd/liburhgpu_cap80.so bash tools/r3_prof.sh r3e/prof_cap80
URH_HOT_LDS_KB=24 bash tools/r3_prof.sh r3e/prof_lds24

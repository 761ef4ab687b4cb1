#!/bin/bash
mkdir -p gpurun_out/r3g
bash tools/r3_ab.sh gpurun_out/r3g/ab.txt 2 default pack64 pack2k cpw8 cpw8p64 pack64:URH_HOT_LDS_KB=24 default:URH_HOT_LDS_KB=33
bash tools/r3_prof.sh r3g/prof_stream --no-device-loop

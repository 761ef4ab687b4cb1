#!/bin/bash
# A/B of estimator builds on the GPU box: tools/hist_ab.sh TAG [TAG ...] -- each TAG names urh_amd/liburhgpu_TAG.so (python -m urh_amd.build --tag);
# prints est_probe's wall times (config 3's estimate on 1 GiB OOK, detect_center on 1 GiB demodulated PSK) for the default build and each TAG
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for round in 1 2; do
for tag in "" "$@"; do
  lib=$R/urh_amd/liburhgpu${tag:+_$tag}.so
  echo "${tag:-default}: $(URHGPU_LIB=$lib python tools/est_probe.py 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('estimate', round(d['estimate_ms'],4), 'centers+plateaus', d['estimate_stages_ms']['centers_and_plateaus_ms'], 'detect_center', round(d['detect_center_ms'],4), d['center'], d['estimate']['center'], d['estimate']['bit_length'])")"
done
done

#!/bin/bash
# A/B of one tuning key (urhgpu_ctx_set_tuning, include/urhgpu.h) or one build flag on the default bench line (GPU box):
#   tools/ab.sh key=VALUE [key=VALUE ...]          e.g.  tools/ab.sh stream_policy=0 hot_cus_removed_per_xcd=6
#   tools/ab.sh --build TAG -DURH_X=1 [...]        builds urh_amd/liburhgpu_TAG.so first (python -m urh_amd.build --tag) and runs it against the default
# prints ms_per_step / kernel ms / single-capture ms of the default run and of the variant, three alternating runs each
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
ENVV=""; LIB=""
if [ "$1" = "--build" ]; then
  TAG=$2; shift; shift
  python -m urh_amd.build --tag $TAG "$@" > /dev/null || exit 1
  LIB="URHGPU_LIB=$R/urh_amd/liburhgpu_$TAG.so"
else
  for kv in "$@"; do k=${kv%%=*}; v=${kv#*=}; ENVV="$ENVV URH_TUNE_$(echo $k | tr a-z A-Z)=$v"; done
fi
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print(d['ms_per_step'], d['roofline']['kernel_ms'], c.get('single_capture_incl_compact_d2h_ms'), c.get('ms_per_step_with_device_positions'))"; }
for i in 1 2 3; do
  echo "default : $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-upload --no-pmc --no-variants 2>/dev/null | line)"
  echo "variant : $(env $ENVV $LIB python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-upload --no-pmc --no-variants 2>/dev/null | line)   [$ENVV $LIB]"
done

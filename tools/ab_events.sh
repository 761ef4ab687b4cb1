for rep in 1 2 3; do
for t in new bracket prev; do
  case $t in new) L="X=1";; bracket) L="URH_PROFILE_BRACKET=1";; prev) L="URHGPU_LIB=/root/repo/urh_amd/liburhgpu_prev.so";; esac
  env $L python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done; done

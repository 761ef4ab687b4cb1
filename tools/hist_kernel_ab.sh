#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of est_probe's PSK and OOK parts for the default build and each TAG: tools/hist_kernel_ab.sh REGEX TAG...
RX=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
for tag in "" "$@"; do
  lib=$R/urh_amd/liburhgpu${tag:+_$tag}.so
  for part in psk ook; do
    other=$([ $part = ook ] && echo --no-psk || echo --no-ook)
    D=/tmp/hk_${tag:-default}_$part; rm -rf $D
    URHGPU_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o e -- python $R/tools/est_probe.py $other > $D.log 2>&1
    f=$(find $D -name "*kernel_stats.csv" | head -1)
    echo "${tag:-default} $part: $(python3 -c "
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']): print(r['Name'].replace('urh::', '').replace('void ', '')[:24], round(float(r['AverageNs']) / 1e3, 1), end='  ')
" $f "$RX")"
  done
done

#!/bin/bash
# tools/kstats.sh <file.hip> [filter-regex] [extra hipcc flags...]: register / scratch statistics of every kernel of one source file
# (hipcc cross-compiles for gfx950 without a GPU).  Columns: VGPRs SGPRs spilled-VGPRs scratch-bytes LDS-bytes name
SRC=$1; FILTER=${2:-.}; shift; shift
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math --offload-device-only -c "$SRC" -o $T/k.bundle "$@" || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    if re.search(sys.argv[1], name):
        print(g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), name[:150])
" "$FILTER"
[ -n "$KEEP" ] && cp $T/k.co "$KEEP"
rm -rf $T

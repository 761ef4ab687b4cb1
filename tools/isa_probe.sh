#!/bin/bash
# tools/isa_probe.sh [instantiation] : gfx950 ISA of ONE instantiation of the hot kernel (default: the headline's
# k_demod_runs_bp<SRC_IQ, F32, FSK, qad, runs, one plane>), compiled from the kernel half of demod_runs.hip alone (seconds instead of
# minutes; no GPU needed), with a per-basic-block VALU / SALU / memory instruction count.  Output: /tmp/isa/probe.s
INST=${1:-"k_demod_runs_bp<0, 4, 1, true, true, 1, false>"}
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/isa; mkdir -p $OUT
awk '/^\/\/ ---- host-side launchers/{exit} {print}' $HERE/urh_amd/csrc/demod_runs.hip > $OUT/probe.hip
cat >> $OUT/probe.hip <<EOT
bool g_force_state_bytes = false; bool g_stamp_probe = false; thread_local HotEvents g_hot_events;
template __global__ void $INST(const RunArgs);
}
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math --offload-device-only -S \
  -I$HERE/urh_amd/csrc -I$HERE/include $OUT/probe.hip -o $OUT/probe.s "${@:2}" 2>&1 | grep -v "hip-link" 
python3 - <<'EOT'
import re
txt = open('/tmp/isa/probe.s').read()
m = re.search(r'\n(_ZN3urh15k_demod_runs_bp\w+):.*?s_endpgm', txt, re.S)
body = m.group(0)
for k in ('vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size'):
    print(k, re.findall(r'\.' + k + r':\s+(\d+)', txt))
blocks, cur = [], None
for l in body.split('\n'):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm or cur is None:
        cur = [mm.group(1) if mm else 'entry', 0, 0, 0, 0]; blocks.append(cur)
        if mm: continue
    s = l.strip()
    if not s or s[0] in ';.': continue
    op = s.split()[0]
    if op.startswith('v_'): cur[1] += 1
    elif op == 's_nop': cur[4] += 1
    elif op.startswith('s_'): cur[2] += 1
    else: cur[3] += 1
print('blocks with >= 20 VALU: label valu salu mem nop')
for b in blocks:
    if b[1] >= 20: print('  ', *b)
print('static totals: VALU', sum(b[1] for b in blocks), 'v_mov', len(re.findall(r'\n\s+v_mov_b32', body)), 'v_cndmask', len(re.findall(r'v_cndmask', body)), 's_nop', len(re.findall(r's_nop', body)), 'v_pk', len(re.findall(r'\n\s+v_pk_', body)))
EOT

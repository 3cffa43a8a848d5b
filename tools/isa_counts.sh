#!/bin/bash
# static instruction counts of kernels of obj/har_kernels.o whose demangled name contains one of the given substrings
# usage: tools/isa_counts.sh 'k_trace_closest<2, false, false>' 'k_resolve<0, 2, false, false>' ...
set -e
L=/opt/rocm/lib/llvm/bin; D=$(mktemp -d); O=${HAR_OBJ:-$(dirname $0)/../mitsuba3_amd/csrc/obj/har_kernels.o}
$L/llvm-objcopy --dump-section .hip_fatbin=$D/fat.bin $O
T=$($L/clang-offload-bundler --list --type=o --input=$D/fat.bin | grep gfx950)
$L/clang-offload-bundler --unbundle --type=o --targets=$T --input=$D/fat.bin --output=$D/k.co
$L/llvm-objdump -d $D/k.co > $D/k.s
python3 - "$D/k.s" "$@" <<'PY'
import re, subprocess, collections, sys
txt = open(sys.argv[1]).read(); want = sys.argv[2:]
for b in re.split(r'\n(?=[0-9a-f]{16} <)', txt):
    m = re.match(r'[0-9a-f]{16} <(\S+)>:', b)
    if not m: continue
    name = subprocess.check_output(['c++filt', m.group(1)]).decode().strip()
    if not any(w in name for w in want): continue
    c = collections.Counter()
    for line in b.split('\n'):
        t = line.split()
        if len(t) > 1:
            op = t[0]
            for k in ('global_load_dwordx4', 'global_load_dwordx2', 'global_load_dword ', 'flat_load', 'ds_read_b128', 'ds_read_b64', 'ds_read2', 'ds_write', 'global_store', 's_cbranch', 'v_mov_b32', 'v_cndmask', 'scratch'):
                if (op + ' ').startswith(k): c[k.strip()] += 1
            c['total'] += 1
            if op.startswith('v_'): c['valu'] += 1
    print(name.split('(')[0][:70], dict(c))
PY
rm -rf $D

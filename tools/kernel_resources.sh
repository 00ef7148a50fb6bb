#!/bin/bash
# VGPRs / SGPRs / scratch / LDS of the kernels in a built librpt_hip.so (from the code object's metadata).
# usage: tools/kernel_resources.sh [lib.so|""] [name filter]
LIB=$(readlink -f "${1:-$(dirname "$0")/../rust-path-tracer_amd/lib/librpt_hip.so}")
T=$(mktemp -d); cp "$LIB" "$T/lib.so"
(cd "$T" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for CO in "$T"/lib.so.*gfx950*; do
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$CO" | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
rows = []
for blk in re.split(r'\n  - ', txt):
    if '.vgpr_count' not in blk: continue
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    rows.append((g('name'), g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
names = subprocess.run(['c++filt'], input='\n'.join(r[0] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    if flt in n: print('%4s vgpr %4s sgpr %6s scratch %6s lds  %s' % (r[1], r[2], r[3], r[4], n[:110]))
" "$2"
done
rm -rf "$T"

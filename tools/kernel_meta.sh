#!/bin/bash
# Prints VGPR / SGPR / scratch / LDS of every kernel in a compiled translation unit: tools/kernel_meta.sh snarkjs_amd/build/msm_bn254.o [filter]
set -e
obj=$(readlink -f "$1"); filt=${2:-.}
tmp=$(mktemp -d); cp "$obj" $tmp/u.o; cd $tmp
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading u.o >/dev/null 2>&1
co=$(ls u.o.*gfx950* | head -1)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | python3 -c '
import sys,re
name=None; d={}
for ln in sys.stdin:
    m=re.match(r"\s*[-]?\s*\.(\w+):\s*(.*)",ln)
    if not m: continue
    k,v=m.group(1),m.group(2).strip()
    if k in ("vgpr_count","sgpr_count","private_segment_fixed_size","group_segment_fixed_size","agpr_count","vgpr_spill_count"): d[k]=v
    if k=="symbol": sym=v
    if k=="wavefront_size":          # last key of a kernel entry (keys are sorted)
        print(sym, d); d={}
' | c++filt | grep -E "$filt" | sed "s/'//g"
rm -rf $tmp

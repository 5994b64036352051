#!/bin/bash
# r03: what the called-product k_plonk_t costs on a HEALTHY box (leaves a slow-fetch box at once)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
ratio=$(python -c "
import ctypes as C
from snarkjs_amd import zkmi
zkmi.init(0); L = zkmi.lib(); a, b = C.c_double(0), C.c_double(0)
L.zkmi_calibrate_code_fetch(C.byref(a), C.byref(b)); print(round(b.value / a.value, 3))" 2>/dev/null | tail -1)
echo "code fetch big/small = $ratio"
if python -c "import sys; sys.exit(0 if float('$ratio') >= 0.85 else 1)"; then
for m in 16 0 16 0; do echo -n "plonk compact=$m: "; ZKMI_COMPACT_CODE=$m timeout 600 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('latency_ms_single_proof'))"; done
fi

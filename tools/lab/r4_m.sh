#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4m; mkdir -p $O
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('stages_ms',{}).get('ntt_x6'), d.get('submetrics',{}).get('ntt_ms'))"; }
for rep in 1 2; do
$B 2>/dev/null | line "groth16 default(ntt29<=2^22)"
ZKMI_NTT29=0 $B 2>/dev/null | line "groth16 ntt32"
done | tee $O/ab.txt
for rep in 1 2; do
python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | line "plonk default"
ZKMI_NTT29=0 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | line "plonk ntt32"
done | tee -a $O/ab.txt
python bench.py --workload fflonk --log-n 18 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line "fflonk default" | tee -a $O/ab.txt
ZKMI_NTT29=0 python bench.py --workload fflonk --log-n 18 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | line "fflonk ntt32" | tee -a $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plonk.py -m gpu -x -q -k "ntt or fused or golden_proof or plonk_golden or fflonk" 2>&1 | tail -3

#!/bin/bash
# r06: level-1 scatter ranked in LDS (k_rsort_scatter1_staged) against the direct scatter (ZKMI_RSORT_STAGED=0), one box: parity first, the standalone
# kernel times of one table MSM, then the PLONK and Groth16 lines interleaved
O=$GRAFT_REPO_ROOT/gpurun_out/r6staged; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or skew or groth16 or cache or variants" 2>&1 | tail -3) | tee $O/pytest.log
(timeout 400 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q -k "golden or two_proofs or full_size" 2>&1 | tail -3) | tee -a $O/pytest.log
bash tools/lab/r6_sort_probe.sh | tee $O/probe_staged.txt
for mode in "ZKMI_RSORT_STAGED=0" "ZKMI_RSORT_STAGED=1" "ZKMI_RSORT_STAGED=0" "ZKMI_RSORT_STAGED=1"; do
  env $mode python bench.py --workload plonk --steps 16 --warmup 3 --no-cpu-baseline --no-napi-wall --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk   $mode', d['value'], d['latency_ms_single_proof'])" | tee -a $O/ab.txt
  env $mode python bench.py --steps 20 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groth16 $mode', d['value'], 'latency', d['latency_ms_single_proof'])" | tee -a $O/ab.txt
done

#!/bin/bash
# r05: the fused level 2 of the digit sort (msm.cuh: k_rsort_part) against the chunked kernels, one box, one call: parity first (default build), then
# PLONK and Groth16 lines under ZKMI_RSORT_FUSED=0 (r04's path), ZKMI_RSORT_LB=10 (the default: 512 partitions of 2^10 buckets, 152 KB of LDS per block)
# and ZKMI_RSORT_LB=9 (1 024 partitions, 66 KB per block). usage: tools/lab/r5_rsort_ab.sh out_dir
out=${1:-gpurun_out/r5_rsort}; mkdir -p $out
(timeout 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or skew or groth16 or cache or variants" 2>&1 | tail -3) | tee $out/pytest.log
(timeout 400 python -m pytest tests/test_gpu_plonk.py -m gpu -x -q -k "golden or two_proofs or full_size" 2>&1 | tail -3) | tee -a $out/pytest.log
for mode in "ZKMI_RSORT_FUSED=0" "ZKMI_RSORT_LB=10" "ZKMI_RSORT_LB=9" "ZKMI_RSORT_FUSED=0" "ZKMI_RSORT_LB=10"; do
  env $mode python bench.py --workload plonk --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plonk   $mode', d['value'], d['latency_ms_single_proof'])" | tee -a $out/ab.txt
  env $mode python bench.py --steps 16 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('groth16 $mode', d['repeats']['proofs_per_s'], 'table msm ms', d['submetrics']['g1_msm_resident_tables_ms'], 'latency', d['latency_ms_single_proof'])" | tee -a $out/ab.txt
done

#!/bin/bash
# r06: level 1 of the digit sort with atomic range reservation (default) against the scanned matrix of r02 - r05 (ZKMI_RSORT_SCAN=1), same box, interleaved runs
out=gpurun_out/r6_l1; mkdir -p $out
for rep in 1 2; do for v in 0 1; do
  ZKMI_RSORT_SCAN=$v python bench.py --workload plonk --steps 12 --warmup 3 --no-cpu-baseline > $out/plonk_scan${v}_$rep.json 2>/dev/null
  ZKMI_RSORT_SCAN=$v python bench.py --workload plonk --steps 12 --warmup 3 --no-cpu-baseline --pipeline 1 > $out/plonk_serial_scan${v}_$rep.json 2>/dev/null
  ZKMI_RSORT_SCAN=$v python bench.py --steps 20 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs > $out/g16_scan${v}_$rep.json 2>/dev/null
done; done
ZKMI_RSORT_SCAN=0 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-napi-wall --no-cpu-baseline > $out/bls_scan0.json 2>/dev/null
ZKMI_RSORT_SCAN=1 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-napi-wall --no-cpu-baseline > $out/bls_scan1.json 2>/dev/null
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r6_l1/*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d.get('latency_ms_single_proof'))
    except Exception as e: print(f,'ERR',e)
PY

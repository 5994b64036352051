#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
for pl in 1 2; do timeout 600 python bench.py --workload plonk --log-n 20 --steps 16 --warmup 4 --pipeline $pl --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], 'timed', d.get('timed_region_ms_per_proof'), 'lat', d.get('latency_ms_serial_proofs'), 'up', d.get('latency_ms_with_witness_upload'), d['box_calibration'].get('code_fetch',{}).get('big_over_small'))"; done

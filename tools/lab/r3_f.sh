#!/bin/bash
# r03 sixth GPU pass: workgroup placement under the auxiliary stream — accumulation block size and the cap of the Fq2 row/column sums (BLS12-381, BN254)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT" || exit 1
run() { python bench.py "$@" --no-cpu-baseline --no-napi-wall 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages_ms']; print(d['value'], d['ms_per_step'], 'B2', round(s['accum_B2'],2), 'B1', round(s['accum_B1+sort_witness'],2), 'A', round(s['accum_A'],2), 'C', round(s['accum_C'],2), 'H', round(s['accum_H'],2), 'red', round(s['reduce_g1'],2))"; }
for blk in 128 256 64; do for cap in 512 256; do echo -n "bls acc_block=$blk aux_cap=$cap: "; ZKMI_ACC29_BLOCK=$blk ZKMI_AUX_RC_SUMS=$cap run --curve bls12381 --steps 8 --warmup 2; done; done
for blk in 256 128; do for cap in 512 256; do echo -n "bn acc_block=$blk aux_cap=$cap: "; ZKMI_ACC29_BLOCK=$blk ZKMI_AUX_RC_SUMS=$cap run --steps 20 --warmup 3; done; done

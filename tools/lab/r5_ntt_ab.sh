#!/bin/bash
# r05 A/B of the 29-bit NTT passes: the shipped library against variant builds (ZKMI_BUILD_VARIANT=<v> ZKMI_EXTRA_FLAGS=-DZKMI_NTT_VARIANT=1|2 python -m snarkjs_amd.build
# -> snarkjs_amd/libzkmi_<v>.so, loaded with ZKMI_LIB; csrc/ntt29.cuh says what the variants are):
# standalone transforms (tools/lab/r4_ntt_probe.py: device events, sha of the last output — must be equal across builds), the all-sizes parity test,
# and the in-proof chain / proofs per second of a short bench run. usage: tools/lab/r5_ntt_ab.sh out_dir variant...
out=$1; shift
mkdir -p $out
for v in product "$@"; do
  if [ $v = product ]; then unset ZKMI_LIB; else export ZKMI_LIB=$PWD/snarkjs_amd/libzkmi_$v.so; fi
  echo "== $v" | tee -a $out/ntt_ab.txt
  for rep in 1 2; do python tools/lab/r4_ntt_probe.py 2>&1 | tail -1 | tee -a $out/ntt_ab.txt; done
  if [ $v != product ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or fft or prescale or golden" 2>&1 | tail -2 | tee -a $out/ntt_ab.txt; fi
  python bench.py --steps 12 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['repeats']['proofs_per_s'], 'ntt_x6', d['stages_ms'].get('ntt_x6'), 'ntt_ms', d['submetrics']['ntt_ms'])" | tee -a $out/ntt_ab.txt
done

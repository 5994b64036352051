#!/bin/bash
# r06: window width of the pre-computed tables, measured ("measure c in {16,17,18} for the G2 MSM"). One box, same call.
#   Groth16 2^20 dense: ZKMI_PRECOMP=<c> moves EVERY table of the key to c (stage times show what each MSM pays; rocprofv3 gives the reductions)
#   PLONK 2^20: ZKMI_TABLE_C=<c> moves the SRS table
# c = 18 / 19 are left out: the top window of a 254-bit scalar would hold 2 / 7 bits (msm_host.hpp: msm_precomp_c).
out=gpurun_out/r6_window; mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
for c in 20 17 16; do
  ZKMI_PRECOMP=$c python bench.py --steps 12 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs > $out/groth16_c$c.json 2> $out/groth16_c$c.err
  ZKMI_PRECOMP=$c rocprofv3 --kernel-trace --stats -d $out/prof_c$c -o g16 -- python bench.py --steps 8 --warmup 2 --repeats 1 --no-napi-wall --no-cpu-baseline --no-other-configs > $out/prof_c$c.json 2> $out/prof_c$c.err
  f=$(find $out/prof_c$c -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -40 "$f" > $out/kernel_stats_c$c.csv
  rm -rf $out/prof_c$c
done
for c in 20 17; do
  ZKMI_TABLE_C=$c python bench.py --workload plonk --steps 10 --warmup 2 --no-cpu-baseline > $out/plonk_c$c.json 2> $out/plonk_c$c.err
done
python - <<'PY'
import json,glob,os
o='gpurun_out/r6_window'
for f in sorted(glob.glob(o+'/groth16_c*.json'))+sorted(glob.glob(o+'/plonk_c*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d.get('stages_ms'), d.get('latency_ms_single_proof'))
    except Exception as e: print(f, 'ERR', e)
PY

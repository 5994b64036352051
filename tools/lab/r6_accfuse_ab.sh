#!/bin/bash
# A/B: G1 accumulations sharing one digit sort in one launch (ZKMI_ACC_FUSE=1) vs one launch per table
mkdir -p gpurun_out/r6acc; O=gpurun_out/r6acc
ZKMI_ACC_FUSE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_groth16_build.py -m gpu -x -q -k "groth16 or g16 or build" > $O/parity_fused.log 2>&1; echo "parity rc=$?" >> $O/parity_fused.log
for rep in 1 2 3; do
  for f in 0 1; do
    ZKMI_ACC_FUSE=$f timeout 300 python bench.py --steps 20 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs > $O/dense_f${f}_$rep.json 2>$O/dense_f${f}_$rep.err
    ZKMI_ACC_FUSE=$f timeout 300 python bench.py --steps 20 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs --coef-dist real --witness mixed > $O/real_f${f}_$rep.json 2>$O/real_f${f}_$rep.err
  done
done
python - <<'PY'
import json,glob
for w in ("dense","real"):
    for f in (0,1):
        v=[]
        for p in sorted(glob.glob(f"gpurun_out/r6acc/{w}_f{f}_*.json")):
            try: v.append(json.loads(open(p).read().strip().splitlines()[-1])["value"])
            except Exception as e: v.append(str(e)[:40])
        print(w,"fuse",f,v)
PY
tail -3 $O/parity_fused.log

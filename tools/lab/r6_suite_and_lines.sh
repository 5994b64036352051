mkdir -p gpurun_out/r6s1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6s1/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r6s1/pytest_gpu.log; tail -4 gpurun_out/r6s1/pytest_gpu.log
for rep in 1 2; do
timeout 300 python bench.py --workload plonk --steps 16 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs > gpurun_out/r6s1/plonk_two_$rep.json 2>gpurun_out/r6s1/plonk_two.err
timeout 300 python bench.py --steps 20 --warmup 3 --no-napi-wall --no-cpu-baseline --no-other-configs > gpurun_out/r6s1/g16_$rep.json 2>gpurun_out/r6s1/g16.err
done
timeout 300 python bench.py --workload plonk --pipeline 1 --steps 12 --warmup 2 --no-napi-wall --no-cpu-baseline --no-other-configs > gpurun_out/r6s1/plonk_serial.json 2>gpurun_out/r6s1/plonk_serial.err
python - <<'PY'
import json,glob
for p in sorted(glob.glob("gpurun_out/r6s1/*.json")):
    try:
        j=json.loads(open(p).read().strip().splitlines()[-1]); print(p.split('/')[-1], j["value"], j.get("latency_ms_single_proof"))
    except Exception as e: print(p, "ERR", str(e)[:80])
PY

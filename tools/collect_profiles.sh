#!/bin/bash
# Collect the round's judged evidence on the GPU box (run through gpurun): bench JSON (dense B, 2 proofs in flight = the default line),
# serial and sparse-B variants, rocprofv3 kernel stats of the SERIAL command, the two PMC passes (separate runs, --kernel-trace only),
# PLONK / BLS12-381 / 2^24 / FFLONK side benches, the PLONK kernel trace, the mul ceilings of the library's own field arithmetic.
# usage (repo root):  gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r05'   then   python tools/publish_profiles.py r05
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
# the default line exactly as the driver runs it (r05: same-box reference WASM baseline, three timed regions, the three other configs as child runs)
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
timeout 300 python bench.py --steps 20 --warmup 3 --pipeline 1 --no-cpu-baseline --no-napi-wall --no-other-configs > $O/bench_serial.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --b-zero-every 3 --no-cpu-baseline --no-napi-wall --no-other-configs > $O/bench_sparse_b.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 3 --witness mixed --no-cpu-baseline --no-napi-wall --no-other-configs > $O/bench_mixed_witness.json 2>/dev/null
timeout 600 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 > $O/bench_plonk_2p20.json 2>/dev/null      # r05: 524 285 additions on the device inside every proof; same-box reference at 2^14 / 2^16
timeout 600 python bench.py --workload plonk --log-n 20 --steps 12 --warmup 4 --pipeline 1 --no-cpu-baseline > $O/bench_plonk_2p20_serial.json 2>/dev/null
timeout 600 python bench.py --curve bls12381 --steps 8 --warmup 2 --no-cpu-baseline --no-napi-wall > $O/bench_bls12381_2p20.json 2>/dev/null
timeout 300 python bench.py --workload fflonk --log-n 18 --steps 5 --warmup 1 > $O/bench_fflonk_2p18.json 2>/dev/null
timeout 900 python bench.py --log-n 24 --steps 3 --warmup 1 --no-cpu-baseline --no-napi-wall > $O/bench_bn128_2p24.json 2>/dev/null
# kernel durations are judged on the SERIAL command (--pipeline 1): with two proofs in flight kernels of different proofs share the chip and a
# per-launch average would not be the kernel's own time (bench.py measures its live roofline time on serial proofs for the same reason)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python bench.py --steps 20 --warmup 3 --pipeline 1 --no-cpu-baseline --no-napi-wall --no-other-configs > $O/bench_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bls -o bls -- python bench.py --curve bls12381 --steps 6 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_bls_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/stats_plonk -o plonk -- python bench.py --workload plonk --log-n 20 --steps 4 --warmup 3 --pipeline 1 --no-cpu-baseline > $O/bench_plonk_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall --no-other-configs > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall --no-other-configs > /dev/null 2>&1
# r06: the same two passes for EVERY other config of the driver line (roofline.traffic non-null in all of them), each with the bench line of the
# counted run itself (its accum_mixed_additions feed the per-access-class correction), and the circuit-shaped key
pmc() {   # pmc <name> <bench args...>
  local nm=$1; shift
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_${nm}_fetch -o f -- python bench.py "$@" --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall --no-other-configs > $O/pmc_${nm}_bench.json 2>/dev/null
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_${nm}_write -o w -- python bench.py "$@" --steps 2 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall --no-other-configs > /dev/null 2>&1
  rm -f $O/pmc_${nm}_fetch/*/*kernel_trace.csv $O/pmc_${nm}_write/*/*kernel_trace.csv $O/pmc_${nm}_fetch/*kernel_trace.csv $O/pmc_${nm}_write/*kernel_trace.csv
}
pmc bls --curve bls12381
pmc plonk --workload plonk --log-n 20
pmc real --coef-dist real --witness mixed
pmc p24 --log-n 24
# FETCH_SIZE calibration in the gather widths of both curves' table entries (64 / 128 B: BN254 G1 / G2; 96 / 192 B: BLS12-381) and a coalesced stream
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_gather -o g -- tools/bin/gatherbench > $O/gatherbench.txt 2>&1
rm -f $O/pmc_gather/*/*kernel_trace.csv $O/pmc_gather/*kernel_trace.csv
timeout 300 python bench.py --coef-dist real --witness mixed --steps 20 --warmup 3 --no-cpu-baseline --no-napi-wall > $O/bench_real.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_real -o real -- python bench.py --coef-dist real --witness mixed --steps 10 --warmup 2 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_real_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_p24 -o p24 -- python bench.py --log-n 24 --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline --no-napi-wall > $O/bench_p24_under_rocprof.json 2>/dev/null
rm -f $O/stats_real/*/*kernel_trace.csv $O/stats_p24/*/*kernel_trace.csv $O/stats_real/*kernel_trace.csv $O/stats_p24/*kernel_trace.csv
# tools/bin/fieldbench29 = the shipped arithmetic (9-limb fields: one asm statement per column; 14-limb: plain C); _plain = -DZK_MAD_PLAIN (plain C
# everywhere); _asm = -DZK_MAD_PLAIN -DZK_MAD_ASM (one asm statement per multiply-add: the r03 build)
{ echo "== the shipped build: one asm statement per COLUMN of the product scanning (every field)"; tools/bin/fieldbench29;
  echo "== -DZK_MAD_PLAIN: multiply-adds in plain C everywhere"; tools/bin/fieldbench29_plain;
  echo "== -DZK_MAD_PLAIN -DZK_MAD_ASM: one inline-asm statement per multiply-add (the r03 build: one s_nop behind each)"; tools/bin/fieldbench29_asm; tools/bin/maddbench29; tools/bin/maddbench29_g2; } > $O/fieldbench29.txt 2>&1
# r06: every kernel of ONE table MSM by itself (sort and accumulation on one stream: the sort kernels' own durations)
bash tools/lab/r6_sort_probe.sh > $O/sort_probe.txt 2>&1
# the multi-rank code path of bench.py on this ONE GPU (a 1-rank RCCL communicator): sharded MSM over resident tables, one proof over all ranks at 2^20 and at 2^24 (BASELINE configs[2])
ZKMI_FORCE_DIST=1 timeout 1200 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-napi-wall --no-other-configs > $O/bench_force_dist.json 2>/dev/null
# the whole GPU suite on this box
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest_gpu.log
# keep the merge under the gpurun limit: the per-dispatch traces are large, the stats files are what gets published
rm -f $O/stats/*kernel_trace.csv $O/stats_bls/*kernel_trace.csv $O/pmc_fetch/*kernel_trace.csv $O/pmc_write/*kernel_trace.csv
for f in bench bench_serial bench_sparse_b bench_mixed_witness bench_real bench_plonk_2p20 bench_plonk_2p20_serial bench_bls12381_2p20 bench_fflonk_2p18 bench_bn128_2p24 bench_force_dist; do python - "$O/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms")
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
du -sh $O

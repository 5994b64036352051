#!/bin/bash
# r04 first contact: the whole GPU suite, then the default bench line (new: 29-bit standalone MSM, napi wall with small-n and 2-process sharded proofs)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4a/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["submetrics"])); print(json.dumps(d.get("wall_through_napi"))); print(json.dumps(d["roofline"])); print(json.dumps(d["stages_ms"]))
PY
tail -5 $O/bench.err

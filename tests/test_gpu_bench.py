"""bench.py on the GPU box: the line's contract, and the multi-rank (RCCL) path kept alive on a communicator of one rank.

The scaling runs (N = 2, 4, 8) are the driver's; what can rot unnoticed between them is the code only they reach — bench.py: multi_rank_extras and
snarkjs_amd/distributed.py over the real RCCL backend. ZKMI_FORCE_DIST=1 makes a single process initialise torch.distributed ("nccl" = RCCL) with
world size 1 and run exactly that code: sharded table MSM + all_gather, one proof over "all" ranks on key shards with the chain exchange, the
BASELINE configs[2] leg with the key synthesised once and mapped from shared memory."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry ONE JSON line, got {len(lines)}: {r.stdout[-1500:]}"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract_small():
    """the default code path at 2^16: every key the driver and the judge read, the spread over repeated regions, the live C-port baseline with its
    in-run parity check (the same-box WASM leg is exercised by the default-size run; here it is switched off to keep the test short)"""
    d = _run(["--log-n", "16", "--steps", "6", "--warmup", "1", "--no-napi-wall", "--no-ref-wasm", "--cpu-log-n", "14"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "repeats"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["value"] > 0
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-5
    assert d["cpu_baseline"]["parity_on_sample"] is True and d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] >= 1
    rp = d["repeats"]
    assert len(rp["proofs_per_s"]) == 3 and rp["min"] <= rp["median"] <= rp["max"] and abs(rp["proofs_per_s"][0] - d["value"]) / d["value"] < 1e-3


@pytest.mark.gpu
def test_bench_multi_rank_path_on_a_one_rank_rccl_communicator():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, ZKMI_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    d = _run(["--gpus", "1", "--steps", "4", "--warmup", "1", "--configs2-log-n", "16", "--no-napi-wall", "--no-cpu-baseline", "--no-other-configs", "--repeats", "1"], env=env)
    sh = d["g1_msm_sharded"]
    assert sh is not None and "error" not in sh, sh
    assert sh["world_size_rccl"] == 1 and sh["backend"] == "nccl" and sh["mscalar_per_s"] > 0
    one = sh["groth16_one_proof_over_all_ranks"]
    assert one["equals_single_device_proof"] is True and one["ms_per_proof"] > 0 and set(one["timeline_ms_per_rank"]) == {"0"}
    c2 = sh["groth16_configs2"]
    assert "skipped" not in c2 and c2["log_n"] == 16 and c2["ms_per_proof"] > 0 and "mapped by every rank" in c2["key_source"]
    # the budget guard: with nothing left for the extras the 2^k leg says so instead of running
    d2 = _run(["--gpus", "1", "--steps", "4", "--warmup", "1", "--configs2-log-n", "16", "--extras-budget", "0", "--no-napi-wall", "--no-cpu-baseline", "--no-other-configs", "--repeats", "1"], env=env)
    assert "budget" in d2["g1_msm_sharded"]["groth16_configs2"]["skipped"]


@pytest.mark.gpu
def test_bench_parity_modes_of_the_other_configs_small():
    """r06: the per-config parity legs of bench.py at sizes that take seconds — a circuit-shaped key (--coef-dist real) against the C restatement, the
    closed form of the full-size proof (the configs[2] leg), and the SAME key opened by offset and proved from Node (wall_through_napi: proof equal to
    the Python mirror's for the same draws)."""
    d = _run(["--log-n", "14", "--steps", "4", "--warmup", "1", "--coef-dist", "real", "--witness", "mixed", "--cpu-baseline-mode", "port", "--no-napi-wall", "--no-other-configs", "--repeats", "1"])
    assert d["cpu_baseline"]["parity_on_sample"] is True and d["config"]["coef_dist"] == "real" and d["coef_layout"]["cut_rows"] >= 2
    d = _run(["--log-n", "14", "--steps", "4", "--warmup", "1", "--cpu-baseline-mode", "closed", "--napi-wall-reps", "1", "--no-other-configs", "--repeats", "1"],
             env=dict(os.environ, ZKMI_NAPI_WALL_BIG="1"))
    assert d["cpu_baseline"]["closed_form_at_bench_size"] is True and d["cpu_baseline"]["parity_on_sample"] is True
    nw = d["wall_through_napi"]
    if "skipped" not in nw:
        assert nw.get("proof_equals_python_mirror") is True and nw["groth16_pipelined_equals_serial"] is True, nw


@pytest.mark.gpu
def test_bench_preflight_and_plonk_replica_line_on_a_one_rank_communicator():
    """r06 first-contact hardening: --preflight (device ordinal, free HBM, peer-access row, one all-reduce over RCCL) and the PLONK replica line
    (--workload plonk --gpus N) through torch.distributed, both on a communicator of one rank."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, ZKMI_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = _run(["--gpus", "1", "--preflight"], env=env)
    assert p["preflight"] is True and p["ok"] is True and p["world_size_rccl"] == 1 and p["backend"] == "nccl" and p["distinct_devices"] is True
    r0 = p["ranks"][0]
    assert r0["device"] == 0 and r0["hbm_free_gb"] > 100 and r0["peer_access"][0] is True and r0["allreduce_ok"] is True and r0["zkmi_device_count"] >= 1
    p1 = _run(["--preflight"])                                # without a launcher / communicator
    assert p1["ok"] is True and p1["world_size_rccl"] == 1 and p1["backend"] is None
    d = _run(["--gpus", "1", "--workload", "plonk", "--log-n", "12", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], env=env)
    assert d["metric"] == "plonk_proofs_per_sec" and d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "replica x1"

"""Host-side logic of bench.py that needs no GPU: the budget guard of other_configs (a config that does not fit what is left of --other-configs-budget / --total-budget is
reported as skipped, never run and never silently dropped) and the per-workload lookup of measured HBM traffic."""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_other_configs_budget_guard(monkeypatch):
    bench = importlib.import_module("bench")
    calls = []
    monkeypatch.setattr("subprocess.run", lambda *a, **k: calls.append(a) or (_ for _ in ()).throw(AssertionError("a child was started")))
    args = argparse.Namespace(other_configs_budget=0.0, total_budget=10_000.0, no_cpu_baseline=False, no_ref_wasm=False)
    res = bench.other_configs(args)
    assert set(res) == {"configs[4]", "configs[3]", "configs[1] on a circuit-shaped key", "configs[2] at N=1"}
    assert all("skipped" in v and "budget" in v["skipped"] for v in res.values()) and not calls
    # the whole-run budget binds too, counted from process start
    args = argparse.Namespace(other_configs_budget=10_000.0, total_budget=0.0, no_cpu_baseline=False, no_ref_wasm=False)
    res = bench.other_configs(args)
    assert all("skipped" in v for v in res.values()) and not calls


def test_pmc_traffic_is_looked_up_per_workload():
    bench = importlib.import_module("bench")
    t = bench.pmc_traffic("groth16:bn128:2^20:b_zero_every=0:uniform", "k_msm_accum29_g2s<Bn254Fq>")     # r06: one Fq2 component per lane
    assert t is not None and 2.0e9 < t < 3.5e9
    assert bench.pmc_traffic("groth16:bls12381:2^20:b_zero_every=0:uniform", "k_msm_accum29_g2s<Bls12381Fq>") > 3.0e9
    assert bench.pmc_traffic("plonk:bn128:2^20:additions=524285", "k_msm_accum29<Bn254Fq>") > 1.0e9
    assert bench.pmc_traffic("groth16:bn128:2^24:b_zero_every=0:uniform", "k_msm_accum29_g2s<Bn254Fq>") > 3.5e10
    # a workload that was never counted gets no figure from another one
    assert bench.pmc_traffic("groth16:bn128:2^16:b_zero_every=0:uniform", "k_msm_accum29_g2s<Bn254Fq>") is None
    assert bench.pmc_traffic("groth16:bn128:2^20:b_zero_every=3:uniform", "k_msm_accum29_g2s<Bn254Fq>") is None

"""Shim: the generators live in snarkjs_amd/workloads (shared with bench.py)."""
from snarkjs_amd.workloads.synth import *  # noqa: F401,F403

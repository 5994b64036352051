"""Shim: the generators live in snarkjs_amd/workloads (shared with bench.py)."""
from snarkjs_amd.workloads.synth_plonk import *  # noqa: F401,F403
from snarkjs_amd.workloads.synth_plonk import make, make_fflonk  # noqa: F401

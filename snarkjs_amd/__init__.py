"""snarkjs_amd — MI355X-native proving backend for the snarkjs prover hot path.

Only what the path needs: csrc/ (hand-written HIP kernels + the C-ABI of include/zkmi.h), zkmi.py (ctypes binding),
curves.py (host-side mirror of the ffjavascript curve surface snarkjs calls), groth16.py (fused prover driver),
napi/ + js/ (the Node addon and register.js glue for unmodified snarkjs).
"""
from . import zkmi  # noqa: F401
from .curves import get_curve_from_name, get_curve_from_r  # noqa: F401

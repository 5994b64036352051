"""Re-export of the product's zkey / wtns container readers (snarkjs_amd/binfile.py) for the tests."""
from snarkjs_amd.binfile import *  # noqa: F401,F403
from snarkjs_amd.binfile import read_sections, read_groth16_zkey, read_wtns, proof_json  # noqa: F401

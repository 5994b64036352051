"""Shim over snarkjs_amd/workloads/synth_zkey.py (shared with bench.py) that adds the GPU-less path: with use_device=False the base
tables come from the CPU oracle (test infrastructure)."""
from snarkjs_amd.workloads import synth_zkey as _z
from snarkjs_amd.workloads.synth_zkey import PRIMES, _binfile, _section  # noqa: F401


def _oracle_tables(cid, group, n):
    import oracle_lib as O
    return O.geom_bases(cid, group, n)


def make(name, lg, seed=1, n_public=2, witness="mixed", use_device=True, coef_per_row=1, b_zero_every=3, coef_dist="flat", n_vars=None):
    return _z.make(name, lg, seed=seed, n_public=n_public, witness=witness, tables=None if use_device else _oracle_tables,
                   coef_per_row=coef_per_row, b_zero_every=b_zero_every, coef_dist=coef_dist, n_vars=n_vars)

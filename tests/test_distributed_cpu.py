"""World-size-2 gloo test of the sharded-MSM exchange path (snarkjs_amd/distributed.py) on CPU.

The per-shard compute is stubbed with the CPU oracle (tests may use the oracle; there is no GPU here); what is covered
is the N>1 plumbing: index-range sharding, the all_gather of partial Jacobian points, the host fold (zkmi_point_add from
the product library, which needs no device) and that every rank ends with the same, correct group element.
"""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, n, out_dir):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_lib as O
    import synth
    from snarkjs_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class FakeCurve:            # stands in for snarkjs_amd.curves.Curve: only .id is read when `compute` is given
        id = O.BN128
        G1 = G2 = None

    for group in (1, 2):
        m = n if group == 1 else n // 4
        bases = O.geom_bases(O.BN128, group, m)
        scalars = synth.witness_like(0xD15C0 + group, m)
        pb = 64 * group
        lo, hi = D.shard_range(m, rank, world)
        res = D.msm_sharded(FakeCurve, group, bases[lo * pb:hi * pb], scalars[lo * 32:hi * 32],
                            compute=lambda b, s: O.msm(O.BN128, group, b, s, len(s) // 32))
        want = O.to_affine(O.BN128, group, O.msm(O.BN128, group, bases, scalars, m))
        assert np.array_equal(O.to_affine(O.BN128, group, res), want)
        np.save(os.path.join(out_dir, f"r{rank}_g{group}.npy"), res)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_covers_everything():
    from snarkjs_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 9, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_sharded_msm_gloo_world2(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), 600, str(tmp_path)), nprocs=2, join=True)
    for g in (1, 2):       # every rank holds the same bytes (fold in rank order)
        assert np.array_equal(np.load(tmp_path / f"r0_g{g}.npy"), np.load(tmp_path / f"r1_g{g}.npy"))

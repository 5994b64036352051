"""Multi-GPU MSM: one process per GPU, shards by base-index range, partial points exchanged over RCCL/xGMI.

The reference already splits one multiExp into contiguous index chunks across its workers and adds the chunk results on
the host (ffjavascript engine_multiexp, bundle min.js:1@214651).  Here a chunk = the slice [lo, hi) owned by one rank:
its bases are resident on that rank's GPU (zkey sections are static), every rank runs the full device Pippenger on its
slice, and the g partial Jacobian points (96..288 bytes each) are exchanged with ONE all_gather — RCCL has no
user-defined reduction, so "reduce" = all-gather + local fold.  Payload is a few hundred bytes: the
collective is latency-only; xGMI bandwidth is irrelevant for this path.

Backend: torch.distributed ("nccl" = RCCL on ROCm; "gloo" for the CPU tests).
"""
import numpy as np

from . import zkmi


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n terms owned by `rank` (ceil split, like the reference's chunking)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def all_gather_bytes(local, group=None):
    """all_gather of equally sized byte strings -> list of numpy uint8 arrays (one per rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.ascontiguousarray(local, dtype=np.uint8).copy()).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return [o.cpu().numpy() for o in outs]


def fold_points(curve_id, group_id, partials):
    """Sum of Jacobian points (host, zkmi_point_add): ranks fold in rank order, so every rank gets identical bytes."""
    acc = np.zeros_like(np.ascontiguousarray(partials[0], dtype=np.uint8))
    L = zkmi.lib()
    for p in partials:
        p = np.ascontiguousarray(p, dtype=np.uint8)
        out = np.zeros_like(acc)
        zkmi.check(L.zkmi_point_add(curve_id, group_id, zkmi.ptr(acc), zkmi.ptr(p), zkmi.ptr(out)))
        acc = out
    return acc


def msm_sharded(curve, group_id, local_bases, local_scalars, process_group=None, compute=None):
    """multiExpAffine over bases/scalars whose slice [lo, hi) lives on this rank. Returns the full result on every rank.

    compute(bases, scalars) -> Jacobian bytes: defaults to the device MSM of `curve` (snarkjs_amd.curves.Curve)."""
    G = curve.G1 if group_id == 1 else curve.G2
    fn = compute or G.multiExpAffine
    part = np.ascontiguousarray(fn(local_bases, local_scalars), dtype=np.uint8)
    return fold_points(curve.id, group_id, all_gather_bytes(part, process_group))


def fold_groth16_sums(curve_id, all_sums):
    """Add the per-rank MSM sums of a sharded Groth16 proof (ProvingKey.sums_raw: jA | jB1 | jB2 | jC | jH) point by point."""
    q = 32 if curve_id == 0 else 48
    j1 = 3 * q
    cuts = [(0, j1, 1), (j1, 2 * j1, 1), (2 * j1, 4 * j1, 2), (4 * j1, 5 * j1, 1), (5 * j1, 6 * j1, 1)]
    out = np.zeros(7 * j1, np.uint8)
    for a, b, grp in cuts:
        out[a:b] = fold_points(curve_id, grp, [np.ascontiguousarray(s[a:b]) for s in all_sums])
    return out


def chain_owner(chain, world):
    """Rank that runs chain 0 (A), 1 (B) or 2 (C) of the iNTT -> coset -> NTT step (src/groth16_prove.js:64-76)."""
    return chain % world


class DeviceShard:
    """Backend of groth16_prove_sharded on the MI355X: a ProvingKey shard + torch CUDA tensors for the exchanged slices."""

    def __init__(self, pk, witness=None, d_witness=None):
        import torch
        self.pk, self.torch = pk, torch
        self.curve_id = pk.curve_id
        self.n, self.m = pk.zk["domainSize"], pk.zk["nVars"]
        self._wbuf = None
        if d_witness is None:
            self._wbuf = zkmi.DeviceBuffer.from_host(zkmi.u8(witness))
            d_witness = self._wbuf.ptr
        self.d_witness = d_witness
        self.dev = torch.device("cuda", torch.cuda.current_device())

    def empty(self, nbytes):
        return self.torch.empty(max(nbytes, 1), dtype=self.torch.uint8, device=self.dev)

    def chains(self, owned):
        """{chain: tensor of n*32 bytes} for the chains this rank owns (buildABC + iNTT + coset NTT on the full domain)."""
        if not owned:
            return {}
        out = {c: self.empty(self.n * 32) for c in owned}
        ptr = lambda c: out[c].data_ptr() if c in out else None
        mask = sum(1 << c for c in owned)
        zkmi.check(zkmi.lib().zkmi_groth16_chains_dev(self.pk.key, self.d_witness, mask, ptr(0), ptr(1), ptr(2)))      # synchronises the library stream
        return out

    def sums_w(self):
        """Enqueue the witness-side half of this shard's MSMs (digit sorts of the witness, accumulations B2, B1, A, C, the G2 bucket
        reduction): needs the witness only, returns at once — the kernels run while the chain slices travel (zkmi_groth16_sums_w_dev)."""
        zkmi.check(zkmi.lib().zkmi_groth16_sums_w_dev(self.pk.key, self.d_witness))

    def join(self, a, b, c, cnt):
        """joinABC (:79, :320-374) on this rank's slice -> H-MSM scalars (normal form)"""
        h = self.empty(cnt * 32)
        # the exchanged slices were written on torch's streams: wait for THOSE (a device-wide synchronize would also wait for the
        # witness-side accumulations already running on the library's stream)
        self.torch.cuda.current_stream().synchronize()
        if cnt:
            zkmi.check(zkmi.lib().zkmi_groth16_join_abc_dev(self.curve_id, a.data_ptr(), b.data_ptr(), c.data_ptr(), h.data_ptr(), cnt))
        return h

    def sums(self, h):
        q = 32 if self.curve_id == 0 else 48
        out = np.zeros(7 * 3 * q, np.uint8)
        zkmi.check(zkmi.lib().zkmi_groth16_sums_h_dev(self.pk.key, self.d_witness, h.data_ptr(), zkmi.ptr(out)))
        return out

    def finish(self, sums, r_mont, s_mont):
        return self.pk.finish_raw(sums, r_mont, s_mont)

    def close(self):
        if self._wbuf is not None:
            self._wbuf.free()
            self._wbuf = None


def start_chain_exchange(be, outs, rank, world, h_ranges, process_group=None):
    """Starts the reduce-scatter-shaped exchange of the chain-parallel proof: the owner of chain c sends rank j the byte slice
    [32*h_lo_j, 32*h_hi_j) of its output. Returns (this rank's three slice buffers [A', B', C'], pending requests): the buffers are
    complete after wait_chain_exchange. Point-to-point (batch_isend_irecv: RCCL send/recv over xGMI, or gloo on CPU); domain*32 bytes
    leave each owner in total, nothing is replicated."""
    import torch.distributed as dist
    lo, hi = h_ranges[rank]
    mine = [None, None, None]
    ops = []
    for c in range(3):
        o = chain_owner(c, world)
        if o == rank:
            mine[c] = outs[c][32 * lo:32 * hi].clone() if hi > lo else be.empty(0)
            for j in range(world):
                jl, jh = h_ranges[j]
                if j != rank and jh > jl:
                    ops.append(dist.P2POp(dist.isend, outs[c][32 * jl:32 * jh].contiguous(), j, group=process_group))
        elif hi > lo:
            mine[c] = be.empty(32 * (hi - lo))
            ops.append(dist.P2POp(dist.irecv, mine[c], o, group=process_group))
        else:
            mine[c] = be.empty(0)
    return mine, (dist.batch_isend_irecv(ops) if ops else [])


def wait_chain_exchange(reqs):
    for req in reqs:
        req.wait()


def exchange_chain_slices(be, outs, rank, world, h_ranges, process_group=None):
    """start_chain_exchange + wait_chain_exchange in one call."""
    mine, reqs = start_chain_exchange(be, outs, rank, world, h_ranges, process_group)
    wait_chain_exchange(reqs)
    return mine


def groth16_prove_sharded(pk, witness, r_mont, s_mont, process_group=None, d_witness=None, backend=None, timeline=None):
    """One Groth16 proof over the ranks of `process_group` (BASELINE configs[2]), nothing replicated but buildABC:

      1. chain-parallel transforms: rank c % world runs buildABC + the iNTT -> coset -> NTT chain c (A, B, C are independent until
         joinABC, src/groth16_prove.js:64-79);
      2. every chain owner starts sending rank j the slice [h_lo_j, h_hi_j) of its output (point-to-point over xGMI);
      3. WHILE the slices travel every rank runs the witness-side half of its shard's MSMs (A, B1, B2, C depend on the witness only:
         ranks without a chain start here at time 0, nobody idles through the transforms and the exchange);
      4. rank j joins its slices into ITS H-MSM scalars and runs the H half (digit sort, accumulation H, G1 bucket reductions);
      5. ONE all_gather of the 7*3*n8q-byte partial sums, folded in rank order: every rank returns the identical (pi_a, pi_b, pi_c).

    pk: ProvingKey(zkey, shard=(rank, world)) on every rank; witness: the FULL witness on every rank; r_mont, s_mont: the same
    blinding draws on every rank. backend: object with the DeviceShard interface (the gloo CPU test passes an oracle-backed one).
    timeline: optional dict that receives this rank's host-clock stage boundaries in ms since the call (bench.py --gpus N prints them per
    rank): chains_done (the rank's own chains complete, 0 for ranks without one), w_enqueued, slices_requested_done, exchange_done (every slice of
    this rank has arrived: the copies share the device with the witness-side accumulations already running), sums_done (H half finished = the witness-side half finished too), gathered (all_gather + fold + blinding)."""
    import time
    import torch.distributed as dist
    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
    be = backend or DeviceShard(pk, witness, d_witness)
    t0 = time.perf_counter()
    mark = (lambda k: timeline.__setitem__(k, round((time.perf_counter() - t0) * 1e3, 3))) if timeline is not None else (lambda k: None)
    try:
        h_ranges = [shard_range(be.n, j, world) for j in range(world)]
        outs = be.chains([c for c in range(3) if chain_owner(c, world) == rank])
        mark("chains_done")
        (a, b, c), reqs = start_chain_exchange(be, outs, rank, world, h_ranges, process_group)
        be.sums_w()                                            # witness-side MSMs underneath the exchange
        mark("w_enqueued")
        wait_chain_exchange(reqs)
        mark("slices_requested_done")                          # every send / receive of this rank has been waited for (host side)
        lo, hi = h_ranges[rank]
        h = be.join(a, b, c, hi - lo)                          # waits for the slices on torch's stream, then enqueues joinABC
        mark("exchange_done")
        part = be.sums(h)
        mark("sums_done")
        out = be.finish(fold_groth16_sums(be.curve_id, all_gather_bytes(part, process_group)), r_mont, s_mont)
        mark("gathered")
        return out
    finally:
        if backend is None:
            be.close()


def groth16_prove_sharded_local(make_shard, world, r_mont, s_mont):
    """The same proof with all `world` ranks simulated one after the other on ONE device (tests / single-GPU checks of the
    multi-GPU path): make_shard(rank) -> backend (DeviceShard interface); the exchange is done in-process."""
    bes = [make_shard(j) for j in range(world)]
    try:
        n = bes[0].n
        h_ranges = [shard_range(n, j, world) for j in range(world)]
        outs = {}
        for j, be in enumerate(bes):
            outs.update(be.chains([c for c in range(3) if chain_owner(c, world) == j]))
        parts = []
        for j, be in enumerate(bes):
            lo, hi = h_ranges[j]
            sl = [outs[c][32 * lo:32 * hi].clone() if hi > lo else be.empty(0) for c in range(3)]
            be.sums_w()                                        # the same split order as the ranks run it: witness-side half, join, H half
            parts.append(be.sums(be.join(sl[0], sl[1], sl[2], hi - lo)))
        return bes[-1].finish(fold_groth16_sums(bes[0].curve_id, parts), r_mont, s_mont)
    finally:
        for be in bes:
            be.close()

"""The C-ABI's GPU-to-GPU exchange layer (include/zkmi.h: zkmi_ipc_export / _open / _close, zkmi_peer_copy; snarkjs_amd/csrc/peer.hip) — what the
processes of a multi-GPU proof move their chain slices with (js/groth16_shards.js). The reference's analogue hands chunk buffers to worker threads
(ffjavascript, build/snarkjs.min.js:1@214651); here a worker is a PROCESS bound to one GPU, so the test uses two processes. On a one-GPU box both
sit on device 0: the export / open / copy protocol is the multi-GPU one, the placement is not."""
import ctypes as C
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HB = 96                                                       # ZKMI_IPC_HANDLE_BYTES


def _child(handle_bytes, lo, cnt, device, with_torch, q):
    try:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if with_torch:
            import torch  # noqa: F401  — see the note in the test: both ends of a handle must run the same HIP runtime
        from snarkjs_amd import zkmi
        zkmi.init(device)
        L = zkmi.lib()
        h = np.frombuffer(handle_bytes, np.uint8).copy()
        p, vis = C.c_void_p(), C.c_size_t()
        zkmi.check(L.zkmi_ipc_open(zkmi.ptr(h), C.byref(p), C.byref(vis)))
        dst = zkmi.DeviceBuffer(cnt)
        zkmi.check(L.zkmi_peer_copy(dst.ptr, p.value + lo, cnt))          # a SLICE of the exported buffer, device to device
        got = dst.to_host()
        zkmi.check(L.zkmi_ipc_close(p))
        zkmi.check(L.zkmi_ipc_close(p))                                   # closing twice is harmless
        q.put(("ok", got.tobytes(), int(vis.value)))
    except Exception as e:                                                # noqa: BLE001
        q.put(("err", repr(e), 0))


def test_ipc_export_open_peer_copy_between_processes():
    from snarkjs_amd import zkmi
    zkmi.init()
    L = zkmi.lib()
    n = 3 << 20
    data = (np.arange(n, dtype=np.uint64) * 2654435761 >> 7).astype(np.uint8)
    big = zkmi.DeviceBuffer(n + 4096)
    zkmi.check(L.zkmi_memcpy_h2d(big.ptr + 4096, zkmi.ptr(data), n))      # an INTERIOR pointer is exported
    h = np.zeros(HB, np.uint8)
    zkmi.check(L.zkmi_ipc_export(big.ptr + 4096, zkmi.ptr(h)))
    # the exporting process resolves its own handle to the original pointer
    p, vis = C.c_void_p(), C.c_size_t()
    zkmi.check(L.zkmi_ipc_open(zkmi.ptr(h), C.byref(p), C.byref(vis)))
    assert p.value == big.ptr + 4096 and vis.value >= n
    zkmi.check(L.zkmi_ipc_close(p))
    same = zkmi.DeviceBuffer(1000)
    zkmi.check(L.zkmi_peer_copy(same.ptr, p.value + 77, 1000))
    assert np.array_equal(same.to_host(), data[77:1077])
    # queued copies + fence: what is enqueued on the library stream after the fence sees the data (to_host copies on that stream)
    two = zkmi.DeviceBuffer(4096)
    zkmi.check(L.zkmi_peer_copy_async(two.ptr, p.value + 1000, 2048))
    zkmi.check(L.zkmi_peer_copy_async(two.ptr + 2048, p.value + 5000, 2048))
    zkmi.check(L.zkmi_peer_fence())
    assert np.array_equal(two.to_host(), np.concatenate([data[1000:3048], data[5000:7048]]))
    two.free()
    # another process (on the second GPU when there is one) pulls a slice. A pytest session has torch imported (conftest, other tests), and the
    # torch wheel brings its own HIP runtime into the process: a handle exported under it cannot be opened by a process that runs the system
    # runtime alone (r04 probe: hipIpcOpenMemHandle "invalid argument" exactly then, tools/lab/r4_ipc_probe.py). The product's shard processes are
    # Node processes without torch; here the child simply loads what the parent has loaded.
    dev = 1 if L.zkmi_device_count() > 1 else 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    lo, cnt = 123 * 32, 65536 * 32
    import sys
    pr = ctx.Process(target=_child, args=(h.tobytes(), lo, cnt, dev, "torch" in sys.modules, q))
    pr.start()
    status, payload, visible = q.get(timeout=300)
    pr.join(60)
    assert status == "ok", payload
    assert payload == data[lo:lo + cnt].tobytes() and visible >= n
    # the exporter still owns the memory afterwards
    assert np.array_equal(big.to_host(64, 4096), data[:64])
    assert L.zkmi_peer_copy(None, None, 16) != 0 and L.zkmi_peer_copy(None, None, 0) == 0
    big.free(); same.free()


def test_groth16_key_curve_and_reset(golden_dir):
    from snarkjs_amd import groth16, zkmi
    L = zkmi.lib()
    zkey = open(os.path.join(golden_dir, "groth16_bls12381_n1024.zkey"), "rb").read()
    pk = groth16.ProvingKey(zkey)
    assert L.zkmi_groth16_key_curve(pk.key) == 1 and L.zkmi_groth16_key_curve(0xDEAD) == -1
    assert L.zkmi_groth16_reset(0xDEAD) != 0
    zkmi.check(L.zkmi_groth16_reset(pk.key))
    pk.release()


def test_reset_clears_a_half_enqueued_proof(golden_dir):
    """error recovery of the split proof (js/groth16_shards.js after a failed worker): a witness-side half without its H half blocks the slot until
    zkmi_groth16_reset; the next whole proof is the golden one"""
    import json
    from snarkjs_amd import binfile, groth16, zkmi
    L = zkmi.lib()
    g = json.load(open(os.path.join(golden_dir, "groth16_bn128_n1024.json")))
    zkey = open(os.path.join(golden_dir, "groth16_bn128_n1024.zkey"), "rb").read()
    w = binfile.read_wtns(open(os.path.join(golden_dir, "groth16_bn128_n1024.wtns"), "rb").read())["witness"]
    r_m, s_m = np.frombuffer(bytes.fromhex(g["r_mont"]), np.uint8), np.frombuffer(bytes.fromhex(g["s_mont"]), np.uint8)
    pk = groth16.ProvingKey(zkey)
    want = pk.prove_raw(w, r_m, s_m)
    d_w = zkmi.DeviceBuffer.from_host(w)
    zkmi.check(L.zkmi_groth16_sums_w_dev(pk.key, d_w.ptr))
    with pytest.raises(zkmi.ZkmiError, match="witness-side half is waiting"):
        pk.prove_raw(w, r_m, s_m)
    zkmi.check(L.zkmi_groth16_reset(pk.key))
    got = pk.prove_raw(w, r_m, s_m)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    d_w.free(); pk.release()

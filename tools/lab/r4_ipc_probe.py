"""r04 probe (GPU box): which hipIpc export / open patterns work between two processes here, and window widths for the standalone MSM."""
import ctypes as C, multiprocessing as mp, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def child(hb, off, cnt, q):
    try:
        from snarkjs_amd import zkmi
        zkmi.init(0)
        L = zkmi.lib()
        h = np.frombuffer(hb, np.uint8).copy()
        p, vis = C.c_void_p(), C.c_size_t()
        rc = L.zkmi_ipc_open(zkmi.ptr(h), C.byref(p), C.byref(vis))
        if rc:
            q.put(("open failed", L.zkmi_last_error().decode())); return
        d = zkmi.DeviceBuffer(cnt)
        rc = L.zkmi_peer_copy(d.ptr, p.value + off, cnt)
        q.put(("ok" if rc == 0 else "copy failed", d.to_host()[:8].tolist() if rc == 0 else L.zkmi_last_error().decode()))
    except Exception as e:
        q.put(("exc", repr(e)))


def main():
    from snarkjs_amd import zkmi
    import_torch = "--torch" in sys.argv
    if import_torch:
        import torch; torch.cuda.device_count()
    zkmi.init(0)
    L = zkmi.lib()
    ctx = mp.get_context("spawn")
    for name, size, interior in (("32KB@0", 32768, 0), ("3MB@0", 3 << 20, 0), ("3MB+4096 interior", (3 << 20) + 4096, 4096), ("4MB interior 2MB", 4 << 20, 2 << 20), ("64MB@0", 64 << 20, 0)):
        b = zkmi.DeviceBuffer(size)
        data = np.arange(size, dtype=np.uint8)
        zkmi.check(L.zkmi_memcpy_h2d(b.ptr, zkmi.ptr(data), size))
        h = np.zeros(96, np.uint8)
        rc = L.zkmi_ipc_export(b.ptr + interior, zkmi.ptr(h))
        if rc:
            print(name, "export failed", L.zkmi_last_error().decode()); continue
        q = ctx.Queue()
        pr = ctx.Process(target=child, args=(h.tobytes(), 64, 4096, q)); pr.start()
        print(name, "off", int.from_bytes(h[64:72].tobytes(), "little"), q.get(timeout=120), "expect", data[interior + 64:interior + 72].tolist(), flush=True)
        pr.join(30)
        b.free()
    # window width of the standalone MSM (caller-owned plain bases, 2^20)
    from snarkjs_amd.workloads import synth
    n = 1 << 20
    d_b = zkmi.DeviceBuffer(n * 64)
    zkmi.check(L.zkmi_gen_geometric_bases_dev(0, 1, n, 7, 11, d_b.ptr))
    d_s = zkmi.DeviceBuffer.from_host(synth.elems(0x5EED, n))
    jac = np.zeros(96, np.uint8)
    for c in (0, 13, 14, 15, 16, 17, 18):
        zkmi.check(L.zkmi_msm_set_window_bits(c))
        ts, ta = [], []
        for _ in range(4):
            zkmi.check(L.zkmi_msm_dev(0, 1, d_b.ptr, d_s.ptr, n, 32, zkmi.ptr(jac)))
            ts.append(L.zkmi_last_kernel_ms()); ta.append(L.zkmi_msm_accum_ms(0))
        print("c", c, "msm_ms", round(min(ts[1:]), 4), "accum_ms", round(min(ta[1:]), 4), jac[:4].tolist(), flush=True)
    zkmi.check(L.zkmi_msm_set_window_bits(0))


if __name__ == "__main__":
    main()

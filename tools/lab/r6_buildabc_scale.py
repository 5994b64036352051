"""r06 one-off: the sliced buildABC layout at scale — circuit-shaped keys at 2^22 and 2^24 (49 M coefficient records, rows up to 10^5 terms), A_T / B_T / C_T of
zkmi_groth16_build_abc_dev against the oracle's restatement of the reference loop (oracle/zk_oracle.c: orc_groth16_build_abc), layout statistics, load and kernel time.
usage: gpurun -- 'python tools/lab/r6_buildabc_scale.py 22 24'"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
from snarkjs_amd import binfile, groth16, zkmi  # noqa: E402
from snarkjs_amd.workloads import synth_zkey  # noqa: E402

zkmi.init(0)
for lg in [int(x) for x in sys.argv[1:]] or [22]:
    t0 = time.perf_counter()
    zkey, wtns = synth_zkey.make("bn128", lg, seed=0x5EED, witness="mixed", coef_dist="real")
    t_syn = time.perf_counter() - t0
    zk, w = binfile.read_groth16_zkey(zkey), binfile.read_wtns(wtns)["witness"]
    rec = np.frombuffer(zk["coeffs"][4:].tobytes(), dtype=[("m", "<u4"), ("c", "<u4"), ("s", "<u4"), ("v", "u1", 32)])
    cnt = np.bincount(rec["m"].astype(np.int64) * zk["domainSize"] + rec["c"])
    t0 = time.perf_counter()
    pk = groth16.ProvingKey(zkey)
    t_load = time.perf_counter() - t0
    d_w = zkmi.DeviceBuffer.from_host(w)
    ms = []
    for _ in range(5):
        got = pk.build_abc(d_witness=d_w.ptr)
        ms.append(zkmi.lib().zkmi_last_kernel_ms())
    t0 = time.perf_counter()
    want = O.build_abc(0, zk["coeffs"], w, zk["nVars"], zk["domainSize"])
    t_or = time.perf_counter() - t0
    ok = all(np.array_equal(a, b.reshape(-1)) for a, b in zip(got, want))
    print(json.dumps({"log_n": lg, "n_coef": int(rec.size), "n_coef_over_n": round(rec.size / zk["domainSize"], 3), "longest_row": int(cnt.max()), "rows_beyond_32": int((cnt > 32).sum()),
                      "rows_beyond_1000": int((cnt > 1000).sum()), "layout": pk.coef_layout(), "key_synthesis_s": round(t_syn, 1), "key_load_s": round(t_load, 1),
                      "build_abc_ms": [round(x, 3) for x in ms], "oracle_build_abc_s": round(t_or, 1), "equals_oracle": bool(ok)}), flush=True)
    pk.release(); d_w.free()
    del zkey, wtns, zk, w, rec, got, want

"""FFLONK on BLS12-381: pinned as UNSUPPORTED BY THE REFERENCE, not as a gap of this backend.

oracle/gen_golden.js (fflonkBlsProbe) ran the reference's own fflonk.setup + fflonk.prove on a satisfied Multiplier(40) circuit over a seeded
BLS12-381 ptau: the setup hard-codes BN254 constants (src/fflonk_setup.js:533-556), the key it writes has w3^3 != 1, and the reference's prover
throws "Polynomial is not divisible". There is no proof to be bit-identical to. The device drivers hold a non-bn128 key to what the protocol needs
of it (w3^3 = 1, w3 != 1, w4 / w8 of order 4 / 8, wr^3 = the domain's root) and refuse an inconsistent one up front with the reference's own
"Polynomial is not divisible"; a consistent key (a repaired or third-party setup) is not refused for its curve (r05)."""
import json
import os
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLS_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_R = 52435875175126190479447740508185965837690552500527637822603658699938581184513


def test_reference_rejects_fflonk_on_bls12381():
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "fflonk_bls12381_unsupported.json")))
    assert d["reference_prove_error"] == "Polynomial is not divisible"
    assert d["header"]["curve"] == "bls12381" and d["header"]["protocol"] == "fflonk"
    # the constants the reference's setup wrote are not what the protocol needs on this curve
    assert d["header"]["w3_cubed_is_one"] is False and d["header"]["wr_cubed_is_w_power"] is False
    # independent of the bundle: BN254's generator / exponent of computeW3 do not give a cube root of unity in BLS12-381's Fr
    w3 = pow(31624, 3648040478639879203707734290876212514758060733402672390616367364429301415936 // 3, BLS_R)
    assert w3 == int(d["header"]["w3"]) and pow(w3, 3, BLS_R) != 1


def test_device_driver_refuses_a_bls12381_fflonk_key_before_touching_the_device():
    from snarkjs_amd import fflonk
    sec = lambda t, b: struct.pack("<IQ", t, len(b)) + b
    hdr = struct.pack("<I", 48) + BLS_Q.to_bytes(48, "little") + struct.pack("<I", 32) + BLS_R.to_bytes(32, "little") + bytes(20 + 6 * 32 + 6 * 48)
    zkey = b"zkey" + struct.pack("<II", 1, 2) + sec(1, struct.pack("<I", 10)) + sec(2, hdr)
    with pytest.raises(ValueError, match="Polynomial is not divisible"):
        fflonk.FflonkKey(zkey)
    # the reference's own inconsistent header values (w3 from BN254's generator) are refused the same way
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "fflonk_bls12381_unsupported.json")))
    R = pow(2, 256, BLS_R)
    m = lambda v: (int(v) * R % BLS_R).to_bytes(32, "little")
    h = d["header"]
    body = struct.pack("<IIIII", 10, 1, 64, 0, 8) + m(2) + m(3) + m(h["w3"]) + bytes(3 * 32) + bytes(6 * 48)
    hdr2 = struct.pack("<I", 48) + BLS_Q.to_bytes(48, "little") + struct.pack("<I", 32) + BLS_R.to_bytes(32, "little") + body
    with pytest.raises(ValueError, match="Polynomial is not divisible"):
        fflonk.FflonkKey(b"zkey" + struct.pack("<II", 1, 2) + sec(1, struct.pack("<I", 10)) + sec(2, hdr2))

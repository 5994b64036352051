"""Readers for the iden3 binfile containers the prover consumes (zkey / wtns).

Format (@iden3/binfileutils): magic[4], u32 version, u32 nSections, then
{u32 type, u64 len, bytes} per section.  Groth16 zkey sections: src/zkey_utils.js:20-45, header layout
src/zkey_utils.js:229-259; wtns: src/wtns_utils.js:62-72.  Host-side test plumbing only.
"""
import struct

import numpy as np


def read_sections(data: bytes, magic: bytes):
    assert data[:4] == magic, f"bad magic {data[:4]!r}"
    _version, nsec = struct.unpack_from("<II", data, 4)
    off, sections = 12, {}
    for _ in range(nsec):
        typ, ln = struct.unpack_from("<IQ", data, off)
        off += 12
        sections.setdefault(typ, []).append((off, ln))
        off += ln
    return sections


def _sec(data, sections, typ):
    off, ln = sections[typ][0]
    return np.frombuffer(data, dtype=np.uint8, count=ln, offset=off)


def read_groth16_zkey(data: bytes):
    s = read_sections(data, b"zkey")
    assert struct.unpack_from("<I", data, s[1][0][0])[0] == 1, "not a groth16 zkey"
    off = s[2][0][0]
    n8q = struct.unpack_from("<I", data, off)[0]; off += 4
    q = int.from_bytes(data[off:off + n8q], "little"); off += n8q
    n8r = struct.unpack_from("<I", data, off)[0]; off += 4
    r = int.from_bytes(data[off:off + n8r], "little"); off += n8r
    n_vars, n_public, domain = struct.unpack_from("<III", data, off); off += 12
    zk = dict(n8q=n8q, n8r=n8r, q=q, r=r, nVars=n_vars, nPublic=n_public, domainSize=domain)
    for name, size in (("vk_alpha_1", 2), ("vk_beta_1", 2), ("vk_beta_2", 4), ("vk_gamma_2", 4), ("vk_delta_1", 2), ("vk_delta_2", 4)):
        zk[name] = np.frombuffer(data, dtype=np.uint8, count=size * n8q, offset=off); off += size * n8q
    zk["coeffs"] = _sec(data, s, 4)
    for name, typ in (("A", 5), ("B1", 6), ("B2", 7), ("C", 8), ("H", 9)):
        zk[name] = _sec(data, s, typ)
    return zk


def read_wtns(data: bytes):
    s = read_sections(data, b"wtns")
    off = s[1][0][0]
    n8 = struct.unpack_from("<I", data, off)[0]; off += 4
    q = int.from_bytes(data[off:off + n8], "little"); off += n8
    n_witness = struct.unpack_from("<I", data, off)[0]
    return dict(n8=n8, q=q, nWitness=n_witness, witness=_sec(data, s, 2))


def proof_json(curve_name, n8q, pi_a_norm, pi_b_norm, pi_c_norm):
    """Render normal-form LE coordinate bytes the way snarkjs does (src/groth16_prove.js:130-141 → JSON.stringify)."""
    import json

    def c(b, i):
        return str(int.from_bytes(bytes(b[i * n8q:(i + 1) * n8q]), "little"))

    proof = {
        "pi_a": [c(pi_a_norm, 0), c(pi_a_norm, 1), "1"],
        "pi_b": [[c(pi_b_norm, 0), c(pi_b_norm, 1)], [c(pi_b_norm, 2), c(pi_b_norm, 3)], ["1", "0"]],
        "pi_c": [c(pi_c_norm, 0), c(pi_c_norm, 1), "1"],
        "protocol": "groth16",
        "curve": curve_name,
    }
    return proof, json.dumps(proof, separators=(",", ":"))

// tools/lab/r5_setup_at_scale.js — r05 one-off (GPU box; needs oracle/_ref): the SETUP-side callers of the bulk entry points at a size where the
// device matters (VERDICT r04 "Missing #5": src/powersoftau_*.js, src/plonk_setup.js:323-403, src/zkey_new.js:497), through the drop-in boundary:
// unmodified snarkjs (the reference's bundle) runs  newAccumulator -> contribute -> preparePhase2 -> plonk.setup -> zKey.newZKey  twice in one
// process — on its own WASM + worker threads, then with registerAll() (group FFTs, point conversions, G.batchApplyKey, MSMs and Fr FFTs on the
// MI355X through the real addon) — with the same entropy stream: ptau and zkey BYTES must be identical; wall times of every step side by side.
// The circuit is the Multiplier(n) chain of the reference's test/groth16/circuit.circom (x_i = x_{i-1}^2 + b) written straight into the r1cs
// container (src/r1cs format: header, constraints, wire map), n = 2^(power-2): plonk.setup turns it into 2n rows + additions.
// usage: node --harmony-optional-chaining --harmony-nullish --max-old-space-size=32000 tools/lab/r5_setup_at_scale.js <power> [<power> ...]
"use strict";
const path = require("path"), crypto = require("crypto");
const ROOT = path.join(__dirname, "..", "..");
const snarkjs = require(path.join(ROOT, "oracle", "ref_shim.js"));
const { register, unregister } = require(path.join(ROOT, "snarkjs_amd", "js", "register.js"));
const sha = (b) => crypto.createHash("sha256").update(b).digest("hex").slice(0, 16);
const bytes = (m) => { const d = m.data; return (d instanceof Uint8Array) ? d : d.slice(0, d.byteLength); };

function chainR1cs(r, n) {
    const le = (v, k) => { const o = Buffer.alloc(k); let x = BigInt(v); for (let i = 0; i < k; i++) { o[i] = Number(x & 255n); x >>= 8n; } return o; };
    const nWires = n + 3, wire = (i) => (i == n - 1 ? 1 : 4 + i);                  // wires: 1, out = x_{n-1}, a, b, x_0 .. x_{n-2}
    const one = le(1n, 32), minus1 = le(r - 1n, 32), u32 = (v) => le(v, 4);
    const parts = [];
    for (let i = 0; i < n; i++) {
        const prev = i == 0 ? 2 : wire(i - 1);
        parts.push(u32(1), u32(prev), one, u32(1), u32(prev), one, u32(2), u32(3), minus1, u32(wire(i)), one);    // prev * prev = x_i - b
    }
    const cs = Buffer.concat(parts);
    const hdr = Buffer.concat([u32(32), le(r, 32), u32(nWires), u32(1), u32(1), u32(1), le(nWires, 8), u32(n)]);
    const map = Buffer.alloc(8 * nWires);
    for (let i = 0; i < nWires; i++) map.writeUInt32LE(i, 8 * i);
    const sec = (t, b) => Buffer.concat([u32(t), le(b.length, 8), b]);
    return new Uint8Array(Buffer.concat([Buffer.from("r1cs"), u32(1), u32(3), sec(1, hdr), sec(2, cs), sec(3, map)]));
}

(async () => {
    const curve = await snarkjs.curves.getCurveFromName("bn128");
    const mem = () => ({ type: "mem" });
    for (const power of process.argv.slice(2).map(Number)) {
        const r1cs = chainR1cs(curve.Fr.p, 1 << (power - 2));
        const rows = [];
        const out = {};
        for (const mode of (process.env.MODES || "wasm,device").split(",")) {
            if (mode === "device") register(curve, {}); else unregister(curve);
            snarkjs.reseed();
            const t = {}, tick = async (k, f) => { const t0 = process.hrtime.bigint(); await f(); t[k] = Number(process.hrtime.bigint() - t0) / 1e9; };
            const p0 = mem(), p1 = mem(), pf = mem(), zp = mem(), zg = mem();
            await tick("newAccumulator", () => snarkjs.powersOfTau.newAccumulator(curve, power, p0));
            await tick("contribute", () => snarkjs.powersOfTau.contribute(p0, p1, "C1", "Entropy1"));
            await tick("preparePhase2", () => snarkjs.powersOfTau.preparePhase2(p1, pf));
            await tick("plonk.setup", () => snarkjs.plonk.setup(r1cs, pf, zp));
            await tick("zKey.newZKey", () => snarkjs.zKey.newZKey(r1cs, pf, zg));
            out[mode] = { t, ptau: sha(bytes(pf)), plonk: sha(bytes(zp)), groth16: sha(bytes(zg)), bytes: { ptau: bytes(pf).length, plonk: bytes(zp).length, groth16: bytes(zg).length } };
        }
        unregister(curve);
        if (!out.device) out.device = out.wasm;                       // MODES=wasm: a dry run of the script's own logic on a GPU-less box
        const same = ["ptau", "plonk", "groth16"].every((k) => out.wasm[k] === out.device[k]);
        console.log(`power ${power} (r1cs: ${1 << (power - 2)} constraints; ${snarkjs.nThreads} worker threads): ptau / plonk zkey / groth16 zkey bytes identical: ${same}` +
                    ` (${out.wasm.bytes.ptau} / ${out.wasm.bytes.plonk} / ${out.wasm.bytes.groth16} bytes; sha ${out.device.ptau} ${out.device.plonk} ${out.device.groth16})`);
        for (const k of Object.keys(out.wasm.t)) console.log(`   ${k.padEnd(16)} wasm ${out.wasm.t[k].toFixed(2).padStart(8)} s   device ${out.device.t[k].toFixed(2).padStart(8)} s   x${(out.wasm.t[k] / out.device.t[k]).toFixed(1)}`);
        if (!same) process.exitCode = 1;
    }
    process.exit(process.exitCode || 0);
})().catch((e) => { console.error(e && e.stack || e); process.exit(2); });

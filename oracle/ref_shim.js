// oracle/ref_shim.js — TEST INFRASTRUCTURE ONLY (never shipped, never imported by the product path).
//
// Evaluates the reference's own self-contained browser bundle (/root/reference/build/snarkjs.min.js =
// snarkjs 0.7.6 + ffjavascript 0.3.1 + wasmcurves 0.2.2) under this container's Node 12, so that the
// real reference implementation can (a) pin the C restatement in oracle/*.c and (b) emit the golden
// vectors committed under tests/golden/.  Recipe: SURVEY.md Appendix A.  The bundle is only READ: from
// /root/reference in the build container, or from oracle/_ref/ (`make -C oracle _ref`: the git-ignored staging copy of
// the reference's build output + test fixtures that travels to the GPU box, where /root/reference does not exist).
// There it serves tests/js/unmodified_gpu.js (unmodified snarkjs + the REAL addon in one process) and bench.py's
// cpu_baseline leg (the reference's WASM path timed on the GPU box's host cores) — never the product path.
//
// Run:  node --harmony-optional-chaining --harmony-nullish script.js
'use strict';
const fs = require('fs'), vm = require('vm'), nodeCrypto = require('crypto'), os = require('os');
const STAGED = require('path').join(__dirname, '_ref');
const REF_ROOT = process.env.SNARKJS_REF_ROOT || (fs.existsSync(require('path').join(STAGED, 'build', 'snarkjs.min.js')) ? STAGED : '/root/reference');
const REF = process.env.SNARKJS_REF_BUNDLE || require('path').join(REF_ROOT, 'build', 'snarkjs.min.js');
if (!fs.existsSync(REF)) throw new Error(`reference bundle not found at ${REF}: run \`make -C oracle _ref\` where /root/reference exists`);

// Deterministic byte stream = xorshift32 of SURVEY.md Appendix C.3 (seed 0x12345678).  ffjavascript seeds its
// process-wide ChaCha from ONE 32-byte getRandomValues call, so fixing this stream fixes every proof.
let st = 0x12345678;
function xorshiftFill(a) {
    const b = new Uint8Array(a.buffer, a.byteOffset, a.byteLength);
    for (let i = 0; i < b.length; i++) {
        st ^= st << 13; st >>>= 0; st ^= st >>> 17; st ^= st << 5; st >>>= 0;
        b[i] = st & 255;
    }
    return a;
}
const seeded = !process.env.ORACLE_UNSEEDED;
globalThis.crypto = { getRandomValues: a => seeded ? xorshiftFill(a) : (nodeCrypto.randomFillSync(a), a) };
globalThis.btoa = s => Buffer.from(s, 'binary').toString('base64');
globalThis.window = globalThis;
process.browser = true;

const nThreads = parseInt(process.env.NTHREADS || os.cpus().length);
if (!process.env.SINGLE) {
    const { Worker: NW } = require('worker_threads');
    globalThis.navigator = { hardwareConcurrency: nThreads };
    // ffjavascript does `new Worker("data:application/javascript;base64,…")` (web-worker API); adapt to worker_threads.
    globalThis.Worker = class WorkerShim {
        constructor(url) {
            const code = Buffer.from(url.split('base64,')[1], 'base64').toString();
            this.w = new NW(`const {parentPort}=require('worker_threads');
              const self={postMessage:m=>parentPort.postMessage(m),close:()=>process.exit(0),onmessage:null};
              parentPort.on('message',m=>self.onmessage({data:m}));\n${code}`, { eval: true });
        }
        addEventListener(ev, fn) { this.w.on(ev, m => fn({ data: m })); }
        postMessage(m, t) { this.w.postMessage(m, t); }
        terminate() { this.w.terminate(); }
    };
}
vm.runInThisContext(fs.readFileSync(REF, 'utf8') + ';globalThis.snarkjs=snarkjs;');
module.exports = globalThis.snarkjs;
module.exports.nThreads = process.env.SINGLE ? 1 : nThreads;
module.exports.reseed = (v) => { st = (v === undefined ? 0x12345678 : v) >>> 0; };   // restart the getRandomValues stream (ceremony code mixes it into its entropy hash)
module.exports.refRoot = REF_ROOT;                // where test/groth16, test/circuit2, ... of the reference's own tree are read from

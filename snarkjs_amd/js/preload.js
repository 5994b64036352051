// snarkjs_amd/js/preload.js — `node -r ./snarkjs_amd/js/preload.js <snarkjs cli> groth16 prove ...`
// Patches snarkjs's curve getters so that every curve object the CLI obtains is registered with the MI355X backend
// before its first bulk operation (src/curves.js:9-53 are the only places snarkjs builds curves).
"use strict";
const { register } = require("./register.js");
const Module = require("module");
const origLoad = Module._load;
Module._load = function (request, parent, isMain) {
    const m = origLoad.apply(this, arguments);
    if (request === "snarkjs" && m && m.curves && !m.curves.__zkmi) {
        for (const fn of ["getCurveFromName", "getCurveFromQ", "getCurveFromR"]) {
            const orig = m.curves[fn];
            if (typeof orig === "function") m.curves[fn] = async function () { return register(await orig.apply(this, arguments)); };
        }
        m.curves.__zkmi = true;
    }
    return m;
};

"""Static instruction mix of the bucket-accumulation kernels' hot loops (gfx950 code object inside snarkjs_amd/build/msm_bn254.o):
python tools/isa_counts.py > profiles/rNN_isa_counts.md.  The hot loop = the outermost backward branch of the kernel; the rare
equal-points (doubling) path inside it is the largest forward-skipped region and is reported separately."""
import collections, os, re, subprocess, sys, tempfile

OBJ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snarkjs_amd", "build", "msm_bn254.o")
LLVM = "/opt/rocm/lib/llvm/bin"


def disasm(obj):
    d = tempfile.mkdtemp()
    subprocess.run(["cp", obj, os.path.join(d, "u.o")], check=True)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "u.o"], cwd=d, capture_output=True)
    co = [f for f in os.listdir(d) if "gfx950" in f][0]
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(d, co)], capture_output=True, text=True).stdout


def kernels(text):
    for p in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:\n)", text):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", p)
        if not m:
            continue
        ins = []
        for l in p.split("\n")[1:]:
            mm = re.match(r"\s+(\S.*?)\s+// ([0-9A-F]+):", l)
            if mm:
                ins.append((int(mm.group(2), 16), mm.group(1)))
        yield m.group(1), ins


def branches(ins):
    a2i = {a: i for i, (a, _) in enumerate(ins)}
    for i, (a, t) in enumerate(ins):
        mm = re.match(r"s_c?branch\w* (\d+)", t)
        if mm:
            off = int(mm.group(1))
            off -= 65536 if off >= 32768 else 0
            j = a2i.get(a + 4 + off * 4)
            if j is not None:
                yield i, j


def mix(seg):
    h = collections.Counter(t.split()[0] for _, t in seg)
    valu = sum(v for k, v in h.items() if k.startswith("v_"))
    return {"instructions": len(seg), "VALU": valu, "v_mad_u64_u32": h["v_mad_u64_u32"], "s_nop": h["s_nop"], "LDS": sum(v for k, v in h.items() if k.startswith("ds_")),
            "global/scratch loads": sum(v for k, v in h.items() if k.startswith(("global_load", "scratch_load"))), "scratch stores": h["scratch_store_dword"]}


def main():
    text = disasm(sys.argv[1] if len(sys.argv) > 1 else OBJ)
    print("# Static instruction mix of the accumulation hot loops (llvm-objdump of the gfx950 code object; tools/isa_counts.py)\n")
    print("| kernel | part | instructions | VALU | v_mad_u64_u32 | s_nop | LDS | global/scratch loads | scratch stores |\n|---|---|---|---|---|---|---|---|---|")
    for name, ins in kernels(text):
        if "accum29" not in name:
            continue
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void zkmi::", "")
        macs = [k for k, (_, t) in enumerate(ins) if "v_mad_u64" in t]
        back = [(i, j) for i, j in branches(ins) if j < i and i - j > 2000 and any("v_mad_u64" in t for _, t in ins[j:i])]
        fwd = [j for i, j in branches(ins) if j > i and j - i > 2000]
        if back:
            lo, hi = min(j for _, j in back), max(i for i, _ in back)
        elif fwd and sum(1 for _, t in ins if t.startswith("s_setpc")) <= 4:      # back-edge out of the 16-bit branch range (s_setpc): from the loads before the first product
            lo = max(k for k, (_, t) in enumerate(ins[:macs[0]]) if t.startswith("global_load_dwordx4")) - 12
            hi = max(fwd)
        else:
            # every long jump is an s_setpc (14-limb Fq2 kernel: > 128 KiB of code): split at them. Reading for k_msm_accum29_g2<Bls12381Fq>:
            # segment 1 = gather, unpack, ZZ, ZZZ, U2, S2 (every addition); the next big segment = the rare doubling; the one after = the main path;
            # the last = the once-per-lane store (Jacobian -> XYZZ words)
            cuts = [0] + [k for k, (_, t) in enumerate(ins) if t.startswith("s_setpc")] + [len(ins)]
            for a, b in zip(cuts, cuts[1:]):
                if b - a < 1500:
                    continue
                m = mix(ins[a:b])
                print(f"| `{dn}` | s_setpc-delimited segment [{a}, {b}) | " + " | ".join(str(m[k]) for k in ("instructions", "VALU", "v_mad_u64_u32", "s_nop", "LDS", "global/scratch loads", "scratch stores")) + " |")
            continue
        # big forward-skipped regions inside the loop, outermost first, non-overlapping
        regs = []
        for i, j in sorted((x for x in branches(ins) if lo <= x[0] < x[1] <= hi + 1 and x[1] - x[0] > 600), key=lambda x: (x[0], -x[1])):
            if j - i > 0.9 * (hi - lo):
                continue                           # the whole body (loop guard)
            if regs and i < regs[-1][1]:
                continue
            regs.append((i, j))
        cuts = [lo] + [x for r in regs for x in r] + [hi + 1]
        segs = [("straight-line part %d" % (k // 2 + 1) if k % 2 == 0 else "conditionally skipped region %d" % (k // 2 + 1), ins[cuts[k]:cuts[k + 1]]) for k in range(len(cuts) - 1)]
        for part, seg in segs:
            if not seg:
                continue
            m = mix(seg)
            print(f"| `{dn}` | {part} | " + " | ".join(str(m[k]) for k in ("instructions", "VALU", "v_mad_u64_u32", "s_nop", "LDS", "global/scratch loads", "scratch stores")) + " |")

if __name__ == "__main__":
    main()

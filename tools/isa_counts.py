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
    """One row per straight-line SEGMENT of every inlined accumulation kernel: the kernel is cut at every long jump (s_setpc) and at the source and
    target of every 16-bit branch that spans more than 600 instructions; segments under 400 instructions are folded into a count. The main
    path of one mixed addition = the segments executed on every iteration of the lane loop (see the reading under the table)."""
    text = disasm(sys.argv[1] if len(sys.argv) > 1 else OBJ)
    print("# Static instruction mix of the accumulation kernels, segment by segment (llvm-objdump of the gfx950 code object; tools/isa_counts.py)\n")
    print("| kernel | segment [first, last) | instructions | VALU | v_mad_u64_u32 | s_nop | LDS | global/scratch loads | scratch stores | entered by |\n|---|---|---|---|---|---|---|---|---|---|")
    for name, ins in kernels(text):
        if "accum29" not in name or "Compact" in name:
            continue
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void zkmi::", "")
        cuts, why = {0: "kernel entry", len(ins): ""}, {}
        for k, (_, t) in enumerate(ins):
            if t.startswith("s_setpc"):
                cuts[k + 1] = "after s_setpc"
        for i, j in branches(ins):
            if abs(j - i) > 600:
                cuts.setdefault(i + 1, f"fall-through of the branch at {i} (-> {j})")
                cuts[j] = (cuts.get(j, "") + f" target of the branch at {i}").strip()
        ks = sorted(cuts)
        small = 0
        for a, b in zip(ks, ks[1:]):
            if b - a < 400:
                small += b - a
                continue
            m = mix(ins[a:b])
            print(f"| `{dn}` | [{a}, {b}) | " + " | ".join(str(m[k]) for k in ("instructions", "VALU", "v_mad_u64_u32", "s_nop", "LDS", "global/scratch loads", "scratch stores")) + f" | {cuts[a]} |")
        print(f"| `{dn}` | segments under 400 instructions | {small} | | | | | | | |")


if __name__ == "__main__":
    main()

"""Builds libzkmi.so (hand-written HIP for gfx950) in-tree: snarkjs_amd/csrc/*.hip -> snarkjs_amd/libzkmi.so.

hipcc cross-compiles gfx950 without a GPU. Translation units are compiled in parallel and only when stale.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# ZKMI_BUILD_VARIANT=<name> ZKMI_EXTRA_FLAGS="-D..." builds an A/B variant beside the product: objects in build_<name>/, libzkmi_<name>.so (loaded
# with ZKMI_LIB=<path>, snarkjs_amd/zkmi.py); the addon is only built for the product library
VARIANT = os.environ.get("ZKMI_BUILD_VARIANT", "")
OBJ = os.path.join(HERE, "build" + ("_" + VARIANT if VARIANT else ""))
LIB = os.path.join(HERE, "libzkmi" + ("_" + VARIANT if VARIANT else "") + ".so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"] + os.environ.get("ZKMI_EXTRA_FLAGS", "").split()
UNITS = ["zkmi_api.hip", "ntt.hip", "msm_sort.hip", "msm_bn254.hip", "msm_bls12381.hip", "groth16.hip", "plonk.hip", "gfft.hip", "gconv.hip", "calib.hip", "peer.hip"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def _depfile(obj):
    """headers a translation unit actually includes, from the compiler's own -MD output of its last build (None: not built yet)"""
    d = obj[:-2] + ".d"
    if not os.path.exists(d):
        return None
    txt = open(d).read().replace("\\\n", " ")
    deps = txt.split(":", 1)[1].split() if ":" in txt else []
    # a unit compiled from another working directory lists its own headers RELATIVE to it (r04: calib.d / plonk.d did, were filtered out below and
    # the two objects never went stale): resolve against this package directory first
    deps = [x if os.path.isabs(x) else os.path.normpath(os.path.join(HERE, x)) for x in deps]
    own = [x for x in deps if x.startswith(os.path.dirname(HERE)) or x.startswith(CSRC)]
    return own if own else None                           # no header of ours in the list: treat as unknown (depend on every header)


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".hpp"))]
    headers += [os.path.join(os.path.dirname(HERE), "include", h) for h in ("zkmi.h", "zkmi_diag.h")]
    units = [u for u in UNITS if os.path.exists(os.path.join(CSRC, u))]
    jobs = []
    for u in units:
        src, obj = os.path.join(CSRC, u), os.path.join(OBJ, u.replace(".hip", ".o"))
        deps = _depfile(obj)                               # per-unit dependencies once known; every header before the first build
        if force or _stale(obj, [src] + (deps if deps is not None else headers)):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-MD", "-MF", obj[:-2] + ".d", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, u.replace(".hip", ".o")) for u in units]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if not VARIANT:
        build_addon(verbose)
    return LIB


def build_addon(verbose=False):
    """The N-API addon (plain C): snarkjs_amd/napi/zkmi_napi.node, linked against ../libzkmi.so. Skipped (with a note) when
    the Node headers are not installed."""
    src = os.path.join(HERE, "napi", "zkmi_napi.c")
    out = os.path.join(HERE, "napi", "zkmi_napi.node")
    inc = next((d for d in ("/usr/include/node", "/usr/local/include/node") if os.path.exists(os.path.join(d, "node_api.h"))), None)
    if inc is None:
        if verbose:
            print("node_api.h not found: N-API addon not built")
        return None
    if not _stale(out, [src, LIB, os.path.join(os.path.dirname(HERE), "include", "zkmi.h")]):
        return out
    cmd = ["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-DNODE_GYP_MODULE_NAME=zkmi_napi", "-I" + inc, src, "-o", out,
           "-L" + HERE, "-lzkmi", "-ldl", "-lpthread", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"addon build failed:\n{r.stderr[-4000:]}")
    return out


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))

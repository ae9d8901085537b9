"""Builds libsmnngp.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot.

    python build.py                    -> libsmnngp.so
    python build.py --variant s1 -DSMN_STAGES=1   -> libsmnngp_s1.so (A/B builds; select with SMNNGP_LIB=...)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["api.hip", "kernel_build.hip", "cholesky.hip", "heads.hip", "comm.hip", "cnn.hip", "cnn_resnet.hip", "grad.hip", "mixture.hip"]
# -fno-slp-vectorize: hipcc's SLP pass packs adjacent f32 ops into v_pk_*_f32, which on gfx950 issue slower
# than the two plain ops they replace (MI355X_MICROARCH.md, "packed f32 VALU"); measured here: recursion
# kernel -9 %, panel kernel -8 % with packing off (profiles/README.md).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=fast", "-fno-slp-vectorize"]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def stale_sources(variant=""):
    """Names of the sources / headers / includes newer than the built library (all of them when it does not exist)."""
    lib = os.path.join(HERE, "libsmnngp%s.so" % ("_" + variant if variant else ""))
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".inc"))]
    deps.append(os.path.join(HERE, "..", "include", "smnngp.h"))
    if not os.path.exists(lib):
        return [os.path.basename(d) for d in deps]
    t = os.path.getmtime(lib)
    return [os.path.basename(d) for d in deps if os.path.getmtime(d) > t]


def build(force=False, verbose=False, variant="", defines=()):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    suffix = "_" + variant if variant else ""
    objdir = os.path.join(HERE, "build" + suffix)
    lib = os.path.join(HERE, "libsmnngp%s.so" % suffix)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "smnngp.h"))
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer([src] + headers, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + list(defines) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(objs, lib):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return lib


def compiler_version():
    """First line of `hipcc --version`: the leaf of the panel kernel is inline asm with hand-placed wait states, validated
    against the instructions THIS compiler emits around it (tested: HIP 7.2 / AMD clang 22.0.0git roc-7.2.0)."""
    try:
        out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True).stdout
        return next((ln.strip() for ln in out.splitlines() if "clang version" in ln or "HIP version" in ln), out.strip()[:80])
    except OSError as e:
        return "hipcc not found (%s)" % e


if __name__ == "__main__":
    print("compiler:", compiler_version())
    argv = sys.argv[1:]
    variant = ""
    if "--variant" in argv:
        i = argv.index("--variant")
        variant = argv[i + 1]
        del argv[i:i + 2]
    defs = [a for a in argv if a.startswith("-D")]
    print(build(force="--force" in argv, verbose=True, variant=variant, defines=defs))

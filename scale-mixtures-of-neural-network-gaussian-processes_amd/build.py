"""Builds libsmnngp.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsmnngp.so")
SOURCES = ["api.hip", "kernel_build.hip", "cholesky.hip", "heads.hip", "comm.hip", "cnn.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=fast"]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "smnngp.h"))
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _newer([src] + headers, obj):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
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
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

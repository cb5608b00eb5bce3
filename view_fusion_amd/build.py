"""Build libvf_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python -m view_fusion_amd.build            # incremental
    python -m view_fusion_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the
GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "vf_hip")
LIB = os.path.join(HERE, "lib", "libvf_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"),
         "-Wno-unused-result"]


def _hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _digest(paths, extra=()):
    """Content hash of a translation unit's inputs (source, every header, flags): an object is reused only when the
    hash stored next to it matches -- mtimes do not survive repo snapshots."""
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update("\0".join(extra).encode())
    return h.hexdigest()


def build(force=False, verbose=True, report=None):
    """Compile csrc/*.hip -> build/vf_hip/*.o -> lib/libvf_hip.so.  Incremental on mtimes (objects and the library are
    git-ignored but travel with a gpurun snapshot).  `report` (a dict, optional) receives what was done:
    {"compiled": [sources], "reused": [sources], "linked": bool}; a one-line summary is always printed."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "vf_hip.h"))
    jobs, objs, compiled, reused = [], [], [], []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        dig = _digest([s] + sorted(headers), [f.replace(ROOT, ".") for f in FLAGS])
        try:
            old = open(o + ".sha256").read().strip()
        except OSError:
            old = None
        if force or not os.path.exists(o) or old != dig:
            jobs.append(([_hipcc(), *FLAGS, "-c", s, "-o", o], o, dig))
            compiled.append(src)
        else:
            reused.append(src)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    def compile_one(job):
        cmd, o, dig = job
        if os.path.exists(o + ".sha256"):
            os.remove(o + ".sha256")
        run(cmd)
        with open(o + ".sha256", "w") as f:
            f.write(dig)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(compile_one, jobs))
    linked = bool(force or jobs or _stale(LIB, objs))
    if linked:
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    print(f"[view_fusion_amd.build] compiled {len(compiled)} ({', '.join(compiled) or '-'}), reused {len(reused)} "
          f"up-to-date objects, {'linked' if linked else 'kept'} {os.path.relpath(LIB, ROOT)}", flush=True)
    if report is not None:
        report.update(compiled=compiled, reused=reused, linked=linked)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

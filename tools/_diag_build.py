"""Shared by the in-kernel-stamp tools: build a private diagnostic library and never reuse a stale one.

A diagnostic .so (`-DVF_..._STAMPS` build of one or two csrc files) is reused only while the sha256 of its sources, of
every header under csrc/ and include/, and of its flags equals the digest stored next to it -- mtimes do not survive a
repo snapshot to the GPU box, and a phase clock read from a build of an older kernel is worse than no clock."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "view_fusion_amd", "csrc")


def diag_build(name, sources, defines, extra=()):
    """name: file name under build/; sources: csrc file names; defines: ["-DVF_X", ...].  Returns the .so path."""
    so = os.path.join(ROOT, "build", name)
    srcs = [os.path.join(CSRC, s) for s in sources]
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + \
        [os.path.join(ROOT, "include", "vf_hip.h")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
             *defines, *extra]
    h = hashlib.sha256()
    for p in srcs + headers:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update("\0".join(f.replace(ROOT, ".") for f in flags).encode())
    dig = h.hexdigest()
    try:
        old = open(so + ".sha256").read().strip()
    except OSError:
        old = None
    if not os.path.exists(so) or old != dig:
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["hipcc", *flags, *srcs, "-o", so], check=True)
        with open(so + ".sha256", "w") as f:
            f.write(dig)
    return so

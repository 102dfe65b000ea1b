"""Build the gfx950 C-ABI shared library (liblvae_hip.so) in-tree with hipcc.

`python -m vae_lagging_encoder_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles
without a GPU, so this runs on the CPU-only CI box; the .so travels to the GPU box with the tree.
"""
import hashlib
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_NAME = "liblvae_hip.so"
LIB_PATH = os.path.join(CSRC, LIB_NAME)
STAMP = os.path.join(CSRC, ".liblvae_hip.stamp")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(extra=""):
    h = hashlib.sha256(extra.encode())
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode())
                h.update(fh.read())
    return h.hexdigest()


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked at $HIPCC, PATH, /opt/rocm/bin/hipcc)")


def _file_digest(path, flags):
    h = hashlib.sha256(" ".join(flags).encode())
    for f in sorted(os.listdir(CSRC)):          # every header may be included by every source
        if f.endswith(".h"):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(f.encode())
                h.update(fh.read())
    with open(path, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def build_hip(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 into csrc/liblvae_hip.so; returns the path.  One hipcc process per source file, in
    parallel, objects cached under csrc/build/ by content digest (a one-file edit recompiles one file), then one link."""
    cflags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", CSRC]
    dg = _digest(" ".join(cflags) + " per-file")
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dg:
                return LIB_PATH
    hipcc = hipcc_path()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0]
        obj, tag = os.path.join(objdir, base + ".o"), os.path.join(objdir, base + ".digest")
        objs.append(obj)
        fd = _file_digest(src, cflags)
        if not force and os.path.exists(obj) and os.path.exists(tag) and open(tag).read().strip() == fd:
            continue
        cmd = [hipcc] + cflags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        jobs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), tag, fd, src))
    errors = []
    for proc, tag, fd, src in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            errors.append("hipcc failed on %s:\n%s" % (src, out))
            if os.path.exists(tag):
                os.remove(tag)
        else:
            with open(tag, "w") as fh:
                fh.write(fd)
    if errors:
        raise RuntimeError("\n".join(errors))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    with open(STAMP, "w") as fh:
        fh.write(dg)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))

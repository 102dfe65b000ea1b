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


def build_hip(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 into csrc/liblvae_hip.so; returns the path."""
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off", "-I", CSRC]
    dg = _digest(" ".join(flags))
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dg:
                return LIB_PATH
    cmd = [hipcc_path()] + flags + ["-o", LIB_PATH] + sources()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    with open(STAMP, "w") as fh:
        fh.write(dg)
    return LIB_PATH


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))

"""TEST-ONLY: compile the kernel sources (csrc/*.hip) with g++ against the thread-level emulator.

Produces tests/emu/liblvae_emu.so exposing the same C ABI as liblvae_hip.so but operating on host
pointers.  Used only by `-m "not gpu"` tests to check kernel index math and host launch sequencing
on the GPU-less CI box.  The package never loads this library (see vae_lagging_encoder_amd/_lib.py).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "vae_lagging_encoder_amd", "csrc")
LIB = os.path.join(HERE, "liblvae_emu.so")
STAMP = os.path.join(HERE, ".liblvae_emu.stamp")


def _digest():
    h = hashlib.sha256()
    for d in (CSRC, HERE):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h", ".cpp")):
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(f.encode())
                    h.update(fh.read())
    return h.hexdigest()


def build_emu(force=False):
    dg = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            if fh.read().strip() == dg:
                return LIB
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = ["g++", "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-DLV_EMU", "-ffp-contract=off",
               "-Wno-attributes", "-Wno-unknown-pragmas", "-I", HERE, "-I", CSRC, "-c", s, "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("g++ (emu) failed on %s:\n%s" % (s, out))
    impl = os.path.join(HERE, "hip_emu_impl.cpp")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-DLV_EMU", "-I", HERE,
           impl] + objs + ["-o", LIB]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("g++ (emu link) failed:\n" + res.stdout + res.stderr)
    with open(STAMP, "w") as fh:
        fh.write(dg)
    return LIB


if __name__ == "__main__":
    print(build_emu(force="--force" in sys.argv))

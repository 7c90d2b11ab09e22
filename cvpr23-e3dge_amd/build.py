"""Builds libe3dge_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library is a
plain C-ABI shared object (include/e3dge_hip.h) loaded with ctypes."""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "libe3dge_hip.so"
SOURCES = ["stream_ops.hip", "upfirdn2d.hip", "siren.hip", "siren_bwd.hip", "resblock.hip", "modconv.hip", "decoder2.hip", "local_query.hip", "metrics.hip", "align_volume.hip", "hitprob.hip", "siren_ws.hip", "wgrad.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=on", "-fno-slp-vectorize",
         "-Wno-unused-result"]  # no SLP: packed-f32 VALU next to MFMAs is slower and un-does the epilogue interleave


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def _digest():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ["../../include/e3dge_hip.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    """Compile every .hip source for gfx950 and link the shared library.  Returns its path."""
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, ".build_digest")
    dig = _digest()
    if not force and os.path.exists(lib_path()) and os.path.exists(stamp) and open(stamp).read() == dig:
        return lib_path()
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[e3dge build]", " ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", lib_path()]
    if verbose:
        print("[e3dge build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return lib_path()


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

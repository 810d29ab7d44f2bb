"""Build libffq_hip.so (gfx950) in-tree with hipcc.

    python fastq-and-furious_amd/build.py

The library is the product's only compute path: hand-written HIP kernels for
CDNA4 behind the C ABI of include/ffq.h.  hipcc cross-compiles without a GPU.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libffq_hip.so")
SOURCES = ["ffq_hip.hip"]
HEADERS = ["ffq_dev.h", "ffq_kernels.h", "ffq_chain.h", os.path.join("..", "..", "include", "ffq.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libffq_hip.so cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    return any(os.path.getmtime(d) > t for d in deps)


ENTRIES_SRC = os.path.join(CSRC, "ffq_entries.c")
ENTRIES_LIB = os.path.join(CSRC, "_ffq_entries.so")


def build_entries(force=False, verbose=False):
    """The batched default entryfunc (csrc/ffq_entries.c, CPython C API): host glue of the iterator,
    compiled with the C compiler against this interpreter's headers."""
    import sysconfig
    if not force and os.path.exists(ENTRIES_LIB) and os.path.getmtime(ENTRIES_LIB) >= os.path.getmtime(ENTRIES_SRC):
        return ENTRIES_LIB
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler: _ffq_entries.so cannot be built")
    cmd = [cc, "-O2", "-std=c99", "-Wall", "-Wextra", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"],
           "-o", ENTRIES_LIB, ENTRIES_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return ENTRIES_LIB


def build(force=False, verbose=False):
    build_entries(force, verbose)
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

"""Build libffq_hip.so (gfx950) in-tree with hipcc.

    python fastq-and-furious_amd/build.py [--force] [--probe]

The library is the product's only compute path: hand-written HIP kernels for
CDNA4 behind the C ABI of include/ffq.h.  hipcc cross-compiles without a GPU.

Every build bakes in a BUILD ID: the hash of the sources it was compiled from
(csrc/*.h, *.hip and include/ffq.h; the instrumented build also include/ffq_probe.h).  `ffq_build_id()` returns it, hip.lib()
recomputes it from the tree when the library is loaded and rebuilds (or refuses
to run) on a mismatch -- an in-tree .so that is older or newer than the sources
cannot stand in for them (the .so files are git-ignored but travel with the
working tree).

libffq_probe.so is the same sources with -DFFQ_PROBES: the ablation switches and
the read / look-back / pipeline probes (include/ffq_probe.h).  Tools only; the
product never loads it.
"""
import hashlib
import os
import shutil
import subprocess
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(CSRC, "libffq_hip.so")
PROBE_LIB = os.path.join(CSRC, "libffq_probe.so")
SOURCES = ["ffq_hip.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libffq_hip.so cannot be built")


def source_files(probe=False):
    """What libffq_hip.so is compiled from: csrc/*.h, *.hip and include/ffq.h (ffq_entries.c is the CPython glue, a
    library of its own; include/ffq_probe.h belongs to the instrumented build only)."""
    fs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    fs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h") and (probe or f != "ffq_probe.h")]
    return sorted(fs)


def source_id(probe=False):
    """16 hex digits over names and contents of every source the library is compiled from."""
    h = hashlib.sha256()
    for f in source_files(probe):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def built_id(path=LIB):
    """The id baked into a built library (read out of the file, without loading it), or None."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    at = blob.find(b"FFQ_BUILD_ID=")
    return blob[at + 13:at + 29].decode("ascii", "replace") if at >= 0 else None


def needs_build(path=LIB):
    return built_id(path) != source_id(probe=(path == PROBE_LIB))


ENTRIES_SRC = os.path.join(CSRC, "ffq_entries.c")


def entries_lib():
    """_ffq_entries<EXT_SUFFIX>: the interpreter ABI is part of the name, so a build made by another
    Python is not picked up."""
    import sysconfig
    return os.path.join(CSRC, "_ffq_entries" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_entries(force=False, verbose=False):
    """The batched default entryfunc (csrc/ffq_entries.c, CPython C API): host glue of the iterator,
    compiled with the C compiler against this interpreter's headers."""
    import sysconfig
    out = entries_lib()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(ENTRIES_SRC):
        return out
    cc = os.environ.get("CC") or shutil.which("gcc") or shutil.which("cc")
    if not cc:
        raise RuntimeError("no C compiler: %s cannot be built" % os.path.basename(out))
    cmd = [cc, "-O2", "-std=c99", "-Wall", "-Wextra", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"],
           "-o", out, ENTRIES_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


def _have_zlib():
    for d in ("/usr/include", "/usr/local/include", "/opt/rocm/include"):
        if os.path.exists(os.path.join(d, "zlib.h")):
            return True
    return False


class _BuildLock:
    """One builder at a time (ranks of one torchrun that all find the library stale: one compiles, the others wait
    and then find it fresh)."""

    def __enter__(self):
        import fcntl
        self.fh = open(os.path.join(CSRC, ".build.lock"), "w")
        fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *a):
        import fcntl
        fcntl.flock(self.fh, fcntl.LOCK_UN)
        self.fh.close()


def _compile(out, extra, verbose):
    if not _have_zlib():
        raise RuntimeError("zlib.h not found: csrc/ffq_stream.h (the gzip feeder of the stream front end) needs the zlib "
                           "development headers; libffq_hip.so links -lz")
    sid = source_id(probe=(out == PROBE_LIB))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function", '-DFFQ_BUILD_ID="%s"' % sid] + extra + \
          ["-o", out + ".tmp.%d" % os.getpid()] + [os.path.join(CSRC, s) for s in SOURCES] + ["-lz"]      # zlib: the gzip feeder (ffq_stream.h)
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp.%d" % os.getpid(), out)      # (a name of its own per process: ranks that rebuild together do not write into one file)
    return out


def build(force=False, verbose=False):
    try:
        # an optional accelerator of the host iterator (the Python slices do the same): a missing
        # compiler or Python.h must not take the scan library down with it
        build_entries(force, verbose)
    except Exception as e:      # noqa: BLE001
        warnings.warn("csrc/ffq_entries.c not built (%s): the iterator cuts its slices in Python" % (e,))
    if force or needs_build(LIB):
        with _BuildLock():
            if force or needs_build(LIB):          # (another process may have built it while this one waited)
                _compile(LIB, [], verbose)
    return LIB


def build_probe(force=False, verbose=False):
    """The instrumented build (tools only)."""
    if force or needs_build(PROBE_LIB):
        with _BuildLock():
            if force or needs_build(PROBE_LIB):
                _compile(PROBE_LIB, ["-DFFQ_PROBES=1"], verbose)
    return PROBE_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--probe" in sys.argv:
        print(build_probe(force="--force" in sys.argv, verbose=True))

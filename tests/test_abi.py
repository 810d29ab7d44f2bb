"""The C-ABI library: it loads, exports everything include/ffq.h declares, and
fails loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def hip(pkg):
    from fastqandfurious_amd import build, hip
    build.build()
    return hip


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ffq.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ffq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip):
    L = ctypes.CDLL(hip.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 29
    for name in names:
        assert hasattr(L, name), name
    assert sorted(hip.SYMBOLS) == names


def test_probe_library_exports_its_header(hip):
    """include/ffq_probe.h is served by the instrumented build only (tools; bench.py's hbm_read_probe)."""
    from fastqandfurious_amd import build
    text = open(os.path.join(ROOT, "include", "ffq_probe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(ffq_[a-z0-9_]+)\s*\(", text)))
    assert names == ["ffq_read_probe"]
    L = ctypes.CDLL(build.build_probe())
    for name in names + declared_symbols():
        assert hasattr(L, name), name
    L.ffq_build_id.restype = ctypes.c_char_p
    assert L.ffq_build_id().decode() == build.source_id(probe=True) + "+probes"


def test_build_id_is_the_hash_of_the_sources(hip, tmp_path):
    """ffq_build_id() = hash of csrc/ + include/ baked in at compile time; the loader rebuilds or refuses
    on a mismatch, so the GPU box cannot run a stale in-tree binary in place of the sources."""
    from fastqandfurious_amd import build
    want = build.source_id()
    assert len(want) == 16 and build.built_id(hip.LIB_PATH) == want
    assert hip.lib().ffq_build_id().decode() == want == hip.build_id()
    # a library built from other sources is recognised without loading it
    other = tmp_path / "libother.so"
    blob = open(hip.LIB_PATH, "rb").read()
    other.write_bytes(blob.replace(b"FFQ_BUILD_ID=" + want.encode(), b"FFQ_BUILD_ID=" + b"0" * 16))
    assert build.built_id(str(other)) == "0" * 16 and build.needs_build(str(other))


def test_probes_are_not_in_the_product_library(hip):
    """Probe kernels / entry points live in libffq_probe.so (include/ffq_probe.h), not in the product."""
    L = ctypes.CDLL(hip.LIB_PATH)
    assert not hasattr(L, "ffq_read_probe")
    blob = open(hip.LIB_PATH, "rb").read()
    for name in (b"k_pipe_probe", b"k_read_probe"):
        assert name not in blob, name
    text = open(os.path.join(ROOT, "include", "ffq.h")).read()
    assert "ffq_read_probe" not in text


def test_abi_version_and_status_codes(hip):
    assert hip.lib().ffq_abi_version() == hip.ABI_VERSION
    text = open(os.path.join(ROOT, "include", "ffq.h")).read()
    for name, val in (("FFQ_INVALID", "(-1)"), ("FFQ_COMPLETE", "6"), ("FFQ_MISSING_QUALHEADER_END", "7"),
                      ("FFQ_POS_QUAL_END", "5")):
        assert re.search(r"#define\s+%s\s+%s" % (name, re.escape(val)), text), name


def test_scan_result_layout(hip):
    # mirrors struct ffq_scan_result in include/ffq.h
    assert ctypes.sizeof(hip.ScanResult) == 8 * 3 + 48 + 4 * 4 + 8 + 4 * 4


def test_no_gpu_fails_loudly(hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip.FFQError) as ei:
        hip.Context(0)
    assert ei.value.code == hip.E_NODEVICE
    assert "no CPU fallback" in str(ei.value)
    from fastqandfurious_amd import _fastqandfurious
    from array import array
    with pytest.raises(hip.FFQError):
        _fastqandfurious.entrypos(b"\n@a\nA\n+\nI\n@", 0, array("q", [-1] * 6))
    with pytest.raises(hip.FFQError):
        _fastqandfurious.arrayadd_b(array("b", [1, 2, 3]), -33)


def test_product_never_imports_oracle():
    """The product package must not reference the oracle (or any CPU scanner
    library) anywhere."""
    pk = os.path.join(ROOT, "fastq-and-furious_amd")
    for dirpath, _, files in os.walk(pk):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dirpath, fn)


def test_mirror_module_surface(pkg):
    from fastqandfurious_amd import _fastqandfurious as C
    assert (C.INVALID, C.POS_HEAD_BEG, C.POS_HEAD_END, C.POS_SEQ_BEG, C.POS_SEQ_END, C.POS_QUAL_BEG,
            C.POS_QUAL_END, C.COMPLETE, C.MISSING_QUALHEADER_END) == (-1, 0, 1, 2, 3, 4, 5, 6, 7)
    assert callable(C.entrypos) and callable(C.arrayadd_b) and callable(C.arrayadd_q)
    from array import array
    # argument checking happens before any device work (reference messages)
    with pytest.raises(ValueError, match="format type q"):
        C.entrypos(b"x", 0, array("b", [0] * 48))
    with pytest.raises(ValueError, match="format type b"):
        C.arrayadd_b(array("q", [1]), 1)
    with pytest.raises(ValueError, match="format type q"):
        C.arrayadd_q(array("b", [1]), 1)
    with pytest.raises(OverflowError):
        C.arrayadd_b(array("b", [1]), 40000)
    with pytest.raises(TypeError):
        C.arrayadd_b(array("b", [1]), 1.5)


def build_c_example(out_dir):
    """examples/ffq_count.c compiled as C99 and linked against the in-tree library: the boundary is a C ABI, not a
    Python extension.  Returns the binary's path (None without gcc)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        return None
    libdir = os.path.join(ROOT, "fastq-and-furious_amd", "csrc")
    exe = os.path.join(str(out_dir), "ffq_count")
    subprocess.run([gcc, "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "ffq_count.c"), "-o", exe, "-L", libdir, "-lffq_hip", "-Wl,-rpath," + libdir],
                   check=True, capture_output=True, text=True)
    return exe


def test_plain_c_program_links_against_the_library(hip, tmp_path):
    import subprocess
    exe = build_c_example(tmp_path)
    if exe is None:
        pytest.skip("no gcc")
    r = subprocess.run([exe, "-v"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split()[:3] == ["ffq", "abi", str(hip.ABI_VERSION)]

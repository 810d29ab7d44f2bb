"""The checker itself under ASan + UBSan ("ASan/UBSan build in tests"): the
oracle's C restatement is compiled with -fsanitize=address,undefined (oracle/Makefile, target
`asan`) and driven over the golden corpora -- every prefix of the templates, the edge cases, the
fuzz cases, the reference fixtures -- in a subprocess with the ASan runtime preloaded.  Any
out-of-bounds read (the reference's own scanner has one, _fastqandfurious.c:70-71 with length
(size_t)-1), signed overflow or misaligned access aborts the subprocess."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import ctypes, json, os, sys
import numpy as np
root = sys.argv[1]
L = ctypes.CDLL(os.path.join(root, "oracle", "libffq_oracle_asan.so"))
i64, vp = ctypes.c_int64, ctypes.c_void_p
L.ffq_oracle_entrypos_c.argtypes = [vp, i64, i64, vp]; L.ffq_oracle_entrypos_c.restype = ctypes.c_int
L.ffq_oracle_entrypos_py.argtypes = [vp, i64, i64, vp]; L.ffq_oracle_entrypos_py.restype = ctypes.c_int
L.ffq_oracle_scan.argtypes = [vp, i64, ctypes.c_int, i64, ctypes.c_int, ctypes.c_int, i64, vp, i64, vp]
L.ffq_oracle_entrypos_fasta.argtypes = [vp, i64, i64, vp]; L.ffq_oracle_entrypos_fasta.restype = ctypes.c_int
L.ffq_oracle_decode_quals.argtypes = [vp, vp, i64, ctypes.c_int, vp, vp]
L.ffq_oracle_gather_column.argtypes = [vp, vp, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
libc = ctypes.CDLL(None)
libc.malloc.restype = vp; libc.malloc.argtypes = [ctypes.c_size_t]; libc.free.argtypes = [vp]
class exact:                       # a malloc'ed copy of exactly len(b) bytes: ASan traps the first byte past it
    def __init__(self, b):
        self.size = len(b)
        self.p = libc.malloc(max(self.size, 1))
        ctypes.memmove(self.p, bytes(b), self.size)
        self.ctypes = self
        self.data = self.p
    def __del__(self):
        libc.free(self.p)
g = json.load(open(os.path.join(root, "tests", "golden", "golden.json")))
calls = 0
def scan_all(data):
    global calls
    a = exact(data)
    for sentinel in (0, 1):
        for eof in (0, 1):
            for variant in (0, 1):
                cap = a.size // 4 + 2
                table = np.empty((cap, 6), dtype=np.int64); out = np.zeros(4, dtype=np.int64)
                L.ffq_oracle_scan(a.ctypes.data if a.size else None, a.size, sentinel, 0, eof, variant, 0, table.ctypes.data, cap, out.ctypes.data)
                calls += 1
    n = int(out[0])
    if n:
        t = np.ascontiguousarray(table[:n]); base = exact(b"\n" + bytes(data))
        tot = int((t[:, 5] - t[:, 4]).sum())
        q = np.empty(max(tot, 1), dtype=np.int8); off = np.empty(n + 1, dtype=np.int64)
        L.ffq_oracle_decode_quals(base.ctypes.data, t.ctypes.data, n, -33, q.ctypes.data, off.ctypes.data)
        L.ffq_oracle_gather_column(base.ctypes.data, t.ctypes.data, n, 0, 1, 1, 0, q.ctypes.data, off.ctypes.data) if tot >= int((t[:, 1] - t[:, 0] - 1).clip(0).sum()) else None
for tpl in g["templates"]:
    buf = bytes.fromhex(tpl["buf"])
    for cut in range(len(buf) + 1):
        a = exact(buf[:cut]); pos = np.full(6, -1, dtype=np.int64)
        for off in range(0, cut + 1, max(1, cut // 7)):
            L.ffq_oracle_entrypos_c(a.ctypes.data if cut else None, cut, off, pos.ctypes.data)
            L.ffq_oracle_entrypos_py(a.ctypes.data if cut else None, cut, off, pos.ctypes.data)
            calls += 2
for tpl in g["fasta"]:
    buf = bytes.fromhex(tpl["buf"])
    for cut in range(len(buf) + 1):
        a = exact(buf[:cut]); pos = np.full(6, -1, dtype=np.int64)
        L.ffq_oracle_entrypos_fasta(a.ctypes.data if cut else None, cut, 0, pos.ctypes.data); calls += 1
for ent in g["edge"].values():
    scan_all(bytes.fromhex(ent["data"]))
for ent in g["fuzz"]:
    scan_all(bytes.fromhex(ent["data"]))
for fn in ("test.fq", "test_longqualityheader.fq", "test_multiline.fq"):
    data = open(os.path.join(root, "tests", "golden", "data", fn), "rb").read()
    for cut in range(0, len(data) + 1, 7):
        scan_all(data[:cut])
print("sanitized calls:", calls)
'''


def test_oracle_under_asan_ubsan(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no ASan runtime beside gcc")
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan + ((":" + ubsan) if os.path.isabs(ubsan) and os.path.exists(ubsan) else "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER)
    r = subprocess.run([sys.executable, str(drv), ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitized calls:" in r.stdout and int(r.stdout.rsplit(":", 1)[1]) > 5000
    # and the harness does trap: one byte read past a malloc'ed buffer must abort the subprocess
    bad = tmp_path / "bad.py"
    bad.write_text(DRIVER.split("g = json.load")[0] +
                   "a = exact(b'\\n@abc\\nACGT'); pos = np.full(6, -1, dtype=np.int64)\n"
                   "L.ffq_oracle_entrypos_c(a.ctypes.data, a.size + 40, 0, pos.ctypes.data)\nprint('survived')\n")
    r = subprocess.run([sys.executable, str(bad), ROOT], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "survived" not in r.stdout and "AddressSanitizer" in r.stderr

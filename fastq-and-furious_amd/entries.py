"""The default entryfunc over a whole offset table.

The reference builds one entry per scanner call in the interpreter
(/root/reference/src/fastqandfurious.py:161-171, called at :255).  With a scanner that returns
the table of a whole buffer fill, the three slices of every row are cut by one native call
(csrc/ffq_entries.c, CPython C API) and the iterator yields from the resulting list; without the
compiled module the same slices are cut here, in Python.  Host glue: the scan itself never runs
on the CPU.
"""
import importlib.machinery
import importlib.util
import os

import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# the interpreter's ABI tag is part of the file name (build.entries_lib): another Python's build is not loaded
_LIB = os.path.join(_HERE, "csrc", "_ffq_entries" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
_native = None
_tried = False


def native():
    """The compiled module, or None when csrc/_ffq_entries<EXT_SUFFIX> is not there or does not load (build.build_entries())."""
    global _native, _tried
    if not _tried:
        _tried = True
        if os.path.exists(_LIB):
            try:
                loader = importlib.machinery.ExtensionFileLoader("_ffq_entries", _LIB)
                spec = importlib.util.spec_from_loader("_ffq_entries", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _native = mod
            except ImportError:
                _native = None           # not loadable here: the Python slices below do the same
    return _native


def entries_python(buf, rows, shift=0, hskip=1):
    """[(buf[p0+hskip:p1], buf[p2:p3], buf[p4:p5]) for every row]; rows: flat sequence of positions."""
    it = iter(rows)
    return [(buf[p0 - shift + hskip:p1 - shift], buf[p2 - shift:p3 - shift], buf[p4 - shift:p5 - shift])
            for p0, p1, p2, p3, p4, p5 in zip(it, it, it, it, it, it)]


def entries(buf, rows, shift=0, hskip=1, cls=None):
    """List of (header, sequence, quality) `bytes` tuples, one per row of six int64 positions.
    `rows`: a C-contiguous buffer of int64 (array('q'), numpy); positions minus `shift` index `buf`;
    the header slice starts at pos[0] + hskip (1: without the '@', as entryfunc cuts it); cls: a namedtuple
    class the entries are instances of (entryfunc_namedtuple's `Entry`)."""
    mod = native()
    if mod is not None:
        return mod.entries(buf, rows, shift, hskip, cls)
    if not isinstance(buf, bytes):
        buf = bytes(buf)
    flat = memoryview(rows).cast("B").cast("q")
    out = entries_python(buf, flat, shift, hskip)
    return out if cls is None else [cls(*t) for t in out]

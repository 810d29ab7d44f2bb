"""Load the REAL reference (Python module + C extension) for validating the
oracle and for generating golden vectors.  load_py works only where
/root/reference is mounted (the build container); load_ext only needs oracle/_ref/, which
travels to the GPU box (bench.py's cpu_baseline leg times it there).

The Python module is imported from where it lies (never copied); the C
extension is the one oracle/Makefile builds into oracle/_ref/.
"""
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("FFQ_REFERENCE", "/root/reference")


def have_reference_py():
    return os.path.exists(os.path.join(REF_ROOT, "src", "fastqandfurious.py"))


def have_reference_ext():
    return os.path.exists(os.path.join(_HERE, "_ref", "_fastqandfurious.so"))


def load_py():
    path = os.path.join(REF_ROOT, "src", "fastqandfurious.py")
    spec = importlib.util.spec_from_file_location("_reference_fastqandfurious", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ext():
    path = os.path.join(_HERE, "_ref", "_fastqandfurious.so")
    loader = importlib.machinery.ExtensionFileLoader("_fastqandfurious", path)
    spec = importlib.util.spec_from_file_location("_fastqandfurious", path, loader=loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod

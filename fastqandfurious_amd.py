"""Import shim: makes the package directory `fastq-and-furious_amd/` importable
as `fastqandfurious_amd` (a hyphen cannot appear in a module name).

    import fastqandfurious_amd
    from fastqandfurious_amd import fastqandfurious, _fastqandfurious
"""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fastq-and-furious_amd")
_spec = importlib.util.spec_from_file_location(
    "fastqandfurious_amd", os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fastqandfurious_amd"] = _mod
_spec.loader.exec_module(_mod)

"""fastqandfurious_amd -- MI355X-native FASTQ buffer-scan path behind the
plug-in API of lgautier/fastq-and-furious.

    fastqandfurious    mirror of the reference Python module (iterator,
                       entryfunc*, pure-Python entrypos, status constants)
    _fastqandfurious   mirror of the reference C extension, on the GPU
                       (entrypos, arrayadd_b, arrayadd_q)
    hip                ctypes binding of libffq_hip.so (include/ffq.h)
    synth              synthetic FASTQ generators (numpy twins of the device ones)
    build              compiles csrc/ for gfx950

The directory is named fastq-and-furious_amd (not importable as written); the
repo root carries fastqandfurious_amd.py which registers it under this name.
"""
__version__ = "0.1.0"

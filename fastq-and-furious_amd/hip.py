"""ctypes binding of libffq_hip.so -- the C ABI declared in include/ffq.h.

This module is plumbing only: it loads the in-tree shared library, declares the
argument types and turns error codes into exceptions.  There is no CPU
fallback: if the library is missing or no gfx950 device is usable, everything
here raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FFQ_HIP_LIB: another build of the same library (same-box A/B measurements, tools/ab_run.sh)
LIB_PATH = os.environ.get("FFQ_HIP_LIB") or os.path.join(_HERE, "csrc", "libffq_hip.so")

# status codes (include/ffq.h; reference: _fastqandfurious.c:7-15)
INVALID = -1
POS_HEAD_BEG = 0
POS_HEAD_END = 1
POS_SEQ_BEG = 2
POS_SEQ_END = 3
POS_QUAL_BEG = 4
POS_QUAL_END = 5
COMPLETE = 6
MISSING_QUALHEADER_END = 7

END_OK, END_REFILL, END_ERR_FINAL_QUAL, END_ERR_INCOMPLETE, END_ERR_INVALID = range(5)

ABI_VERSION = 6
OK = 0
E_NODEVICE, E_HIP, E_ARG, E_NOMEM, E_TABLE_FULL, E_INTERNAL, E_TIMEOUT = -1, -2, -3, -4, -5, -6, -7
STAGE_NONE, STAGE_HANDOFF, STAGE_SCAN, STAGE_GATHER = 0, 1, 2, 3      # FFQ_SHARD_STAGE_*
STAGE_NAMES = ("none", "hand-off", "scan", "gather")

F_DECODE_QUAL = 1
F_FORCE_SERIAL = 2
F_FORCE_RANKED = 4
F_POLL_RESULT = 8
F_SINGLE_PASS = 16
SEG_STRIDE = 8704            # FFQ_F_SINGLE_PASS: bytes of the quality buffer every 16 KiB tile of the input owns (include/ffq.h)
INPLACE_STRIDE = 16384       # ... and what admits the in-place layout as well (lines of any length)
PATH_IN_PLACE = 8            # ScanResult.path bit (FFQ_PATH_IN_PLACE): the general path's index pass decoded every byte in place
F_NO_TIMING = 32
F_FORCE_GENERAL = 64        # tests: skip the four-line fast path

# every symbol include/ffq.h declares (tests check the library exports them all)
SYMBOLS = (
    "ffq_abi_version", "ffq_build_id", "ffq_last_error", "ffq_device_count", "ffq_ctx_create", "ffq_ctx_create_shared",
    "ffq_ctx_destroy", "ffq_ctx_reserve", "ffq_ctx_forget", "ffq_ctx_stream", "ffq_dev_alloc", "ffq_dev_free",
    "ffq_pinned_alloc", "ffq_pinned_free", "ffq_copy_h2d", "ffq_copy_d2h", "ffq_sync",
    "ffq_scan_device", "ffq_scan_submit", "ffq_scan_wait", "ffq_scan_host", "ffq_entrypos", "ffq_arrayadd_b_device",
    "ffq_arrayadd_b", "ffq_arrayadd_q_device", "ffq_arrayadd_q", "ffq_table_lower_bound",
    "ffq_table_select_seqlen", "ffq_table_cut", "ffq_table_gather_column", "ffq_stream_open", "ffq_stream_next", "ffq_stream_close",
    "ffq_stream_open2", "ffq_stream_quals", "ffq_stream_tell", "ffq_stream_path",
    "ffq_shard_unique_id", "ffq_shard_create", "ffq_shard_create_lane", "ffq_shard_world_create", "ffq_shard_world_abort",
    "ffq_shard_world_destroy", "ffq_shard_create_local", "ffq_shard_destroy", "ffq_shard_halo", "ffq_shard_exchange_halo",
    "ffq_shard_step_submit", "ffq_shard_step_wait", "ffq_shard_transport", "ffq_shard_self_exchange", "ffq_stream_open_gzip", "ffq_gunzip_fd", "ffq_gunzip_stats", "ffq_bgzf_range", "ffq_stream_open_push",
    "ffq_stream_push_buffer", "ffq_stream_push", "ffq_scan_fasta_device", "ffq_scan_fasta_host",
    "ffq_synth_single",
    "ffq_synth_wrapped_size", "ffq_synth_wrapped", "ffq_selftest", "ffq_shard_load_fd", "ffq_load_fd",
    "ffq_shard_host_step", "ffq_shard_host_free", "ffq_stream_set_filter", "ffq_stream_selected",
    "ffq_shard_create_hosted",
    "ffq_shard_create2", "ffq_shard_scan_fd_slabs", "ffq_table_select_seqlen_idx", "ffq_shard_get_info", "ffq_shard_set_timeout", "ffq_shard_set_serial", "ffq_shard_abort", "ffq_shard_inject_stall",
)


class ScanResult(ctypes.Structure):
    _fields_ = [
        ("n_records", ctypes.c_int64),
        ("n_qual_bytes", ctypes.c_int64),
        ("end_offset", ctypes.c_int64),
        ("last_pos", ctypes.c_int64 * 6),
        ("last_status", ctypes.c_int32),
        ("end_state", ctypes.c_int32),
        ("path", ctypes.c_int32),
        ("retries", ctypes.c_int32),
        ("n_lines", ctypes.c_int64),
        ("ms_index", ctypes.c_float),
        ("ms_chain", ctypes.c_float),
        ("ms_decode", ctypes.c_float),
        ("ms_total", ctypes.c_float),
    ]


class ShardResult(ctypes.Structure):
    """ffq_shard_result (include/ffq.h)."""
    _fields_ = [
        ("scan", ScanResult),
        ("n_rows", ctypes.c_int64), ("row_lo", ctypes.c_int64), ("row_hi", ctypes.c_int64),
        ("exit_pos", ctypes.c_int64), ("first_pos", ctypes.c_int64),
        ("n_own_records", ctypes.c_int64), ("record_base", ctypes.c_int64), ("total_records", ctypes.c_int64),
        ("err_byte", ctypes.c_int64),
        ("err_state", ctypes.c_int32), ("rounds", ctypes.c_int32), ("regathers", ctypes.c_int32), ("halo_source", ctypes.c_int32),
        ("handoff_bytes", ctypes.c_int64),
        ("handoff_ms", ctypes.c_float), ("allgather_ms", ctypes.c_float),
        ("d_ext", ctypes.c_void_p), ("tail", ctypes.c_int64), ("head", ctypes.c_int64),
        ("nranks", ctypes.c_int32), ("serial", ctypes.c_int32),
        ("n_slabs", ctypes.c_int64), ("bytes_read", ctypes.c_int64),
    ]


class ShardInfo(ctypes.Structure):
    """ffq_shard_info (include/ffq.h): who is there -- the communicators' rank counts, every rank's PCI bus id --, the
    mode, the watchdog's deadline and where its last trip found the step."""
    _fields_ = [
        ("rank", ctypes.c_int32), ("world", ctypes.c_int32),
        ("nranks_handoff", ctypes.c_int32), ("nranks_gather", ctypes.c_int32),
        ("serial", ctypes.c_int32), ("last_stage", ctypes.c_int32), ("poisoned", ctypes.c_int32), ("n_bus", ctypes.c_int32),
        ("timeout_s", ctypes.c_double),
        ("bus_id", ctypes.c_int64 * 64),
    ]


class ShardPiece(ctypes.Structure):
    """ffq_shard_piece: stream bytes [a, b) go from rank src to rank dst; ptr: where this rank reads / writes them."""
    _fields_ = [("src", ctypes.c_int32), ("dst", ctypes.c_int32), ("a", ctypes.c_int64), ("b", ctypes.c_int64), ("ptr", ctypes.c_void_p)]


_SCAN_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                            ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(ScanResult))
_EXCHANGE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ShardPiece), ctypes.c_int)
_GATHER_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64))


class ShardHostOps(ctypes.Structure):
    """ffq_shard_host_ops (include/ffq.h)."""
    _fields_ = [("user", ctypes.c_void_p), ("scan", _SCAN_CB), ("exchange", _EXCHANGE_CB), ("allgather", _GATHER_CB)]


class FFQError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libffq_hip error %d: %s" % (code, msg))
        self.code = code


class FFQTimeout(FFQError, TimeoutError):
    """A shard step's watchdog (E_TIMEOUT): the message names the stage (hand-off / scan / gather), the transport, the mode
    and the ranks whose words are missing.  The shard is poisoned: abort() + close(), then build a new one."""


class FFQGzipError(FFQError, OSError):
    """Corrupt gzip input met by the stream front end's reader: also an OSError, as gzip.BadGzipFile is."""


class FFQGzipTruncated(FFQError, EOFError):
    """A gzip file that ends before its end-of-stream marker: also an EOFError, as Python's gzip raises."""


_lib = None
_probe = False      # use_probe_build(): tools load the instrumented build instead of the product library


def _share_torch_hip_runtime():
    """One process, one HIP runtime.  PyTorch-ROCm bundles its own libamdhip64; if this
    library's (the system ROCm one) initialises first, torch later finds "No HIP GPUs".  When
    torch is installed, load ITS runtime before libffq_hip.so so that both resolve to it --
    the order that `import torch` first gives anyway.  Without torch nothing happens."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass


def use_probe_build():
    """tools/ only: load libffq_probe.so (the same sources compiled with -DFFQ_PROBES: ablation
    switches, read / look-back / pipeline probes; include/ffq_probe.h) in place of the product
    library.  Must be called before the first use of lib()."""
    global _probe, LIB_PATH
    assert _lib is None, "use_probe_build() must come before the library is loaded"
    from . import build as _build
    _probe = True
    LIB_PATH = _build.build_probe()


def _checked_build():
    """The in-tree library must be the build of the sources in the tree: compare the id baked
    into it (ffq_build_id) with the hash of csrc/ + include/ and rebuild on a mismatch; if that
    is not possible, refuse.  (FFQ_HIP_LIB names another build on purpose: not checked.)"""
    if os.environ.get("FFQ_HIP_LIB") or _probe:
        return
    from . import build as _build
    want = _build.source_id()
    if os.path.exists(LIB_PATH) and _build.built_id(LIB_PATH) == want:
        return
    if os.environ.get("FFQ_TRUST_BUILD") == "1" and os.path.exists(LIB_PATH):
        # a read-only install / a box without hipcc: the caller takes responsibility (the ABI version is still checked)
        import warnings
        warnings.warn("%s carries build id %s, the sources in this tree hash to %s: loaded as it is (FFQ_TRUST_BUILD=1)"
                      % (LIB_PATH, _build.built_id(LIB_PATH), want))
        return
    try:
        _build.build()
    except Exception as e:      # noqa: BLE001
        raise FFQError(E_NODEVICE,
                       "%s is missing or was not built from the sources in this tree (build id %s, "
                       "sources %s) and cannot be rebuilt here: %s.  Build it with `python "
                       "fastq-and-furious_amd/build.py` (there is no CPU fallback for the scan path)"
                       % (LIB_PATH, _build.built_id(LIB_PATH), want, e))
    if _build.built_id(LIB_PATH) != want:
        raise FFQError(E_INTERNAL, "%s does not carry the build id of its sources after a rebuild" % LIB_PATH)


class _MissingExport:
    """An export the library named by FFQ_HIP_LIB does not have: an error when CALLED, not when bound."""

    def __init__(self, name):
        self._name = name
        self.argtypes = self.restype = None

    def __call__(self, *a):
        raise FFQError(E_INTERNAL, "%s: the library named by FFQ_HIP_LIB (%s) does not export it" % (self._name, LIB_PATH))


class _OtherBuild:
    """FFQ_HIP_LIB names another build on purpose (tools/ab_*.sh: an OLDER commit's library under this tree's Python for a
    same-box A/B): what it lacks is bound to _MissingExport.  The in-tree library never goes through this."""

    def __init__(self, cdll):
        self.__dict__["_cdll"] = cdll

    def __getattr__(self, name):
        try:
            f = getattr(self._cdll, name)
        except AttributeError:
            f = _MissingExport(name)
        self.__dict__[name] = f
        return f


def build_id():
    """The source hash baked into the loaded library (== build.source_id() of this tree)."""
    return lib().ffq_build_id().decode("ascii")


def lib():
    """The loaded library.  Raises if it has not been built and cannot be."""
    global _lib
    if _lib is None:
        if os.environ.get("FFQ_USE_PROBE_BUILD") == "1" and not _probe:
            use_probe_build()            # (tools/*.sh: the ablation switches live in the instrumented build)
        _share_torch_hip_runtime()
        _checked_build()
        if not os.path.exists(LIB_PATH):
            raise FFQError(E_NODEVICE,
                           "%s is missing: build it with `python fastq-and-furious_amd/build.py` "
                           "(there is no CPU fallback for the scan path)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        if os.environ.get("FFQ_HIP_LIB"):
            L = _OtherBuild(L)
        vp, i64, i32, u32, u64 = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_uint32,
                                  ctypes.c_uint64)
        P = ctypes.POINTER
        L.ffq_abi_version.restype = i32
        L.ffq_build_id.restype = ctypes.c_char_p
        L.ffq_last_error.restype = ctypes.c_char_p
        L.ffq_device_count.restype = i32
        L.ffq_ctx_create.argtypes = [i32, P(vp)]
        L.ffq_ctx_create_shared.argtypes = [vp, P(vp)]
        L.ffq_ctx_destroy.argtypes = [vp]
        L.ffq_ctx_destroy.restype = None
        L.ffq_ctx_reserve.argtypes = [vp, i64]
        L.ffq_ctx_forget.argtypes = [vp]
        L.ffq_ctx_forget.restype = None
        L.ffq_ctx_stream.argtypes = [vp]
        L.ffq_ctx_stream.restype = vp
        L.ffq_dev_alloc.argtypes = [vp, i64, P(vp)]
        L.ffq_dev_free.argtypes = [vp, vp]
        L.ffq_pinned_alloc.argtypes = [i64, P(vp)]
        L.ffq_pinned_free.argtypes = [vp]
        L.ffq_copy_h2d.argtypes = [vp, vp, vp, i64, i32]
        L.ffq_copy_d2h.argtypes = [vp, vp, vp, i64, i32]
        L.ffq_sync.argtypes = [vp]
        L.ffq_scan_device.argtypes = [vp, vp, i64, i32, i64, i32, i64, u32, i32, vp, i64, vp, i64, vp,
                                      P(ScanResult)]
        L.ffq_scan_submit.argtypes = [vp, vp, i64, i32, i64, i32, i64, u32, i32, vp, i64, vp, i64, vp]
        L.ffq_scan_wait.argtypes = [vp, P(ScanResult)]
        L.ffq_scan_host.argtypes = [vp, vp, i64, i32, i64, i32, i64, u32, i32, vp, i64, vp, i64, vp,
                                    P(ScanResult)]
        L.ffq_entrypos.argtypes = [vp, vp, i64, i64, vp, P(i32)]
        L.ffq_arrayadd_b_device.argtypes = [vp, vp, i64, i32]
        L.ffq_arrayadd_b.argtypes = [vp, vp, i64, i32]
        L.ffq_arrayadd_q_device.argtypes = [vp, vp, i64, i64]
        L.ffq_arrayadd_q.argtypes = [vp, vp, i64, i64]
        L.ffq_table_lower_bound.argtypes = [vp, vp, i64, i32, i64, P(i64)]
        L.ffq_table_select_seqlen.argtypes = [vp, vp, i64, i64, i64, vp, P(i64)]
        L.ffq_table_select_seqlen_idx.argtypes = [vp, vp, i64, i64, i64, vp, vp, P(i64)]
        L.ffq_table_cut.argtypes = [vp, vp, i64, i64, i64, P(i64)]
        L.ffq_table_gather_column.argtypes = [vp, vp, i64, i32, i64, vp, i64, i32, i32, i32, i32, vp, i64, vp, P(i64)]
        L.ffq_stream_open.argtypes = [vp, i32, i64, P(vp)]
        L.ffq_stream_next.argtypes = [vp, P(vp), P(i64), P(i32), P(i64), P(vp), P(i64), P(i64)]
        L.ffq_stream_close.argtypes = [vp]
        L.ffq_scan_fasta_device.argtypes = [vp, vp, i64, i32, i64, i64, vp, i64, P(ScanResult)]
        L.ffq_scan_fasta_host.argtypes = [vp, vp, i64, i32, i64, i64, vp, i64, P(ScanResult)]
        L.ffq_stream_open2.argtypes = [vp, i32, i64, u32, i32, i64, P(vp)]
        L.ffq_stream_open_gzip.argtypes = [vp, i32, i64, u32, i32, i64, P(vp)]
        L.ffq_gunzip_fd.argtypes = [i32, vp, i64, i64, i32, P(i64)]
        L.ffq_gunzip_fd.restype = i64
        L.ffq_gunzip_stats.argtypes = [P(i64)]
        L.ffq_gunzip_stats.restype = None
        L.ffq_bgzf_range.argtypes = [i32, i64, i64, vp, i64, i32, P(i64), P(i64), P(i64), P(i64)]
        L.ffq_stream_open_push.argtypes = [vp, i64, u32, i32, P(vp)]
        L.ffq_stream_push_buffer.argtypes = [vp, P(vp), P(i64)]
        L.ffq_stream_push.argtypes = [vp, i64, i32]
        L.ffq_stream_tell.argtypes = [vp]
        L.ffq_stream_tell.restype = i64
        L.ffq_stream_path.argtypes = [vp]
        L.ffq_stream_path.restype = i32
        L.ffq_shard_unique_id.argtypes = [vp]
        L.ffq_shard_create.argtypes = [vp, vp, i32, i32, P(i64), i64, i64, P(vp)]
        L.ffq_shard_create2.argtypes = [vp, vp, i32, i32, P(i64), i64, i64, u32, P(vp)]
        L.ffq_shard_create_lane.argtypes = [vp, vp, P(vp)]
        L.ffq_shard_world_create.argtypes = [i32, P(vp)]
        L.ffq_shard_world_abort.argtypes = [vp]
        L.ffq_shard_world_abort.restype = None
        L.ffq_shard_world_destroy.argtypes = [vp]
        L.ffq_shard_world_destroy.restype = None
        L.ffq_shard_create_local.argtypes = [vp, vp, i32, P(i64), i64, i64, P(vp)]
        L.ffq_shard_destroy.argtypes = [vp]
        L.ffq_shard_destroy.restype = None
        L.ffq_shard_halo.argtypes = [vp, P(i64), P(i64)]
        L.ffq_shard_exchange_halo.argtypes = [vp, vp, i32]
        L.ffq_shard_step_submit.argtypes = [vp, vp, i32, u32, i32, vp, i64, vp, i64, vp]
        L.ffq_shard_step_wait.argtypes = [vp, P(ShardResult)]
        L.ffq_shard_self_exchange.argtypes = [vp, vp, vp, i64]
        L.ffq_shard_transport.argtypes = [vp]
        L.ffq_shard_transport.restype = ctypes.c_char_p
        L.ffq_shard_get_info.argtypes = [vp, P(ShardInfo)]
        L.ffq_shard_set_timeout.argtypes = [vp, ctypes.c_double]
        L.ffq_shard_set_serial.argtypes = [vp, i32]
        L.ffq_shard_abort.argtypes = [vp]
        L.ffq_shard_inject_stall.argtypes = [vp, i32, ctypes.c_double]
        L.ffq_shard_load_fd.argtypes = [vp, i32, vp, P(i64)]
        L.ffq_shard_scan_fd_slabs.argtypes = [vp, i32, i64, u32, vp, i64, P(ShardResult)]
        L.ffq_load_fd.argtypes = [vp, i32, i64, i64, vp, P(i64)]
        L.ffq_shard_host_step.argtypes = [P(ShardHostOps), vp, i32, i32, P(i64), i64, i64, vp, vp, i64, P(ShardResult)]
        L.ffq_shard_create_hosted.argtypes = [vp, P(ShardHostOps), i32, i32, P(i64), i64, i64, P(vp)]
        L.ffq_shard_host_free.argtypes = [vp]
        L.ffq_shard_host_free.restype = None
        L.ffq_stream_set_filter.argtypes = [vp, i64, i64, i32, i32]
        L.ffq_stream_selected.argtypes = [vp, P(vp), P(i64), P(vp), P(vp), P(i64)]
        L.ffq_stream_quals.argtypes = [vp, P(vp), P(vp), P(i64)]
        L.ffq_stream_close.restype = None
        L.ffq_synth_single.argtypes = [vp, vp, i64, i64, u64]
        L.ffq_synth_wrapped_size.argtypes = [i64, u64]
        L.ffq_synth_wrapped_size.restype = i64
        L.ffq_synth_wrapped.argtypes = [vp, vp, vp, i64, i64, u64]
        if _probe:
            L.ffq_read_probe.argtypes = [vp, vp, i64, i32, i32, P(ctypes.c_float)]
        L.ffq_selftest.argtypes = [vp]
        _lib = L
    return _lib


def last_error():
    """the calling thread's last error text (ffq_last_error)"""
    return lib().ffq_last_error().decode("utf-8", "replace")


def check(rc, allow=()):
    if rc != OK and rc not in allow:
        msg = lib().ffq_last_error().decode("utf-8", "replace")
        if "gzip: " in msg:
            # what a drop-in caller of gzip.open() catches: EOFError for a file cut short, BadGzipFile (an OSError) otherwise
            raise (FFQGzipTruncated if "ended before the end-of-stream marker" in msg else FFQGzipError)(rc, msg)
        raise (FFQTimeout if rc == E_TIMEOUT else FFQError)(rc, msg)
    return rc


class Context:
    """One GPU context (stream + scratch).  Not thread-safe; one per thread."""

    def __init__(self, device=0, share=None):
        """share: another Context whose HIP streams this one uses (own scratch); scans of
        the two then execute in submission order (see scan_submit / scan_wait)."""
        self._h = ctypes.c_void_p()
        self._parent = share
        self._children = []          # contexts on this one's streams: they are closed first
        if share is not None:
            self.device = share.device
            check(lib().ffq_ctx_create_shared(share.handle, ctypes.byref(self._h)))
            import weakref
            share._children.append(weakref.ref(self))
        else:
            self.device = device
            check(lib().ffq_ctx_create(int(device), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            for ref in self._children:
                child = ref()
                if child is not None:
                    child.close()
            self._children = []
            lib().ffq_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        if not self._h:
            raise FFQError(E_ARG, "context is closed")
        return self._h

    def reserve(self, max_bytes):
        check(lib().ffq_ctx_reserve(self.handle, int(max_bytes)))

    def stream(self):
        """The HIP stream (hipStream_t as an integer) every scan of this context runs on."""
        return int(lib().ffq_ctx_stream(self.handle) or 0)

    def forget(self):
        """Drop the context's memory of what its recent input looked like."""
        lib().ffq_ctx_forget(self.handle)

    def selftest(self):
        check(lib().ffq_selftest(self.handle))

    # ---- raw device memory (for hosts without torch) -------------------
    def dev_alloc(self, nbytes):
        p = ctypes.c_void_p()
        check(lib().ffq_dev_alloc(self.handle, int(nbytes), ctypes.byref(p)))
        return p.value

    def dev_free(self, ptr):
        check(lib().ffq_dev_free(self.handle, ctypes.c_void_p(ptr)))

    def h2d(self, dptr, arr):
        a = np.ascontiguousarray(arr)
        check(lib().ffq_copy_h2d(self.handle, ctypes.c_void_p(dptr), a.ctypes.data, a.nbytes, 0))

    def d2h(self, arr, dptr):
        assert arr.flags.c_contiguous
        check(lib().ffq_copy_d2h(self.handle, arr.ctypes.data, ctypes.c_void_p(dptr), arr.nbytes, 0))

    def load_fd(self, fd, pos, n_bytes, d_dst):
        """Bytes [pos, pos + n_bytes) of the file behind fd into device memory (ffq_load_fd); returns the bytes loaded
        (less than n_bytes: the file ended)."""
        n = ctypes.c_int64()
        check(lib().ffq_load_fd(self.handle, int(fd), int(pos), int(n_bytes), ctypes.c_void_p(d_dst), ctypes.byref(n)))
        return n.value

    def sync(self):
        check(lib().ffq_sync(self.handle))

    # ---- hot path --------------------------------------------------------
    def scan_device(self, d_buf, n_bytes, d_table, table_cap, sentinel=True, offset=0, eof=True,
                    add=None, flags=0, qual_add=-33, d_qual=None, qual_cap=0, d_qoff=None):
        """Record chain over a device-resident buffer (raw device pointers).

        Returns (rc, ScanResult); rc is OK or E_TABLE_FULL."""
        if add is None:
            add = -1 if sentinel else 0
        res = ScanResult()
        rc = lib().ffq_scan_device(self.handle, ctypes.c_void_p(d_buf), int(n_bytes), int(bool(sentinel)),
                                   int(offset), int(bool(eof)), int(add), int(flags), int(qual_add),
                                   ctypes.c_void_p(d_table), int(table_cap),
                                   ctypes.c_void_p(d_qual) if d_qual else None, int(qual_cap),
                                   ctypes.c_void_p(d_qoff) if d_qoff else None, ctypes.byref(res))
        check(rc, allow=(E_TABLE_FULL,))
        return rc, res

    def scan_submit(self, d_buf, n_bytes, d_table, table_cap, sentinel=True, offset=0, eof=True,
                    add=None, flags=0, qual_add=-33, d_qual=None, qual_cap=0, d_qoff=None):
        """Enqueue a scan and return at once; scan_wait() completes it."""
        if add is None:
            add = -1 if sentinel else 0
        check(lib().ffq_scan_submit(self.handle, ctypes.c_void_p(d_buf), int(n_bytes), int(bool(sentinel)),
                                    int(offset), int(bool(eof)), int(add), int(flags), int(qual_add),
                                    ctypes.c_void_p(d_table), int(table_cap),
                                    ctypes.c_void_p(d_qual) if d_qual else None, int(qual_cap),
                                    ctypes.c_void_p(d_qoff) if d_qoff else None))

    def scan_wait(self):
        res = ScanResult()
        rc = lib().ffq_scan_wait(self.handle, ctypes.byref(res))
        check(rc, allow=(E_TABLE_FULL,))
        return rc, res

    def scan_host(self, buf, sentinel=True, offset=0, eof=True, add=None, flags=0, qual_add=-33,
                  table_cap=None, qual_room=None):
        """Record chain over a host bytes-like object.

        qual_room: with F_SINGLE_PASS, bytes of the quality buffer per 16 KiB tile of input (SEG_STRIDE; INPLACE_STRIDE
        admits the in-place layout, i.e. long lines, as well).
        Returns (table int64[n,6], ScanResult[, qual int8[], qoff int64[n+1]])."""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
        if add is None:
            add = -1 if sentinel else 0
        decode = bool(flags & F_DECODE_QUAL)
        cap = int(table_cap) if table_cap is not None else max(a.size // 64 + 16, 16)
        while True:
            table = np.empty((cap, 6), dtype=np.int64)
            nq = a.size if decode else 0
            if decode and (flags & F_SINGLE_PASS):
                nq = max(nq, ((a.size + 16383) >> 14) * int(qual_room or SEG_STRIDE))     # room for every tile's segment
            qual = np.empty(nq, dtype=np.int8)
            qoff = np.empty(cap + 1 if decode else 0, dtype=np.int64)
            res = ScanResult()
            rc = lib().ffq_scan_host(self.handle, a.ctypes.data if a.size else None, a.size,
                                     int(bool(sentinel)), int(offset), int(bool(eof)), int(add),
                                     int(flags), int(qual_add), table.ctypes.data, cap,
                                     qual.ctypes.data if decode else None, qual.size,
                                     qoff.ctypes.data if decode else None, ctypes.byref(res))
            check(rc, allow=(E_TABLE_FULL,))
            if rc == E_TABLE_FULL and table_cap is None:
                cap = int(res.n_records) + 1
                continue
            break
        n = min(int(res.n_records), cap)
        if decode:
            return table[:n], res, qual[:int(res.n_qual_bytes)], qoff[:n + 1]
        return table[:n], res

    def scan_fasta_host(self, buf, sentinel=False, offset=0, add=0, table_cap=None):
        """Every COMPLETE FASTA entry of a host buffer (the repeated entrypos_fasta call,
        reference fastqandfurious.py:103-143): (table int64[n,6] with pos4 = pos5 = -1,
        ScanResult with the last call's status / posbuffer / offset)."""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
        cap = int(table_cap) if table_cap is not None else max(a.size // 64 + 16, 16)
        while True:
            table = np.empty((cap, 6), dtype=np.int64)
            res = ScanResult()
            rc = lib().ffq_scan_fasta_host(self.handle, a.ctypes.data if a.size else None, a.size, int(bool(sentinel)),
                                           int(offset), int(add), table.ctypes.data, cap, ctypes.byref(res))
            check(rc, allow=(E_TABLE_FULL,))
            if rc == E_TABLE_FULL and table_cap is None:
                cap = int(res.n_records) + 1
                continue
            break
        return table[:min(int(res.n_records), cap)], res

    def scan_fasta_device(self, d_buf, n_bytes, d_table, table_cap, sentinel=False, offset=0, add=0):
        res = ScanResult()
        rc = lib().ffq_scan_fasta_device(self.handle, ctypes.c_void_p(d_buf), int(n_bytes), int(bool(sentinel)),
                                         int(offset), int(add), ctypes.c_void_p(d_table), int(table_cap),
                                         ctypes.byref(res))
        check(rc, allow=(E_TABLE_FULL,))
        return rc, res

    def entrypos(self, buf, offset, pos):
        """One scanner call on the GPU: fills pos (6 x int64), returns status."""
        a = np.frombuffer(buf, dtype=np.uint8)
        st = ctypes.c_int(0)
        p = np.empty(6, dtype=np.int64)
        check(lib().ffq_entrypos(self.handle, a.ctypes.data if a.size else None, a.size, int(offset),
                                 p.ctypes.data, ctypes.byref(st)))
        for i in range(6):
            pos[i] = int(p[i])
        return st.value

    def arrayadd_b(self, arr, value):
        a = np.frombuffer(arr, dtype=np.int8) if not isinstance(arr, np.ndarray) else arr
        check(lib().ffq_arrayadd_b(self.handle, a.ctypes.data, a.size, int(value)))

    def arrayadd_q(self, arr, value):
        a = np.frombuffer(arr, dtype=np.int64) if not isinstance(arr, np.ndarray) else arr
        v = (int(value) + 2**63) % 2**64 - 2**63
        check(lib().ffq_arrayadd_q(self.handle, a.ctypes.data, a.size, v))

    def arrayadd_b_device(self, dptr, n, value):
        check(lib().ffq_arrayadd_b_device(self.handle, ctypes.c_void_p(dptr), int(n), int(value)))

    def arrayadd_q_device(self, dptr, n, value):
        v = (int(value) + 2**63) % 2**64 - 2**63
        check(lib().ffq_arrayadd_q_device(self.handle, ctypes.c_void_p(dptr), int(n), v))

    def table_lower_bound(self, d_table, n_rows, col, value):
        idx = ctypes.c_int64(0)
        check(lib().ffq_table_lower_bound(self.handle, ctypes.c_void_p(d_table), int(n_rows), int(col),
                                          int(value), ctypes.byref(idx)))
        return idx.value

    def read_probe(self, dptr, n_bytes, mode=0, reps=10):
        """Instrumented build only (use_probe_build(); include/ffq_probe.h)."""
        if not _probe:
            raise FFQError(E_ARG, "ffq_read_probe exists only in libffq_probe.so: call hip.use_probe_build() first (tools)")
        ms = ctypes.c_float(0)
        check(lib().ffq_read_probe(self.handle, ctypes.c_void_p(dptr), int(n_bytes), int(mode), int(reps),
                                   ctypes.byref(ms)))
        return ms.value

    def table_cut(self, d_table, n_rows, lo, hi):
        """(i0, i1, pos0[i0], pos0[i1], pos5[i0 - 1], pos5[i1 - 1]): first rows with pos0 >= lo / >= hi,
        their pos0, and pos5 of the rows in front of them (-1: none)."""
        out = (ctypes.c_int64 * 6)()
        check(lib().ffq_table_cut(self.handle, ctypes.c_void_p(d_table), int(n_rows), int(lo), int(hi), out))
        return [int(x) for x in out]

    def table_select_seqlen(self, d_table, n_rows, min_len, max_len, d_out):
        """Rows with min_len <= pos3 - pos2 <= max_len of a device table, in order, into
        d_out (raw device pointers); returns the number kept."""
        k = ctypes.c_int64(0)
        check(lib().ffq_table_select_seqlen(self.handle, ctypes.c_void_p(d_table), int(n_rows), int(min_len),
                                            int(max_len), ctypes.c_void_p(d_out), ctypes.byref(k)))
        return k.value

    def table_select_seqlen_idx(self, d_table, n_rows, min_len, max_len, d_out, d_idx):
        """... and d_idx[i] = the ordinal in the table of kept row i (ffq_table_select_seqlen_idx)."""
        k = ctypes.c_int64(0)
        check(lib().ffq_table_select_seqlen_idx(self.handle, ctypes.c_void_p(d_table), int(n_rows), int(min_len), int(max_len),
                                                ctypes.c_void_p(d_out), ctypes.c_void_p(d_idx), ctypes.byref(k)))
        return k.value

    COLUMNS = {"header": (0, 1, 1), "sequence": (2, 0, 3), "quality": (4, 0, 5)}

    def table_gather_column(self, d_buf, n_bytes, d_table, n_rows, which, d_out, out_cap, d_off, sentinel=True,
                            add=None, value_add=0):
        """Packed component `which` ("header" = buf[pos0 + 1:pos1], "sequence", "quality", or a
        (begin column, shift, end column) triple) of every row of a device table, + value_add per
        byte, into d_out with CSR offsets d_off (n_rows + 1); raw device pointers.  Returns
        (rc, bytes of the stream); rc is OK or E_TABLE_FULL (out_cap too small)."""
        ca, sh, cb = self.COLUMNS[which] if isinstance(which, str) else which
        if add is None:
            add = -1 if sentinel else 0
        nb = ctypes.c_int64(0)
        rc = lib().ffq_table_gather_column(self.handle, ctypes.c_void_p(d_buf), int(n_bytes), int(bool(sentinel)),
                                           int(add), ctypes.c_void_p(d_table), int(n_rows), int(ca), int(sh), int(cb),
                                           int(value_add), ctypes.c_void_p(d_out) if d_out else None, int(out_cap),
                                           ctypes.c_void_p(d_off), ctypes.byref(nb))
        check(rc, allow=(E_TABLE_FULL,))
        return rc, nb.value

    def synth_single(self, dptr, first, count, seed=42):
        check(lib().ffq_synth_single(self.handle, ctypes.c_void_p(dptr), int(first), int(count), int(seed)))

    def synth_wrapped(self, dptr, d_start, first, count, seed=43):
        check(lib().ffq_synth_wrapped(self.handle, ctypes.c_void_p(dptr), ctypes.c_void_p(d_start),
                                      int(first), int(count), int(seed)))


def shard_unique_id():
    """128 bytes of communicator id (ncclGetUniqueId): rank 0 draws it, every rank gets it from rank 0."""
    buf = (ctypes.c_uint8 * 128)()
    check(lib().ffq_shard_unique_id(buf))
    return bytes(buf)


def shard_host_step(rank, world, bounds, tail_bytes, head_bytes, ext_ptr, table_ptr, table_cap, exchange, allgather, scan=None, ctx=None):
    """One step of the byte-range shards over HOST memory (ffq_shard_host_step): the protocol is the library's
    (csrc/ffq_shard_proto.h), the transport the caller's --
        exchange(pieces)   pieces = [(src, dst, a, b, ptr)]: the same list on every rank; ptr is this rank's end (or None)
        allgather(words)   eight ints in, [world][8] out
        scan(buf_ptr, n, sentinel, offset, eof, add, table_ptr, cap, res) -> FFQ code, res (a ScanResult) filled;
                           None: ffq_scan_host on `ctx` (the GPU)
    An exception raised inside a callback is re-raised here.  Returns (rc, ShardResult); rc is OK or E_TABLE_FULL."""
    err = []

    def guard(fn):
        def run(*a):
            try:
                return int(fn(*a) or 0)
            except BaseException as e:      # noqa: BLE001  (must not propagate through the C frames)
                err.append(e)
                return E_INTERNAL
        return run

    def c_exchange(_user, pieces, n):
        exchange([(pieces[i].src, pieces[i].dst, pieces[i].a, pieces[i].b, pieces[i].ptr) for i in range(n)])

    def c_gather(_user, mine, out):
        allv = allgather([mine[i] for i in range(8)])
        for r in range(world):
            for k in range(8):
                out[r * 8 + k] = int(allv[r][k])

    def c_scan(_user, buf, n, sentinel, offset, eof, add, table, cap, res):
        return scan(buf, n, sentinel, offset, eof, add, table, cap, res.contents)

    ops = ShardHostOps(None, _SCAN_CB(guard(c_scan)) if scan is not None else _SCAN_CB(), _EXCHANGE_CB(guard(c_exchange)),
                       _GATHER_CB(guard(c_gather)))
    b = (ctypes.c_int64 * (world + 1))(*[int(x) for x in bounds])
    res = ShardResult()
    rc = lib().ffq_shard_host_step(ctypes.byref(ops), ctx.handle if ctx is not None else None, int(rank), int(world), b,
                                   int(tail_bytes), int(head_bytes), ctypes.c_void_p(ext_ptr), ctypes.c_void_p(table_ptr),
                                   int(table_cap), ctypes.byref(res))
    if err:
        raise err[0]
    check(rc, allow=(E_TABLE_FULL,))
    return rc, res


class ShardWorld:
    """k logical ranks as threads of one process (ffq_shard_world_*): the in-process transport of the native step."""

    def __init__(self, world):
        self.world = int(world)
        self._h = ctypes.c_void_p()
        check(lib().ffq_shard_world_create(self.world, ctypes.byref(self._h)))

    def abort(self):
        if self._h:
            lib().ffq_shard_world_abort(self._h)

    def close(self):
        if self._h:
            lib().ffq_shard_world_destroy(self._h)
            self._h = ctypes.c_void_p()


class Shard:
    """One rank's byte-range shard of a stream, the whole step behind the C ABI (ffq_shard_*, include/ffq.h): halo
    hand-off over RCCL (or the in-process transport), scan, cut, one gather of the hand-off words."""

    def __init__(self, ctx, bounds, rank, world, tail_bytes, head_bytes, unique_id=None, local_world=None, parent=None, hosted=None,
                 serial=None):
        """serial: True -- the serial step (ONE communicator, ONE stream) from the start; None: as FFQ_SHARD_SERIAL says.
        unique_id: RCCL between processes; local_world: a ShardWorld (threads of this process); hosted: a transport object
        with exchange(pieces) / allgather(words) as shard_host_step takes them -- the device step over the caller's own
        transport (ffq_shard_create_hosted: several processes on one GPU, a group over gloo)."""
        self._ctx = ctx
        self._h = ctypes.c_void_p()
        self._keep = (local_world, parent)
        self._err = []
        b = (ctypes.c_int64 * (world + 1))(*[int(x) for x in bounds])
        if hosted is not None:
            err = self._err

            def c_exchange(_user, pieces, n):
                try:
                    hosted.exchange([(pieces[i].src, pieces[i].dst, pieces[i].a, pieces[i].b, pieces[i].ptr) for i in range(n)])
                    return 0
                except BaseException as e:      # noqa: BLE001
                    err.append(e)
                    return E_INTERNAL

            def c_gather(_user, mine, out):
                try:
                    allv = hosted.allgather([mine[i] for i in range(8)])
                    for r in range(world):
                        for k in range(8):
                            out[r * 8 + k] = int(allv[r][k])
                    return 0
                except BaseException as e:      # noqa: BLE001
                    err.append(e)
                    return E_INTERNAL
            ops = ShardHostOps(None, _SCAN_CB(), _EXCHANGE_CB(c_exchange), _GATHER_CB(c_gather))
            self._keep = (ops, hosted)                     # (the callbacks live as long as the shard)
            check(lib().ffq_shard_create_hosted(ctx.handle, ctypes.byref(ops), int(rank), int(world), b, int(tail_bytes), int(head_bytes),
                                                ctypes.byref(self._h)))
        elif parent is not None:
            check(lib().ffq_shard_create_lane(parent._h, ctx.handle, ctypes.byref(self._h)))
        elif local_world is not None:
            check(lib().ffq_shard_create_local(ctx.handle, local_world._h, int(rank), b, int(tail_bytes), int(head_bytes),
                                               ctypes.byref(self._h)))
        else:
            idb = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
            if serial is None:
                check(lib().ffq_shard_create(ctx.handle, idb, int(rank), int(world), b, int(tail_bytes), int(head_bytes),
                                             ctypes.byref(self._h)))
            else:
                check(lib().ffq_shard_create2(ctx.handle, idb, int(rank), int(world), b, int(tail_bytes), int(head_bytes),
                                              1 if serial else 0, ctypes.byref(self._h)))
                serial = None
        if serial is not None and parent is None:
            check(lib().ffq_shard_set_serial(self._h, 1 if serial else 0))
        import weakref
        ctx._children.append(weakref.ref(self))      # (a shard lives on its context: closed with it, before it)

    def lane(self, ctx):
        """A second shard of the same rank on another context (same scan stream): steps queued one ahead."""
        return Shard(ctx, [], 0, 0, 0, 0, parent=self)

    def halo(self):
        t, h = ctypes.c_int64(), ctypes.c_int64()
        check(lib().ffq_shard_halo(self._h, ctypes.byref(t), ctypes.byref(h)))
        return t.value, h.value

    def transport(self):
        return lib().ffq_shard_transport(self._h).decode()

    def info(self):
        """Who is there and how the steps run (ffq_shard_get_info): dict with nranks_handoff / nranks_gather
        (ncclCommCount), bus_ids (every rank's GPU as "dddd:bb:dd.f", None: unknown), serial, timeout_s, last_stage
        (where the last watchdog trip found the step), poisoned."""
        i = ShardInfo()
        check(lib().ffq_shard_get_info(self._h, ctypes.byref(i)))

        def bdf(v):
            return None if v < 0 else "%04x:%02x:%02x.%d" % (v >> 16, (v >> 8) & 0xFF, (v >> 3) & 0x1F, v & 7)
        return {"rank": i.rank, "world": i.world, "nranks_handoff": i.nranks_handoff, "nranks_gather": i.nranks_gather,
                "serial": bool(i.serial), "mode": "serial" if i.serial else "pipelined", "timeout_s": float(i.timeout_s),
                "last_stage": STAGE_NAMES[i.last_stage], "poisoned": bool(i.poisoned),
                "bus_ids": [bdf(int(i.bus_id[r])) for r in range(i.n_bus)]}

    def set_timeout(self, seconds):
        """The step watchdog's deadline (0: none; default FFQ_SHARD_TIMEOUT_S or 30 s): every lane of the rank."""
        check(lib().ffq_shard_set_timeout(self._h, float(seconds)))

    def set_serial(self, on=True):
        """Between steps (no lane pending), every rank alike: the serial step -- ONE communicator, ONE stream."""
        check(lib().ffq_shard_set_serial(self._h, 1 if on else 0))

    def abort(self):
        """After E_TIMEOUT: ncclCommAbort on the communicators, the streams drained.  True: drained; only close() is left."""
        return lib().ffq_shard_abort(self._h) == OK

    def inject_stall(self, stage, seconds):
        """diagnostics: the next step hangs at STAGE_* for up to `seconds` (released by abort / close)."""
        check(lib().ffq_shard_inject_stall(self._h, int(stage), float(seconds)))

    def self_exchange(self, d_src, d_dst, n):
        check(lib().ffq_shard_self_exchange(self._h, ctypes.c_void_p(d_src), ctypes.c_void_p(d_dst), int(n)))

    def exchange_halo(self, d_ext, overlap=False):
        check(lib().ffq_shard_exchange_halo(self._h, ctypes.c_void_p(d_ext), 1 if overlap else 0))

    def load_fd(self, fd, d_ext):
        """This rank's bytes [lo - tail, hi + head) of the file behind fd (bounds are file offsets) into d_ext; steps over
        that buffer then hand off nothing (ffq_shard_load_fd).  Returns the bytes loaded.  fd < 0: detach."""
        n = ctypes.c_int64()
        check(lib().ffq_shard_load_fd(self._h, int(fd), ctypes.c_void_p(d_ext), ctypes.byref(n)))
        return n.value

    def scan_fd_slabs(self, fd, slab_bytes, d_table, table_cap, flags=0):
        """This rank's range of the file behind fd through ONE device buffer of slab_bytes, slab after slab
        (ffq_shard_scan_fd_slabs: a range that does not fit the GPU); collective like a step.  (rc, ShardResult)."""
        res = ShardResult()
        rc = lib().ffq_shard_scan_fd_slabs(self._h, int(fd), int(slab_bytes), int(flags), ctypes.c_void_p(d_table), int(table_cap), ctypes.byref(res))
        if self._err:
            raise self._err.pop(0)
        check(rc, allow=(E_TABLE_FULL,))
        return rc, res

    def step_submit(self, d_ext, d_table, table_cap, flags=0, qual_add=-33, d_qual=None, qual_cap=0, d_qoff=None, overlap=False):
        rc = lib().ffq_shard_step_submit(self._h, ctypes.c_void_p(d_ext), 1 if overlap else 0, int(flags), int(qual_add),
                                         ctypes.c_void_p(d_table), int(table_cap), ctypes.c_void_p(d_qual), int(qual_cap),
                                         ctypes.c_void_p(d_qoff))
        if self._err:                                   # (an exception inside a hosted transport's callback: the hand-off)
            raise self._err.pop(0)
        check(rc)

    def step_wait(self):
        res = ShardResult()
        rc = lib().ffq_shard_step_wait(self._h, ctypes.byref(res))
        if self._err:                                   # (an exception inside a hosted transport's callback)
            raise self._err.pop(0)
        check(rc, allow=(E_TABLE_FULL,))
        return rc, res

    def close(self):
        if self._h:
            lib().ffq_shard_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Stream:
    """What the native stream front ends (ffq_stream_*) have in common.  Iterating yields
    (rows, fill, fill_offset, end_state, err_offset): `rows` int64[n][6] absolute offsets and
    `fill` (uint8 array, fill[i] = stream byte fill_offset + i) are views of memory the stream
    owns -- valid until the next iteration step."""
    _h = None
    decode = False
    on_close = None        # called once, before the native stream goes away
    consumed = 0
    at_end = False
    filtered = False

    def tell(self):
        """File position behind the last chunk handed out (-1: the source has none)."""
        return int(lib().ffq_stream_tell(self._h)) if self._h else -1

    def path(self):
        """ScanResult.path of the fill the iteration has just yielded (6: index and decoded qualities in one pass)."""
        return int(lib().ffq_stream_path(self._h)) if self._h else -1

    def quals(self):
        """(qual int8[], qoff int64[n + 1]) of the fill the iteration has just yielded (streams
        opened with decode=True); views, valid until the next iteration step.  Record i's decoded
        bytes are qual[qoff[i] : qoff[i] + pos5 - pos4] (row i's columns): packed back to back, or --
        single_pass streams, where the input allows it -- with gaps between the index tiles' segments
        (include/ffq.h, FFQ_F_SINGLE_PASS); qoff[n] is where the last record's bytes end."""
        qp, op, nq = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        check(lib().ffq_stream_quals(self._h, ctypes.byref(qp), ctypes.byref(op), ctypes.byref(nq)))
        n = self._last_rows
        qual = (np.ctypeslib.as_array((ctypes.c_int8 * nq.value).from_address(qp.value)) if nq.value
                else np.zeros(0, dtype=np.int8))
        qoff = np.ctypeslib.as_array((ctypes.c_int64 * (n + 1)).from_address(op.value))
        return qual, qoff

    COLUMNS = {None: 0, "entry": 0, "header": 1, "sequence": 2, "quality": 3}

    def set_filter(self, min_seq_len=None, max_seq_len=None, column=None, value_add=0):
        """Push-down (ffq_stream_set_filter; doc/user-guide.rst:153-180): from the next fill on the iteration yields only the
        rows with min_seq_len <= pos3 - pos2 <= max_seq_len; column = "header" | "sequence" | "quality": that component
        of the kept rows is gathered on the device (selected())."""
        lo = -(1 << 62) if min_seq_len is None else int(min_seq_len)
        hi = (1 << 62) if max_seq_len is None else int(max_seq_len)
        check(lib().ffq_stream_set_filter(self._h, lo, hi, self.COLUMNS[column], int(value_add)))
        self.filtered = True

    def selected(self):
        """(index int64[kept], n_scanned, col int8[] or None, coloff int64[kept + 1] or None) of the fill the iteration has
        just yielded: index[i] = ordinal of kept row i among the fill's n_scanned records; bytes of kept row i =
        col[coloff[i] : coloff[i + 1]].  Views, valid until the next iteration step."""
        ip, cp, op = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        ns, nb = ctypes.c_int64(), ctypes.c_int64()
        check(lib().ffq_stream_selected(self._h, ctypes.byref(ip), ctypes.byref(ns), ctypes.byref(cp), ctypes.byref(op), ctypes.byref(nb)))
        k = self._last_rows
        idx = np.ctypeslib.as_array((ctypes.c_int64 * k).from_address(ip.value)) if k else np.zeros(0, dtype=np.int64)
        col = off = None
        if op.value:
            off = np.ctypeslib.as_array((ctypes.c_int64 * (k + 1)).from_address(op.value))
            col = np.ctypeslib.as_array((ctypes.c_int8 * nb.value).from_address(cp.value)) if nb.value else np.zeros(0, dtype=np.int8)
        return idx, int(ns.value), col, off

    def close(self):
        if self._h:
            if self.on_close is not None:
                cb, self.on_close = self.on_close, None
                cb()
            lib().ffq_stream_close(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _next(self):
        """One fill: (rows, fill, fill_offset, end_state, err_offset)."""
        rows_p, fill_p = ctypes.c_void_p(), ctypes.c_void_p()
        n, nb, off, err = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        end = ctypes.c_int32()
        check(lib().ffq_stream_next(self._h, ctypes.byref(rows_p), ctypes.byref(n), ctypes.byref(end),
                                    ctypes.byref(err), ctypes.byref(fill_p), ctypes.byref(nb), ctypes.byref(off)))
        self._last_rows = n.value
        self.consumed = off.value + nb.value     # stream bytes handed out so far (byte i of the fill is stream offset off + i)
        self.at_end = end.value != END_REFILL    # the stream is through (cleanly or with its error): nothing follows
        rows = (np.ctypeslib.as_array((ctypes.c_int64 * (n.value * 6)).from_address(rows_p.value)).reshape(-1, 6)
                if n.value else np.zeros((0, 6), dtype=np.int64))
        fill = (np.ctypeslib.as_array((ctypes.c_uint8 * nb.value).from_address(fill_p.value))
                if nb.value else np.zeros(0, dtype=np.uint8))
        return rows, fill, off.value, end.value, err.value


def _decode_flags(decode, single_pass):
    return (F_DECODE_QUAL | (F_SINGLE_PASS if single_pass else 0)) if decode else 0


class FileStream(_Stream):
    """The stream front end over a file descriptor: buffer fills read ahead into pinned memory (reader
    threads; a gzip file is inflated by them) while the previous fill is scanned."""

    def __init__(self, ctx, fd, fbufsize=1 << 24, decode=False, qual_add=-33, start=None, gzip=False, single_pass=True):
        """start: byte of the file the stream begins at (None: the descriptor's current position).
        A descriptor that can seek is read with pread: its own position does not move.
        gzip: the descriptor is a gzip file; the stream's reader thread inflates it into the pinned
        chunk buffers (fbufsize and every offset count DECOMPRESSED bytes).
        single_pass (with decode): the qualities of a fill of four-line records are decoded by the pass
        that builds the line index (one read of the bytes) and come segmented: see quals()."""
        self._ctx = ctx
        self._h = ctypes.c_void_p()
        self.decode = bool(decode)
        opener = lib().ffq_stream_open_gzip if gzip else lib().ffq_stream_open2
        check(opener(ctx.handle, int(fd), int(fbufsize), _decode_flags(decode, single_pass),
                     int(qual_add), -1 if start is None else int(start), ctypes.byref(self._h)))

    def __iter__(self):
        while True:
            t = self._next()
            yield t
            if t[3] != END_REFILL:
                return


class PushStream(_Stream):
    """The stream front end over any object with readinto() / read() -- BytesIO, bz2 / lzma / gzip
    file objects, sockets: no reader thread; every chunk is read by THIS thread straight into the
    stream's pinned chunk buffer (ffq_stream_push_buffer / ffq_stream_push; no bytes object, no
    copy), then scanned.  Iterates like FileStream."""

    def __init__(self, ctx, fh, fbufsize=1 << 23, decode=False, qual_add=-33, single_pass=True, min_fill=None):
        """min_fill: a chunk is handed over SHORT (which is not the end of the stream) as soon as it holds this many
        bytes and a read comes back with less than was asked for -- a live source (a socket, stdin, a decompressor
        over a pipe) has nothing more for now, and the reference's loop yields after every read of fbufsize bytes
        (fastqandfurious.py:222-232 advises small buffers for latency).  None: fbufsize (a chunk is always filled)."""
        self._ctx = ctx
        self._h = ctypes.c_void_p()
        self.decode = bool(decode)
        self._fh = fh
        self._min_fill = int(fbufsize if min_fill is None else max(1, min(int(min_fill), int(fbufsize))))
        self._readinto = getattr(fh, "readinto", None)
        check(lib().ffq_stream_open_push(ctx.handle, int(fbufsize), _decode_flags(decode, single_pass), int(qual_add),
                                         ctypes.byref(self._h)))

    def _pump(self):
        """One chunk of the source into the next slot (reference read(), fastqandfurious.py:30-36: the
        source is exhausted when it returns less than was asked for -- here after asking again until
        the chunk is full or nothing comes)."""
        dst, cap = ctypes.c_void_p(), ctypes.c_int64()
        check(lib().ffq_stream_push_buffer(self._h, ctypes.byref(dst), ctypes.byref(cap)))
        mv = memoryview((ctypes.c_uint8 * cap.value).from_address(dst.value)).cast("B")
        got, eof = 0, False
        while got < cap.value:
            want = cap.value - got
            if self._readinto is not None:
                n = self._readinto(mv[got:]) or 0
            else:
                data = self._fh.read(want)
                n = len(data)
                mv[got:got + n] = data
            if n <= 0:
                eof = True
                break
            got += n
            if n < want and got >= self._min_fill:
                break                       # a short read with enough in hand: hand the chunk over, the rest comes with the next
        check(lib().ffq_stream_push(self._h, got, 1 if eof else 0))

    def __iter__(self):
        self._pump()
        while True:
            t = self._next()
            yield t
            if t[3] != END_REFILL:
                return
            self._pump()


_default_ctx = {}


def gunzip_fd(fd, cap, chunk=16 << 20, threads=0):
    """The stream front end's gzip reader on its own (ffq_gunzip_fd; host only, no device): the file behind
    `fd` inflated into a new uint8 array of at most `cap` bytes, `chunk` bytes per call of the reader.
    Returns (array, members inflated side by side)."""
    import numpy as np
    out = np.empty(max(int(cap), 1), dtype=np.uint8)
    npar = ctypes.c_int64(0)
    n = lib().ffq_gunzip_fd(int(fd), out.ctypes.data, int(cap), int(chunk), int(threads), ctypes.byref(npar))
    if n < 0:
        check(int(n))
    return out[:n], int(npar.value)


def bgzf_range(fd, c_lo, c_hi, out=None, threads=0):
    """ffq_bgzf_range (host only): the BGZF members whose first byte lies in [c_lo, c_hi) of the file behind fd.
    out=None: nothing is inflated -> (c_first, c_end, n_bytes, n_members), n_bytes = what the members' trailers promise;
    out = a writable uint8 array of at least that many bytes: inflated into it, the same tuple.  FFQGzipError (an OSError)
    for what is not BGZF or does not inflate to its trailer, FFQGzipTruncated (an EOFError) for a file cut short."""
    cf, ce, no, nm = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    check(lib().ffq_bgzf_range(int(fd), int(c_lo), int(c_hi), ctypes.c_void_p(out.ctypes.data) if out is not None else None,
                               int(out.size) if out is not None else 0, int(threads), ctypes.byref(cf), ctypes.byref(ce),
                               ctypes.byref(no), ctypes.byref(nm)))
    return int(cf.value), int(ce.value), int(no.value), int(nm.value)


def gunzip_stats():
    """Counters of the several-thread inflate of plain gzip members (ffq_gunzip_stats; csrc/ffq_pgz.h)."""
    a = (ctypes.c_int64 * 5)()
    lib().ffq_gunzip_stats(a)
    return dict(zip(("batches", "chunks", "rejected", "giveups", "members"), (int(x) for x in a)))


def default_context(device=None):
    """Process-wide context for `device` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    ctx = _default_ctx.get(device)
    if ctx is None:
        ctx = Context(device)
        _default_ctx[device] = ctx
    return ctx


class ReadProbe:
    """bench.py's `hbm_read_probe`: what this box's memory system gives a pure streaming read, measured
    by the INSTRUMENTED build (libffq_probe.so, include/ffq_probe.h) loaded beside the product library
    -- its own handle, its own context; the product library has no such entry point."""

    def __init__(self, device=0):
        from . import build as _build
        _share_torch_hip_runtime()
        self._L = ctypes.CDLL(_build.build_probe())
        L = self._L
        vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        L.ffq_last_error.restype = ctypes.c_char_p
        L.ffq_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
        L.ffq_ctx_destroy.argtypes = [vp]
        L.ffq_ctx_destroy.restype = None
        L.ffq_read_probe.argtypes = [vp, vp, i64, i32, i32, ctypes.POINTER(ctypes.c_float)]
        self._h = vp()
        rc = L.ffq_ctx_create(int(device), ctypes.byref(self._h))
        if rc != OK:
            raise FFQError(rc, L.ffq_last_error().decode("utf-8", "replace"))

    def read_ms(self, dptr, n_bytes, mode=6, reps=10):
        ms = ctypes.c_float(0)
        rc = self._L.ffq_read_probe(self._h, ctypes.c_void_p(dptr), int(n_bytes), int(mode), int(reps), ctypes.byref(ms))
        if rc != OK:
            raise FFQError(rc, self._L.ffq_last_error().decode("utf-8", "replace"))
        return ms.value

    def close(self):
        if self._h:
            self._L.ffq_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

"""ctypes loader of libfinch_hip.so (the C ABI in include/finch_hip.h).

The product path has no CPU implementation: if the HIP library is missing or no device is usable,
everything here fails loudly instead of falling back."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("FH_LIB", os.path.join(_HERE, "libfinch_hip.so"))

FH_OK = 0
FH_ERR_INVALID, FH_ERR_NO_DEVICE, FH_ERR_HIP, FH_ERR_STATE, FH_ERR_CAPACITY, FH_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
KIND_MASH, KIND_SCALED = 0, 1


class FhParams(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("k", C.c_uint32), ("size", C.c_uint64), ("seed", C.c_uint64),
                ("scale", C.c_double), ("max_launch", C.c_uint64), ("hash_mask", C.c_uint64), ("stage_bytes", C.c_uint64)]


class FinchHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("finch_hip error %d: %s" % (code, msg))
        self.code = code


# every symbol include/finch_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_U64P, _U32P, _U8P = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
SYMBOLS = {
    "fh_device_count": (C.c_int, []),
    "fh_last_error": (C.c_char_p, []),
    "fh_abi_version": (C.c_int, []),
    "fh_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "fh_get_option": (C.c_char_p, [C.c_char_p]),
    "fh_option_list": (C.c_char_p, []),
    "fh_new": (_P, [C.POINTER(FhParams), C.c_int]),
    "fh_free": (None, [_P]),
    "fh_release_cached": (None, []),
    "fh_reset": (C.c_int, [_P]),
    "fh_set_stream_offset": (C.c_int, [_P, C.c_uint64]),
    "fh_process": (C.c_int, [_P, _P, C.c_uint64]),
    "fh_process_records": (C.c_int, [_P, _P, _P, _P, C.c_uint64]),
    "fh_process_records_in": (C.c_int, [_P, _P, C.c_uint64, _P, _P, C.c_uint64, _P]),
    "fh_total_bases": (C.c_int, [_P, _U64P]),
    "fh_push_block": (C.c_int, [_P, _P, C.c_uint64]),
    "fh_push_block_ex": (C.c_int, [_P, _P, C.c_uint64, C.c_uint32]),
    "fh_text_buffer": (C.c_int, [_P, C.POINTER(_P), _U64P]),
    "fh_text_buffers": (C.c_int, [_P, C.POINTER(_P), _U64P, C.POINTER(C.c_int)]),
    "fh_text_prefetch": (C.c_int, [_P, C.c_int, C.c_uint64]),
    "fh_push_fastq_text": (C.c_int, [_P, C.c_uint64]),
    "fh_push_bgzf_fastq": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint32]),
    "fh_push_gzip_fastq": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), _U64P]),
    "fh_gzip_batch_capacity": (C.c_int, [_P, _U64P]),
    "fh_bgzf_text_capacity": (C.c_int, [_P, _U64P]),
    "fh_push_fasta_text": (C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint32]),
    "fh_push_staged": (C.c_int, [_P, C.c_uint64, C.c_uint32]),
    "fh_set_text_halo": (C.c_int, [_P, _P, C.c_uint32]),
    "fh_text_bases": (C.c_int, [_P, _U64P]),
    "fh_push_device": (C.c_int, [_P, _P, C.c_uint64]),
    "fh_set_record_stride": (C.c_int, [_P, C.c_uint32]),
    "fh_debug_segments": (C.c_int, [_P, _U64P, _U64P, C.POINTER(C.c_uint32)]),
    "fh_sync": (C.c_int, [_P]),
    "fh_finish": (C.c_int, [_P, _U64P, _U64P]),
    "fh_copy_out": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "fh_copy_out_records": (C.c_int, [_P, _P, _P, _P]),
    "fh_copy_out_kmers": (C.c_int, [_P, _P, C.c_uint64, _P]),
    "fh_copy_out_rows": (C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    "fh_result_counts": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), _U64P]),
    "fh_merge": (C.c_int, [_P, _P]),
    "fh_merge_arrays": (C.c_int, [_P, C.c_uint64, _P, _P, _P, _P, _P, C.c_uint64]),
    "fh_merge_partials": (C.c_int, [C.c_uint32, C.c_uint64, C.c_double, C.c_uint32,
                                    C.c_uint64, _P, _P, _P, _P, _P, C.c_uint64, _P, _P, _P, _P, _P,
                                    _U64P, _P, _P, _P, _P, _P]),
    "fh_merge_wire": (C.c_int, [C.c_uint32, C.c_uint64, C.c_double, C.c_uint32, C.c_uint64, C.c_uint32, _P, _U64P, _P, _P, _P, _P, _P,
                      _U64P]),
    "fh_sketch_device_blocks": (C.c_int, [_P, _P, _U64P, _U64P, C.c_uint32]),
    "fh_batch_new": (_P, [C.POINTER(FhParams), C.c_int, C.c_uint32, C.c_uint64]),
    "fh_batch_free": (None, [_P]),
    "fh_batch_stage": (C.c_int, [_P, C.c_int, C.POINTER(_P), _U64P]),
    "fh_batch_submit": (C.c_int, [_P, C.c_int, _P, _P, C.c_uint32]),
    "fh_batch_submit_packed": (C.c_int, [_P, C.c_int, _P, _P, C.c_uint32]),
    "fh_batch_packed_bytes": (C.c_uint64, [C.c_uint64]),
    "fh_batch_pack": (C.c_int, [_P, C.c_uint64, _P, C.c_uint64]),
    "fh_batch_wait": (C.c_int, [_P, C.c_int, _P]),
    "fh_batch_result": (C.c_int, [_P, C.c_int, C.c_uint32, _U64P, _U64P]),
    "fh_batch_copy_out": (C.c_int, [_P, C.c_int, C.c_uint32, _P, _P, _P, _P, _P]),
    "fh_batch_copy_out_records": (C.c_int, [_P, C.c_int, C.c_uint32, _P, _P]),
    "fh_batch_set_profiling": (C.c_int, [_P, C.c_int]),
    "fh_batch_kernel_time": (C.c_int, [_P, C.POINTER(C.c_double), _U64P, _U64P]),
    "fh_batch_counters": (C.c_int, [_P, _U64P, _U64P]),
    "fh_set_profiling": (C.c_int, [_P, C.c_int]),
    "fh_kernel_time": (C.c_int, [_P, C.POINTER(C.c_double), _U64P, _U64P]),
    "fh_debug_counters": (C.c_int, [_P, _U64P, _U64P, _U64P]),
    "fh_debug_add_counts": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "fh_debug_gzip_feed_timeouts": (C.c_uint64, []),
    "fh_debug_speculation": (C.c_int, [_P, _U64P, _U64P]),
    "fh_debug_fast_path": (C.c_int, [_P, _U64P, _U64P, _U64P]),
    "fh_measure_read_bandwidth": (C.c_int, [C.c_int, _P, C.c_uint64, C.c_int, C.POINTER(C.c_double)]),
    "fh_device_alloc": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(_P)]),
    "fh_device_free": (C.c_int, [C.c_int, _P]),
    "fh_copy_to_device": (C.c_int, [C.c_int, _P, _P, C.c_uint64]),
    "fh_copy_from_device": (C.c_int, [C.c_int, _P, _P, C.c_uint64]),
    "fh_synth_genome_host": (C.c_int, [_P, C.c_uint64, C.c_uint64]),
    "fh_synth_genome_device": (C.c_int, [C.c_int, _P, C.c_uint64, C.c_uint64]),
    "fh_synth_reads_host": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64,
                                      C.c_uint32, C.c_uint32]),
    "fh_synth_reads_device": (C.c_int, [C.c_int, _P, _P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64,
                                        C.c_uint32, C.c_uint32]),
}

_lib = None


def load():
    """dlopen libfinch_hip.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH) and "FH_LIB" not in os.environ and not os.environ.get("FH_NO_AUTOBUILD"):
            # source checkout without the built artefact: compile it (hipcc, ~20 s).  This builds the HIP library
            # itself -- there is no alternative implementation to fall back to.
            import importlib.util
            spec = importlib.util.spec_from_file_location("fh_build", os.path.join(_HERE, "csrc", "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build()
        if not os.path.exists(SO_PATH):
            raise FinchHipError(FH_ERR_NO_DEVICE, "%s not built -- run `python finch_rs_amd/csrc/build.py` "
                                "(or __graft_entry__.build())" % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != FH_OK:
        raise FinchHipError(rc, (load().fh_last_error() or b"").decode(errors="replace"))


# --- the library's one configuration surface (include/finch_hip.h: fh_set_option, FH_DEBUG; csrc/fh_options.cpp) ---
def set_option(name: str, value=None) -> None:
    """process-wide option `name` = value (None: back to "not set"); FinchHipError for a name the library does not know"""
    check(load().fh_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name: str):
    v = load().fh_get_option(name.encode())
    return None if v is None else v.decode()


def option_list():
    """[(name, what it does)] for every option"""
    return [tuple(line.split("\t", 1)) for line in load().fh_option_list().decode().splitlines() if line]


def _debug_parse(s: str) -> dict:
    d = {}
    for item in (s or "").replace(";", ",").replace(" ", ",").split(","):
        if item:
            k, _, v = item.partition("=")
            d[k] = v if _ else "1"
    return d


def debug_env(env=None, **opts) -> dict:
    """a copy of `env` (default: this process's environment) whose FH_DEBUG -- the ONE environment variable the library reads --
    carries `opts` on top of what it held (value None removes an option): for child processes of tests and tools"""
    e = dict(os.environ if env is None else env)
    d = _debug_parse(e.get("FH_DEBUG", ""))
    for k, v in opts.items():
        if v is None:
            d.pop(k, None)
        else:
            d[k] = str(v)
    if d:
        e["FH_DEBUG"] = ",".join("%s=%s" % kv for kv in d.items())
    else:
        e.pop("FH_DEBUG", None)
    return e


def debug_set(**opts) -> None:
    """the same on this process's own environment: the library looks at FH_DEBUG whenever it asks for an option, so options
    that are read per call follow at once (those read once per process / per handle do not: use a child process)"""
    e = debug_env(**opts)
    if "FH_DEBUG" in e:
        os.environ["FH_DEBUG"] = e["FH_DEBUG"]
    else:
        os.environ.pop("FH_DEBUG", None)

"""ctypes loader for the C-ABI library (cuclark_amd/lib/libmi_clark.so, include/mi_clark.h).

There is no CPU fallback: if the HIP library is missing, or no gfx950 device is present when an engine is
created, the caller gets an exception — never a silently slower path.
"""
import ctypes as C
import importlib.util
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# MIC_LIB_PATH selects a measuring build (make variant ...: csrc/Makefile); the product library is the default
LIB_PATH = os.environ.get("MIC_LIB_PATH") or os.path.join(_PKG, "lib", "libmi_clark.so")
CSRC = os.path.join(_PKG, "csrc")

MIC_RESULT_WORDS = 8
MIC_FLAG_ROW_OVERFLOW = 1
MIC_FLAG_DENSE_PATH = 2
MIC_ROW_INVALID = 0xFFFFFFFF
MIC_LAYOUT_AUTO, MIC_LAYOUT_DIRECT, MIC_LAYOUT_MINIMIZER, MIC_LAYOUT_SUPER, MIC_LAYOUT_SUPER2 = 0, 1, 2, 3, 4


class MicError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmi_clark error {code}: {msg}")
        self.code = code


class MicConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("k", C.c_int32), ("num_targets", C.c_uint32), ("num_batches", C.c_uint32),
                ("row_words", C.c_uint32), ("layout", C.c_uint32)]


class MicDbInfo(C.Structure):
    _fields_ = [("htsize", C.c_uint64), ("shard_start", C.c_uint64), ("shard_end", C.c_uint64),
                ("n_elems", C.c_uint64), ("n_elems_file", C.c_uint64), ("n_slots", C.c_uint64),
                ("n_overflow", C.c_uint64), ("hbm_bytes", C.c_uint64), ("key_bytes", C.c_int32),
                ("slot_class", C.c_int32), ("max_bucket", C.c_uint32), ("sampling", C.c_uint32), ("layout", C.c_int32),
                ("minimizer_len", C.c_int32), ("max_chain", C.c_uint32), ("reserved", C.c_uint32), ("n_entries", C.c_uint64),
                ("part", C.c_uint32), ("n_parts", C.c_uint32), ("part_slot_lo", C.c_uint64), ("part_slot_hi", C.c_uint64),
                ("n_slots_whole", C.c_uint64), ("side_kmers", C.c_uint64), ("side_bytes", C.c_uint64)]


class MicSynthSpec(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("htsize", C.c_uint64), ("genome_nt", C.c_uint64), ("n_targets", C.c_uint32),
                ("n_genomes", C.c_uint32), ("k", C.c_int32), ("key_bytes", C.c_int32), ("keep_ppm", C.c_uint32), ("run_len", C.c_uint32),
                ("repeat_ppm", C.c_uint32), ("mosaic_ppm", C.c_uint32)]


class MicIngestResult(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("csv_bytes", C.c_uint64), ("csv", C.c_void_p), ("results", C.c_void_p),
                ("status", C.c_uint32), ("n_lines", C.c_uint32)]


MIC_INGEST_OK, MIC_INGEST_FALLBACK, MIC_INGEST_ODD_RECORD, MIC_INGEST_TRUNCATED = 0, 1, 2, 4
MIC_INGEST_LONG_READ, MIC_INGEST_TOO_MANY, MIC_INGEST_DENSE = 8, 16, 32

# every symbol include/mi_clark.h declares: (name, restype, argtypes)
_VP, _SZ, _U32P, _U16P, _U64P = C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p
SYMBOLS = [
    ("mic_create", C.c_int, [C.POINTER(MicConfig), C.POINTER(_VP)]),
    ("mic_destroy", C.c_int, [_VP]),
    ("mic_last_error", C.c_char_p, []),
    ("mic_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("mic_db_load_files", C.c_int, [_VP, C.c_char_p, C.c_int, C.c_uint32, C.c_uint64, C.c_uint64]),
    ("mic_db_load_host", C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_int, _VP, C.c_uint32, C.c_uint64, C.c_uint64]),
    ("mic_db_load_device", C.c_int, [_VP, _VP, C.c_uint64, _VP, C.c_int, _VP, C.c_uint32, C.c_uint64, C.c_uint64]),
    ("mic_db_set_part", C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    ("mic_db_reserve_hbm", C.c_int, [_VP, C.c_uint64]),
    ("mic_peer_matrix", C.c_int, [C.POINTER(C.c_int), C.c_int]),
    ("mic_device_memory", C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("mic_db_load_files_multi", C.c_int, [C.POINTER(C.c_void_p), _SZ, C.c_char_p, C.c_int, C.c_uint32]),
    ("mic_db_kernel_name", C.c_int, [_VP, C.c_char_p, _SZ]),
    ("mic_db_get_info", C.c_int, [_VP, C.POINTER(MicDbInfo)]),
    ("mic_db_unload", C.c_int, [_VP]),
    ("mic_db_last_build_report", C.c_char_p, []),
    ("mic_batches_alloc", C.c_int, [_VP, _SZ, _SZ, _SZ, _U32P, C.c_int, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_VP),
                                    C.POINTER(_VP)]),
    ("mic_batch_ready", C.c_int, [_VP, _SZ, _SZ, _SZ]),
    ("mic_batch_query", C.c_int, [_VP, _SZ, C.c_int, C.c_int]),
    ("mic_batch_wait", C.c_int, [_VP, _SZ]),
    ("mic_batch_dense_counts", C.c_int, [_VP, _SZ, _SZ, _VP]),
    ("mic_batch_check", C.c_int, [_VP, _SZ, C.POINTER(C.c_int)]),
    ("mic_batch_merge_shards", C.c_int, [C.POINTER(C.c_void_p), _SZ, _SZ]),
    ("mic_batch_query_group", C.c_int, [C.POINTER(C.c_void_p), _SZ, _SZ, C.c_int]),
    ("mic_sync", C.c_int, [_VP]),
    ("mic_thread_bind_near_device", C.c_int, [_VP, C.c_int]),
    ("mic_batches_free", C.c_int, [_VP]),
    ("mic_query_device", C.c_int, [_VP, _VP, _VP, _SZ, _VP, _VP, _VP]),
    ("mic_resolve_flagged_device", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(_SZ)]),
    ("mic_merge_rows_device", C.c_int, [_VP, _VP, _VP, _VP, _SZ, _VP]),
    ("mic_result_from_rows_device", C.c_int, [_VP, _VP, _VP, _SZ, _VP]),
    ("mic_count_dense_device", C.c_int, [_VP, _VP, _VP, _VP, _SZ, _VP, _VP]),
    ("mic_result_from_dense_device", C.c_int, [_VP, _VP, _VP, _SZ, _VP, _VP, _VP]),
    ("mic_probe_stats_device", C.c_int, [_VP, _VP, _VP, _SZ, C.POINTER(C.c_uint64)]),
    ("mic_last_query_ms", C.c_int, [_VP, C.POINTER(C.c_float)]),
    ("mic_last_crowd_stats", C.c_int, [_VP, C.POINTER(C.c_uint32)]),
    ("mic_debug_fetch_crowd", C.c_int, [_VP, _VP, _SZ, C.POINTER(C.c_uint32)]),
    ("mic_debug_stream_pool", C.c_int, [C.POINTER(C.c_uint32)]),
    ("mic_ingest_alloc", C.c_int, [_VP, _SZ, _SZ, C.POINTER(C.c_char_p), C.c_uint32, C.c_int, C.POINTER(_VP)]),
    ("mic_ingest_classify", C.c_int, [_VP, _SZ, _SZ, C.c_int, C.POINTER(MicIngestResult)]),
    ("mic_ingest_classify_group", C.c_int, [C.POINTER(C.c_void_p), _SZ, _SZ, _SZ, _SZ, C.c_int, C.POINTER(MicIngestResult)]),
    ("mic_ingest_fetch_group_rows", C.c_int, [_VP, _SZ, _SZ, _VP, _SZ, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("mic_ingest_group_stats", C.c_int, [_VP, C.POINTER(C.c_double), _SZ]),
    ("mic_ingest_fetch_packed", C.c_int, [_VP, _SZ, _VP, _SZ, _VP, _SZ, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("mic_ingest_free", C.c_int, [_VP]),
    ("mic_gz_inflate_device", C.c_int, [_VP, _VP, _SZ, C.POINTER(_VP), C.POINTER(_SZ), C.POINTER(C.c_uint32)]),
    ("mic_gz_copy_text", C.c_int, [_VP, _VP, _SZ, _SZ, _VP]),
    ("mic_gz_free_text", C.c_int, [_VP, _VP]),
    ("mic_gz_reserve", C.c_int, [_VP, _SZ, C.c_uint32]),
    ("mic_gz_reserve_bytes", C.c_uint64, [_SZ, C.c_uint32]),
    ("mic_gz_release", C.c_int, [_VP]),
    ("mic_pairs_index_device", C.c_int, [_VP, _VP, _SZ, _VP, _SZ, C.POINTER(_VP), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("mic_pairs_offsets", C.c_int, [_VP, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(_SZ), C.POINTER(C.c_uint32)]),
    ("mic_pairs_merge_to_slot", C.c_int, [_VP, _VP, C.c_uint64, C.c_uint64, _SZ, C.POINTER(_SZ)]),
    ("mic_pairs_text", C.c_int, [_VP, _VP, C.c_uint64, C.c_uint64, _VP, _SZ, C.POINTER(_SZ)]),
    ("mic_pairs_free", C.c_int, [_VP, _VP]),
    ("mic_text_index_device", C.c_int, [_VP, _VP, _SZ, C.POINTER(_VP), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("mic_text_index_front_device", C.c_int, [_VP, _VP, _SZ, C.POINTER(_VP), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("mic_gz_stream_open", C.c_int, [_VP, _VP, _SZ, C.c_uint32, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(_SZ)]),
    ("mic_gz_stream_next", C.c_int, [_VP, C.POINTER(_SZ), C.POINTER(C.c_int)]),
    ("mic_gz_stream_close", C.c_int, [_VP, C.c_int]),
    ("mic_text_offsets", C.c_int, [_VP, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(_SZ), C.POINTER(C.c_uint32)]),
    ("mic_text_to_slot", C.c_int, [_VP, _VP, C.c_uint64, C.c_uint64, _SZ, C.POINTER(_SZ)]),
    ("mic_text_copy", C.c_int, [_VP, _VP, C.c_uint64, C.c_uint64, _VP, _SZ, C.POINTER(_SZ)]),
    ("mic_text_free", C.c_int, [_VP, _VP]),
    ("mic_text_format", C.c_int, [_VP]),
    ("mic_format_ratio_g", C.c_int, [C.c_uint32, C.c_uint32, C.c_char_p]),
    ("mic_key_bytes_rule", C.c_int, [C.c_uint64, C.c_int]),
    ("mic_index_reads", C.c_long, [_VP, _SZ, _SZ, _U64P, _U64P, _U64P, _U64P, _U64P]),
    ("mic_index_reads_parallel", C.c_long, [_VP, _SZ, C.c_int, _SZ, _U64P, _U64P, _U64P, _U64P, _U64P]),
    ("mic_find_record_start", _SZ, [_VP, _SZ, _SZ]),
    ("mic_find_record_start_in", _SZ, [_VP, _SZ, C.c_int, _SZ]),
    ("mic_pack_bound", _SZ, [_U64P, _U64P, _SZ, C.c_int]),
    ("mic_pack_reads", _SZ, [_VP, _U64P, _U64P, _U64P, _SZ, C.c_int, _U32P, _U16P, _SZ]),
    ("mic_csv_header", C.c_int, [C.c_char_p, _SZ, C.c_int, C.POINTER(C.c_char_p), C.c_uint32]),
    ("mic_csv_line", C.c_int, [C.c_char_p, _SZ, _VP, _SZ, C.c_uint64, C.c_int, C.c_int, _VP, C.POINTER(C.c_char_p),
                               C.c_uint32, C.c_int, _VP, _VP]),
    ("mic_db_build", C.c_int, [C.POINTER(C.c_char_p), _VP, _SZ, C.c_int, C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p,
                               C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("mic_db_build_error", C.c_char_p, []),
    ("mic_synth_db_device", C.c_int, [C.POINTER(MicSynthSpec), _VP, _VP, _VP, C.c_uint64, C.POINTER(C.c_uint64), _VP]),
    ("mic_synth_read_pitch", C.c_uint32, [C.c_uint32, C.c_int]),
    ("mic_synth_reads_device", C.c_int, [C.POINTER(MicSynthSpec), C.c_uint64, _SZ, C.c_uint32, C.c_double, C.c_double,
                                         C.c_double, _VP, _VP, _SZ, _VP, _VP]),
    ("mic_synth_reads_device2", C.c_int, [C.POINTER(MicSynthSpec), C.c_uint64, _SZ, C.c_uint32, C.c_int, C.c_double, C.c_double,
                                          C.c_double, _VP, _VP, _SZ, _VP, _VP]),
    ("mic_synth_text_record_bytes", _SZ, [C.c_uint32, C.c_int]),
    ("mic_synth_reads_text_device", C.c_int, [C.POINTER(MicSynthSpec), C.c_uint64, _SZ, C.c_uint32, C.c_double, C.c_double,
                                              C.c_double, C.c_int, C.c_int, _VP, _SZ, _VP]),
]

_lib = None


def build(verbose=False):
    """Compile libmi_clark.so (and the CLI) for gfx950 with hipcc — works without a GPU."""
    r = subprocess.run(["make", "-C", CSRC, "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libmi_clark.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout)
    return LIB_PATH


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes in one
    process cannot both own the GPU, so when torch is installed its copy is loaded first (by full path, RTLD_GLOBAL):
    libmi_clark.so's DT_NEEDED libamdhip64.so.7 then resolves to it, and a later `import torch` reuses it too.
    The standalone CLI (exe/cuCLARK) links /opt/rocm's runtime directly."""
    if os.environ.get("MIC_NO_TORCH_HIP"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            C.CDLL(p, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load the library and bind every declared symbol; raises if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the k-mer query path)")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise MicError(rc, load().mic_last_error().decode(errors="replace"))

"""ctypes binding of oracle/liboracle.so (the CPU restatement, clark_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (cuclark_amd) must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
# ORACLE_LIB_PATH: another build of the same sources (tools/sanitize/oracle_rig.sh: AddressSanitizer + UBSan)
_LIB_OVERRIDE = os.environ.get("ORACLE_LIB_PATH")


class _OrcDb(C.Structure):
    _fields_ = [("htsize", C.c_uint64), ("key_bytes", C.c_int), ("n_elems", C.c_uint64),
                ("bucket_off", C.POINTER(C.c_uint64)), ("keys", C.c_void_p), ("labels", C.POINTER(C.c_uint16)),
                ("borrowed", C.c_int)]


class _OrcIndex(C.Structure):
    _fields_ = [("n_reads", C.c_size_t), ("name_s", C.POINTER(C.c_uint64)), ("name_e", C.POINTER(C.c_uint64)),
                ("seq_s", C.POINTER(C.c_uint64)), ("seq_e", C.POINTER(C.c_uint64)), ("length", C.POINTER(C.c_uint64))]


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    srcs = [os.path.join(_HERE, f) for f in ("clark_oracle.c", "part_rule.c", "clark_oracle.h")]
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)
    return _LIB


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self):
        if not _LIB_OVERRIDE:
            build()
        L = C.CDLL(_LIB_OVERRIDE or _LIB)
        self.L = L
        L.orc_nt_code.restype = C.c_int
        L.orc_nt_code.argtypes = [C.c_uint8]
        L.orc_revcomp.restype = C.c_uint64
        L.orc_revcomp.argtypes = [C.c_uint64, C.c_int]
        L.orc_kmer_from_ascii.restype = C.c_uint64
        L.orc_kmer_from_ascii.argtypes = [C.c_char_p, C.c_int]
        L.orc_canonical.restype = C.c_uint64
        L.orc_canonical.argtypes = [C.c_uint64, C.c_int]
        L.orc_key_bytes_rule.restype = C.c_int
        L.orc_key_bytes_rule.argtypes = [C.c_uint64, C.c_int]
        L.orc_db_load.restype = C.POINTER(_OrcDb)
        L.orc_db_load.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_uint32]
        L.orc_db_from_arrays.restype = C.POINTER(_OrcDb)
        L.orc_db_from_arrays.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
        L.orc_db_wrap_arrays.restype = C.POINTER(_OrcDb)
        L.orc_db_wrap_arrays.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_classify_batch_fast.restype = C.c_uint64
        L.orc_classify_batch_fast.argtypes = [C.POINTER(_OrcDb), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                              C.c_void_p, C.c_int]
        L.orc_db_copy_spread.restype = C.POINTER(_OrcDb)
        L.orc_db_copy_spread.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_numa_db_create.restype = C.c_void_p
        L.orc_numa_db_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_numa_db_free.argtypes = [C.c_void_p]
        L.orc_classify_batch_numa.restype = C.c_uint64
        L.orc_classify_batch_numa.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
        L.orc_classify_batch.restype = C.c_uint64
        L.orc_classify_batch.argtypes = [C.POINTER(_OrcDb), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                         C.c_void_p, C.c_int]
        L.orc_db_free.argtypes = [C.POINTER(_OrcDb)]
        L.orc_db_find.restype = C.c_int
        L.orc_db_find.argtypes = [C.POINTER(_OrcDb), C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint16)]
        L.orc_probe_stats.argtypes = [C.POINTER(_OrcDb), C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double),
                                      C.POINTER(C.c_uint64)]
        L.orc_pack_batch.restype = C.c_size_t
        L.orc_pack_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_size_t]
        L.orc_query_batch.restype = C.c_uint64
        L.orc_query_batch.argtypes = [C.POINTER(_OrcDb), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64,
                                      C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_count_read_ascii.restype = C.c_uint64
        L.orc_count_read_ascii.argtypes = [C.POINTER(_OrcDb), C.c_int, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64,
                                           C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_sparse_row.restype = C.c_uint32
        L.orc_sparse_row.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_merge_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_result_from_row.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_result_from_counts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_index_reads.restype = C.POINTER(_OrcIndex)
        L.orc_index_reads.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_index_free.argtypes = [C.POINTER(_OrcIndex)]
        L.orc_classify_file.restype = C.c_long
        L.orc_classify_file.argtypes = [C.POINTER(_OrcDb), C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p),
                                        C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_void_p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_part_slot_of_kmer.restype = C.c_uint32
        L.orc_part_slot_of_kmer.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint32]
        L.orc_query_batch_slot_part.restype = C.c_uint64
        L.orc_query_batch_slot_part.argtypes = [C.POINTER(_OrcDb), C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]

    # -- codec
    def revcomp(self, x, k):
        return int(self.L.orc_revcomp(int(x), k))

    def canonical(self, x, k):
        return int(self.L.orc_canonical(int(x), k))

    def key_bytes_rule(self, htsize, k):
        return int(self.L.orc_key_bytes_rule(int(htsize), k))

    # -- db
    def db_load(self, prefix, htsize=0, key_bytes=4, sampling=1):
        p = self.L.orc_db_load(prefix.encode(), int(htsize), key_bytes, sampling)
        if not p:
            raise RuntimeError(f"orc_db_load failed for {prefix}")
        return OracleDb(self, p)

    def db_from_arrays(self, sizes, keys, labels, sampling=1):
        sizes = np.ascontiguousarray(sizes, np.uint8)
        keys = np.ascontiguousarray(keys)
        labels = np.ascontiguousarray(labels, np.uint16)
        p = self.L.orc_db_from_arrays(sizes.ctypes.data, sizes.size, keys.ctypes.data, keys.dtype.itemsize,
                                      labels.ctypes.data, sampling)
        if not p:
            raise RuntimeError("orc_db_from_arrays failed")
        return OracleDb(self, p)

    def db_wrap_arrays(self, sizes, keys, labels):
        """Zero-copy: the numpy arrays must stay alive as long as the returned OracleDb."""
        assert sizes.dtype == np.uint8 and labels.dtype == np.uint16 and sizes.flags.c_contiguous
        p = self.L.orc_db_wrap_arrays(sizes.ctypes.data, sizes.size, keys.ctypes.data, keys.dtype.itemsize,
                                      labels.ctypes.data)
        if not p:
            raise RuntimeError("orc_db_wrap_arrays failed")
        db = OracleDb(self, p)
        db._keep = (sizes, keys, labels)
        return db

    def db_copy_spread(self, sizes, keys, labels, threads=0):
        """A private copy written by all threads (pages on every NUMA node, huge pages requested): the CPU baseline's table."""
        assert sizes.dtype == np.uint8 and labels.dtype == np.uint16 and sizes.flags.c_contiguous
        p = self.L.orc_db_copy_spread(sizes.ctypes.data, sizes.size, keys.ctypes.data, keys.dtype.itemsize, labels.ctypes.data,
                                      int(threads))
        if not p:
            raise RuntimeError("orc_db_copy_spread failed")
        return OracleDb(self, p)

    def numa_db(self, sizes, keys, labels, threads=0):
        """One replica of the table per NUMA node (threads pinned to their node's replica): the CPU baseline's table."""
        assert sizes.dtype == np.uint8 and labels.dtype == np.uint16 and sizes.flags.c_contiguous
        p = self.L.orc_numa_db_create(sizes.ctypes.data, sizes.size, keys.ctypes.data, keys.dtype.itemsize, labels.ctypes.data, int(threads))
        if not p:
            raise RuntimeError("orc_numa_db_create failed")
        return NumaDb(self, p)

    # -- rows / results
    def sparse_row(self, counts, max_pairs=64):
        counts = np.ascontiguousarray(counts, np.uint32)
        row = np.zeros(1 + 2 * max_pairs, np.uint16)
        n = self.L.orc_sparse_row(counts.ctypes.data, counts.size, row.ctypes.data, max_pairs)
        return int(n), row

    def merge_rows(self, a, b):
        a = np.ascontiguousarray(a, np.uint16)
        b = np.ascontiguousarray(b, np.uint16)
        out = np.zeros(a.size + b.size, np.uint16)
        self.L.orc_merge_rows(a.ctypes.data, b.ctypes.data, out.ctypes.data)
        return out

    def result_from_row(self, row):
        row = np.ascontiguousarray(row, np.uint16)
        out = np.zeros(5, np.uint32)
        self.L.orc_result_from_row(row.ctypes.data, out.ctypes.data)
        return out

    def result_from_counts(self, counts):
        counts = np.ascontiguousarray(counts, np.uint32)
        if counts.ndim == 1:
            out = np.zeros(5, np.uint32)
            self.L.orc_result_from_counts(counts.ctypes.data, counts.size, out.ctypes.data)
            return out
        out = np.zeros((counts.shape[0], 5), np.uint32)
        for i in range(counts.shape[0]):
            self.L.orc_result_from_counts(counts[i].ctypes.data, counts.shape[1], out[i].ctypes.data)
        return out

    # -- reads
    def index_reads(self, data):
        buf = np.frombuffer(data, np.uint8)
        ix = self.L.orc_index_reads(buf.ctypes.data, buf.size)
        if not ix:
            return None
        n = ix.contents.n_reads
        out = {f: np.ctypeslib.as_array(getattr(ix.contents, f), (n,)).copy() if n else np.zeros(0, np.uint64)
               for f in ("name_s", "name_e", "seq_s", "seq_e", "length")}
        self.L.orc_index_free(ix)
        return out

    def pack_batch(self, data, seq_s, seq_e, length, k):
        buf = np.frombuffer(data, np.uint8)
        n = len(seq_s)
        seq_s = np.ascontiguousarray(seq_s, np.uint64)
        seq_e = np.ascontiguousarray(seq_e, np.uint64)
        length = np.ascontiguousarray(length, np.uint64)
        cap = int((seq_e - seq_s).sum() // 4 + 4 * n + 64)
        rp = np.zeros(n + 1, np.uint32)
        cont = np.zeros(cap, np.uint16)
        m = self.L.orc_pack_batch(buf.ctypes.data, seq_s.ctypes.data, seq_e.ctypes.data, length.ctypes.data, n, k,
                                  rp.ctypes.data, cont.ctypes.data, cap)
        if m == C.c_size_t(-1).value:
            raise RuntimeError("orc_pack_batch: capacity")
        return rp, cont[:m].copy()


class NumaDb:
    def __init__(self, orc, p):
        self.orc, self.p = orc, p

    def classify_batch(self, k, reads_pointer, containers, n_targets):
        rp = np.ascontiguousarray(reads_pointer, np.uint32)
        ct = np.ascontiguousarray(containers, np.uint16)
        n = rp.size - 1
        res = np.zeros((n, 5), np.uint32)
        bad = self.orc.L.orc_classify_batch_numa(self.p, k, rp.ctypes.data, ct.ctypes.data, n, n_targets, res.ctypes.data)
        assert bad == 0
        return res

    def close(self):
        if self.p:
            self.orc.L.orc_numa_db_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleDb:
    def __init__(self, orc, p):
        self.orc, self.p = orc, p
        self.htsize = int(p.contents.htsize)
        self.key_bytes = int(p.contents.key_bytes)
        self.n_elems = int(p.contents.n_elems)

    def close(self):
        if self.p:
            self.orc.L.orc_db_free(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def find(self, kmer, k, part=(0, None)):
        lab = C.c_uint16(0)
        pe = self.htsize if part[1] is None else part[1]
        f = self.orc.L.orc_db_find(self.p, int(kmer), k, int(part[0]), int(pe), C.byref(lab))
        return bool(f), int(lab.value)

    def find_many(self, kmers, k, part=(0, None)):
        found = np.zeros(len(kmers), np.uint8)
        label = np.zeros(len(kmers), np.uint16)
        for i, v in enumerate(kmers):
            f, l = self.find(int(v), k, part)
            found[i], label[i] = f, l if f else 0
        return found, label

    def probe_stats(self, kmers, k):
        kmers = np.ascontiguousarray(kmers, np.uint64)
        m, h = C.c_double(0), C.c_uint64(0)
        self.orc.L.orc_probe_stats(self.p, kmers.ctypes.data, kmers.size, k, C.byref(m), C.byref(h))
        return float(m.value), int(h.value)

    def query_batch(self, k, reads_pointer, containers, n_targets, part=(0, None)):
        rp = np.ascontiguousarray(reads_pointer, np.uint32)
        ct = np.ascontiguousarray(containers, np.uint16)
        ct = np.concatenate([ct, np.zeros(8, np.uint16)])
        n = rp.size - 1
        counts = np.zeros((n, n_targets), np.uint32)
        pe = self.htsize if part[1] is None else part[1]
        bad = self.orc.L.orc_query_batch(self.p, k, rp.ctypes.data, ct.ctypes.data, n, int(part[0]), int(pe), n_targets,
                                         counts.ctypes.data)
        return counts, int(bad)

    def query_batch_slot_part(self, k, m, both_strands, n_slots, part, n_parts, reads_pointer, containers, n_targets):
        """dense counts of the k-mer occurrences that part `part` of `n_parts` of the product's resident slot range answers for
        (part_rule.c); the parts sum to query_batch's counts"""
        rp = np.ascontiguousarray(reads_pointer, np.uint32)
        ct = np.concatenate([np.ascontiguousarray(containers, np.uint16), np.zeros(8, np.uint16)])
        n = rp.size - 1
        counts = np.zeros((n, n_targets), np.uint32)
        bad = self.orc.L.orc_query_batch_slot_part(self.p, k, m, int(bool(both_strands)), int(n_slots), int(part), int(n_parts),
                                                   rp.ctypes.data, ct.ctypes.data, n, n_targets, counts.ctypes.data)
        return counts, int(bad)

    def classify_batch(self, k, reads_pointer, containers, n_targets, threads=0):
        rp = np.ascontiguousarray(reads_pointer, np.uint32)
        ct = np.ascontiguousarray(containers, np.uint16)
        n = rp.size - 1
        res = np.zeros((n, 5), np.uint32)
        bad = self.orc.L.orc_classify_batch(self.p, k, rp.ctypes.data, ct.ctypes.data, n, n_targets, res.ctypes.data, threads)
        assert bad == 0
        return res

    def classify_batch_fast(self, k, reads_pointer, containers, n_targets, threads=0):
        """orc_classify_batch's results through the prefetching form (the CPU baseline bench.py times)."""
        rp = np.ascontiguousarray(reads_pointer, np.uint32)
        ct = np.ascontiguousarray(containers, np.uint16)
        n = rp.size - 1
        res = np.zeros((n, 5), np.uint32)
        bad = self.orc.L.orc_classify_batch_fast(self.p, k, rp.ctypes.data, ct.ctypes.data, n, n_targets, res.ctypes.data, threads)
        assert bad == 0
        return res

    def count_read_ascii(self, k, seq, length, n_targets, part=(0, None)):
        buf = np.frombuffer(seq, np.uint8)
        counts = np.zeros(n_targets, np.uint32)
        pe = self.htsize if part[1] is None else part[1]
        self.orc.L.orc_count_read_ascii(self.p, k, buf.ctypes.data, buf.size, int(length), int(part[0]), int(pe),
                                        n_targets, counts.ctypes.data)
        return counts

    def classify_file(self, k, data, target_names, paired=False, extended=False):
        buf = np.frombuffer(data, np.uint8)
        names = (C.c_char_p * len(target_names))(*[t.encode() for t in target_names])
        csv, n, res = C.c_void_p(), C.c_size_t(0), C.c_void_p()
        nr = self.orc.L.orc_classify_file(self.p, k, buf.ctypes.data, buf.size, names, len(target_names), int(paired),
                                          int(extended), C.byref(csv), C.byref(n), C.byref(res))
        if nr < 0:
            raise RuntimeError("orc_classify_file: unrecognised format")
        text = C.string_at(csv.value, n.value)
        results = np.ctypeslib.as_array(C.cast(res.value, C.POINTER(C.c_uint32)), (nr, 5)).copy() if nr else np.zeros((0, 5), np.uint32)
        self.orc.L.orc_free(csv)
        self.orc.L.orc_free(res)
        return text, results

"""MiClarkDB — host-side mirror of the reference's CuClarkDB<HKMERr> interface over the C ABI.

Method names and argument meaning follow CuClarkDB.cuh:98-150 (read, malloc, readyBatch, queryBatch,
swapDbParts, waitForBatch, checkBatch, sync, freeBatchMemory) so that tests read like calls the reference's
CuCLARK_hh.hh makes.  Differences, all stated in include/mi_clark.h: the key width is a runtime value, results
are u32 rows of 8 words, the whole table is resident (swapDbParts never has another part), and errors raise
MicError instead of exit(1).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MicConfig, MicDbInfo, MicError, check, MIC_RESULT_WORDS


class MiClarkUnsupported(RuntimeError):
    """mic_gz_*: an input the device path does not take (MIC_E_UNSUPPORTED); the caller uses its other path."""


def _as_np(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


class MiClarkDB:
    def __init__(self, k, num_targets, num_batches=1, device=-1, row_words=16, layout=0):
        self.L = _lib.load()
        self.k = int(k)
        self.num_targets = int(num_targets)
        self.num_batches = int(num_batches)
        self.row_words = int(row_words)
        cfg = MicConfig(device, self.k, self.num_targets, self.num_batches, self.row_words, int(layout))
        h = C.c_void_p()
        check(self.L.mic_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._bufs = None

    # -- lifetime
    def close(self):
        if getattr(self, "h", None):
            self.L.mic_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- database (CuClarkDB::read + swapDbParts)
    def read(self, filename, sampling=1, key_bytes=0, shard=(0, 0)):
        """Returns False when a DB file cannot be opened (reference: read() returns false, CuClarkDB.cu:490-495)."""
        rc = self.L.mic_db_load_files(self.h, filename.encode(), int(key_bytes), int(sampling), int(shard[0]), int(shard[1]))
        if rc == -2:
            return False
        check(rc)
        return True

    def read_arrays(self, sizes, keys, labels, sampling=1, shard=(0, 0)):
        sizes = np.ascontiguousarray(sizes, np.uint8)
        keys = np.ascontiguousarray(keys)
        labels = np.ascontiguousarray(labels, np.uint16)
        assert keys.dtype.itemsize in (2, 4, 8) and keys.size == labels.size
        check(self.L.mic_db_load_host(self.h, sizes.ctypes.data, sizes.size, keys.ctypes.data if keys.size else sizes.ctypes.data,
                                      keys.dtype.itemsize, labels.ctypes.data if labels.size else sizes.ctypes.data,
                                      int(sampling), int(shard[0]), int(shard[1])))

    def read_device(self, d_sizes, htsize, d_keys, key_bytes, d_labels, sampling=1, shard=(0, 0)):
        check(self.L.mic_db_load_device(self.h, d_sizes, int(htsize), d_keys, int(key_bytes), d_labels, int(sampling),
                                        int(shard[0]), int(shard[1])))

    def set_part(self, part, n_parts):
        """Table-sharded runs (CuClarkDB.cu:566-574): answer for part `part` of `n_parts` of the database; call before
        read*().  Super-k-mer layouts cut the resident table by slot range, the others by on-disk bucket range."""
        check(self.L.mic_db_set_part(self.h, int(part), int(n_parts)))

    def swapDbParts(self):
        """The table is fully resident: there is never another part to swap in (CuClarkDB.cu:813-858)."""
        return False

    def info(self):
        i = MicDbInfo()
        check(self.L.mic_db_get_info(self.h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in MicDbInfo._fields_}

    # -- batch API
    def malloc(self, num_reads, max_reads, max_containers, index_batches, extended=False):
        ib = np.ascontiguousarray(index_batches, np.uint32)
        assert ib.size == self.num_batches + 1
        res, rows = C.c_void_p(), C.c_void_p()
        rp = (C.c_void_p * self.num_batches)()
        ct = (C.c_void_p * self.num_batches)()
        check(self.L.mic_batches_alloc(self.h, num_reads, max_reads, max_containers, ib.ctypes.data, int(bool(extended)),
                                       C.byref(res), C.byref(rows), rp, ct))
        self._bufs = dict(
            results=_as_np(res.value, (num_reads, MIC_RESULT_WORDS), np.uint32),
            rows=_as_np(rows.value, (num_reads, self.row_words), np.uint32) if extended else None,
            reads_pointer=[_as_np(rp[b], (max_reads + 1,), np.uint32) for b in range(self.num_batches)],
            containers=[_as_np(ct[b], (max_containers,), np.uint16) for b in range(self.num_batches)],
        )
        return self._bufs

    def readyBatch(self, batch, num_reads, container_count):
        check(self.L.mic_batch_ready(self.h, batch, num_reads, container_count))
        return True

    def queryBatch(self, batch, extended=False, followup=False):
        check(self.L.mic_batch_query(self.h, batch, int(bool(extended)), int(bool(followup))))
        return True

    def waitForBatch(self, batch):
        check(self.L.mic_batch_wait(self.h, batch))
        return True

    @staticmethod
    def merge_shards(engines, batch):
        """Table-sharded batches (CuClarkDB.cu:934-1001): sum the sparse rows of `batch` of every engine into engines[0]
        and finish best / second-best there.  Every engine must have queried the same reads with extended=True."""
        L = engines[0].L
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        check(L.mic_batch_merge_shards(arr, len(engines), batch))

    @staticmethod
    def query_group(engines, batch, extended=True):
        """queryBatch on every engine of a table-sharded group from ONE upload: the reads are engines[0]'s (its lent buffers, its
        readyBatch); the others take the packed reads device to device (mic_batch_query_group; CuClarkDB.cu:886-890)."""
        L = engines[0].L
        arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
        check(L.mic_batch_query_group(arr, len(engines), batch, int(bool(extended))))

    def checkBatch(self, batch):
        d = C.c_int(0)
        check(self.L.mic_batch_check(self.h, batch, C.byref(d)))
        return bool(d.value)

    def sync(self):
        check(self.L.mic_sync(self.h))
        return True

    def freeBatchMemory(self):
        self._bufs = None
        check(self.L.mic_batches_free(self.h))

    # -- convenience: one batch through the batch API
    def classify_packed(self, reads_pointer, containers, extended=False):
        assert self.num_batches == 1
        rp = np.ascontiguousarray(reads_pointer, np.uint32)
        ct = np.ascontiguousarray(containers, np.uint16)
        n = rp.size - 1
        bufs = self.malloc(n, n, max(ct.size, 1), [0, n], extended)
        bufs["reads_pointer"][0][: n + 1] = rp
        bufs["containers"][0][: ct.size] = ct
        self.readyBatch(0, n, ct.size)
        self.queryBatch(0, extended)
        self.waitForBatch(0)
        res = bufs["results"].copy()
        rows = bufs["rows"].copy() if extended else None
        self.freeBatchMemory()
        return (res, rows) if extended else res

    # -- compressed input: one gzip member inflated on the device (mic_gz_*)
    def gunzip(self, gz_bytes):
        """The text of a one-member .gz file, inflated on the device and copied back.  Returns (text bytes, crc32 of the
        member's trailer), or raises MiClarkUnsupported when the file is not of the kind this path takes."""
        buf = np.frombuffer(gz_bytes, np.uint8)
        d_text, n, crc = C.c_void_p(), C.c_size_t(0), C.c_uint32(0)
        rc = self.L.mic_gz_inflate_device(self.h, buf.ctypes.data, buf.size, C.byref(d_text), C.byref(n), C.byref(crc))
        if rc == -7:
            raise MiClarkUnsupported(self.L.mic_last_error().decode())
        check(rc)
        out = np.empty(n.value, np.uint8)
        try:
            if n.value:
                check(self.L.mic_gz_copy_text(self.h, d_text, 0, n.value, out.ctypes.data))
        finally:
            self.L.mic_gz_free_text(self.h, d_text)
        return out.tobytes(), int(crc.value)

    def gunzip_device(self, gz_bytes):
        """As gunzip, but the text stays on the device: (device pointer, size, crc32); free_text() releases it."""
        buf = np.frombuffer(gz_bytes, np.uint8)
        d_text, n, crc = C.c_void_p(), C.c_size_t(0), C.c_uint32(0)
        rc = self.L.mic_gz_inflate_device(self.h, buf.ctypes.data, buf.size, C.byref(d_text), C.byref(n), C.byref(crc))
        if rc == -7:
            raise MiClarkUnsupported(self.L.mic_last_error().decode())
        check(rc)
        return d_text.value, int(n.value), int(crc.value)

    def free_text(self, d_text):
        self.L.mic_gz_free_text(self.h, d_text)

    def gunzip_stripes(self, gz_bytes, stripes, on_stripe=None):
        """The same member in stripes (mic_gz_stream_*): yields nothing, calls on_stripe(d_text, n_final, done) after every stripe and
        returns (text bytes, [n_final of every stripe]).  Raises like gunzip - possibly after some stripes went through."""
        buf = np.frombuffer(gz_bytes, np.uint8)
        h, d_text, n = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        rc = self.L.mic_gz_stream_open(self.h, buf.ctypes.data, buf.size, int(stripes), C.byref(h), C.byref(d_text), C.byref(n))
        if rc == -7:
            raise MiClarkUnsupported(self.L.mic_last_error().decode())
        check(rc)
        finals = []
        try:
            done, nf = C.c_int(0), C.c_size_t(0)
            while not done.value:
                rc = self.L.mic_gz_stream_next(h, C.byref(nf), C.byref(done))
                if rc == -7:
                    raise MiClarkUnsupported(self.L.mic_last_error().decode())
                check(rc)
                finals.append(int(nf.value))
                if on_stripe:
                    on_stripe(d_text.value, int(nf.value), bool(done.value))
            out = np.empty(finals[-1], np.uint8)
            if out.size:
                check(self.L.mic_gz_copy_text(self.h, d_text, 0, out.size, out.ctypes.data))
            return out.tobytes(), finals
        finally:
            self.L.mic_gz_stream_close(h, 0)

    def text_index_front(self, d_text, n):
        """Whole FASTQ records at the front of a text that is still growing: (handle or None, records, bytes used, status)."""
        h, nr, used, st = C.c_void_p(), C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        check(self.L.mic_text_index_front_device(self.h, d_text, n, C.byref(h), C.byref(nr), C.byref(used), C.byref(st)))
        return h.value, int(nr.value), int(used.value), int(st.value)

    # -- paired-end FASTQ texts on the device: the reference's merge (file.cc:205-268) without the host (mic_pairs_*)
    def pairs_index(self, d_text1, n1, d_text2, n2):
        """Returns (handle, n_records, offsets, stride), or (None, status, None, None) when the texts need the host reader."""
        h, n, st = C.c_void_p(), C.c_uint64(0), C.c_uint32(0)
        check(self.L.mic_pairs_index_device(self.h, d_text1, n1, d_text2, n2, C.byref(h), C.byref(n), C.byref(st)))
        if st.value:
            return None, int(st.value), None, None
        sp, ns, stride = C.POINTER(C.c_uint64)(), C.c_size_t(0), C.c_uint32(0)
        check(self.L.mic_pairs_offsets(h, C.byref(sp), C.byref(ns), C.byref(stride)))
        off = np.ctypeslib.as_array(sp, shape=(ns.value,)).copy()
        return h, int(n.value), off, int(stride.value)

    def pairs_text(self, handle, r0, r1, cap=1 << 30):
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        check(self.L.mic_pairs_text(self.h, handle, r0, r1, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].tobytes()

    def pairs_classify(self, handle, slot, r0, r1):
        """Records [r0, r1) merged into the slot on the device and classified there: as ingest_classify(paired=True)."""
        n = C.c_size_t(0)
        check(self.L.mic_pairs_merge_to_slot(self.h, handle, r0, r1, slot, C.byref(n)))
        out = _lib.MicIngestResult()
        check(self.L.mic_ingest_classify(self.h, slot, n.value, 1 | 4, C.byref(out)))
        r = dict(status=int(out.status), n_reads=int(out.n_reads), n_lines=int(out.n_lines), csv=None, results=None, n_bytes=int(n.value))
        if out.status == 0:
            r["csv"] = C.string_at(out.csv, out.csv_bytes) if out.csv_bytes else b""
            if out.results:
                r["results"] = _as_np(out.results, (int(out.n_reads), MIC_RESULT_WORDS), np.uint32).copy()
        return r

    def pairs_free(self, handle):
        self.L.mic_pairs_free(self.h, handle)

    # -- one FASTQ text on the device (mic_text_*)
    def text_index(self, d_text, n):
        """Returns (handle, n_records, offsets, stride), or (None, status, None, None) when the text needs the host reader."""
        h, nr, st = C.c_void_p(), C.c_uint64(0), C.c_uint32(0)
        check(self.L.mic_text_index_device(self.h, d_text, n, C.byref(h), C.byref(nr), C.byref(st)))
        if st.value:
            return None, int(st.value), None, None
        sp, ns, stride = C.POINTER(C.c_uint64)(), C.c_size_t(0), C.c_uint32(0)
        check(self.L.mic_text_offsets(h, C.byref(sp), C.byref(ns), C.byref(stride)))
        off = np.ctypeslib.as_array(sp, shape=(ns.value,)).copy()
        return h, int(nr.value), off, int(stride.value)

    def text_copy(self, handle, r0, r1, cap=1 << 30):
        out = np.empty(cap, np.uint8)
        n = C.c_size_t(0)
        check(self.L.mic_text_copy(self.h, handle, r0, r1, out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].tobytes()

    def text_format(self, handle):
        return chr(self.L.mic_text_format(handle))

    def text_classify(self, handle, slot, r0, r1):
        n = C.c_size_t(0)
        check(self.L.mic_text_to_slot(self.h, handle, r0, r1, slot, C.byref(n)))
        out = _lib.MicIngestResult()
        check(self.L.mic_ingest_classify(self.h, slot, n.value, 12 if self.text_format(handle) == "@" else 4, C.byref(out)))
        r = dict(status=int(out.status), n_reads=int(out.n_reads), n_lines=int(out.n_lines), csv=None, results=None, n_bytes=int(n.value))
        if out.status == 0:
            r["csv"] = C.string_at(out.csv, out.csv_bytes) if out.csv_bytes else b""
            if out.results:
                r["results"] = _as_np(out.results, (int(out.n_reads), MIC_RESULT_WORDS), np.uint32).copy()
        return r

    def text_free(self, handle):
        self.L.mic_text_free(self.h, handle)

    # -- device-side ingest: raw FASTA/FASTQ bytes -> CSV text (mic_ingest_*)
    def ingest_alloc(self, n_slots, max_bytes, target_names, want_results=False):
        names = (C.c_char_p * len(target_names))(*[t.encode() for t in target_names])
        raw = (C.c_void_p * n_slots)()
        check(self.L.mic_ingest_alloc(self.h, n_slots, max_bytes, names, len(target_names), int(bool(want_results)), raw))
        self._ingest = dict(raw=[_as_np(raw[i], (max_bytes,), np.uint8) for i in range(n_slots)], max_bytes=max_bytes)
        return self._ingest["raw"]

    def ingest_classify(self, slot, data, paired=False):
        """data: bytes of whole records.  Returns dict(status, n_reads, csv (bytes), results (u32[n,8] or None))."""
        buf = np.frombuffer(data, np.uint8)
        self._ingest["raw"][slot][: buf.size] = buf
        out = _lib.MicIngestResult()
        check(self.L.mic_ingest_classify(self.h, slot, buf.size, int(bool(paired)), C.byref(out)))
        r = dict(status=int(out.status), n_reads=int(out.n_reads), n_lines=int(out.n_lines), csv=None, results=None)
        if out.status == 0:
            r["csv"] = C.string_at(out.csv, out.csv_bytes) if out.csv_bytes else b""
            if out.results:
                r["results"] = _as_np(out.results, (int(out.n_reads), MIC_RESULT_WORDS), np.uint32).copy()
        return r

    @staticmethod
    def ingest_classify_group(group, owner, slot, data, paired=False):
        """Table-sharded ingest (mic_ingest_classify_group): group[p] holds part p of len(group) parts, the slot belongs to group[owner]."""
        e = group[owner]
        buf = np.frombuffer(data, np.uint8)
        e._ingest["raw"][slot][: buf.size] = buf
        out = _lib.MicIngestResult()
        hs = (C.c_void_p * len(group))(*[g.h for g in group])
        check(e.L.mic_ingest_classify_group(hs, len(group), owner, slot, buf.size, int(bool(paired)), C.byref(out)))
        r = dict(status=int(out.status), n_reads=int(out.n_reads), n_lines=int(out.n_lines), csv=None, results=None)
        if out.status == 0:
            r["csv"] = C.string_at(out.csv, out.csv_bytes) if out.csv_bytes else b""
            if out.results:
                r["results"] = _as_np(out.results, (int(out.n_reads), MIC_RESULT_WORDS), np.uint32).copy()
        return r

    def ingest_fetch_group_rows(self, slot, part):
        """test hook: the partial rows engine `part` of the group computed for the slot's last table-sharded batch"""
        n, rw = C.c_uint64(0), C.c_uint32(0)
        check(self.L.mic_ingest_fetch_group_rows(self.h, slot, part, None, 0, C.byref(n), C.byref(rw)))
        rows = np.zeros((n.value, rw.value), np.uint32)
        check(self.L.mic_ingest_fetch_group_rows(self.h, slot, part, rows.ctypes.data, rows.size, None, None))
        return rows

    def ingest_fetch_packed(self, slot):
        n, m = C.c_uint64(0), C.c_uint64(0)
        check(self.L.mic_ingest_fetch_packed(self.h, slot, None, 0, None, 0, C.byref(n), C.byref(m)))
        rp = np.zeros(n.value + 1, np.uint32)
        ct = np.zeros(max(m.value, 1), np.uint16)
        check(self.L.mic_ingest_fetch_packed(self.h, slot, rp.ctypes.data, rp.size, ct.ctypes.data, ct.size, None, None))
        return rp, ct[: m.value]

    def ingest_free(self):
        self._ingest = None
        check(self.L.mic_ingest_free(self.h))

    # -- device-resident entry points (pointers are raw device addresses, e.g. torch.Tensor.data_ptr())
    def query_device(self, d_reads_pointer, d_containers, n_reads, d_results, d_rows=0, stream=0):
        check(self.L.mic_query_device(self.h, d_reads_pointer, d_containers, n_reads, d_results, d_rows or None,
                                      stream or None))

    def resolve_flagged_device(self, d_reads_pointer, d_containers, d_results, d_rows=0, stream=0):
        n = C.c_size_t(0)
        check(self.L.mic_resolve_flagged_device(self.h, d_reads_pointer, d_containers, d_results, d_rows or None,
                                                stream or None, C.byref(n)))
        return int(n.value)

    def merge_rows_device(self, d_a, d_b, d_out, n_reads, stream=0):
        check(self.L.mic_merge_rows_device(self.h, d_a, d_b, d_out, n_reads, stream or None))

    def result_from_rows_device(self, d_rows, d_results, n_reads, stream=0):
        check(self.L.mic_result_from_rows_device(self.h, d_rows, d_results, n_reads, stream or None))

    def count_dense_device(self, d_reads_pointer, d_containers, d_ids, n_ids, d_counts, stream=0):
        check(self.L.mic_count_dense_device(self.h, d_reads_pointer, d_containers, d_ids or None, n_ids, d_counts,
                                            stream or None))

    def result_from_dense_device(self, d_counts, d_ids, n_ids, d_results, d_rows=0, stream=0):
        check(self.L.mic_result_from_dense_device(self.h, d_counts, d_ids or None, n_ids, d_results, d_rows or None, stream or None))

    def probe_stats_device(self, d_reads_pointer, d_containers, n_reads):
        out = (C.c_uint64 * 4)()
        check(self.L.mic_probe_stats_device(self.h, d_reads_pointer, d_containers, n_reads, out))
        return dict(kmers=int(out[0]), probed=int(out[1]), hits=int(out[2]), bucket_len_sum=int(out[3]))

    def last_query_ms(self):
        ms = C.c_float(0)
        check(self.L.mic_last_query_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def last_crowd_stats(self):
        """reads / runs of crowded minimizers the last query_device launch handed to its follow-up kernel"""
        out = (C.c_uint32 * 4)()
        check(self.L.mic_last_crowd_stats(self.h, out))
        return dict(reads=int(out[0]), runs=int(out[1]), row_words=int(out[2]), reads_to_dense_path=int(out[3]))

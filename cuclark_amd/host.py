"""Host-side pieces of the path exposed by the C ABI: key-width rule, read indexer, packer, CSV.

These wrap the C++ implementations in cuclark_amd/csrc/mic_host.cpp (the same code the cuCLARK CLI uses);
nothing here touches the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib


def key_bytes_rule(htsize, k):
    """main.cc:274-316."""
    return int(_lib.load().mic_key_bytes_rule(int(htsize), int(k)))


def index_reads(data, threads=1):
    """CuCLARK_hh.hh:1339-1534 (one batch).  Returns dict of u64 arrays or None for an unknown format."""
    L = _lib.load()
    buf = np.frombuffer(data, np.uint8)
    cap = max(16, buf.size // 64)
    while True:
        arrs = [np.zeros(cap, np.uint64) for _ in range(5)]
        if threads > 1:
            n = L.mic_index_reads_parallel(buf.ctypes.data, buf.size, threads, cap, *[a.ctypes.data for a in arrs])
        else:
            n = L.mic_index_reads(buf.ctypes.data, buf.size, cap, *[a.ctypes.data for a in arrs])
        if n < 0:
            return None
        if n <= cap:
            return dict(zip(("name_s", "name_e", "seq_s", "seq_e", "length"), [a[:n].copy() for a in arrs]))
        cap = n


def pack_reads(data, seq_s, seq_e, length, k):
    """CuCLARK_hh.hh:1616-1716.  Returns (reads_pointer u32[n+1], containers u16[m])."""
    L = _lib.load()
    buf = np.frombuffer(data, np.uint8)
    seq_s = np.ascontiguousarray(seq_s, np.uint64)
    seq_e = np.ascontiguousarray(seq_e, np.uint64)
    length = np.ascontiguousarray(length, np.uint64)
    n = seq_s.size
    cap = int(L.mic_pack_bound(seq_s.ctypes.data, seq_e.ctypes.data, n, k))
    rp = np.zeros(n + 1, np.uint32)
    cont = np.zeros(cap, np.uint16)
    m = L.mic_pack_reads(buf.ctypes.data, seq_s.ctypes.data, seq_e.ctypes.data, length.ctypes.data, n, k, rp.ctypes.data,
                         cont.ctypes.data, cap)
    if m == C.c_size_t(-1).value:
        raise RuntimeError("mic_pack_reads: bound too small")
    return rp, cont[:m].copy()


def format_csv(data, idx, results, target_names, k, paired=False, extended=False, rows=None, dense=None):
    """CuCLARK_hh.hh:1951-2139: header + one line per read.  dense: optional dict read index -> u32[T] counts."""
    L = _lib.load()
    buf = np.frombuffer(data, np.uint8)
    names = (C.c_char_p * len(target_names))(*[t.encode() for t in target_names])
    T = len(target_names)
    cap = 512 + (T * 48 if extended else 0) + sum(len(t) + 1 for t in target_names)
    line = C.create_string_buffer(cap)
    out = []
    n = L.mic_csv_header(line, cap, int(extended), names, T)
    out.append(line.raw[:n])
    results = np.ascontiguousarray(results, np.uint32)
    for r in range(results.shape[0]):
        row_p = rows[r].ctypes.data if (extended and rows is not None) else None
        dn = None
        if extended and dense is not None and r in dense:
            dn = np.ascontiguousarray(dense[r], np.uint32)
        ns, ne = int(idx["name_s"][r]), int(idx["name_e"][r])
        n = L.mic_csv_line(line, cap, buf.ctypes.data + ns, ne - ns, int(idx["length"][r]), int(paired), k,
                           results[r].ctypes.data, names, T, int(extended), row_p, dn.ctypes.data if dn is not None else None)
        assert n >= 0
        out.append(line.raw[:n])
    return b"".join(out)


def build_db(target_files, target_labels, k, htsize, out_prefix, key_bytes=0, min_count=0, device=-1, threads=4, parts=0,
             light_gap=0):
    """GPU database builder (mic_db_build): target_labels[i] is the label index of target_files[i].  Returns #k-mers.
    light_gap > 0 builds cuCLARK-l's light database (non-overlapping k-blocks, every light_gap-th one)."""
    L = _lib.load()
    files = (C.c_char_p * len(target_files))(*[f.encode() for f in target_files])
    labels = np.ascontiguousarray(target_labels, np.uint16)
    n = C.c_uint64(0)
    rc = L.mic_db_build(files, labels.ctypes.data, len(target_files), int(k), int(htsize), int(key_bytes), int(min_count),
                        int(light_gap), out_prefix.encode(), int(device), int(threads), int(parts), C.byref(n))
    if rc != 0:
        raise RuntimeError(f"mic_db_build failed ({rc}): {L.mic_db_build_error().decode(errors='replace')}")
    return int(n.value)

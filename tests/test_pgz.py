"""The parallel gzip inflater (cuclark_amd/csrc/pgz.hpp: speculative block starts, marker symbols, window hand-over) against
zlib on everything a gzip file may hold: every compression level, stored and fixed-Huffman blocks, several members, header
fields, long matches, tiny and empty inputs; damaged files must end with an error as with zlib.  Built with
AddressSanitizer + UBSan: a false block start must never read or write out of bounds.  CPU only."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import golden_util as gu


@pytest.fixture(scope="module")
def cli(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("pgz"))
    exe = os.path.join(d, "pgz_cli")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        f"-I{os.path.join(gu.ROOT, 'cuclark_amd', 'csrc')}", "-o", exe, os.path.join(gu.ROOT, "tools", "pgz_cli.cpp"),
                        "-lz", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return d, exe


def _fastq(rng, n, L=150):
    nt = np.frombuffer(b"ACGTN", np.uint8)
    out = bytearray()
    for i in range(n):
        s = nt[rng.choice(5, L, p=[0.248, 0.248, 0.248, 0.248, 0.008])].tobytes()
        q = bytes(rng.integers(35, 74, L, dtype=np.uint8))
        out += b"@read%08d/1 lane=3\n" % i + s + b"\n+\n" + q + b"\n"
    return bytes(out)


def _member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flags=0):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    body = c.compress(data) + c.flush()
    hdr = b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\0\xff"
    if flags & 4:
        hdr += struct.pack("<H", 6) + b"XY\x02\0ab"
    if flags & 8:
        hdr += b"name.fq\0"
    if flags & 16:
        hdr += b"a comment\0"
    if flags & 2:
        hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    return hdr + body + struct.pack("<II", zlib.crc32(data), len(data) & 0xFFFFFFFF)


def _run(cli, gz, threads, chunk):
    d, exe = cli
    src, dst = os.path.join(d, "in.gz"), os.path.join(d, "out.bin")
    open(src, "wb").write(gz)
    if os.path.exists(dst):
        os.remove(dst)
    r = subprocess.run([exe, src, dst, str(threads), str(chunk)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    return r.returncode, (open(dst, "rb").read() if os.path.exists(dst) else b"")


def test_inflates_like_zlib(cli):
    rng = np.random.default_rng(5)
    fq = _fastq(rng, 60000)
    cases = {f"fastq_level{l}": (gzip.compress(fq, l), fq) for l in (1, 6, 9)}
    cases["gzip_tool_level1"] = (subprocess.run(["gzip", "-1", "-c"], input=fq, capture_output=True, check=True).stdout, fq)
    rnd = os.urandom(3_000_000)
    cases["stored_blocks"] = (gzip.compress(rnd, 6), rnd)
    zeros = bytes(20_000_000)
    cases["long_matches"] = (gzip.compress(zeros + b"x" + zeros[:1000], 9), zeros + b"x" + zeros[:1000])
    cases["fixed_huffman"] = (_member(fq[:200000], 6, zlib.Z_FIXED), fq[:200000])
    cases["huffman_only"] = (_member(fq[:3_000_000], 6, zlib.Z_HUFFMAN_ONLY), fq[:3_000_000])
    cases["rle"] = (_member(fq[:3_000_000], 6, zlib.Z_RLE), fq[:3_000_000])
    cases["header_fields"] = (_member(fq[:500000], 1, flags=4 | 8 | 16 | 2), fq[:500000])
    parts = [fq[:1_000_000], b"", fq[1_000_000:1_000_010], fq[1_000_010:9_000_000], rnd[:70000], fq[9_000_000:]]
    cases["six_members"] = (b"".join(gzip.compress(p, 1 + i % 9) for i, p in enumerate(parts)), b"".join(parts))
    cases["zero_padding_after_member"] = (gzip.compress(fq[:100000]) + bytes(512), fq[:100000])
    cases["tiny"] = (gzip.compress(b"ACGT\n"), b"ACGT\n")
    cases["empty"] = (gzip.compress(b""), b"")
    for name, (gz, want) in cases.items():
        assert gzip.decompress(gz) == want
        for threads, chunk in ((1, 1 << 20), (3, 1 << 16), (8, 1 << 18), (5, 4096)):
            rc, got = _run(cli, gz, threads, chunk)
            assert rc == 0 and got == want, (name, threads, chunk, rc, len(got), len(want))


def test_damaged_files_end_with_an_error(cli):
    rng = np.random.default_rng(6)
    fq = _fastq(rng, 40000)
    gz = gzip.compress(fq, 1)
    bad = {"truncated": gz[: len(gz) // 2], "truncated_in_trailer": gz[:-3], "not_gzip": fq[:1000],
           "trailing_garbage": gz + b"garbage here",
           "wrong_crc": gz[:-8] + struct.pack("<I", 12345) + gz[-4:], "wrong_length": gz[:-4] + struct.pack("<I", 7)}
    for pos in (len(gz) // 3, len(gz) // 2, len(gz) - 5000):           # a flipped bit in the middle of the stream
        b = bytearray(gz)
        b[pos] ^= 0x10
        bad[f"bit_flip_at_{pos}"] = bytes(b)
    for name, data in bad.items():
        for threads, chunk in ((1, 1 << 20), (4, 1 << 16), (8, 1 << 18)):
            rc, got = _run(cli, data, threads, chunk)
            assert rc == 1, (name, threads, chunk, rc)

"""gzip members inflated on the device (mic_gz_*, csrc/mic_gz.hip) against zlib: every kind of block, sizes from empty to tens of
megabytes, every compression level; what the path does not take (several members, damage) must come back as "unsupported" or as an
error - never as wrong text."""
import gzip
import io
import zlib

import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


def _fastq(rng, n, L=150):
    nt = np.frombuffer(b"ACGT", np.uint8)
    out = io.BytesIO()
    for i in range(n):
        s = nt[rng.integers(0, 4, L)].tobytes()
        q = bytes(rng.choice(np.frombuffer(b"FFFFFFFF:,#", np.uint8), L))
        out.write(b"@read_%09d/1 some description\n" % i + s + b"\n+\n" + q + b"\n")
    return out.getvalue()


@pytest.fixture(scope="module")
def engine():
    from cuclark_amd import MiClarkDB
    with MiClarkDB(31, 4) as e:
        yield e


def _gz(data, level=6, **kw):
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", compresslevel=level, mtime=0, **kw) as f:
        f.write(data)
    return buf.getvalue()


@pytest.mark.parametrize("level", [1, 6, 9])
def test_fastq_text_of_many_blocks(engine, level):
    rng = np.random.default_rng(level)
    data = _fastq(rng, 60000)                       # ~20 MB of text, hundreds of deflate blocks
    text, crc = engine.gunzip(_gz(data, level))
    assert text == data
    assert crc == zlib.crc32(data)


def _with_extra_field(gz, n_extra):
    """The same member with an FEXTRA field of n_extra bytes in its header: the deflate data starts that much later."""
    import struct
    assert gz[3] & 4 == 0
    sub = b"XX" + struct.pack("<H", n_extra - 4) + bytes(n_extra - 4)
    return gz[:3] + bytes([gz[3] | 4]) + gz[4:10] + struct.pack("<H", n_extra) + sub + gz[10:]


def test_a_header_longer_than_two_finder_chunks(engine):
    """20 000 bytes of FEXTRA: the first block starts in the finder's third 8-KiB chunk (the chunks in front of it hold no deflate
    data at all), and zlib reads the same member."""
    rng = np.random.default_rng(21)
    data = _fastq(rng, 20000)
    gz = _with_extra_field(_gz(data, 1), 20000)
    assert gzip.decompress(gz) == data
    text, crc = engine.gunzip(gz)
    assert text == data and crc == zlib.crc32(data)


def test_texts_that_leave_the_window_decode(engine):
    """What the decode's 64-offsets-at-a-time path hands to the serial one: codes longer than its tables (a text of many rare
    byte values: 11- to 15-bit codes), matches longer than 64 symbols and runs (distance 1 ... 5), literals only (level 0 is
    stored; a text without repeats at level 1 is literals), and the same FASTQ with every one of them in between."""
    rng = np.random.default_rng(22)
    skew = np.concatenate([np.full(40000, 65, np.uint8), rng.integers(0, 256, 3000).astype(np.uint8), np.full(40000, 67, np.uint8)])
    rng.shuffle(skew)
    rare = np.tile(skew, 12).tobytes()                                               # two common bytes, 254 rare ones: long codes
    runs = b"".join(bytes([65 + i % 4]) * int(n) for i, n in enumerate(rng.integers(1, 700, 4000)))
    period = b"".join((b"ACGTT"[: 1 + i % 5]) * int(n) for i, n in enumerate(rng.integers(20, 200, 3000)))
    noise = rng.integers(0, 256, 400000, dtype=np.uint8).tobytes()
    fq = _fastq(rng, 4000)
    for level in (1, 6, 9):
        for data in (rare, runs, period, fq + rare[:200000] + runs[:200000] + fq + noise[:50000] + period[:100000] + fq):
            text, crc = engine.gunzip(_gz(data, level))
            assert text == data and crc == zlib.crc32(data)


@pytest.mark.parametrize("kind", ["empty", "one byte", "tiny (fixed codes)", "random (stored blocks)", "zeros", "period 3", "text 1 MB",
                                  "binary mix", "long header"])
def test_block_kinds_and_sizes(engine, kind):
    rng = np.random.default_rng(7)
    data = {"empty": b"", "one byte": b"x", "tiny (fixed codes)": b"hello, hello, hello world\n",
            "random (stored blocks)": rng.integers(0, 256, 300000, dtype=np.uint8).tobytes(), "zeros": bytes(5_000_000),
            "period 3": b"ACG" * 700000, "text 1 MB": _fastq(rng, 3000),
            "binary mix": rng.integers(0, 256, 100000, dtype=np.uint8).tobytes() + _fastq(rng, 2000) + bytes(200000) +
                          rng.integers(0, 4, 300000, dtype=np.uint8).tobytes(),
            "long header": _fastq(rng, 500)}[kind]
    gz = _gz(data, 6, filename="x" * 300 + ".fq") if kind == "long header" else _gz(data, 6)
    text, crc = engine.gunzip(gz)
    assert text == data and crc == zlib.crc32(data)


def test_raw_deflate_with_fixed_and_stored_blocks_between_dynamic_ones(engine):
    """a stream put together block by block: Z_FULL_FLUSH / Z_SYNC_FLUSH leave empty stored blocks, Z_FIXED forces fixed codes"""
    rng = np.random.default_rng(11)
    parts = []
    body = b""
    crc = 0
    for i, (strategy, flush) in enumerate([(zlib.Z_DEFAULT_STRATEGY, zlib.Z_SYNC_FLUSH), (zlib.Z_FIXED, zlib.Z_FULL_FLUSH),
                                           (zlib.Z_DEFAULT_STRATEGY, zlib.Z_NO_FLUSH), (zlib.Z_HUFFMAN_ONLY, zlib.Z_SYNC_FLUSH)] * 3):
        parts.append(_fastq(rng, 2500))
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    for i, p in enumerate(parts):
        body += c.compress(p) + c.flush(zlib.Z_FULL_FLUSH if i % 2 else zlib.Z_SYNC_FLUSH)
    body += c.flush()
    data = b"".join(parts)
    gz = b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + body + zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little")
    assert zlib.decompress(gz, 31) == data
    text, crc = engine.gunzip(gz)
    assert text == data


def test_what_the_path_does_not_take(engine):
    from cuclark_amd.db import MiClarkUnsupported
    from cuclark_amd import MicError
    rng = np.random.default_rng(5)
    data = _fastq(rng, 20000)
    gz = _gz(data, 6)
    with pytest.raises(MiClarkUnsupported):                       # two members
        engine.gunzip(gz + gz)
    with pytest.raises(MiClarkUnsupported):                       # not gzip at all
        engine.gunzip(data[:100000])
    import struct
    blk = data[:0xFF00]                                            # one member with a 'BC' subfield, and the same bytes with another subfield
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    body = c.compress(blk) + c.flush()
    bgzf = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk), len(blk))
    assert zlib.decompress(bgzf, 31) == blk
    other_extra = bgzf[:12] + b"XY" + bgzf[14:]                    # (another subfield in the same place is skipped like any header field)
    assert engine.gunzip(other_extra)[0] == blk
    bad = bytearray(gz)
    bad[len(bad) // 2] ^= 0x55                                     # a damaged block: unsupported or an error, never wrong text passing as right
    try:
        text, crc = engine.gunzip(bytes(bad))
        assert zlib.crc32(text) != crc or text == data
    except (MiClarkUnsupported, MicError):
        pass
    bad = bytearray(gz)
    bad[-6] ^= 1                                                   # CRC-32 wrong: what gunzip calls a crc error
    with pytest.raises(MicError):
        engine.gunzip(bytes(bad))
    stored = zlib.compressobj(0, zlib.DEFLATED, 31)                # a byte changed inside a stored block: everything stitches, the CRC does not
    raw = bytearray(stored.compress(data[:200000]) + stored.flush())
    assert engine.gunzip(bytes(raw))[0] == data[:200000]
    raw[100000] ^= 0x20
    with pytest.raises(MicError):
        engine.gunzip(bytes(raw))
    bad = bytearray(gz)
    bad[-2] ^= 1                                                   # ISIZE wrong
    with pytest.raises((MiClarkUnsupported, MicError)):
        engine.gunzip(bytes(bad))


def test_buffers_reserved_ahead_of_the_call(engine):
    """mic_gz_reserve: the call for a file of the reserved size runs out of the reservation (text buffer included) and gives the same
    bytes; a file of another size, a second call for the same size and a file whose blocks are smaller than the reservation assumed
    (more windows than it holds) allocate for themselves."""
    import struct
    rng = np.random.default_rng(77)
    data = _fastq(rng, 40000)
    small_blocks = zlib.compressobj(1, zlib.DEFLATED, 31, 1)            # memLevel 1: a block every 127 symbols' worth of codes
    gz_small = small_blocks.compress(data[:3_000_000]) + small_blocks.flush()
    for gz, want in ((_gz(data, 1), data), (gz_small, data[:3_000_000])):
        isize = struct.unpack("<I", gz[-4:])[0]
        assert engine.L.mic_gz_reserve_bytes(len(gz), isize) > len(want)
        assert engine.L.mic_gz_reserve(engine.h, len(gz), isize) == 0
        for _ in range(2):
            text, crc = engine.gunzip(gz)
            assert text == want and crc == zlib.crc32(want)
        other = _gz(data[:-316], 1)
        text, _ = engine.gunzip(other)
        assert text == data[:-316]
        assert engine.L.mic_gz_release(engine.h) == 0
        text, _ = engine.gunzip(gz)
        assert text == want


def _bgzf(data, block=0xFF00, level=1, eof=True):
    import struct
    out = []
    for o in range(0, len(data), block):
        blk = data[o:o + block]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(blk) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
    if eof:
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
    return b"".join(out)


def test_block_gzip_members_side_by_side(engine):
    """BGZF (bgzip / samtools): every member is a unit of its own - sizes from its header, place in the text from the trailers in front
    of it, CRC-32 and length checked per member.  Against zlib; damage inside a member, in a trailer, a cut file: an error or
    "unsupported", never wrong text."""
    from cuclark_amd.db import MiClarkUnsupported
    from cuclark_amd import MicError
    rng = np.random.default_rng(21)
    data = _fastq(rng, 30000)                                       # ~10 MB: 150 members
    for block, level, eof in ((0xFF00, 1, True), (0xFF00, 6, False), (65536, 9, True), (1000, 1, True), (4096, 0, True)):
        piece = data if block > 5000 else data[:300000]
        gz = _bgzf(piece, block, level, eof)
        assert zlib.decompressobj(31).decompress(gz)[:100] == piece[:100]
        text, _ = engine.gunzip(gz)
        assert text == piece, (block, level, eof)
    assert engine.gunzip(_bgzf(b""))[0] == b""
    whole = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes() + data[:500000] + bytes(300000)      # stored, dynamic and run members
    assert engine.gunzip(_bgzf(whole, 0xFF00, 6))[0] == whole
    gz = bytearray(_bgzf(data[:1_000_000]))
    bad = bytearray(gz); bad[len(bad) // 2] ^= 0x10                 # somewhere inside a member's deflate data
    try:
        text, _ = engine.gunzip(bytes(bad))
        assert text == data[:1_000_000]                             # (a flipped bit that changes nothing is not possible, but say so)
    except (MiClarkUnsupported, MicError):
        pass
    first = 18 + (gz[16] | (gz[17] << 8)) + 1 - 18                  # size of the first member
    bad = bytearray(gz); bad[first - 8] ^= 1                        # its CRC-32
    with pytest.raises(MicError):
        engine.gunzip(bytes(bad))
    bad = bytearray(gz); bad[first - 4] ^= 1                        # its ISIZE
    with pytest.raises((MiClarkUnsupported, MicError)):
        engine.gunzip(bytes(bad))
    with pytest.raises((MiClarkUnsupported, MicError)):             # cut inside a member
        engine.gunzip(bytes(gz[:len(gz) // 2]))
    with pytest.raises((MiClarkUnsupported, MicError)):             # an ordinary member behind block-gzip ones
        engine.gunzip(bytes(gz) + _gz(data[:1000]))


def _random_stream(rng):
    """One deflate stream of a random make (tools/gz_soak.py draws from it too): (text, gzip member, what it was)."""
    kind = int(rng.integers(6))
    n = int(rng.choice([0, 1, 100, 5000, 70000, 400000, 1500000]))
    if kind == 0:
        data = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].tobytes()
    elif kind == 1:
        data = _fastq(rng, n // 316 + 1)[:n]
    elif kind == 2:
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 3:
        data = (b"the quick brown fox jumps over the lazy dog\n" * (n // 44 + 1))[:n]
    elif kind == 4:
        data = bytes(n)
    else:
        parts, left = [], n
        while left > 0:
            m = int(min(left, rng.integers(1, 60000)))
            parts.append(rng.integers(0, int(rng.choice([2, 4, 20, 256])), m, dtype=np.uint8).tobytes() if rng.random() < 0.7 else bytes([int(rng.integers(256))]) * m)
            left -= m
        data = b"".join(parts)
    level = int(rng.integers(0, 10))
    strategy = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    mem = int(rng.integers(1, 10))
    wbits = int(rng.integers(9, 16))
    c = zlib.compressobj(level, zlib.DEFLATED, 16 + wbits, mem, strategy)
    gz = b""
    cuts = sorted(int(x) for x in rng.integers(0, len(data) + 1, int(rng.integers(0, 3))))
    prev = 0
    for cut in cuts:
        gz += c.compress(data[prev:cut]) + c.flush(int(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH])))
        prev = cut
    gz += c.compress(data[prev:]) + c.flush()
    assert zlib.decompress(gz, 31) == data
    return data, gz, (kind, n, level, strategy, mem, wbits)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_streams_against_zlib(engine, seed):
    """Deflate streams of every make: levels 0-9, the strategies (default, filtered, Huffman only, RLE, fixed codes), memLevel 1-9
    (blocks from 127 codes up), window sizes 512 B - 32 KiB, flushes in the middle, data from DNA over text to noise - as one
    gzip member and as block gzip; what comes back is zlib's text or "unsupported", never anything else."""
    from cuclark_amd.db import MiClarkUnsupported
    rng = np.random.default_rng(1000 + seed)
    took = 0
    for trial in range(40):
        data, gz, what = _random_stream(rng)
        n, level = what[1], what[2]
        try:
            text, crc = engine.gunzip(gz)
            assert text == data and crc == zlib.crc32(data), (trial,) + what
            took += 1
        except MiClarkUnsupported:
            pass
        if n and trial % 3 == 0:
            block = int(rng.choice([300, 4096, 0xFF00, 65536]))
            assert engine.gunzip(_bgzf(data, block, level if level else 1))[0] == data, (trial, "bgzf", block)
    assert took >= 30, took                      # (unsupported is for what the finder cannot stitch: the rare exception, not the rule)


@pytest.mark.parametrize("stripes", [2, 3, 5])
def test_a_member_in_stripes(engine, stripes):
    """mic_gz_stream_*: the units of the member decoded, stitched and resolved a stripe at a time, the window in front of a stripe
    the end of the text so far.  After every stripe the text up to n_final is final (compared right then), and its whole FASTQ
    records are what mic_text_index_front_device finds from where the last front ended."""
    rng = np.random.default_rng(100 + stripes)
    data = _fastq(rng, 220000)                      # ~73 MB of text: ~280 deflate blocks at level 1, stripes of >= 64 units
    gz = _gz(data, 1)
    seen = {"carry": 0, "records": 0}

    def on_stripe(d_text, n_final, done):
        part = np.empty(n_final - seen["carry"], np.uint8)
        gu_lib = engine.L
        assert gu_lib.mic_gz_copy_text(engine.h, d_text, seen["carry"], part.size, part.ctypes.data) == 0
        assert part.tobytes() == data[seen["carry"]:n_final]
        if not done:
            h, n_rec, used, st = engine.text_index_front(d_text + seen["carry"], n_final - seen["carry"])
            chunk = data[seen["carry"]:n_final]
            whole_lines = chunk.count(b"\n") // 4 * 4
            end = 0
            for _ in range(whole_lines):
                end = chunk.index(b"\n", end) + 1
            assert st == 0 and n_rec == whole_lines // 4 and used == end, (n_rec, used, whole_lines, end)
            if h:
                gu_lib.mic_text_free(engine.h, h)
            seen["records"] += n_rec
            seen["carry"] += used

    text, finals = engine.gunzip_stripes(gz, stripes, on_stripe)
    assert text == data
    assert len(finals) >= 2 and finals == sorted(finals) and finals[-1] == len(data)
    assert seen["records"] > 0 and data[seen["carry"]:].count(b"\n") % 4 == 0


def test_stripes_meet_what_the_whole_inflate_refuses(engine):
    """a second member behind the first, a damaged block in a later stripe, a wrong CRC: the stripes in front of it go through (their
    text is right), then the call says so - the caller starts over on its CPU inflater (tests/test_cli.py)"""
    from cuclark_amd.db import MiClarkUnsupported
    from cuclark_amd._lib import MicError as MiClarkError
    rng = np.random.default_rng(7)
    data = _fastq(rng, 200000)
    gz = _gz(data, 1)
    got = []
    with pytest.raises(MiClarkUnsupported):
        engine.gunzip_stripes(gz + _gz(b"@x\nACGT\n+\nFFFF\n", 1), 3, lambda d, n, done: got.append(n))
    bad = bytearray(gz)
    bad[-8] ^= 1                                                   # the trailer's CRC-32
    got2 = []
    with pytest.raises(MiClarkError):
        engine.gunzip_stripes(bytes(bad), 3, lambda d, n, done: got2.append(n))
    assert len(got2) >= 1                                          # (the stripes in front of the last went through)
    bad = bytearray(gz)
    for o in range(len(gz) * 3 // 4, len(gz) * 3 // 4 + 64):       # a later stripe's data
        bad[o] ^= 0x5A
    with pytest.raises((MiClarkUnsupported, MiClarkError)):
        engine.gunzip_stripes(bytes(bad), 3)
    assert engine.gunzip(gz)[0] == data                            # (and the engine is fine afterwards)


def test_a_text_of_four_gib_and_more_goes_back_to_the_caller(engine):
    """ISIZE is the length mod 2^32: a member of 2^32 + 1000 bytes says 1000.  Until round 6 the text buffer was sized by that field
    and the stitched length compared with it as 32-bit numbers - equal - so the resolve pass would have written 4 GiB into a buffer of
    1 KiB.  Now more text than the trailer's length is MIC_E_UNSUPPORTED the moment the chain gets there, and the caller's CPU
    inflater (which streams) takes the file."""
    import struct
    from cuclark_amd.db import MiClarkUnsupported
    c = zlib.compressobj(1, zlib.DEFLATED, -15)
    zeros = bytes(1 << 26)
    body = [c.compress(zeros) for _ in range(64)] + [c.compress(bytes(1000)), c.flush()]
    n = (1 << 32) + 1000
    crc = 0
    for _ in range(64):
        crc = zlib.crc32(zeros, crc)
    crc = zlib.crc32(bytes(1000), crc)
    gz = b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + b"".join(body) + struct.pack("<II", crc, n & 0xFFFFFFFF)
    assert zlib.decompressobj(31).decompress(gz, 100) == bytes(100)            # (a valid member)
    with pytest.raises(MiClarkUnsupported):
        engine.gunzip(gz)
    with pytest.raises(MiClarkUnsupported):
        engine.gunzip_stripes(gz, 2)
    rng = np.random.default_rng(5)
    data = _fastq(rng, 2000)
    assert engine.gunzip(_gz(data, 6))[0] == data

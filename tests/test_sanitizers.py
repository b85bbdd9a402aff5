"""AddressSanitizer + UndefinedBehaviorSanitizer (and ThreadSanitizer) runs of the HOST code of the command line: classifier.cpp
(loaders, strip_fastq, the paired-end mergers, the gzip / BGZF inflate streams, the three thread pools of run_stream and their
queues), cli_main.cpp and mic_host.cpp (indexer, packer, CSV), linked against a mock of the device entry points
(tools/sanitize/mock_engine.cpp: every slot "classified" on the CPU into one "<name>,<length>" line per record).  The checks:
no sanitizer report, and every record of the input arrives exactly once and in file order.  CPU only - GPU sanitizers are not
available on this pool."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import golden_util as gu

CSRC = os.path.join(gu.ROOT, "cuclark_amd", "csrc")
SRCS = [os.path.join(CSRC, f) for f in ("classifier.cpp", "classifier_stream.cpp", "classifier_batch.cpp", "cli_main.cpp", "mic_host.cpp")] + [os.path.join(gu.ROOT, "tools", "sanitize", "mock_engine.cpp")]


def _build(tmp, flavour):
    exe = os.path.join(tmp, f"cuCLARK_{flavour}")
    san = {"asan": "-fsanitize=address,undefined", "tsan": "-fsanitize=thread"}[flavour]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", *san.split(), "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
                        f"-I{os.path.join(gu.ROOT, 'include')}", f"-I{CSRC}", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                        "-o", exe, *SRCS, "-lz", "-lpthread"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def rig(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("san"))
    rng = np.random.default_rng(9)
    nt = np.frombuffer(b"ACGTN", np.uint8)

    def seq(n):
        return nt[rng.choice(5, n, p=[0.245, 0.245, 0.245, 0.245, 0.02])].tobytes().decode()
    recs = [(f"read{i}_{'x' * int(rng.integers(0, 50))}", seq(int(rng.integers(1, 400)))) for i in range(30000)]
    fq = "".join(f"@{n} extra words\n{s}\n+\n{'I' * len(s)}\n" for n, s in recs).encode()
    fa = "".join(f">{n}\tdesc\n" + "\n".join(s[i:i + 70] for i in range(0, len(s), 70)) + "\n" for n, s in recs).encode()
    mates = [seq(len(s)) for _, s in recs]
    fq1 = "".join(f"@{n}/1\n{s}\n+\n{'F' * len(s)}\n" for n, s in recs).encode()
    fq2 = "".join(f"@{n}/2\n{s}\n+\n{'F' * len(s)}\n" for (n, _), s in zip(recs, mates)).encode()
    files = {}
    for name, data in (("r.fq", fq), ("r.fa", fa), ("p_1.fq", fq1), ("p_2.fq", fq2)):
        files[name] = os.path.join(tmp, name)
        open(files[name], "wb").write(data)
    with gzip.open(os.path.join(tmp, "r.fq.gz"), "wb", compresslevel=1) as f:          # several members, like `cat a.gz b.gz`
        f.write(fq[:len(fq) // 2])
    with gzip.open(os.path.join(tmp, "r.fq.gz"), "ab", compresslevel=6) as f:
        f.write(fq[len(fq) // 2:])
    with open(os.path.join(tmp, "r.bgzf.fq.gz"), "wb") as f:                            # block gzip (BGZF)
        for o in range(0, len(fq), 0xFF00):
            blk = fq[o:o + 0xFF00]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(blk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
    for i, (a, b) in enumerate(((fq1, "p_1.fq.gz"), (fq2, "p_2.fq.gz"), (fq, "r1.fq.gz"), (fa, "r.fa.gz"))):
        with gzip.open(os.path.join(tmp, b), "wb", compresslevel=1) as f:
            f.write(a)

    def bgzf(path, data):
        with open(path, "wb") as f:
            for o in range(0, len(data), 0xFF00):
                blk = data[o:o + 0xFF00]
                c = zlib.compressobj(1, zlib.DEFLATED, -15)
                body = c.compress(blk) + c.flush()
                f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
    bgzf(os.path.join(tmp, "p_1.bgzf.fq.gz"), fq1)
    bgzf(os.path.join(tmp, "p_2.bgzf.fq.gz"), fq2)
    # a database that only has to exist (the mock engine loads nothing)
    db = os.path.join(tmp, "DB")
    os.makedirs(db)
    genome = os.path.join(tmp, "g.fa")
    open(genome, "w").write(">g\nACGT\n")
    open(os.path.join(tmp, "targets.txt"), "w").write(f"{genome} T0\n{genome} T1\n")
    base = os.path.join(db, "db_central_k31_t2_s64_m0.tsk")
    open(base + ".sz", "wb").write(bytes(64))
    open(base + ".ky", "wb").write(b"")
    open(base + ".lb", "wb").write(b"")
    single = "".join(f"{n[:39]},{len(s)}\n" for n, s in recs)
    pairs = "".join(f"{n[:39]},{len(s) + len(m)}\n" for (n, s), m in zip(recs, mates))
    return dict(tmp=tmp, single=single, pairs=pairs)


def _run(exe, rig, objects, env_extra=(), threads="4", extra_args=(), tag=""):
    tmp = rig["tmp"]
    out = os.path.join(tmp, "out" + tag)
    if os.path.exists(out + ".csv"):
        os.remove(out + ".csv")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1",
               MIC_CLI_ORDERLY_EXIT="1", **dict(env_extra))      # (orderly teardown: the destructors run under the sanitizer too)
    cmd = [exe, "-k", "31", "--htsize", "64", "-T", os.path.join(tmp, "targets.txt"), "-D", os.path.join(tmp, "DB"),
           *objects, "-R", out, "-n", threads, *extra_args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    lines = open(out + ".csv").read().split("\n", 1)
    return lines[1] if len(lines) > 1 else ""


CASES = [
    ("fastq", ["-O", "r.fq"], "single", {"MIC_INGEST_MB": "4"}),
    ("fastq_tiny_slots", ["-O", "r.fq"], "single", {"MIC_INGEST_MB": "2"}),
    ("fasta_multiline", ["-O", "r.fa"], "single", {"MIC_INGEST_MB": "2"}),
    ("gzip_two_members", ["-O", "r.fq.gz"], "single", {"MIC_INGEST_MB": "2"}),
    ("bgzf", ["-O", "r.bgzf.fq.gz"], "single", {"MIC_INGEST_MB": "2"}),
    ("pairs_parallel_merge", ["-P", "p_1.fq", "p_2.fq"], "pairs", {"MIC_INGEST_MB": "2"}),
    ("pairs_serial_reader", ["-P", "p_1.fq", "p_2.fq"], "pairs", {"MIC_SERIAL_PAIRS": "1", "MIC_INGEST_MB": "2"}),
    ("pairs_gzip", ["-P", "p_1.fq.gz", "p_2.fq.gz"], "pairs", {"MIC_INGEST_MB": "2"}),
    # compressed input "on the device" (the mock's CPU stand-ins for mic_gz_* / mic_pairs_* / mic_text_*: the command line's
    # DeviceGzFeeder - batches cut at the sampled offsets, slots filled in place - is what runs under the sanitizer) and the same
    # files through the host inflater
    ("pairs_gzip_small_slots", ["-P", "p_1.fq.gz", "p_2.fq.gz"], "pairs", {"MIC_INGEST_KB": "96"}),
    ("pairs_gzip_host", ["-P", "p_1.fq.gz", "p_2.fq.gz"], "pairs", {"MIC_INGEST_MB": "2", "MIC_GZ_HOST": "1"}),
    ("pairs_bgzf", ["-P", "p_1.bgzf.fq.gz", "p_2.bgzf.fq.gz"], "pairs", {"MIC_INGEST_MB": "2"}),
    ("pairs_bgzf_host", ["-P", "p_1.bgzf.fq.gz", "p_2.bgzf.fq.gz"], "pairs", {"MIC_INGEST_MB": "2", "MIC_GZ_HOST": "1"}),
    ("gzip_one_member", ["-O", "r1.fq.gz"], "single", {"MIC_INGEST_KB": "96"}),
    ("gzip_one_member_host", ["-O", "r1.fq.gz"], "single", {"MIC_INGEST_MB": "2", "MIC_GZ_HOST": "1"}),
    ("fasta_gzip", ["-O", "r.fa.gz"], "single", {"MIC_INGEST_MB": "2"}),
    # the member in stripes (MIC_GZ_STRIPES): the feeder's inflater thread hands records out as they become final; FASTA is not cut in
    # stripes (all of it, then one text); two members: the first stripe goes out, the next gives the file back, the run starts over
    ("gzip_stripes", ["-O", "r1.fq.gz"], "single", {"MIC_INGEST_KB": "96", "MIC_GZ_STRIPES": "5"}),
    ("gzip_stripes_fasta", ["-O", "r.fa.gz"], "single", {"MIC_INGEST_KB": "96", "MIC_GZ_STRIPES": "3"}),
    ("gzip_stripes_given_back", ["-O", "r.fq.gz"], "single", {"MIC_INGEST_KB": "96", "MIC_GZ_STRIPES": "3"}),
    ("bgzf_host", ["-O", "r.bgzf.fq.gz"], "single", {"MIC_INGEST_MB": "2", "MIC_GZ_HOST": "1"}),
    # several engines (MIC_SHARD_ENGINES: two / three mock engines): the slots - and the batches - are dealt over the engines, the
    # "device" text of a compressed input fills slots of every engine; read-sharded (default) and table-sharded (--db-sharded:
    # groups of engines answer a slot together, mic_ingest_classify_group)
    ("fastq_two_engines", ["-O", "r.fq"], "single", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "2"}),
    ("fasta_three_engines", ["-O", "r.fa"], "single", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "3"}),
    ("pairs_two_engines", ["-P", "p_1.fq", "p_2.fq"], "pairs", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "2"}),
    ("pairs_gzip_two_engines", ["-P", "p_1.fq.gz", "p_2.fq.gz"], "pairs", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "2"}),
    ("gzip_three_engines", ["-O", "r1.fq.gz"], "single", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "3"}),
    ("bgzf_two_engines", ["-O", "r.bgzf.fq.gz"], "single", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "2"}),
    ("fastq_table_sharded_2x2", ["-O", "r.fq"], "single", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "4"}, ["--db-sharded", "--parts", "2"]),
    ("pairs_gzip_table_sharded_3", ["-P", "p_1.fq.gz", "p_2.fq.gz"], "pairs", {"MIC_INGEST_KB": "96", "MIC_SHARD_ENGINES": "3"}, ["--parts", "3"]),
]


@pytest.mark.parametrize("flavour", ["asan", "tsan"])
def test_host_pipeline_under_sanitizers(flavour, rig):
    exe = _build(rig["tmp"], flavour)
    jobs = []
    # TSan (5-15 x slower) runs the cases that differ in which threads exist and what they share; ASan + UBSan all of them
    tsan_cases = {"fastq", "fastq_tiny_slots", "gzip_two_members", "pairs_parallel_merge", "pairs_serial_reader", "pairs_gzip",
                  "pairs_gzip_small_slots", "pairs_gzip_host", "gzip_one_member", "fasta_gzip", "gzip_stripes", "gzip_stripes_given_back", "fastq_two_engines", "pairs_gzip_two_engines",
                  "gzip_three_engines", "fastq_table_sharded_2x2", "pairs_gzip_table_sharded_3"}
    assert tsan_cases <= {c[0] for c in CASES}
    for i, (name, objects, want, env, *more) in enumerate(CASES):
        if flavour == "tsan" and name not in tsan_cases:
            continue
        objects = [o if o.startswith("-") else os.path.join(rig["tmp"], o) for o in objects]
        for threads in (("1", "7") if flavour == "asan" and i < 8 else ("7",) if flavour == "asan" else ("5",)):
            jobs.append((name, objects, want, env, threads, more[0] if more else ()))
    # independent runs of the instrumented binary, three side by side (each its own process and output file)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=3) as ex:
        gots = list(ex.map(lambda j: _run(exe, rig, j[1], j[3].items(), j[4], j[5], tag=f"_{j[0]}_{j[4]}"), jobs))
    for (name, _, want, _, threads, _), got in zip(jobs, gots):
        assert got == rig[want], (flavour, name, threads, got[:200], rig[want][:200])
    # the merge verb (no engine at all), golden pair files
    f1, f2 = (os.path.join(gu.GOLDEN, f"pairs_k31_{i}.fq") for i in (1, 2))
    out = os.path.join(rig["tmp"], "m.fa")
    want = open(os.path.join(gu.GOLDEN, "pairs_k31_merged.fa"), "rb").read()
    for mode in ([], ["parallel", "6", "300"]):
        r = subprocess.run([exe, "--merge-pairs", f1, f2, out, *mode], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and open(out, "rb").read() == want, r.stderr[-2000:]


def test_the_librarys_host_code_under_asan_against_a_mock_hip_runtime(tmp_path):
    """mic_engine.hip, mic_build.hip, mic_ingest.hip, mic_gz.hip, mic_synth.hip, mic_dbbuild.hip and
    mic_host.cpp compiled HOST-ONLY (hipcc --offload-host-only) under ASan + UBSan, linked against tools/sanitize/hip_mock.cpp
    (device memory = host memory, kernels do not run, ASYNCHRONOUS COPIES DEFERRED to the next synchronisation so that a host
    buffer that dies with a copy queued is a use-after-free the sanitizer sees), driven by tools/sanitize/host_rig.cpp through the
    fuzzer's call sequences - engines, tables from arrays in all layouts (whole, bucket-range shards, slot-range parts), the
    batch API, the merge over shards, ingest slots and the group ingest, the device inflate's error paths.  No report, exit 0."""
    hipcc = "/opt/rocm/bin/hipcc"
    script = os.path.join(gu.ROOT, "tools", "sanitize", "build_host_rig.sh")
    if not os.path.exists(hipcc) or not os.path.exists(script):
        pytest.skip("no hipcc, or the build script did not travel (.gpurunignore: it is a CPU-only rig and stays off the GPU boxes)")
    out = str(tmp_path / "host_rig")
    r = subprocess.run(["bash", script, out], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and os.path.exists(os.path.join(out, "host_rig")), (r.stdout[-2000:], r.stderr[-3000:])
    env = dict(os.environ, ASAN_OPTIONS="detect_stack_use_after_return=1", UBSAN_OPTIONS="print_stacktrace=1")
    # ... and with the engines of a sharded configuration spread over four (mock) devices: peer copies, per-device streams
    r4 = subprocess.run([os.path.join(out, "host_rig"), "15", "6"], capture_output=True, text=True, timeout=600, env=dict(env, MOCK_HIP_DEVICES="4"))
    assert r4.returncode == 0 and "host rig ok" in r4.stdout and "4 device(s)" in r4.stdout, (r4.stdout[-1500:], r4.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r4.stderr and "runtime error" not in r4.stderr, r4.stderr[-4000:]
    r = subprocess.run([os.path.join(out, "host_rig"), "20", "5"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host rig ok" in r.stdout, (r.stdout[-1500:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    # ... and THE COMMAND LINE ITSELF on four mock devices (classifier.cpp + cli_main.cpp on the same objects): the table-sharded
    # modes (4 parts x 1 group, 2 parts x 2 groups: one read of the files for all devices, the peer matrix, the group ingest
    # across devices) and the read-sharded mode (3 and 2 engines, single and paired files) - every record once, no report.  The
    # GPU boxes of this pool have one device: this is where those paths run with every engine on a device of its own.
    import test_cli as tc
    import test_ingest as ti
    tmp = str(tmp_path)
    d, t = tc._db_dir(tmp, "light_k27_u32", light=True), tc._targets_file(tmp)
    rng = np.random.default_rng(43)
    genomes = ti._genomes()
    fq = os.path.join(tmp, "r.fq")
    open(fq, "wb").write(ti._random_reads(rng, genomes, 4000, fasta=False))
    m1, m2 = tc._pair_files(rng, genomes, 2000)
    p1, p2 = os.path.join(tmp, "m_1.fq"), os.path.join(tmp, "m_2.fq")
    open(p1, "wb").write(m1)
    open(p2, "wb").write(m2)
    cli_env = dict(env, ASAN_OPTIONS="detect_stack_use_after_return=1:detect_leaks=0", MOCK_HIP_DEVICES="4", MIC_INGEST_KB="128", MIC_CLI_ORDERLY_EXIT="1")   # (libomp keeps 128 bytes)
    for args, n_rec, said in ((["-d", "4", "--db-sharded", "--parts", "4", "-O", fq], 4000, "4 engine(s) on 4 device(s), table-sharded: 4 part(s) x 1 read group(s)"),
                              (["-d", "4", "--db-sharded", "--parts", "2", "-O", fq], 4000, "table-sharded: 2 part(s) x 2 read group(s)"),
                              (["-d", "3", "-O", fq], 4000, "3 engine(s) on 3 device(s), read-sharded"),
                              (["-d", "2", "-P", p1, p2], 2000, "2 engine(s) on 2 device(s), read-sharded")):
        res = os.path.join(tmp, "out")
        rc = subprocess.run([os.path.join(out, "cuCLARK_mock-l"), "-T", t, "-D", d, *args, "-R", res, "-n", "5"], capture_output=True, text=True, timeout=600, env=cli_env)
        assert rc.returncode == 0, (args, rc.stdout[-800:], rc.stderr[-3000:])
        assert "AddressSanitizer" not in rc.stderr and "runtime error" not in rc.stderr, (args, rc.stderr[-4000:])
        assert said in rc.stderr, (args, [ln for ln in rc.stderr.splitlines() if "Devices" in ln])
        assert sum(1 for _ in open(res + ".csv", "rb")) == n_rec + 1, args
    calls = {ln.split()[0]: (int(ln.split()[1]), int(ln.split()[3])) for ln in r.stdout.splitlines() if ln.startswith("  mic_")}
    # the rig must get INTO the code: tables load, batches run, ingest slots classify (what fails is what needs a kernel's answer)
    for name in ("mic_db_load_host", "mic_batches_alloc", "mic_batch_query", "mic_batch_merge_shards", "mic_ingest_classify", "mic_ingest_classify_group"):
        n, bad = calls[name]
        assert n > 20 and bad < n // 2, (name, n, bad)


def test_guard_page_allocator_catches_overruns_at_the_store(tmp_path):
    """tools/sanitize/guardalloc.c (LD_PRELOAD; tools/guard_soak.sh runs the command line and the fuzzer's product side under it): a
    block ends at an inaccessible page - a store 16 bytes past a 5000-byte block faults AT THE STORE with the arena named, a one-byte
    overrun into the alignment slack is reported at the free, a use after free faults, clean code runs clean."""
    import subprocess
    so = tmp_path / "guardalloc.so"
    r = subprocess.run(["gcc", "-O2", "-g", "-fPIC", "-shared", "-o", str(so), os.path.join(gu.ROOT, "tools", "sanitize", "guardalloc.c"), "-ldl", "-lpthread"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
int main(int argc, char** argv) {
  int mode = atoi(argv[1]);
  char* p = malloc(1000); memset(p, 1, 1000);
  char* q = realloc(p, 5000); memset(q, 2, 5000);
  char* c = calloc(100, 7); for (int i = 0; i < 700; ++i) if (c[i]) return 3;
  void* a; if (posix_memalign(&a, 256, 3000)) return 4; memset(a, 3, 3000); free(a);
  if (mode == 1) q[5000] = 9;
  if (mode == 2) q[5000 + 16] = 9;
  if (mode == 3) { free(q); q[10] = 1; }
  free(q); free(c);
  printf("ok\n");
  return 0;
}''')
    exe = tmp_path / "t"
    assert subprocess.run(["gcc", "-O0", "-g", "-o", str(exe), str(src)]).returncode == 0
    env = dict(os.environ, LD_PRELOAD=str(so), GUARD_REPORT="1")
    r = subprocess.run([str(exe), "0"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "ok" in r.stdout and "[guardalloc] guarded " in r.stderr and "canary failures 0" in r.stderr, r.stderr
    r = subprocess.run([str(exe), "1"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "overrun by 1 byte" in r.stderr, r.stderr
    for mode in ("2", "3"):
        r = subprocess.run([str(exe), mode], capture_output=True, text=True, env=env)
        assert r.returncode != 0 and "inside the guarded arena" in r.stderr and "ok" not in r.stdout, (mode, r.stderr)

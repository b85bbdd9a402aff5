#!/usr/bin/env python3
"""bench.py — Mreads/s of the k-mer query hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload full|light|tiny] [--mode read|db]

A "step" = one pass of the hot path (fused query kernel: canonical k-mer extraction + bucketed probe + per-read
per-target hit counts + best/second) over one batch of synthetic reads that is already resident in HBM.

Workloads, all synthetic, generated in HBM by libmi_clark.so's generators:
  full   config 3: 10 M x 150 bp reads (80 % sampled from the genomes with 1 % substitutions and 0.1 % N, 20 % random)
         vs. a 36 GB-on-disk-equivalent k=31 table: HTSIZE 1610612741, u32 keys, ~5.7e9 k-mers, 4096 targets,
         resident as 59 GB of 128-byte super-k-mer slots, one strand stored (--layout auto: the table exe/cuCLARK builds; the
         two-strand table - 119 GB, no reverse complement in the query - is timed beside it: `two_strand_table`). [default]
  light27 config 2 proper: the CuCLARK-l table as cuCLARK-l builds it: HTSIZE 57777779, k=27 (forced, main.cc:241-249), u32 keys,
         ~90 M k-mers; 10 M x 150 bp reads.
  light  config 2, k=31 side variant: same reads vs. HTSIZE 57777779, k=31 (u64 keys), ~54 M k-mers (not reachable through the
         reference's binaries; kept because the metric says k=31).
  tiny   plumbing-size case for quick checks.
Multi-GPU (one process per GPU over RCCL: `python bench.py --gpus N` starts its N ranks itself under torch.distributed.run as a
child process before anything touches the GPU; launched BY torch.distributed.run it is one of the ranks):
  read   reads sharded, table replicated, no collective (config 5 shape)      -> "scaling": "weak"   [default]
  db     table sharded (mic_db_set_part: super-k-mer layouts by resident slot range, so a run of a read belongs to one rank;
         other layouts by on-disk bucket range), every rank sees all reads, per-read sparse target-score rows exchanged
         with all_to_all over RCCL, merged and finalised per read range (config 4) -> "scaling": "strong"

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch / mean kernel time measured with HIP
events on the kernel's stream; `cpu_baseline` = the CPU oracle (oracle/, a port of the reference's CPU path) timed on
this box's host cores on a bounded sample, and that sample doubles as a bit-exact parity check of the GPU results.

At N=1 two more legs are measured on the same workload and reported next to `value` (SURVEY.md §8d "what to time"):
  "pipeline"    the batch API (mic_batches_alloc / ready / query / wait): packed reads in pinned host memory -> H2D ->
                kernel -> D2H of the result rows, batches overlapped on their streams; Mreads/s and PCIe GB/s.
  "end_to_end"  exe/cuCLARK on files: the table written as .sz/.ky/.lb, the same reads as a FASTQ file, results as CSV
                (device-side ingest: DESIGN.md §5.2); the reference's own "Assignment time ... objects/min" line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # htsize, genome_nt, n_genomes, n_targets, k, key_bytes, n_reads, read_len
    "full": dict(htsize=1610612741, genome_nt=5_730_000_000, n_genomes=8192, n_targets=4096, k=31, key_bytes=4,
                 n_reads=10_000_000, read_len=150,
                 name="10M x 150bp synthetic reads vs 36GB-scale k=31 table (HTSIZE 1610612741, u32 keys, ~5.7e9 k-mers, "
                      "4096 targets) resident in HBM"),
    "light": dict(htsize=57777779, genome_nt=54_000_000, n_genomes=512, n_targets=512, k=31, key_bytes=8,
                  n_reads=10_000_000, read_len=150,
                  name="10M x 150bp synthetic reads vs CuCLARK-l-scale k=31 table (HTSIZE 57777779, u64 keys, ~54M k-mers)"),
    # SURVEY.md 8d config 2 as the reference's cuCLARK-l builds it: k forced to 27 (main.cc:241-249), u32 keys, ~90 M k-mers
    "light27": dict(htsize=57777779, genome_nt=90_000_000, n_genomes=512, n_targets=512, k=27, key_bytes=4,
                    n_reads=10_000_000, read_len=150,
                    name="10M x 150bp synthetic reads vs the CuCLARK-l table proper (HTSIZE 57777779, k=27, u32 keys, ~90M k-mers)"),
    # config 5 shape: paired-end 2 x 150 bp, every object = read 1 + 'N' + read 2 (file.cc:205-268), same table as "full"
    "paired": dict(htsize=1610612741, genome_nt=5_730_000_000, n_genomes=8192, n_targets=4096, k=31, key_bytes=4,
                   n_reads=10_000_000, read_len=150, paired=True,
                   name="10M pairs of 2x150bp synthetic reads (301-character objects) vs 36GB-scale k=31 table (HTSIZE 1610612741, u32 keys, "
                        "~5.7e9 k-mers, 4096 targets) resident in HBM"),
    # the label space at its limit (dataType.hh:48: ILBL = u16): 65 535 targets, one small genome each
    "full_t65535": dict(htsize=1610612741, genome_nt=5_730_000_000, n_genomes=65535, n_targets=65535, k=31, key_bytes=4,
                        n_reads=10_000_000, read_len=150,
                        name="10M x 150bp synthetic reads vs 36GB-scale k=31 table with 65535 targets (HTSIZE 1610612741, u32 keys, ~5.7e9 k-mers) "
                             "resident in HBM"),
    # Databases of DISCRIMINATIVE k-mers: the same 5.73 G candidate k-mers, of which the removal of k-mers common to
    # several targets (HashTableStorage_hh.hh:241-292) leaves 50 % / 25 % in runs of geometric length, mean 8 (mic_synth_spec.keep_ppm)
    "full_frag50": dict(htsize=1610612741, genome_nt=5_730_000_000, n_genomes=8192, n_targets=4096, k=31, key_bytes=4, n_reads=10_000_000, read_len=150,
                        keep_ppm=500_000, run_len=8,
                        name="10M x 150bp synthetic reads vs a FRAGMENTED 36GB-scale k=31 table: 50 % of ~5.7e9 candidate k-mers kept in runs of mean length 8 "
                             "(HTSIZE 1610612741, u32 keys, 4096 targets)"),
    "full_frag25": dict(htsize=1610612741, genome_nt=5_730_000_000, n_genomes=8192, n_targets=4096, k=31, key_bytes=4, n_reads=10_000_000, read_len=150,
                        keep_ppm=250_000, run_len=8,
                        name="10M x 150bp synthetic reads vs a FRAGMENTED 36GB-scale k=31 table: 25 % of ~5.7e9 candidate k-mers kept in runs of mean length 8 "
                             "(HTSIZE 1610612741, u32 keys, 4096 targets)"),
    # Genomes that are NOT uniformly random (mic_synth_spec.repeat_ppm / mosaic_ppm):
    #   full_repeats  2 % of every genome is tandem repeats, the short units shared by all genomes (microsatellites): crowded minimizers,
    #                 a side table, reads with several times the usual number of runs
    #   full_homolog  15 % of the 2-kb segments of every genome carry mosaic labels (the label of a k-mer changes every 1 / 2 / 4 / 8
    #                 positions): ties between best and second, rows of more than 15 and more than 64 targets (the dense path)
    "full_repeats": dict(htsize=1610612741, genome_nt=5_730_000_000, n_genomes=8192, n_targets=4096, k=31, key_bytes=4, n_reads=10_000_000, read_len=150,
                         repeat_ppm=20_000,
                         name="10M x 150bp synthetic reads vs 36GB-scale k=31 table whose genomes are 2 % tandem repeats, short units shared across genomes "
                              "(HTSIZE 1610612741, u32 keys, 4096 targets) resident in HBM"),
    "full_homolog": dict(htsize=1610612741, genome_nt=5_730_000_000, n_genomes=8192, n_targets=4096, k=31, key_bytes=4, n_reads=10_000_000, read_len=150,
                         mosaic_ppm=150_000,
                         name="10M x 150bp synthetic reads vs 36GB-scale k=31 table with 15 % of every genome in segments of mosaic labels "
                              "(HTSIZE 1610612741, u32 keys, 4096 targets) resident in HBM"),
    "tiny_repeats": dict(htsize=999983, genome_nt=3_000_000, n_genomes=64, n_targets=50, k=31, key_bytes=8, n_reads=100_000, read_len=100, repeat_ppm=50_000,
                         name="100k x 100bp synthetic reads vs a 50-target toy table with 5 % tandem repeats (plumbing)"),
    "tiny_homolog": dict(htsize=9999991, genome_nt=8_000_000, n_genomes=1000, n_targets=1000, k=31, key_bytes=8, n_reads=100_000, read_len=150, mosaic_ppm=150_000,
                         name="100k x 150bp synthetic reads vs a 1000-target toy table with 15 % mosaic-label segments (plumbing)"),
    "tiny_frag": dict(htsize=999983, genome_nt=1_500_000, n_genomes=64, n_targets=50, k=31, key_bytes=8, n_reads=100_000, read_len=100, keep_ppm=500_000, run_len=8,
                      name="100k x 100bp synthetic reads vs a fragmented 50-target toy table (plumbing)"),
    "tiny_paired": dict(htsize=999983, genome_nt=1_500_000, n_genomes=64, n_targets=50, k=31, key_bytes=8,
                        n_reads=100_000, read_len=100, paired=True,
                        name="100k pairs of 2x100bp synthetic reads vs 50-target toy table (plumbing)"),
    "tiny": dict(htsize=999983, genome_nt=1_500_000, n_genomes=64, n_targets=50, k=31, key_bytes=8,
                 n_reads=100_000, read_len=100,
                 name="100k x 100bp synthetic reads vs 50-target toy table (plumbing)"),
}
PART_MODE = {1: "table-sharded by on-disk bucket range", 2: "table-sharded by on-disk bucket range",
             3: "table-sharded by resident slot range (a run of a read belongs to one rank)",
             4: "table-sharded by resident slot range (a run of a read belongs to one rank)"}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
CLOCK_HZ = 2.4e9               # MI355X engine clock (MI355X_MICROARCH.md)
RANDOM_SECTOR_GREQ = 51.4      # measured: random 64-B nontemporal requests/s this chip sustains (tools/gather_runs_bench.hip)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def pipeline_leg(eng, L, d_rp, d_cont, n_reads, res_expect, steps, nb):
    """the device pipeline through the batch API.  Packed reads sit in the engine's pinned host buffers
    (as CuCLARK's packer leaves them, CuCLARK_hh.hh:1616-1716); a step = every batch submitted (H2D + kernel + D2H of the
    result rows on its own stream, CuClarkDB.cu:878-1033) and then awaited."""
    rp = d_rp.cpu().numpy().view(np.uint32)
    cont = d_cont.cpu().numpy().view(np.uint16)
    # The generator lays every read out at a fixed pitch (length slots, containers, a 0 that ends the read, padding).  The batch
    # API's contract is the PACKER's format - reads back to back, no terminator (CuCLARK_hh.hh:1616-1716) - so the reads are
    # compacted to it first (not timed): the leg then ships what a caller of the API ships, ~44 B per 150-bp read.
    pos = rp[:-1].astype(np.int64).copy()
    end = rp[1:].astype(np.int64)
    live = np.ones(n_reads, bool)
    for _ in range(64):                                   # parts of a read: a length slot, then ceil(len / 8) containers
        live &= pos < end
        plen = np.where(live, cont[np.minimum(pos, cont.size - 1)], 0).astype(np.int64)
        live &= plen > 0
        if not live.any():
            break
        pos = np.where(live, pos + 1 + (plen + 7) // 8, pos)
    used = (pos - rp[:-1]).astype(np.int64)
    rp_c = np.zeros(n_reads + 1, np.int64)
    np.cumsum(used, out=rp_c[1:])
    src = np.repeat(rp[:-1].astype(np.int64) - rp_c[:-1], used) + np.arange(int(rp_c[-1]), dtype=np.int64)
    cont = cont[src]
    rp = rp_c.astype(np.uint32)
    del src, pos, end, live, used
    per = (n_reads + nb - 1) // nb
    cuts = [min(n_reads, b * per) for b in range(nb + 1)]
    max_cont = max(int(rp[cuts[b + 1]] - rp[cuts[b]]) for b in range(nb)) + 64
    bufs = eng.malloc(n_reads, per, max_cont, cuts)
    n_cont = []
    for b in range(nb):
        lo, hi = cuts[b], cuts[b + 1]
        c0, c1 = int(rp[lo]), int(rp[hi])
        bufs["reads_pointer"][b][: hi - lo + 1] = rp[lo:hi + 1] - rp[lo]
        bufs["containers"][b][: c1 - c0] = cont[c0:c1]
        n_cont.append(c1 - c0)

    def step():
        for b in range(nb):
            eng.readyBatch(b, cuts[b + 1] - cuts[b], n_cont[b])
            eng.queryBatch(b)
        for b in range(nb):
            eng.waitForBatch(b)
    step()
    equal = bool((bufs["results"][:, :5] == res_expect[:, :5]).all())
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    h2d = sum((cuts[b + 1] - cuts[b] + 1) * 4 + (n_cont[b] + 16) * 2 for b in range(nb))
    d2h = n_reads * 32
    eng.freeBatchMemory()
    return {"value": round(n_reads / dt / 1e6, 1), "unit": "Mreads/s", "ms_per_pass": round(dt * 1e3, 3), "batches": nb, "steps": steps,
            "h2d_GBs": round(h2d / dt / 1e9, 1), "d2h_GBs": round(d2h / dt / 1e9, 1),
            "bytes_per_read": {"h2d": round(h2d / n_reads, 1), "d2h": 32},
            "what": "reads in the packer's format (back to back, no padding) in pinned host memory -> H2D -> query kernel -> D2H of the 32-byte result rows, per-batch streams "
                    "(mic_batches_alloc / mic_batch_ready / mic_batch_query / mic_batch_wait)",
            "results_equal_device_path": equal}


def gzip_sub_leg(cmd, fqs, rec, n, res_base, tmp):
    """BASELINE config 5 names gzip input: the first n pairs of the same files as plain gzip (one zlib stream per file, what
    `gzip` writes) and as block gzip (BGZF, what bgzip / samtools write), through the same command line; the CSV must be the
    first n lines of the plain run's.  The reference gunzips to a temporary file first (classify_metagenome.sh:116-142)."""
    import re
    import struct
    import subprocess
    import zlib
    with open(res_base + ".csv", "rb") as f:
        expect = b"".join(f.readline() for _ in range(n + 1))
    out = {"pairs": n}
    heads = []
    for i, fq in enumerate(fqs):
        with open(fq, "rb") as f:
            heads.append(f.read(n * rec))
    for kind in ("gzip", "bgzf"):
        names = []
        t0 = time.time()
        for i, data in enumerate(heads):
            name = os.path.join(tmp, f"{kind}_{i + 1}.fq.gz")
            names.append(name)
            if kind == "gzip":
                with open(name, "wb") as f:
                    subprocess.run(["gzip", "-1", "-c"], input=data, stdout=f, check=True)
            else:
                with open(name, "wb") as f:
                    for o in range(0, len(data), 0xFF00):
                        blk = data[o:o + 0xFF00]
                        c = zlib.compressobj(1, zlib.DEFLATED, -15)
                        body = c.compress(blk) + c.flush()
                        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body +
                                struct.pack("<II", zlib.crc32(blk), len(blk)))
                    f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")
        t_make = time.time() - t0
        res = os.path.join(tmp, "out_" + kind)
        c2 = list(cmd)
        ip = c2.index("-P")
        c2[ip + 1], c2[ip + 2] = names
        c2[c2.index("-R") + 1] = res
        r = subprocess.run(c2, capture_output=True, text=True, env=dict(os.environ, MIC_CLI_TIMING="1"))
        if r.returncode != 0:
            out[kind] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        m = re.search(r"Assignment time: ([0-9.eE+-]+) s\. Speed: (\d+) objects/min\. \((\d+) objects\)", r.stdout)
        t_assign = float(m.group(1))
        out[kind] = {"Mpairs_s": round(int(m.group(3)) / t_assign / 1e6, 2), "assignment_s": round(t_assign, 3),
                     "compressed_MB": round(sum(os.path.getsize(x) for x in names) / 1e6, 1), "made_in_s": round(t_make, 1),
                     "csv_equals_plain_run": open(res + ".csv", "rb").read() == expect,
                     "inflated_on": "device" if re.search(r"device inflate: [0-9.]+ MB of text", r.stderr) else "host"}
    return out


ASSIGN_RE = r"Assignment time: ([0-9.eE+-]+) s\. Speed: (\d+) objects/min\. \((\d+) objects\)"


def run_cli(cmd, extra_env=None):
    """one run of exe/cuCLARK with its timing lines on; returns (CompletedProcess, parsed dict)"""
    import re
    import subprocess
    env = dict(os.environ, MIC_CLI_TIMING="1", MIC_LOAD_TIMING="1")
    env.update(extra_env or {})
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    out = {"process_wall_s": round(wall, 2)}
    if r.returncode != 0:
        out["error"] = (r.stderr or r.stdout)[-400:]
        return r, out
    m = re.search(ASSIGN_RE, r.stdout)
    out["assignment_s"] = round(float(m.group(1)), 4)
    out["objects"] = int(m.group(3))
    out["value"] = round(int(m.group(3)) / float(m.group(1)) / 1e6, 1)
    out["unit"] = "Mreads/s"
    ing = re.search(r"device ingest: (\d+) batches of <= (\d+) KB on (\d+) slot\(s\), (\d+) through the host path; threads: (\d+) load, (\d+) device, (\d+) write; "
                    r"thread-seconds: load ([0-9.e+-]+), device ([0-9.e+-]+), write ([0-9.e+-]+); input ([0-9.e+]+) MB, over the link ([0-9.e+]+) MB", r.stderr)
    if ing:
        out["ingest"] = {"batches": int(ing.group(1)), "slot_KB": int(ing.group(2)), "slots": int(ing.group(3)), "batches_through_host_path": int(ing.group(4)),
                         "threads": {"load": int(ing.group(5)), "device": int(ing.group(6)), "write": int(ing.group(7))},
                         "thread_seconds": {"load": float(ing.group(8)), "device": float(ing.group(9)), "write": float(ing.group(10))},
                         "input_MB": float(ing.group(11)), "h2d_MB": float(ing.group(12))}
    out["table_load_s"] = {a.strip(): float(b) for a, b in re.findall(r"\[load(?: x\d+)?\] ([^:\n]+): ([0-9.]+) s", r.stderr)}
    kn = re.search(r"\[timing\] query kernel: (.+)", r.stderr)
    if kn:
        out["kernel"] = kn.group(1).strip()
    return r, out


def multi_engine_leg(base_cmd, res_one, tmp, paired_gz=None, only=None):
    """the product binary's multi-device modes at FULL scale on this one GPU (MIC_SHARD_ENGINES=n puts n engines on it):
    the same table files and the same 10 M-read FASTQ as the one-engine run; every CSV must equal that run's byte for byte.
      --db-sharded --parts N on N = 2 / 4 / 8 engines  the reference's mode (CuClarkDB.cu:566-574, 886-1024): N parts of the table, every
                                                       batch probed by all N, rows exchanged read-range owned over peer copies
      --db-sharded --parts 2 on 4 engines              2-D: 2 parts x 2 read groups
      -d 2 on 2 engines                                read-sharded: two whole tables, batches dealt
      -P a.fq.gz b.fq.gz -d 2 on 2 engines             compressed mates inflated on the first engine's device, slots on both engines
    Reported per run: assignment time, the table load split (one read of the files / the builds per device), and - from HIP events on
    every engine's stream (MIC_GROUP_TIMING) - bytes and time of the packed-read fan-out and of the row exchange next to the kernels."""
    import filecmp
    import re
    out = {"what": "exe/cuCLARK with n engines on ONE GPU (MIC_SHARD_ENGINES): same files as the one-engine run, CSVs compared byte for byte; "
                   "'peer' copies are device-local here, so exchange_ms is a floor for what xGMI adds", "runs": {}}
    runs = [("db_sharded_2", ["--db-sharded", "--parts", "2"], 2), ("db_sharded_4", ["--db-sharded", "--parts", "4"], 4),
            ("db_sharded_8", ["--db-sharded", "--parts", "8"], 8), ("db_sharded_4_parts_2", ["--db-sharded", "--parts", "2"], 4),
            ("read_sharded_2", ["-d", "2"], 2)]
    for name, flags, n_eng in runs:
        if only and name not in only:
            continue
        if not only and name == "db_sharded_4":      # (25 s of part builds on one device for a point between 2 and 8: on request only)
            continue
        res = os.path.join(tmp, "out_" + name)
        cmd = [res if a == res_one else a for a in base_cmd] + flags
        r, d = run_cli(cmd, {"MIC_SHARD_ENGINES": str(n_eng), "MIC_GROUP_TIMING": "1"})
        d["engines"] = n_eng
        d["flags"] = " ".join(flags)
        if "error" not in d:
            d["csv_equals_one_engine_run"] = filecmp.cmp(res_one + ".csv", res + ".csv", shallow=False)
            dv = re.search(r"Devices: (.+)", r.stderr)
            d["layout"] = dv.group(1).split(";")[0] if dv else None
            g = re.search(r"table-sharded batches: (\d+) timed, (\d+) reads, (\d+) part\(s\); packed-read fan-out ([0-9.e+-]+) MB, ([0-9.e+-]+) ms summed over the helpers "
                          r"\(slowest helper of each batch: ([0-9.e+-]+) ms\); query kernels ([0-9.e+-]+) ms summed over the engines \(slowest engine of each batch: ([0-9.e+-]+) ms\); "
                          r"row exchange ([0-9.e+-]+) MB, ([0-9.e+-]+) ms summed over the engines \(slowest engine of each batch: ([0-9.e+-]+) ms\)", r.stderr)
            if g:
                nb = max(int(g.group(1)), 1)
                d["per_batch"] = {"batches": nb, "reads": int(g.group(2)) // nb,
                                  "fanout_MB": round(float(g.group(4)) / nb, 2), "fanout_ms_slowest_helper": round(float(g.group(6)) / nb, 3),
                                  "kernel_ms_all_engines": round(float(g.group(7)) / nb, 3), "kernel_ms_slowest_engine": round(float(g.group(8)) / nb, 3),
                                  "exchange_MB": round(float(g.group(9)) / nb, 2), "exchange_ms_all_engines": round(float(g.group(10)) / nb, 3),
                                  "exchange_ms_slowest_engine": round(float(g.group(11)) / nb, 3)}
            try:
                os.unlink(res + ".csv")
            except OSError:
                pass
        out["runs"][name] = d
        log("multi_engine", name + ":", json.dumps(d))
    if paired_gz and (not only or "paired_gzip_read_sharded_2" in only):
        # compressed mates through one engine and through two (read-sharded): the text is inflated on the first engine's device and
        # the slots of both engines are filled from there
        a, b, cmd_p = paired_gz
        res1, res2 = os.path.join(tmp, "out_gz1"), os.path.join(tmp, "out_gz2")
        c1 = [res1 if x == res_one else x for x in cmd_p]
        c2 = [res2 if x == res_one else x for x in cmd_p] + ["-d", "2"]
        r1, d1 = run_cli(c1)
        r2, d2 = run_cli(c2, {"MIC_SHARD_ENGINES": "2"})
        d2["engines"] = 2
        d2["flags"] = "-P a.fq.gz b.fq.gz -d 2"
        if "error" not in d1 and "error" not in d2:
            d2["csv_equals_one_engine_run"] = filecmp.cmp(res1 + ".csv", res2 + ".csv", shallow=False)
            d2["one_engine_value"] = d1["value"]
            d2["inflated_on"] = "device" if re.search(r"device inflate: [0-9.]+ MB of text", r2.stderr) else "host"
        elif "error" in d1:
            d2["error"] = "one-engine run: " + d1["error"]
        out["runs"]["paired_gzip_read_sharded_2"] = d2
        log("multi_engine paired_gzip_read_sharded_2:", json.dumps(d2))
    out["all_csv_equal"] = all(v.get("csv_equals_one_engine_run") is True for v in out["runs"].values())
    return out


def end_to_end_leg(L, spec, w, images, n_el, n_reads, read_len, res_expect, truth, threads, keep=False, paired=False, reps=3, multi=True, multi_reads=0, multi_only=None,
                   leg_allowed=lambda name, est: True, partial=None, current_leg=None):
    """files in, file out, through exe/cuCLARK (reference: CuCLARK_hh.hh:550-563 times index + pack +
    GPU + CSV and prints objects/min, :1938-1944).  The table goes to disk in the reference's format, the same reads as
    FASTQ; the binary loads the table, classifies, writes the CSV.  Checked: every CSV line against the kernel's result
    rows of the same reads."""
    import shutil
    import subprocess
    import tempfile
    k, T = w["k"], w["n_targets"]
    tmp = tempfile.mkdtemp(prefix="mic_e2e_", dir=os.environ.get("MIC_BENCH_TMP", "/tmp"))
    out = {}
    try:
        t0 = time.time()
        d_sizes, d_keys, d_labels = images
        dev = d_sizes.device
        dbdir = os.path.join(tmp, "DB")
        os.makedirs(dbdir)
        base = os.path.join(dbdir, f"db_central_k{k}_t{T}_s{w['htsize']}_m0.tsk")
        d_sizes.cpu().numpy().tofile(base + ".sz")
        CH = 1 << 30
        for arr, ext in ((d_keys, ".ky"), (d_labels, ".lb")):
            with open(base + ext, "wb") as f:
                for o in range(0, n_el, CH):
                    arr[o:min(n_el, o + CH)].cpu().numpy().tofile(f)
        # the table's images are on disk now: this process gives their 36 GB of HBM back before the command line loads and builds
        # its own table next to it (the build takes its fast road when the memory is there)
        del d_sizes, d_keys, d_labels
        images.clear()
        torch.cuda.empty_cache()
        dummy = os.path.join(tmp, "genome.fa")
        open(dummy, "w").write(">g\nACGT\n")
        with open(os.path.join(tmp, "targets.txt"), "w") as f:
            for t in range(T):
                f.write(f"{dummy} TARGET_{t:05d}\n")
        rec = int(L.mic_synth_text_record_bytes(read_len, 0))
        d_text = torch.empty(n_reads * rec, dtype=torch.uint8, device=dev)
        fqs = []
        for mate in ((0, 1) if paired else (-1,)):
            rc = L.mic_synth_reads_text_device(C.byref(spec), 5, n_reads, read_len, 0.2, 0.01, 0.001, 0, mate, d_text.data_ptr(), d_text.numel(), None)
            assert rc == 0, f"mic_synth_reads_text_device failed ({rc})"
            torch.cuda.synchronize()
            fqs.append(os.path.join(tmp, f"reads_{max(mate, 0) + 1}.fq"))
            d_text.cpu().numpy().tofile(fqs[-1])
        del d_text
        t_files = time.time() - t0
        exe = os.path.join(ROOT, "exe", "cuCLARK")
        res_base = os.path.join(tmp, "out")
        cmd = [exe, "-k", str(k), "--htsize", str(w["htsize"]), "-T", os.path.join(tmp, "targets.txt"), "-D", dbdir,
               *(["-P", fqs[0], fqs[1]] if paired else ["-O", fqs[0]]), "-R", res_base, "-n", str(threads)]
        # the run is repeated (--e2e-reps, default 3): `value` is the MEDIAN of the assignment rates, min and max beside it - one run
        # is a draw from a spread of +-20 % on this pool (the loaders' rate inside the job's CPU quota moves from run to run)
        r, first = run_cli(cmd)
        if "error" in first:
            return {"error": first["error"]}
        import re
        reps_out = [first]
        import filecmp
        for i in range(1, max(1, reps)):
            res_i = os.path.join(tmp, f"out_rep{i}")
            ri, di = run_cli([res_i if a == res_base else a for a in cmd])
            if "error" in di:
                return {"error": di["error"]}
            di["csv_equals_first_run"] = filecmp.cmp(res_base + ".csv", res_i + ".csv", shallow=False)
            os.unlink(res_i + ".csv")
            reps_out.append(di)
        by_rate = sorted(reps_out, key=lambda d: d["value"])
        med = by_rate[len(by_rate) // 2]
        t_assign, n_obj = med["assignment_s"], med["objects"]
        opm = int(n_obj / t_assign * 60)
        ing_d = med.get("ingest")
        load = first["table_load_s"]
        wall = first["process_wall_s"]
        # every CSV line against the kernel's rows of the same reads
        import pandas as pd
        names = np.array(["NA"] + [f"TARGET_{t:05d}" for t in range(T)])
        df = pd.read_csv(res_base + ".csv", keep_default_na=False, usecols=["Object_ID", "Length", "1st_assignment", "score1", "2nd_assignment", "score2"],
                         dtype={"Object_ID": str, "Length": np.uint32, "1st_assignment": str, "score1": np.uint32, "2nd_assignment": str, "score2": np.uint32})
        lines = len(df)
        ok = lines == n_reads
        if ok:
            e = res_expect[:n_reads]
            ok = bool((df["1st_assignment"].to_numpy() == names[e[:, 1]]).all() and (df["score1"].to_numpy() == e[:, 2]).all() and
                      (df["2nd_assignment"].to_numpy() == names[e[:, 3]]).all() and (df["score2"].to_numpy() == e[:, 4]).all() and
                      (df["Length"].to_numpy() == (2 * read_len if paired else read_len)).all() and df["Object_ID"].iloc[0] == "r000000000" and
                      df["Object_ID"].iloc[-1] == f"r{n_reads - 1:09d}")
        del df
        fq_bytes = sum(os.path.getsize(f) for f in fqs)
        out = {"value": med["value"], "unit": "Mreads/s", "value_is": f"median of {len(reps_out)} runs of the command",
               "min": by_rate[0]["value"], "max": by_rate[-1]["value"], "runs": [d["value"] for d in reps_out],
               "runs_csv_equal": all(d.get("csv_equals_first_run", True) for d in reps_out),
               "objects_per_min": opm, "objects": n_obj,
               "assignment_s": round(t_assign, 4), "process_wall_s": round(wall, 2), "process_wall_s_runs": [d["process_wall_s"] for d in reps_out],
               "table_load_s": load,
               "input": (f"two FASTQ files (pairs), {fq_bytes / n_reads:.0f} bytes per pair, " if paired else f"FASTQ, {fq_bytes / n_reads:.0f} bytes per record, ") +
                        f"{fq_bytes / 1e9:.2f} GB in the page cache",
               "input_GBs": round(fq_bytes / t_assign / 1e9, 1), "csv_MB": round(os.path.getsize(res_base + ".csv") / 1e6, 1),
               "host_threads": threads, "ingest": (dict(ing_d, h2d_GBs=round(ing_d["h2d_MB"] / 1e3 / t_assign, 1)) if ing_d else None),
               "kernel": first.get("kernel"),
               "command": "exe/cuCLARK -k %d -T targets.txt -D DB/ %s -R out -n %d" % (k, "-P reads_1.fq reads_2.fq" if paired else "-O reads_1.fq", threads),
               "csv_lines_equal_kernel_rows": bool(ok and lines == n_reads), "setup_files_s": round(t_files, 1)}
        if partial is not None:
            partial["end_to_end"] = out          # (the sub-legs below add to it: what is there when the time budget cuts the run short)
        full_scale = n_reads >= 5_000_000
        # ---- what bounds the run: every stage's busy share (thread-seconds / threads / assignment time, the median run's) and - plain FASTQ -
        # the loaders ALONE on the same file with the same thread count and chunk size, no device work (exe/cuCLARK --strip-fastq ...
        # loaders: pread of 256 KiB into a stage that stays in L2, AVX2 strip into a slot-sized buffer; tools/loader_rate.sh)
        if ing_d:
            th, ts = ing_d["threads"], ing_d["thread_seconds"]
            shares = {st: round(ts[st] / max(th[st], 1) / t_assign, 3) for st in ("load", "device", "write")}
            out["stage_busy_share"] = {"loaders": shares["load"], "device_threads": shares["device"], "writer": shares["write"]}
            out["device_busy_share"], out["writer_busy_share"] = shares["device"], shares["write"]
            out["h2d_GBs"] = round(ing_d["h2d_MB"] / 1e3 / t_assign, 1)
            if not paired:
                import subprocess
                lr = subprocess.run([exe, "--strip-fastq", fqs[0], "-", "262144", "loaders", str(th["load"])], capture_output=True, text=True)
                rates = sorted(float(x) for x in re.findall(r": ([0-9.]+) GB/s", lr.stdout))
                if rates:
                    alone = rates[len(rates) // 2]
                    in_run = ing_d["input_MB"] / 1e3 / t_assign
                    out["loaders_alone_GBs"] = round(alone, 1)
                    out["loaders_alone_runs_GBs"] = [round(x, 1) for x in rates]
                    out["loaders_in_run_GBs"] = round(in_run, 1)
                    out["loaders_in_run_vs_alone"] = round(in_run / alone, 3)
            top = max(shares, key=lambda st_: shares[st_])
            bound = {"load": "loaders (page cache -> stripped FASTQ in pinned slots)", "device": "device threads (H2D + kernels + D2H per batch)",
                     "write": "CSV writer (one pwrite stream)"}[top]
            if top == "load" and "loaders_alone_GBs" in out:
                bound += (f": {out['loaders_in_run_GBs']} GB/s of input in the run against {out['loaders_alone_GBs']} GB/s for the same {th['load']} loader threads "
                          f"with nothing else running ({out['loaders_in_run_vs_alone']:.2f} x)")
            out["bound"] = bound
        if multi and not paired and leg_allowed("end_to_end.multi_engine", 110 if full_scale else 40):
            if current_leg is not None:
                current_leg[0] = "end_to_end.multi_engine"
            me_cmd, me_ref, me_n = cmd, res_base, n_reads
            if 0 < multi_reads < n_reads:
                # (tests: the same legs on the first multi_reads reads of the file, with a one-engine run of their own to compare against)
                short = os.path.join(tmp, "reads_head.fq")
                with open(fqs[0], "rb") as f, open(short, "wb") as g:
                    g.write(f.read(multi_reads * rec))
                me_ref, me_n = os.path.join(tmp, "out_head"), multi_reads
                me_cmd = [short if a == fqs[0] else me_ref if a == res_base else a for a in cmd]
                r0, d0 = run_cli(me_cmd)
                if "error" in d0:
                    raise RuntimeError("one-engine run on the head of the file: " + d0["error"])
            # compressed mates for the two-engine run: the first million pairs of the same generator, as `gzip -1` files
            pgz = None
            try:
                if multi_only and "paired_gzip_read_sharded_2" not in multi_only:
                    raise LookupError("not asked for")
                import subprocess
                n_p = min(me_n, 1_000_000)
                d_t = torch.empty(n_p * rec, dtype=torch.uint8, device=dev)
                gz_names = []
                for mate in (0, 1):
                    rc = L.mic_synth_reads_text_device(C.byref(spec), 5, n_p, read_len, 0.2, 0.01, 0.001, 0, mate, d_t.data_ptr(), d_t.numel(), None)
                    assert rc == 0
                    torch.cuda.synchronize()
                    name = os.path.join(tmp, f"mate_{mate + 1}.fq.gz")
                    with open(name, "wb") as f:
                        subprocess.run(["gzip", "-1", "-c"], input=d_t.cpu().numpy().tobytes(), stdout=f, check=True)
                    gz_names.append(name)
                del d_t
                cmd_p = [exe, "-k", str(k), "--htsize", str(w["htsize"]), "-T", os.path.join(tmp, "targets.txt"), "-D", dbdir, "-P", gz_names[0], gz_names[1],
                         "-R", me_ref, "-n", str(threads)]
                pgz = (gz_names[0], gz_names[1], cmd_p)
            except Exception as ex:
                log("multi_engine: no compressed mates:", repr(ex))
            torch.cuda.empty_cache()
            out["multi_engine"] = multi_engine_leg(me_cmd, me_ref, tmp, pgz, only=multi_only)
            out["multi_engine"]["reads"] = me_n
        if paired:
            if leg_allowed("end_to_end.gzip_input", 40 if full_scale else 15):
                if current_leg is not None:
                    current_leg[0] = "end_to_end.gzip_input"
                out["gzip_input"] = gzip_sub_leg(cmd, fqs, rec, min(n_reads, 1_000_000), res_base, tmp)
        elif not os.environ.get("MIC_BENCH_NO_FASTA") and leg_allowed("end_to_end.fasta_input", 20 if full_scale else 8):
            if current_leg is not None:
                current_leg[0] = "end_to_end.fasta_input"
            # the same reads as FASTA (header + sequence, no quality lines: 163 instead of 316 bytes per record): the run is bound by
            # the rate at which the loaders take the file out of the page cache, so half the bytes are nearly twice the reads per second
            text = np.fromfile(fqs[0], np.uint8).reshape(n_reads, rec)
            fa = np.ascontiguousarray(text[:, : rec - read_len - 3])
            del text
            fa[:, 0] = ord(">")
            fa_path = os.path.join(tmp, "reads_1.fa")
            fa.tofile(fa_path)
            fa_bytes = fa.size
            del fa
            res_fa = os.path.join(tmp, "out_fa")
            cmd_fa = [fa_path if a == fqs[0] else res_fa if a == res_base else a for a in cmd]
            r2 = subprocess.run(cmd_fa, capture_output=True, text=True)
            m2 = re.search(r"Assignment time: ([0-9.eE+-]+) s\. Speed: (\d+) objects/min\. \((\d+) objects\)", r2.stdout) if r2.returncode == 0 else None
            if m2:
                import filecmp
                t2 = float(m2.group(1))
                out["fasta_input"] = {"value": round(int(m2.group(3)) / t2 / 1e6, 1), "unit": "Mreads/s", "assignment_s": round(t2, 4),
                                      "input": f"FASTA, {fa_bytes / n_reads:.0f} bytes per record, {fa_bytes / 1e9:.2f} GB in the page cache",
                                      "input_GBs": round(fa_bytes / t2 / 1e9, 1),
                                      "csv_equals_fastq_run": filecmp.cmp(res_base + ".csv", res_fa + ".csv", shallow=False)}
            else:
                out["fasta_input"] = {"error": (r2.stderr or r2.stdout)[-300:]}
    finally:
        if not keep and not os.environ.get("MIC_BENCH_KEEP"):
            shutil.rmtree(tmp, ignore_errors=True)
        elif os.environ.get("MIC_BENCH_KEEP"):
            log("end_to_end: files kept in", tmp)
    return out


def cpu_quota_cpus():
    """CPUs' worth of CPU time the process's cgroup allows (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), or None
    when there is no quota - what csrc/pgz.hpp: usable_cpus() reads for the command line's thread pools."""
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(period), 2)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return round(q / period, 2) if q > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def launch_ranks(args, n_dev):
    """Start `args.gpus` ranks of this script under torch.distributed.run (one process per GPU over RCCL, rendezvous on
    127.0.0.1) as a child process, pass its stdout (rank 0's one JSON line) and stderr through, return its exit code.
    Called before anything in this process has initialised the GPU."""
    import socket
    import subprocess
    if args.backend == "nccl" and n_dev < args.gpus:
        log(f"bench.py: --gpus {args.gpus} over RCCL needs {args.gpus} visible devices, this box has {n_dev}. "
            "RCCL needs one device per rank; --backend gloo lets ranks share a GPU (validation only)")
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: starting", args.gpus, "ranks:", " ".join(cmd[1:8]), "...")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="full", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="read", choices=["read", "db"])
    ap.add_argument("--reads", type=int, default=0, help="override the number of reads per GPU")
    ap.add_argument("--read-len", type=int, default=0, help="override the read length (exploration only: the headline is 150 bp)")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="reads timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages the row exchange through host memory (validation on a box with fewer GPUs than ranks)")
    ap.add_argument("--layout", default="auto", choices=["auto", "direct", "minimizer", "super", "super2"],
                    help="resident table layout (DESIGN.md 3).  auto (default) = what exe/cuCLARK builds: super-k-mer slots, one strand "
                         "(k >= 24).  super2 = both strands stored: the fastest query kernel at twice the table and 2 s more build "
                         "(timed beside the headline as `two_strand_table`); falls back to super when it does not fit")
    ap.add_argument("--time-budget", type=float, default=420.0,
                    help="seconds from the start of this process after which no further extra leg (pipeline, table_sharded_proxy, "
                         "two_strand_table, cross_layouts, end_to_end and its sub-legs) is STARTED; the line names them in \"skipped_legs\". "
                         "A leg still running 90 s past the budget is abandoned and the line is printed without it")
    ap.add_argument("--cross-layouts", default="",
                    help="N=1: comma-separated layouts whose result rows (all reads) are compared with the headline table's, each in a "
                         "process of its own (e.g. super2,direct): `cross_layouts` in the line")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the batch-API pipeline leg (N=1)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the files-in, CSV-out leg through exe/cuCLARK (N=1)")
    ap.add_argument("--e2e-threads", type=int, default=12, help="-n of the end-to-end run")
    ap.add_argument("--e2e-reps", type=int, default=3, help="runs of the end-to-end command; the leg reports their median, min and max")
    ap.add_argument("--multi-engine-reads", type=int, default=2_000_000,
                    help="end_to_end.multi_engine runs on the first N reads of the file (0: all of them - 8 more runs of the command at full size)")
    ap.add_argument("--multi-engine-runs", default="",
                    help="comma-separated subset of end_to_end.multi_engine's runs (db_sharded_2, db_sharded_4, db_sharded_8, db_sharded_4_parts_2, "
                         "read_sharded_2, paired_gzip_read_sharded_2); default: all but db_sharded_4")
    ap.add_argument("--no-multi-engine", action="store_true",
                    help="N=1: skip end_to_end.multi_engine (exe/cuCLARK's multi-device modes with 2 / 4 / 8 engines on this GPU, CSVs against the one-engine run)")
    ap.add_argument("--parts", type=int, default=0,
                    help="--mode db: parts the table is cut into (default: one per rank = the reference's mode).  With fewer parts than "
                         "ranks the ranks form N / parts groups that split the reads (2-D layout, DESIGN.md 6)")
    ap.add_argument("--chunks", type=int, default=4, help="--mode db: read chunks per pass (the row exchange of a chunk overlaps the next chunk's kernel)")
    ap.add_argument("--single-part", type=int, default=0,
                    help="N=1, profiling only: load part 0 of this many parts of the table (mic_db_set_part) and time its kernel on all reads - "
                         "the instantiation one rank of a table-sharded run executes; results are partial, the checks that need the whole table are skipped")
    ap.add_argument("--no-parts-proxy", action="store_true",
                    help="N=1: skip \"table_sharded_proxy\" (kernel time of part 0 of 2/4/8 of the table against all reads)")
    ap.add_argument("--no-default-layout", "--no-two-strand", dest="no_default_layout", action="store_true",
                    help="N=1: skip the kernel leg on the two-strand table (`two_strand_table`)")
    ap.add_argument("--allow-variant-lib", action="store_true",
                    help="accept MIC_LIB_PATH (a measuring build of the library, tools/*_sweep.sh); refused otherwise: the line must "
                         "describe the product library")
    ap.add_argument("--pitch-layout", action="store_true",
                    help="leave the reads as the generator writes them (fixed pitch, a 0 behind every read) instead of the packer's format")
    ap.add_argument("--db-leg-timeout", type=float, default=300.0,
                    help="N > 1, read mode: seconds after which the line is printed without the table-sharded extra leg")
    ap.add_argument("--no-db-leg", action="store_true",
                    help="N > 1, read mode: skip the extra table-sharded measurement reported as \"table_sharded\"")
    args = ap.parse_args()
    t_main0 = time.time()
    skipped_legs, cut_off_legs = [], []

    def leg_allowed(name, est_s):
        """extra legs are started only while they are expected to end inside the time budget"""
        if time.time() - t_main0 + est_s <= args.time_budget:
            return True
        skipped_legs.append(name)
        log(f"bench.py: leg '{name}' skipped: {time.time() - t_main0:.0f} s gone, ~{est_s:.0f} s needed, budget {args.time_budget:.0f} s")
        return False

    if os.environ.get("MIC_LIB_PATH") and not args.allow_variant_lib:
        sys.exit("bench.py: MIC_LIB_PATH is set (a measuring build of the library); unset it or pass --allow-variant-lib")
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    n_dev = torch.cuda.device_count()          # (on this image counting devices does not initialise the GPU; the ranks are a fresh child either way)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher: this process has not touched the GPU yet - it starts the N ranks as a
        # fresh child (one process per GPU under torch.distributed.run), relays their output and exits with their code
        sys.exit(launch_ranks(args, n_dev))
    if os.environ.get("MIC_BENCH_WATCHDOG"):
        # a rank that is still running after this many seconds prints every thread's Python stack and exits (tests of the
        # multi-rank paths set it: a stuck rendezvous or collective then names the call instead of running into the test's timeout)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MIC_BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; pass --gpus {world} "
                 f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts its ranks itself)")
    if args.backend == "nccl" and world > 1 and local_rank >= n_dev:
        sys.exit(f"bench.py: rank {rank} (local rank {local_rank}) has no GPU of its own: {n_dev} device(s) visible for {world} ranks. "
                 "RCCL needs one device per rank; --backend gloo lets ranks share a GPU (validation only)")
    local_rank = local_rank % max(n_dev, 1)    # (gloo validation runs: ranks may share a GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # --mode db at N = 1 still goes through the collectives (a degenerate all_to_all, merge loop of length 0, the overflow
    # all-gather): the RCCL path of the table-sharded mode runs on one GPU exactly as the driver launches it on eight
    db_mode = args.mode == "db"
    use_dist = world > 1 or db_mode
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from cuclark_amd import MiClarkDB, _lib, multi
    L = _lib.load()
    import hashlib
    with open(_lib.LIB_PATH, "rb") as f:
        lib_id = {"path": os.path.relpath(_lib.LIB_PATH, ROOT), "sha256_16": hashlib.sha256(f.read()).hexdigest()[:16],
                  "variant": bool(os.environ.get("MIC_LIB_PATH"))}
    w = dict(WORKLOADS[args.workload])
    if args.workload == "paired" and world > 1 and args.mode == "read":
        # BASELINE config 5 names 100 M pairs on the 8-GPU node: 12.5 M pairs per GPU (weak scaling keeps that per-GPU share)
        w["n_reads"] = 12_500_000
        w["name"] = w["name"].replace("10M pairs", "12.5M pairs per GPU")
    if args.reads:
        w["n_reads"] = args.reads
    if args.read_len:
        w["read_len"] = args.read_len
        w["name"] = w["name"].replace("150bp", f"{args.read_len}bp")
    k, T = w["k"], w["n_targets"]
    n_reads, read_len = w["n_reads"], w["read_len"]
    paired = bool(w.get("paired"))
    obj_len = 2 * read_len + 1 if paired else read_len
    t_setup = time.time()

    # ---- synthetic table in the on-disk layout (.sz/.ky/.lb images), in HBM
    spec = _lib.MicSynthSpec(seed=4, htsize=w["htsize"], genome_nt=w["genome_nt"], n_targets=T, n_genomes=w["n_genomes"], k=k,
                             key_bytes=w["key_bytes"], keep_ppm=w.get("keep_ppm", 0), run_len=w.get("run_len", 0),
                             repeat_ppm=w.get("repeat_ppm", 0), mosaic_ppm=w.get("mosaic_ppm", 0))
    cap = int(w["genome_nt"]) + 1024
    d_sizes = torch.empty(w["htsize"], dtype=torch.uint8, device=dev)
    d_keys = torch.empty(cap, dtype=torch.int32 if w["key_bytes"] == 4 else torch.int64, device=dev)
    d_labels = torch.empty(cap, dtype=torch.int16, device=dev)
    n_el = C.c_uint64(0)
    torch.cuda.synchronize()
    rc = L.mic_synth_db_device(C.byref(spec), d_sizes.data_ptr(), d_keys.data_ptr(), d_labels.data_ptr(), cap, C.byref(n_el), None)
    assert rc == 0, f"mic_synth_db_device failed ({rc})"
    n_el = n_el.value
    t_gen = time.time() - t_setup

    # ---- resident slot table (whole table, or this rank's bucket range in db mode)
    row_words = 16
    PIPE_BATCHES = int(os.environ.get("MIC_PIPE_BATCHES", "16"))
    LAYOUTS = {"auto": 0, "direct": 1, "minimizer": 2, "super": 3, "super2": 4}
    layout = 0 if os.environ.get("MIC_LAYOUT") else LAYOUTS[args.layout]      # MIC_LAYOUT (tools/, tests) wins over the flag
    os.environ.setdefault("MIC_SUPER2_MAY_FALL_BACK", "1")
    eng = MiClarkDB(k, T, num_batches=PIPE_BATCHES, device=local_rank, row_words=row_words, layout=layout)
    P = (args.parts or world) if db_mode else 1
    part_i, group_i, n_groups, group_ranks = multi.grid(world, rank, P)
    groups = None
    if db_mode and P < world:          # every rank creates every group, in the same order
        groups = [dist.new_group(list(range(g * P, (g + 1) * P))) for g in range(n_groups)]
    if db_mode and P > 1:
        eng.set_part(part_i, P)        # this rank's part of the table (super-k-mer layouts: a slot range of the resident table)
    if args.single_part > 1 and world == 1 and not db_mode:
        eng.set_part(0, args.single_part)
        args.no_cpu = args.no_pipeline = args.no_e2e = args.no_parts_proxy = args.no_default_layout = True
    t0 = time.time()
    eng.read_device(d_sizes.data_ptr(), w["htsize"], d_keys.data_ptr(), w["key_bytes"], d_labels.data_ptr())
    info = eng.info()
    t_build = time.time() - t0
    build_stages = {a.strip(): float(b) for a, b in (ln.rsplit(":", 1) for ln in L.mic_db_last_build_report().decode().splitlines() if ":" in ln)}

    # ---- reads (packed containers) in HBM; read-sharded ranks draw different reads
    pitch = L.mic_synth_read_pitch(obj_len, k)
    read_seed = 5 + (rank if args.mode == "read" else 0)
    d_rp = torch.empty(n_reads + 1, dtype=torch.int32, device=dev)
    d_cont = torch.zeros(n_reads * pitch + 64, dtype=torch.int16, device=dev)
    d_truth = torch.empty(n_reads * 2, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    def generate_reads(seed):
        """n_reads packed reads of this workload in HBM, IN THE PACKER'S FORMAT: reads back to back, readsPointer[r + 1] = the end of
        read r, no terminator (CuCLARK_hh.hh:1616-1716, the arrays queryBatch receives).  The generator writes every read at a
        fixed pitch with a 0 behind its last part (it works on all reads at once); the compaction below (device, not timed) removes
        pitch and terminators, which the reference's format does not have (the pitch form costs the kernel 2 %:
        --pitch-layout, tools/compact_reads_probe.py)."""
        nonlocal d_rp, d_cont
        rc_ = L.mic_synth_reads_device2(C.byref(spec), seed, n_reads, read_len, int(paired), 0.2, 0.01, 0.001, d_rp.data_ptr(), d_cont.data_ptr(),
                                        d_cont.numel(), d_truth.data_ptr(), None)
        assert rc_ == 0, f"mic_synth_reads_device failed ({rc_})"
        torch.cuda.synchronize()
        if args.pitch_layout:
            return
        rp64 = d_rp.to(torch.int64) & 0xFFFFFFFF
        cu16 = d_cont.view(torch.int16).to(torch.int32) & 0xFFFF
        pos, end = rp64[:-1].clone(), rp64[1:]
        live = torch.ones(n_reads, dtype=torch.bool, device=dev)
        for _ in range(4096):                         # parts of a read: a length slot, then ceil(len / 8) containers
            live &= pos < end
            plen = torch.where(live, cu16[torch.clamp(pos, max=cu16.numel() - 1)].to(torch.int64), torch.zeros_like(pos))
            live &= plen > 0
            if not bool(live.any()):
                break
            pos = torch.where(live, pos + 1 + (plen + 7) // 8, pos)
        used = pos - rp64[:-1]
        rp_c = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
        torch.cumsum(used, 0, out=rp_c[1:])
        total = int(rp_c[-1].item())
        src = torch.repeat_interleave(rp64[:-1] - rp_c[:-1], used, output_size=total) + torch.arange(total, dtype=torch.int64, device=dev)
        compact = torch.zeros(total + 64, dtype=torch.int16, device=dev)          # (the kernel's read-ahead looks past the last read)
        compact[:total] = d_cont[src]
        del src, cu16
        d_cont = compact
        d_rp = rp_c.to(torch.int32)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    generate_reads(read_seed)
    d_res = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)
    n_flagged = [0]
    n_cu_bench = torch.cuda.get_device_properties(dev).multi_processor_count
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream

    def sharded_ops(engine, res_all):
        """device work of a table-sharded pass (cuclark_amd/multi.py: ShardedPass) bound to one engine"""
        staged = args.backend != "nccl"

        def query(first, count, rows):
            engine.query_device(d_rp.data_ptr() + 4 * first, d_cont.data_ptr(), count, res_all.data_ptr() + 32 * first, rows.data_ptr(), sptr)
            n_flagged[0] += engine.resolve_flagged_device(d_rp.data_ptr() + 4 * first, d_cont.data_ptr(), res_all.data_ptr() + 32 * first,
                                                          rows.data_ptr(), sptr)

        def count_dense(ids):
            d_ids = ids.to(torch.int32).to(dev)
            d_counts = torch.zeros((ids.numel(), T), dtype=torch.int32, device=dev)
            engine.count_dense_device(d_rp.data_ptr(), d_cont.data_ptr(), d_ids.data_ptr(), ids.numel(), d_counts.data_ptr(), sptr)
            torch.cuda.synchronize()
            return d_counts.cpu() if staged else d_counts

        def result_from_dense(counts, idx, res):
            d_idx = idx.to(torch.int32).to(dev)
            d_cnt = counts.to(dev).contiguous()
            engine.result_from_dense_device(d_cnt.data_ptr(), d_idx.data_ptr(), idx.numel(), res.data_ptr(), 0, sptr)
            torch.cuda.synchronize()
        return dict(query=query, count_dense=count_dense, result_from_dense=result_from_dense,
                    merge=lambda a, b, out, n: engine.merge_rows_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, sptr),
                    result=lambda rows, res, n: engine.result_from_rows_device(rows.data_ptr(), res.data_ptr(), n, sptr))

    def sharded_pass(engine, res_all, P_, group_index, n_groups_, group, group_rank):
        lo, hi, _ = multi.read_range(n_reads, n_groups_, group_index)          # this group's reads
        return multi.ShardedPass(sharded_ops(engine, res_all), lo, hi - lo, row_words, dev, group, P_, group_rank, chunks=args.chunks,
                                 staged=args.backend != "nccl")

    def sharded_known_answer(sp, truth_all):
        """constructive known answer over the reads of this rank's group, summed over the groups"""
        got = sp.gather().cpu().numpy().view(np.uint32)
        tr = truth_all[sp.first:sp.first + sp.n]
        g = tr[:, 0] > 0
        ok = (tr[g, 1] == 0) | ((got[g, 1] == tr[g, 0]) & (got[g, 2] >= tr[g, 1]))
        v = torch.tensor([float(ok.sum()), float(g.sum()), float((got[~g, 0] == 0).sum()), float((~g).sum()),
                          float(((got[:, 2] == got[:, 4]) & (got[:, 2] > 0)).sum()), float(sp.n)], dtype=torch.float64,
                         device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(v)
        v = (v / sp.P).tolist()                    # every rank of a group reported the group's reads
        return {"genome_reads": int(v[1]), "label_and_count_ok": v[0] / v[1] if v[1] else 1.0,
                "random_reads_no_hit": v[2] / v[3] if v[3] else 1.0, "tie_rate": v[4] / max(v[5], 1.0)}

    sp = None
    if db_mode:
        sp = sharded_pass(eng, d_res, P, group_i, n_groups, groups[group_i] if groups else None, part_i)

    def step():
        n_flagged[0] = 0
        if not db_mode:
            eng.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n_reads, d_res.data_ptr(), 0, sptr)
            # reads the kernel flagged (more than 64 targets) take the dense path INSIDE the step: part of the timed work
            n_flagged[0] = eng.resolve_flagged_device(d_rp.data_ptr(), d_cont.data_ptr(), d_res.data_ptr(), 0, sptr)
            return
        # table-sharded: per chunk local sparse rows -> all_to_all by read range -> merge (sum by target) -> best/second
        sp.step()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if args.steps <= 64:
            pass
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # mean duration of the query kernel alone (HIP events recorded around it on its own stream), sampled
    # outside the timed loop so the event reads do not serialise it
    q_first, q_n = (sp.first, sp.n) if db_mode else (0, n_reads)       # the reads this rank's kernel sees in one pass
    for _ in range(min(args.steps, 5)):
        if db_mode:      # the pass launches the kernel per chunk: time ONE launch over all of this rank's reads instead
            eng.query_device(d_rp.data_ptr() + 4 * q_first, d_cont.data_ptr(), q_n, d_res.data_ptr() + 32 * q_first, 0, sptr)
        else:
            step()
        kernel_ms.append(eng.last_query_ms())
    torch.cuda.synchronize()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    crowd_stats = eng.last_crowd_stats() if not db_mode else None
    ms_per_step = elapsed / args.steps * 1e3
    total_reads = n_reads * (world if args.mode == "read" else 1)
    value = total_reads / (elapsed / args.steps) / 1e6

    # ---- bookkeeping for the roofline: measured k-mers, hit rate, probed-bucket length (product-side kernel)
    flagged = n_flagged[0]
    st = eng.probe_stats_device(d_rp.data_ptr() + 4 * q_first, d_cont.data_ptr(), q_n)
    kern_s = float(np.mean(kernel_ms)) / 1e3
    h = st["hits"] / max(st["probed"], 1)
    lam_q = st["bucket_len_sum"] / max(st["probed"], 1)
    key_b = w["key_bytes"]
    bytes_per_kmer = 8 + key_b * lam_q + 2 * h                      # bucket begin/end + keys of the bucket + label on hit
    # packed read + pointer as the packer emits them and the kernel addresses them (40 B + 4 B for 150 bp): one
    # length slot + ceil(L / 8) containers + the pointer - not the generator's allocated pitch (76 B, mostly never touched)
    in_bytes = 2 * (1 + (obj_len + 7) // 8) + 4
    alg_bytes = st["probed"] * bytes_per_kmer + q_n * (in_bytes + 32)
    achieved = alg_bytes / kern_s / 1e9
    # HBM traffic per launch comes from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, tools/profile_bench.sh); the
    # counters cannot be read from inside this process, so the committed summary of the same workload is used.
    traffic, traffic_src, rdreq, traffic_note = None, None, None, None
    issue = None       # instruction counters of the committed profile of this very instantiation (they do not depend on the box's clock)
    import glob
    kname = {1: "query_kernel<", 2: "query_kernel_m<", 3: "query_kernel_s<", 4: "query_kernel_s<"}[info["layout"]]
    if info["layout"] in (3, 4):      # the instantiation the launcher picks (mic_kernels.hip: mic_launch_query)
        km = (k, info["minimizer_len"]) if (k in (31, 27, 32) and info["minimizer_len"] == 20) else (0, 0)
        parted = info["n_parts"] > 1
        two = info["layout"] == 4
        kname = f"query_kernel_s<{km[0]}, {km[1]}, {'true' if parted else 'false'}, {'true' if two else 'false'}>"
        if not os.environ.get("MIC_S_PER_KMER") and (two or 32 < 2 * k - info["minimizer_len"] <= 48):
            kname = f"query_kernel_r<{km[0]}, {km[1]}, {'true' if two else 'false'}, {'true' if parted else 'false'}>"   # super-k-mer tables are probed per run
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_query_kernel.json")), reverse=True):
        try:
            pj = json.load(open(f))
        except Exception:
            continue
        if not (pj.get("workload") == w["name"] and pj.get("reads_per_launch") == n_reads and pj.get("layout") == info["layout"]):
            continue
        # a committed counter profile is only used for the kernel it was taken on: same instantiation; its rocprofv3 average
        # must agree within 3 % with the HIP-event time of the very run it was taken in (profiles/*_bench_under_rocprof.json),
        # and that run must be within 5 % of the duration measured now (boxes of the pool differ by +-2 %)
        if kname not in pj.get("kernel", ""):
            traffic_note = f"{os.path.basename(f)} was taken on '{pj.get('kernel', '?')[:60]}', this run timed '{kname}'"
            continue
        if issue is None and pj.get("pmc_per_launch", {}).get("SQ_INSTS_VALU"):
            pm, nrl = pj["pmc_per_launch"], float(pj["reads_per_launch"])
            # a wave64 vector instruction holds its SIMD for 4 cycles, a CU has 4 SIMDs: one vector instruction per CU-cycle at
            # most; the CU's ONE scalar unit issues one scalar instruction or branch per cycle (MI355X_MICROARCH.md)
            cu_cycles = n_cu_bench * CLOCK_HZ * kern_s
            issue = {"source": os.path.basename(f), "valu_per_read": round(pm["SQ_INSTS_VALU"] / nrl, 1), "salu_per_read": round(pm.get("SQ_INSTS_SALU", 0) / nrl, 1),
                     "branches_per_read": round(pm.get("SQ_INSTS_BRANCH", 0) / nrl, 1), "lds_per_read": round(pm.get("SQ_INSTS_LDS", 0) / nrl, 1),
                     # (of the CU-cycles at the nominal 2.4 GHz; a little above 1 is possible: v_readlane and friends do not hold the SIMD for 4 cycles)
                     "valu_issue_frac": round(pm["SQ_INSTS_VALU"] / nrl * q_n / cu_cycles, 3),
                     "scalar_issue_frac": round((pm.get("SQ_INSTS_SALU", 0) + pm.get("SQ_INSTS_BRANCH", 0) + pm.get("SQ_INSTS_SMEM", 0)) / nrl * q_n / cu_cycles, 3)}
        ev_ms = pj.get("bench_hip_event_ms") or pj.get("rocprof_avg_ms", 0)
        off_self = abs(pj.get("rocprof_avg_ms", 0) / max(ev_ms, 1e-9) - 1)
        off = abs(ev_ms / (kern_s * 1e3) - 1)
        if off_self > 0.03:
            traffic_note = f"{os.path.basename(f)}: rocprofv3 average {pj.get('rocprof_avg_ms', 0):.3f} ms and the HIP-event time of the same run {ev_ms:.3f} ms disagree by {off_self * 100:.1f} %"
            continue
        if off > 0.05:
            traffic_note = f"{os.path.basename(f)}: the profiled run's kernel took {ev_ms:.3f} ms, {off * 100:.1f} % off the {kern_s * 1e3:.3f} ms measured now"
            continue
        traffic = (pj["fetch_bytes_per_launch"] + pj["write_bytes_per_launch"]) / kern_s / 1e9
        rdreq = pj["pmc_per_launch"].get("TCC_EA0_RDREQ_sum")
        traffic_src = os.path.basename(f)
        traffic_note = None
        break
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                # `bound` and `frac` are the contract's figures (bytes of the REFERENCE's layout against the HBM peak).  What the counters
                # say limits this kernel is instruction issue: `bound_by_counters` and the per-read instruction counts are the progress
                # meter once `frac` saturates (it passes 1.0 on small tables: the kernel moves fewer bytes than the reference's layout)
                "bound_by_counters": (None if issue is None else "valu_issue" if issue["valu_issue_frac"] >= issue["scalar_issue_frac"] else "scalar_issue"),
                "valu_per_read": issue and issue["valu_per_read"], "salu_per_read": issue and issue["salu_per_read"],
                "valu_issue_frac": issue and issue["valu_issue_frac"], "issue": issue,
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": round(traffic, 1) if traffic else None,
                # what the counters saw against the same peak: the kernel moves FEWER bytes than the reference's layout would
                # (~6 k-mers share one 128-byte request), so this is the honest HBM utilisation; `frac` is the contract's figure
                "frac_of_peak_by_measured_traffic": round(traffic / HBM_PEAK_GBS, 4) if traffic else None,
                "traffic_source": traffic_src, "traffic_note": traffic_note,
                "achieved_is": "algorithmic bytes of the reference's layout (SURVEY.md 8d: 8 + key bytes x probed bucket + 2 x hit rate "
                               "per k-mer, + packed read in + result row out) / kernel time; 'traffic' is what the counters saw",
                "kernel": kname, "kernel_ms": round(kern_s * 1e3, 3),
                "algorithmic_bytes_per_kmer": round(bytes_per_kmer, 2), "kmers_per_launch": st["kmers"],
                "probes_per_launch": st["probed"], "hit_rate": round(h, 4), "mean_probed_bucket_len": round(lam_q, 3),
                "probes_per_s_G": round(st["probed"] / kern_s / 1e9, 2),
                # the chip serves ~51 G random HBM requests/s (64 or 128 B alike): how close is the kernel?
                "hbm_requests_per_s_G": round(rdreq / kern_s / 1e9, 2) if rdreq else None,
                "random_request_roof_G": RANDOM_SECTOR_GREQ,
                "frac_of_random_request_roof": round(rdreq / kern_s / 1e9 / RANDOM_SECTOR_GREQ, 4) if rdreq else None}

    # ---- constructive known answer at full size: genome reads must hit their genome's label
    res = d_res.cpu().numpy().view(np.uint32)
    import hashlib as _hl
    results_digest = _hl.sha256(np.ascontiguousarray(res[:, :5]).tobytes()).hexdigest()[:16]      # (sum, best, its count, second, its count) of every read
    truth = d_truth.cpu().numpy().view(np.uint32).reshape(-1, 2)
    gmask = truth[:, 0] > 0
    if args.single_part > 1 and not db_mode:
        known = {"partial_table": f"part 0 of {args.single_part}", "hits": int(res[:, 0].astype(np.int64).sum())}
    elif not db_mode:
        # a genome read with w unmodified windows must report its genome's label with >= w hits (w = 0: no claim)
        ok = (truth[gmask, 1] == 0) | ((res[gmask, 1] == truth[gmask, 0]) & (res[gmask, 2] >= truth[gmask, 1]))
        known = {"genome_reads": int(gmask.sum()), "label_and_count_ok": float(ok.mean()) if gmask.any() else 1.0,
                 "random_reads_no_hit": float((res[~gmask, 0] == 0).mean()) if (~gmask).any() else 1.0,
                 "tie_rate": float(((res[:, 2] == res[:, 4]) & (res[:, 2] > 0)).mean())}
    else:
        # table-sharded: every rank finalised its sub-ranges; gather inside the group in read order and apply the same check
        torch.cuda.synchronize()
        known = sharded_known_answer(sp, truth)

    # ---- CPU baseline (rank 0, N=1): the oracle on this box's host cores, bounded sample; doubles as parity check
    cpu = None
    if rank == 0 and world == 1 and not db_mode and not args.no_cpu:
        from oracle.binding import Oracle
        o = Oracle()
        t0 = time.time()
        h_sizes = d_sizes.cpu().numpy()
        h_keys = d_keys[:n_el].cpu().numpy().view(np.uint32 if key_b == 4 else np.uint64)
        h_labels = d_labels[:n_el].cpu().numpy().view(np.uint16)
        cores_visible = len(os.sched_getaffinity(0))
        cpu_quota = cpu_quota_cpus()               # CPUs of CPU time the cgroup allows (None: no quota)
        # threads follow what the process may USE: more threads than the quota run slower under CFS throttling
        cores = max(1, min(cores_visible, int(np.ceil(cpu_quota)) if cpu_quota else cores_visible))
        # the table is copied once more: one replica per NUMA node, written (first touch) and probed by threads pinned to that
        # node, huge pages requested - the round-1 baseline had every page on the node of the one thread that received the
        # download and was DRAM-bound on that socket
        ndb = o.numa_db(h_sizes, h_keys, h_labels, threads=cores)
        odb = o.db_wrap_arrays(h_sizes, h_keys, h_labels)
        t_copy = time.time() - t0
        ns = min(args.cpu_sample, n_reads)
        rp = d_rp[: ns + 1].cpu().numpy().view(np.uint32)
        ct = d_cont[: int(rp[-1]) + 64].cpu().numpy().view(np.uint16)
        t0 = time.perf_counter()
        ref = ndb.classify_batch(k, rp, ct, T)
        t_cpu = time.perf_counter() - t0
        equal = bool((ref == res[:ns, :5]).all())
        # the plain restatement (two dependent cache misses per k-mer, dense tally) on a slice, for the record
        ns0 = min(ns, 200_000)
        t0 = time.perf_counter()
        ref0 = odb.classify_batch(k, rp[: ns0 + 1], ct, T, threads=cores)
        t_plain = time.perf_counter() - t0
        equal = equal and bool((ref0 == res[:ns0, :5]).all())
        kmers_sample = st["kmers"] * ns / n_reads
        cpu = {"value": round(ns / t_cpu / 1e6, 4), "unit": "Mreads/s", "cores": cores, "kind": "port",
               "cores_visible": cores_visible, "cpu_quota": cpu_quota, "threads_used": cores,
               "sample": f"first {ns} of the {n_reads} reads of the same workload, same table in host RAM, one replica per NUMA node "
                         f"({t_copy:.0f} s download + copy + prefix sums, not timed); {t_cpu:.2f} s wall",
               "objects_per_min": int(ns / t_cpu * 60), "probes_per_s_per_core_M": round(kmers_sample / t_cpu / cores / 1e6, 3),
               "form": "oracle/clark_oracle.c: orc_classify_batch_numa (probe stages pipelined with software prefetch, sparse tally, "
                       "threads pinned to the NUMA node whose table replica they probe)",
               "plain_form_Mreads_s": round(ns0 / t_plain / 1e6, 4),
               "plain_form": "orc_classify_batch (two dependent misses per k-mer, dense tally) on the arrays as downloaded (one node)",
               "parity_with_gpu_on_sample": equal}
        assert equal, "GPU results differ from the CPU oracle on the sample"

    import threading
    emit_lock = threading.Lock()
    emitted = [False]
    legs = {}                  # the extra legs' results as they finish (what the line carries if the time budget cuts the run short)
    current_leg = [None]

    def emit(ts):
        """rank 0's ONE line; ts = the table-sharded extra leg's result (N > 1, read mode), or None.  Printed once whoever calls
        first (the main path or the extra leg's deadline): the lock and the flag keep the one-line contract."""
        with emit_lock:        # (held until the line is out: whoever comes second waits for the print, then returns)
            if emitted[0]:
                return
            emitted[0] = True
            _emit_locked(ts)

    def _emit_locked(ts):
        pipeline, e2e, proxy = legs.get("pipeline"), legs.get("end_to_end"), legs.get("table_sharded_proxy")
        two_strand, cross = legs.get("two_strand_table"), legs.get("cross_layouts")
        reads_label = f"{n_reads / 1e6:g}M" + (" per GPU" if world > 1 and args.mode == "read" else "")
        out = {
            "metric": f"Mreads/sec ({reads_label} x {'2x' if paired else ''}{read_len}bp{' pairs' if paired else ''}, k={k})", "value": round(value, 3), "unit": "Mreads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong" if db_mode else "weak", "vs_baseline": None, "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": w["name"], "reads_per_gpu": n_reads, "read_len": read_len, "k": k,
                       "reads_format": "generator's fixed pitch, a 0 behind every read" if args.pitch_layout else
                                       "the packer's: reads back to back, readsPointer[r + 1] = end of read r (CuCLARK_hh.hh:1616-1716)",
                       "mode": (f"{P} part(s) x {n_groups} read group(s): " + (PART_MODE[info["layout"]] if P > 1 else "table replicated") +
                                " + all_to_all of sparse rows" if db_mode else
                                ("read-sharded, table replicated" if world > 1 else "single GPU, table resident")),
                       "table": {"htsize": info["htsize"], "kmers": info["n_elems"], "slot_class": info["slot_class"],
                                 "layout": {1: "direct: one 64-B slot per on-disk bucket", 2: "minimizer-keyed 128-B slots", 3: "super-k-mer 128-B slots",
                                            4: "super-k-mer 128-B slots, both strands stored (no reverse complement in the query)"}[info["layout"]],
                                 "minimizer_len": info["minimizer_len"], "largest_minimizer_bucket": info["max_chain"],
                                 "hbm_GB": round(info["hbm_bytes"] / 1e9, 2), "overflow_slots": info["n_overflow"],
                                 "max_bucket": info["max_bucket"], "on_disk_equiv_GB": round((info["htsize"] + n_el * (key_b + 2)) / 1e9, 2)},
                       "flagged_reads_dense_path": flagged, "results_sha256_16": results_digest,
                       "crowded": (None if crowd_stats is None else
                                   {"side_table_kmers": info["side_kmers"], "reads_with_crowded_runs": crowd_stats["reads"], "crowded_runs": crowd_stats["runs"],
                                    "reads_to_dense_path_for_lack_of_room": crowd_stats["reads_to_dense_path"],
                                    "what": "reads that met a crowded minimizer (microsatellites: one minimizer in thousands of contexts, its k-mers in a side "
                                            "table) are finished by crowd_finish_kernel behind the query kernel, inside the timed step"}),
                       "setup_s": {"synth_db": round(t_gen, 1), "table_build": round(t_build, 1), "table_build_stages": build_stages},
                       "library": lib_id},
            "roofline": roofline, "cpu_baseline": cpu, "known_answer": known,
        }
        if pipeline is not None:
            out["pipeline"] = pipeline
        if e2e is not None:
            out["end_to_end"] = e2e
        if proxy is not None:
            out["table_sharded_proxy"] = proxy
        if two_strand is not None:
            out["two_strand_table"] = two_strand
        if cross is not None:
            out["cross_layouts"] = cross
        if world == 1 and not db_mode:
            out["skipped_legs"] = list(skipped_legs)
            out["cut_off_legs"] = list(cut_off_legs)
            out["time_budget_s"] = args.time_budget
            out["wall_s"] = round(time.time() - t_main0, 1)
        if ts is not None:
            out["table_sharded"] = ts
        print(json.dumps(out), flush=True)

    # A leg that is still running 90 s past the time budget (a slow box, a stuck child process) does not take the headline with it:
    # rank 0 prints the line with the legs finished so far and leaves.
    def budget_over():
        if emitted[0]:
            return
        if current_leg[0]:
            cut_off_legs.append(current_leg[0])
        log(f"bench.py: time budget exceeded by 90 s in leg '{current_leg[0]}': the line is printed without it")
        emit(None)
        sys.stdout.flush()
        os._exit(0)
    budget_timer = None
    if rank == 0 and world == 1 and not db_mode:
        budget_timer = threading.Timer(max(1.0, args.time_budget + 90.0 - (time.time() - t_main0)), budget_over)
        budget_timer.daemon = True
        budget_timer.start()

    # ---- N = 1: the pipeline through the batch API and the end-to-end run through the CLI (SURVEY.md 8d ii, iii) -------
    if rank == 0 and world == 1 and not db_mode:
        if not args.no_pipeline and leg_allowed("pipeline", 20):
            current_leg[0] = "pipeline"
            try:
                legs["pipeline"] = pipeline_leg(eng, L, d_rp, d_cont, n_reads, res, max(2, min(args.steps, 5)), PIPE_BATCHES)
            except Exception as ex:
                legs["pipeline"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            log("pipeline:", json.dumps(legs["pipeline"]))
        eng.close()
        torch.cuda.empty_cache()

        def kernel_ms_of(e, reps=3):
            """HIP-event time of the query kernel of engine e on the bench's reads (mean of reps after one warm-up)"""
            ms = []
            for i in range(reps + 1):
                e.query_device(d_rp.data_ptr(), d_cont.data_ptr(), n_reads, d_res.data_ptr(), 0, sptr)
                e.resolve_flagged_device(d_rp.data_ptr(), d_cont.data_ptr(), d_res.data_ptr(), 0, sptr)
                if i:
                    ms.append(e.last_query_ms())
            return float(np.mean(ms))
        full_scale = w["genome_nt"] > 1_000_000_000
        if not args.no_parts_proxy and info["layout"] in (3, 4) and leg_allowed("table_sharded_proxy", 50 if full_scale else 10):
            # BASELINE config 4 on one GPU: what ONE rank of an N-GPU table-sharded run does - part 0 of N of the table
            # (mic_db_set_part: a slot range of the resident table), ALL reads.  Per-rank kernel time is what the exchange and
            # the merges are added to; hits_share says the part really answers for ~1/N of the k-mers.
            current_leg[0] = "table_sharded_proxy"
            try:
                proxy = {"what": "HIP-event time of the query kernel of ONE rank of an N-way table-sharded run (part 0 of N of the table, "
                                 "all reads), next to the whole table's", "whole_table_ms": round(kern_s * 1e3, 3), "parts": {}}
                hits_whole = int(res[:, 0].astype(np.int64).sum())
                for n_parts in (2, 8):
                    with MiClarkDB(k, T, device=local_rank, row_words=row_words, layout=layout) as ep:
                        ep.set_part(0, n_parts)
                        t0 = time.time()
                        ep.read_device(d_sizes.data_ptr(), w["htsize"], d_keys.data_ptr(), w["key_bytes"], d_labels.data_ptr())
                        tb = time.time() - t0
                        ms = kernel_ms_of(ep)
                        pi = ep.info()
                        hits = int(d_res[:, 0].to(torch.int64).sum().item())
                    proxy["parts"][str(n_parts)] = {"kernel_ms": round(ms, 3), "vs_whole": round(ms / (kern_s * 1e3), 3),
                                                     "part_hbm_GB": round(pi["hbm_bytes"] / 1e9, 2), "part_build_s": round(tb, 1),
                                                     "hits_share": round(hits / max(hits_whole, 1), 4)}
                    torch.cuda.empty_cache()
            except Exception as ex:
                proxy = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            legs["table_sharded_proxy"] = proxy
            log("table_sharded_proxy:", json.dumps(proxy))

        def other_layout(name, steps=5):
            """The same workload (same seeds: the same table images and reads) on another resident layout, in a process of its OWN: a
            table allocated in this process right behind a freed one of 60-120 GB ran the same kernel 9 % slower (profiles/r05d_*).
            Rows are compared by their digest."""
            import subprocess
            t0 = time.time()
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--layout", name, "--steps", str(steps), "--warmup", "2", "--no-cpu",
                   "--no-pipeline", "--no-e2e", "--no-parts-proxy", "--no-two-strand", "--reads", str(n_reads), "--read-len", str(args.read_len),
                   "--time-budget", "100000"]
            if args.pitch_layout:
                cmd.append("--pitch-layout")
            r = subprocess.run(cmd, capture_output=True, text=True, env={k_: v for k_, v in os.environ.items() if k_ != "MIC_LAYOUT"})
            dl = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]) if r.returncode == 0 else None
            if dl is None:
                return {"error": (r.stderr or r.stdout)[-300:]}
            wanted = {"super2": "both strands", "direct": "direct", "minimizer": "minimizer-keyed", "super": "super-k-mer"}.get(name, "")
            return {"layout": dl["config"]["table"]["layout"],
                    "built_the_layout_asked_for": wanted in dl["config"]["table"]["layout"] and (name != "super" or "both strands" not in dl["config"]["table"]["layout"]),
                    "value": dl["value"], "unit": "Mreads/s", "ms_per_step": dl["ms_per_step"],
                    "kernel_ms": dl["roofline"]["kernel_ms"], "kernel": dl["roofline"]["kernel"], "roofline_frac": dl["roofline"]["frac"],
                    "hbm_GB": dl["config"]["table"]["hbm_GB"], "table_build_s": dl["config"]["setup_s"]["table_build"],
                    "table_build_stages": dl["config"]["setup_s"]["table_build_stages"],
                    "flagged_reads_dense_path": dl["config"]["flagged_reads_dense_path"],
                    "results_equal_headline_table": dl["config"]["results_sha256_16"] == results_digest,
                    "measured_in": f"a process of its own (bench.py --layout {name}, same seeds)", "leg_s": round(time.time() - t0, 1)}
        if info["layout"] == 3 and not os.environ.get("MIC_LAYOUT") and not args.no_default_layout and k >= 24 and \
                leg_allowed("two_strand_table", 70 if full_scale else 25):
            # `value` is quoted on the table the command line builds (one strand).  The two-strand table (twice the memory, 2 s more
            # build, no reverse complement in the query kernel) on the same reads, for whoever has the memory and the reads to repay it
            current_leg[0] = "two_strand_table"
            try:
                legs["two_strand_table"] = other_layout("super2")
            except Exception as ex:
                legs["two_strand_table"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            log("two_strand_table:", json.dumps(legs["two_strand_table"]))
            torch.cuda.empty_cache()
        if args.cross_layouts:
            cross = {}
            for name in [x for x in args.cross_layouts.split(",") if x]:
                if not leg_allowed("cross_layouts." + name, 90 if full_scale else 25):
                    continue
                current_leg[0] = "cross_layouts." + name
                try:
                    cross[name] = other_layout(name, steps=2)
                except Exception as ex:
                    cross[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
                log("cross_layouts", name + ":", json.dumps(cross[name]))
            cross["all_equal"] = all(v.get("results_equal_headline_table") is True for v in cross.values()) if cross else None
            legs["cross_layouts"] = cross
        if not args.no_e2e and leg_allowed("end_to_end", 150 if full_scale else 40):
            current_leg[0] = "end_to_end"
            try:
                del d_res, d_cont, d_rp
                images = [d_sizes, d_keys, d_labels]
                del d_sizes, d_keys, d_labels
                torch.cuda.empty_cache()
                legs["end_to_end"] = end_to_end_leg(L, spec, w, images, n_el, n_reads, read_len, res, truth, args.e2e_threads, paired=paired,
                                                    reps=args.e2e_reps, multi=not args.no_multi_engine, multi_reads=args.multi_engine_reads,
                                                    multi_only=[x for x in args.multi_engine_runs.split(",") if x] or None,
                                                    leg_allowed=leg_allowed, partial=legs, current_leg=current_leg)
            except Exception as ex:
                legs["end_to_end"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            log("end_to_end:", json.dumps(legs["end_to_end"]))
        current_leg[0] = None
        if budget_timer:
            budget_timer.cancel()

    # ---- N > 1, read mode: the reference's own multi-GPU layout as a second, separate measurement --------------------
    # (BASELINE.json configs[3]): the table is re-built as this rank's bucket range, every rank probes the SAME reads,
    # sparse rows are exchanged with all_to_all by read range, merged and finalised.  Fixed total work: "strong".
    table_sharded = None
    if world > 1 and not db_mode and not args.no_db_leg:
        def sharded_leg(P2):
            """the table cut into P2 parts, world // P2 groups of ranks splitting the SAME 10 M reads (P2 = world: the reference's mode)"""
            part2, group2, n_groups2, _ = multi.grid(world, rank, P2)
            grps = [dist.new_group(list(range(g * P2, (g + 1) * P2))) for g in range(n_groups2)] if P2 < world else None
            eng2 = MiClarkDB(k, T, device=local_rank, row_words=row_words, layout=layout)
            try:
                if P2 > 1:
                    eng2.set_part(part2, P2)
                t0 = time.time()
                eng2.read_device(d_sizes.data_ptr(), w["htsize"], d_keys.data_ptr(), w["key_bytes"], d_labels.data_ptr())
                t_build2 = time.time() - t0
                info2 = eng2.info()
                r_res = torch.zeros((n_reads, 8), dtype=torch.int32, device=dev)
                sp2 = sharded_pass(eng2, r_res, P2, group2, n_groups2, grps[group2] if grps else None, part2)
                steps2 = max(1, min(args.steps, 5))
                for _ in range(min(args.warmup, 2)):
                    sp2.step()
                barrier()
                t0 = time.perf_counter()
                for _ in range(steps2):
                    sp2.step()
                torch.cuda.synchronize()
                barrier()
                el2 = time.perf_counter() - t0
                t = torch.tensor([el2], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el2 = float(t.item())
                truth2 = d_truth.cpu().numpy().view(np.uint32).reshape(-1, 2)
                ka2 = sharded_known_answer(sp2, truth2)
                eng2.query_device(d_rp.data_ptr() + 4 * sp2.first, d_cont.data_ptr(), sp2.n, r_res.data_ptr() + 32 * sp2.first, 0, sptr)
                k_ms = eng2.last_query_ms()
                return {"value": round(n_reads / (el2 / steps2) / 1e6, 3), "unit": "Mreads/s", "scaling": "strong",
                        "steps": steps2, "ms_per_step": round(el2 / steps2 * 1e3, 3), "parts": P2, "read_groups": n_groups2,
                        "mode": (PART_MODE[info2["layout"]] if P2 > 1 else "table replicated") + " + all_to_all of sparse rows + merge", "reads_total": n_reads,
                        "kernel_ms_this_rank_all_its_reads": round(k_ms, 3), "reads_this_rank": sp2.n,
                        "shard_hbm_GB": round(info2["hbm_bytes"] / 1e9, 2), "shard_build_s": round(t_build2, 1), "chunks": len(sp2.per),
                        "exchange_MB_per_rank": round(sp2.n * row_words * 4 * (P2 - 1) / P2 / 1e6, 1), "known_answer": ka2}
            finally:
                eng2.close()
                torch.cuda.empty_cache()
        # The headline line must not depend on this leg - not on its failing (the except below) and not on its HANGING: the leg
        # is the one place where ranks meet in collectives of sub-groups; if it has not finished in time, rank 0 prints the line
        # without it and every rank leaves (no barrier: the others may be the ones that are stuck).
        import threading
        leg_done = threading.Event()

        def give_up():
            # (the main path may be finishing the leg this very moment: emit() prints once, whoever gets there first; a rank that
            # gives up leaves at once - no barrier, no destroy_process_group: the others may be the ones that are stuck.  Exit code 0
            # on every rank so that the launcher relays rank 0's line; the line's "error" field is what says the leg is missing)
            if leg_done.is_set():
                return
            if rank == 0:
                emit({"error": f"the table-sharded leg did not finish within {args.db_leg_timeout} s; the line stands without it"})
            sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(args.db_leg_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            eng.close()
            del d_res
            torch.cuda.empty_cache()
            d_rp = torch.empty(n_reads + 1, dtype=torch.int32, device=dev)
            d_cont = torch.zeros(n_reads * pitch + 64, dtype=torch.int16, device=dev)
            generate_reads(5)                          # the same reads on every rank
            table_sharded = sharded_leg(world)
            if world >= 4:      # the 2-D layout for a table that needs two GPUs: 2 parts x world / 2 read groups
                table_sharded["two_parts_2d"] = sharded_leg(2)
        except Exception as ex:   # the headline line must not depend on this leg
            table_sharded = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        finally:
            leg_done.set()
            watchdog.cancel()

    if rank == 0:
        emit(table_sharded)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

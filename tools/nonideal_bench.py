#!/usr/bin/env python3
"""Query-kernel speed on databases that are NOT ideal for the super-k-mer layout.

The headline workload's genomes are uniformly random: every k-mer of a genome is in the database, so super-k-mers are
whole (6.5 k-mers per entry) and minimizers do not recur.  Real target sets differ:
  * homologous targets: RemoveCommon (HashTableStorage_hh.hh:241-292) drops every k-mer shared between targets, which
    cuts super-k-mers into pieces at the boundaries of the shared stretches;
  * tandem repeats / low complexity: the same minimizer recurs, its slot chain grows;
  * cuCLARK-l's sampled database (CuCLARK_hh.hh:694-895): non-overlapping k-blocks, every gap-th one - no two adjacent
    k-mers, one k-mer per entry.
Each case: target FASTA files written to a scratch directory -> mic_db_build (the product's GPU builder, the
reference's rules) -> engine load in every layout -> reads sampled from the targets (1 % substitutions, 20 % random
reads) packed by the device ingest -> query kernel timed with HIP events.

    python tools/nonideal_bench.py [--gnt 1.0] [--reads 4000000] [--cases ideal,homology10,homology20_repeats,light] > out.json
"""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ACGT = np.frombuffer(b"ACGT", np.uint8)


def make_genomes(rng, n_genomes, length, homology, repeats, n_ancestors=16, seg=2000, divergence=0.02):
    """n_genomes sequences of `length` nt.  A fraction `homology` of each genome's 2-kb segments is a copy of the
    corresponding segment of one of a few ancestor sequences with `divergence` substitutions (so related targets share
    exact k-mers in patches); a fraction `repeats` of the length is tandem repeats of 2..50-nt units."""
    anc = rng.integers(0, 4, (n_ancestors, length), dtype=np.uint8) if homology > 0 else None
    out = []
    n_seg = length // seg
    for g in range(n_genomes):
        x = rng.integers(0, 4, length, dtype=np.uint8)
        if homology > 0:
            pick = np.nonzero(rng.random(n_seg) < homology)[0]
            a = rng.integers(0, n_ancestors, pick.size)
            for s_, a_ in zip(pick, a):
                piece = anc[a_, s_ * seg:(s_ + 1) * seg].copy()
                mut = rng.random(seg) < divergence
                piece[mut] = (piece[mut] + rng.integers(1, 4, int(mut.sum()))) & 3
                x[s_ * seg:(s_ + 1) * seg] = piece
        if repeats > 0:
            done = 0
            while done < repeats * length:
                unit = rng.integers(0, 4, int(rng.integers(2, 51)), dtype=np.uint8)
                copies = int(rng.integers(10, 100))
                rep = np.tile(unit, copies)
                p = int(rng.integers(0, length - rep.size))
                x[p:p + rep.size] = rep
                done += rep.size
        out.append(ACGT[x])
    return out


def write_targets(tmp, genomes):
    files = []
    for g, seq in enumerate(genomes):
        fn = os.path.join(tmp, f"g{g:04d}.fa")
        with open(fn, "wb") as f:
            f.write(b">g%d\n" % g)
            f.write(seq.tobytes())
            f.write(b"\n")
        files.append(fn)
    return files


def sample_reads_fasta(rng, genomes, n_reads, read_len, random_frac=0.2, sub_rate=0.01):
    """FASTA text of reads sampled from the targets (either strand), one record per read, fixed record size."""
    L = read_len
    rec = np.empty((n_reads, 12 + L + 1), np.uint8)
    rec[:, 0] = ord(">")
    rec[:, 1] = ord("r")
    ids = np.arange(n_reads)
    for d in range(9):
        rec[:, 10 - d] = ord("0") + (ids // 10 ** d) % 10
    rec[:, 11] = 10
    rec[:, 12 + L] = 10
    glen = genomes[0].size
    allg = np.stack(genomes)                                    # [G, len] ASCII
    comp = np.zeros(256, np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    CH = 250_000
    for lo in range(0, n_reads, CH):
        hi = min(n_reads, lo + CH)
        n = hi - lo
        g = rng.integers(0, len(genomes), n)
        p = rng.integers(0, glen - L, n)
        seq = allg[g[:, None], p[:, None] + np.arange(L)[None, :]]
        rev = rng.random(n) < 0.5
        seq[rev] = comp[seq[rev][:, ::-1]]
        sub = rng.random((n, L)) < sub_rate
        seq[sub] = ACGT[rng.integers(0, 4, int(sub.sum()))]
        rnd = rng.random(n) < random_frac
        seq[rnd] = ACGT[rng.integers(0, 4, (int(rnd.sum()), L))]
        rec[lo:hi, 12:12 + L] = seq
    return rec.reshape(-1)


def run_case(name, args, L, rng):
    from cuclark_amd import MiClarkDB, host, _lib
    k = 27 if name == "light" else 31
    spec = dict(ideal=(0.0, 0.0), homology10=(0.10, 0.0), homology20_repeats=(0.20, 0.02), light=(0.0, 0.0), repeats5=(0.0, 0.05))[name]
    n_genomes = args.genomes
    glen = int(args.gnt * 1e9) // n_genomes
    tmp = tempfile.mkdtemp(prefix="mic_nonideal_", dir=os.environ.get("MIC_BENCH_TMP", "/tmp"))
    out = {"case": name, "k": k, "genomes": n_genomes, "genome_nt": glen, "homology": spec[0], "tandem_repeat_fraction": spec[1]}
    try:
        t0 = time.time()
        genomes = make_genomes(rng, n_genomes, glen, spec[0], spec[1])
        files = write_targets(tmp, genomes)
        out["gen_s"] = round(time.time() - t0, 1)
        htsize = 57777779 if name == "light" else args.htsize
        prefix = os.path.join(tmp, "db")
        t0 = time.time()
        n_kmers = host.build_db(files, list(range(n_genomes)), k, htsize, prefix, threads=16, light_gap=4 if name == "light" else 0)
        out["db_build_s"] = round(time.time() - t0, 1)
        out["kmers_in_db"] = n_kmers
        out["kmers_in_targets"] = n_genomes * (glen - k + 1)
        text = sample_reads_fasta(rng, genomes, args.reads, 150)
        del genomes
        dev = torch.device("cuda:0")
        names = [f"T{g}" for g in range(n_genomes)]
        layouts = {}
        for layout_name, layout in (("auto", 0), ("super2", 4), ("super", 3), ("minimizer", 2), ("direct", 1)):
            if layout_name not in args.layouts.split(","):
                continue
            with MiClarkDB(k, n_genomes, layout=layout) as e:
                t0 = time.time()
                e.read(prefix)
                info = e.info()
                t_load = time.time() - t0
                # pack on the device (mic_ingest_*), collect the packed batches, time the kernel on all reads at once
                slot = 64 << 20
                e.ingest_alloc(1, slot, names, want_results=False)
                rps, cts, base = [], [], 0
                recb = 12 + 150 + 1
                per = (slot // recb) - 1
                for lo in range(0, args.reads, per):
                    hi = min(args.reads, lo + per)
                    r = e.ingest_classify(0, text[lo * recb:hi * recb].tobytes())
                    assert r["status"] == 0, r
                    rp, ct = e.ingest_fetch_packed(0)
                    rps.append(rp[:-1].astype(np.int64) + base)
                    cts.append(ct)
                    base += ct.size
                e.ingest_free()
                rp = np.concatenate(rps + [np.array([base], np.int64)]).astype(np.uint32)
                ct = np.concatenate(cts + [np.zeros(64, np.uint16)])
                d_rp = torch.from_numpy(rp.view(np.int32)).to(dev)
                d_ct = torch.from_numpy(ct.view(np.int16)).to(dev)
                d_res = torch.zeros((args.reads, 8), dtype=torch.int32, device=dev)
                ms = []
                for it in range(6):
                    e.query_device(d_rp.data_ptr(), d_ct.data_ptr(), args.reads, d_res.data_ptr())
                    ms.append(e.last_query_ms())
                crowd = e.last_crowd_stats()
                flagged = e.resolve_flagged_device(d_rp.data_ptr(), d_ct.data_ptr(), d_res.data_ptr())
                st = e.probe_stats_device(d_rp.data_ptr(), d_ct.data_ptr(), args.reads)
                kern = float(np.median(ms[1:])) / 1e3
                h = st["hits"] / max(st["probed"], 1)
                lam = st["bucket_len_sum"] / max(st["probed"], 1)
                bpk = 8 + info["key_bytes"] * lam + 2 * h
                alg = st["probed"] * bpk + args.reads * (2 * ct.size / args.reads + 4 + 32)
                res = d_res.cpu().numpy().view(np.uint32)
                layouts[layout_name] = {
                    "Mreads_s": round(args.reads / kern / 1e6, 1), "kernel_ms": round(kern * 1e3, 3), "Gkmers_s": round(st["kmers"] / kern / 1e9, 1),
                    "roofline_frac": round(alg / kern / 8e12, 4), "hit_rate": round(h, 4), "mean_probed_bucket": round(lam, 3),
                    "hbm_GB": round(info["hbm_bytes"] / 1e9, 2), "slots": info["n_slots"], "continuation_slots": info["n_overflow"],
                    "continuation_slot_rate": round(info["n_overflow"] / max(info["n_slots"], 1), 5),
                    "entries": info["n_entries"], "kmers_per_entry": round(info["n_elems"] / max(info["n_entries"], 1), 2),
                    "fullest_chain_entries": info["max_chain"], "mean_continuation_slots_before_a_kmer": info["reserved"] / 1e6, "reads_through_dense_path": flagged, "load_s": round(t_load, 1),
                    "classified": float((res[:, 0] > 0).mean()),
                    "side_table_kmers": info["side_kmers"], "reads_with_crowded_runs": crowd["reads"], "crowded_runs": crowd["runs"],
                    "kernel_ms_runs": [round(x, 3) for x in ms]}
                layouts[layout_name]["layout_built"] = {1: "direct", 2: "minimizer", 3: "super", 4: "super2"}[info["layout"]]
                if layout_name == "auto":
                    ref = res[:, :5].copy()
                else:
                    layouts[layout_name]["equal_to_auto_layout"] = bool((res[:, :5] == ref).all())
        out["layouts"] = layouts
        if "direct" in layouts and "auto" in layouts:
            out["auto_over_direct"] = round(layouts["auto"]["Mreads_s"] / layouts["direct"]["Mreads_s"], 2)
        if "auto" in layouts:
            out["auto_over_best"] = round(layouts["auto"]["Mreads_s"] / max(l["Mreads_s"] for l in layouts.values()), 3)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gnt", type=float, default=1.0, help="total target nucleotides, in 1e9")
    ap.add_argument("--genomes", type=int, default=256)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--htsize", type=int, default=268435399)
    ap.add_argument("--cases", default="ideal,homology10,homology20_repeats,repeats5,light")
    ap.add_argument("--layouts", default="auto,super2,super,minimizer,direct", help="'auto' first: the others are compared with it")
    args = ap.parse_args()
    from cuclark_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(17)
    results = []
    for name in args.cases.split(","):
        r = run_case(name, args, L, rng)
        print(json.dumps(r), file=sys.stderr, flush=True)
        results.append(r)
    print(json.dumps({"what": "query kernel on non-ideal databases (tools/nonideal_bench.py)", "reads": args.reads, "read_len": 150, "cases": results}, indent=1))


if __name__ == "__main__":
    main()

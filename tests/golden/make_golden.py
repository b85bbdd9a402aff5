#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own CPU hash table.

Run in the build container only (needs /root/reference and oracle/_ref built by oracle/Makefile):

    python tests/golden/make_golden.py            # light-HTSIZE fixtures (seconds)
    python tests/golden/make_golden.py --full     # + the HTSIZE=1610612741 fixture (~4 min, ~26 GB RAM)

What is produced (all small, committed):
  targets/*.fa, targets.tsv      seeded toy target genomes (shared segments, N, lower case, wrapped lines)
  db_<name>.npz                  the database the reference WROTE for those targets
                                 (EHashtable::addElement/SortAllHashTable/RemoveCommon/Write), stored as
                                 .ky/.lb bytes + the non-zero entries of .sz (it is >99.99 % zeros)
  queries_<name>.npz             forward k-mers + the reference's queryElement() answer (found, label)
                                 [+ answers under sampling factor 3 for the k=27 database]
  reads_*.fa / reads_*.fq        read sets covering the edge cases of SURVEY.md §8c
  expected_*.csv                 result CSVs of the CPU restatement (oracle/clark_oracle.c) for those reads;
                                 the restatement's per-k-mer answers are pinned by queries_<name>.npz

Fixtures are data only: no reference source text is stored.
"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_LIGHT = os.path.join(ROOT, "oracle", "_ref", "ref_table_light")
REF_FULL = os.path.join(ROOT, "oracle", "_ref", "ref_table_full")
sys.path.insert(0, os.path.join(ROOT, "tests"))

CODE = {"A": 3, "C": 2, "G": 1, "T": 0, "U": 0}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def kmer_value(s):
    v = 0
    for ch in s.upper():
        v = (v << 2) | CODE[ch]
    return v


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s.upper().replace("U", "T")))


def wrap(seq, width):
    return "\n".join(seq[i:i + width] for i in range(0, len(seq), width))


def make_targets(rng):
    """6 labels, 8 FASTA files: two labels have two genomes each; a 150-nt segment is shared by three
    labels (removed by RemoveCommon), a 120-nt segment is shared by the two genomes of ONE label (kept)."""
    tdir = os.path.join(HERE, "targets")
    os.makedirs(tdir, exist_ok=True)
    shared_all = "".join(rng.choice(list("ACGT"), 150))
    shared_same = "".join(rng.choice(list("ACGT"), 120))
    files = []
    genomes = {}
    spec = [("T_alpha", 1), ("T_beta", 2), ("T_gamma", 1), ("T_delta", 2), ("T_epsilon", 1), ("S6", 1)]
    idx = 0
    for label, n_genomes in spec:
        for g in range(n_genomes):
            L = int(rng.integers(2200, 3400))
            seq = list(rng.choice(list("ACGT"), L))
            if label in ("T_alpha", "T_beta", "T_gamma") and g == 0:
                p = int(rng.integers(100, L - 400))
                seq[p:p + 150] = list(shared_all)
            if label == "T_delta":
                p = int(rng.integers(100, L - 400))
                seq[p:p + 120] = list(shared_same)
            # a few N and lower-case stretches
            for _ in range(3):
                p = int(rng.integers(0, L - 10))
                seq[p] = "N"
            p = int(rng.integers(0, L - 60))
            seq[p:p + 40] = [c.lower() for c in seq[p:p + 40]]
            seq = "".join(seq)
            name = f"genome_{idx}.fa"
            half = L // 2
            with open(os.path.join(tdir, name), "w") as f:
                f.write(f">rec{idx}a some description\n{wrap(seq[:half], 60)}\n")
                f.write(f">rec{idx}b\n{wrap(seq[half:], 70)}\n")
            files.append((name, label))
            genomes.setdefault(label, []).append(seq)
            idx += 1
    with open(os.path.join(HERE, "targets.tsv"), "w") as f:
        for name, label in files:
            f.write(f"targets/{name}\t{label}\n")
    return files, genomes


def run(cmd, **kw):
    r = subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)
    return r.stdout


def build_db(binary, k, key_bytes, files, tmp, name, gap=0):
    tsv = os.path.join(tmp, f"{name}.tsv")
    with open(tsv, "w") as f:
        for fn, label in files:
            f.write(f"{os.path.join(HERE, 'targets', fn)}\t{label}\n")
    prefix = os.path.join(tmp, name)
    n = int(run([binary, "build", str(k), str(key_bytes), prefix, tsv, "0", str(gap)]).strip().splitlines()[-1])
    sz = np.fromfile(prefix + ".sz", dtype=np.uint8)
    ky = np.fromfile(prefix + ".ky", dtype={2: np.uint16, 4: np.uint32, 8: np.uint64}[key_bytes])
    lb = np.fromfile(prefix + ".lb", dtype=np.uint16)
    assert ky.size == n and lb.size == n and int(sz.sum(dtype=np.uint64)) == n
    nz = np.flatnonzero(sz)
    np.savez_compressed(os.path.join(HERE, f"db_{name}.npz"), htsize=np.uint64(sz.size), k=np.int32(k),
                        key_bytes=np.int32(key_bytes), sz_idx=nz.astype(np.uint64), sz_val=sz[nz], ky=ky, lb=lb)
    return prefix, n


def make_queries(rng, genomes, k, n_present=1500, n_absent=700):
    qs = []
    labels = list(genomes)
    for _ in range(n_present):
        g = genomes[labels[int(rng.integers(len(labels)))]]
        s = g[int(rng.integers(len(g)))]
        p = int(rng.integers(0, len(s) - k))
        w = s[p:p + k].upper()
        if "N" in w:
            continue
        if rng.random() < 0.5:
            w = revcomp(w)
        qs.append(kmer_value(w))
    for _ in range(n_absent):
        qs.append(kmer_value("".join(rng.choice(list("ACGT"), k))))
    # palindromes (k even) and extreme values
    if k % 2 == 0:
        for _ in range(20):
            h = "".join(rng.choice(list("ACGT"), k // 2))
            qs.append(kmer_value(h + revcomp(h)))
    qs += [0, (1 << (2 * k)) - 1 if k < 32 else (1 << 64) - 1, kmer_value("A" * k), kmer_value(("ACGT" * 8)[:k])]
    return np.array(qs, dtype=np.uint64)


def ref_query(binary, k, key_bytes, prefix, kmers, tmp, sampling=1, use_mmap=0):
    qf = os.path.join(tmp, "q.txt")
    with open(qf, "w") as f:
        f.write("\n".join(str(int(v)) for v in kmers) + "\n")
    out = run([binary, "query", str(k), str(key_bytes), prefix, qf, str(sampling), str(use_mmap)])
    found = np.zeros(kmers.size, np.uint8)
    label = np.zeros(kmers.size, np.uint16)
    lines = out.strip().splitlines()
    assert len(lines) == kmers.size
    for i, line in enumerate(lines):
        v, f_, l_ = line.split()
        assert int(v) == int(kmers[i])
        found[i] = int(f_)
        label[i] = int(l_)
    return found, label


def make_reads(rng, genomes, k):
    labels = list(genomes)

    def sample(L, sub=0.01):
        g = genomes[labels[int(rng.integers(len(labels)))]]
        s = g[int(rng.integers(len(g)))]
        p = int(rng.integers(0, len(s) - L))
        w = list(s[p:p + L])
        for i in range(L):
            if rng.random() < sub:
                w[i] = "ACGT"[int(rng.integers(4))]
        w = "".join(w)
        return revcomp(w.replace("N", "A").replace("n", "a")) if rng.random() < 0.5 else w

    fa = []
    for i in range(120):
        fa.append((f"read{i}", sample(int(rng.integers(90, 160)))))
    fa.append(("short_lt_k some comment", "ACGTACGTAC"))                     # len < k  -> gamma "-0"
    fa.append(("len_k_minus_1", "ACGT" * 8))                                # placeholder, trimmed below
    fa.append(("exact_k", sample(k, 0)))
    fa.append(("with_N_split", sample(60, 0) + "N" + sample(20, 0) + "NN" + sample(45, 0)))
    fa.append(("lower_and_U", sample(100, 0).lower().replace("t", "u")))
    fa.append(("a_very_long_read_name_that_exceeds_the_thirty_nine_character_limit_of_cuclark", sample(100)))
    fa.append(("random_nohit", "".join(rng.choice(list("ACGT"), 140))))
    fa.append(("iupac_R\tcomment", sample(50, 0) + "R" + sample(50, 0)))
    g0, g1 = genomes[labels[0]][0], genomes[labels[1]][0]
    fa.append(("two_targets_tie", g0[300:300 + k + 9].upper().replace("N", "A") + "N" + g1[500:500 + k + 9].upper().replace("N", "A")))
    fa.append(("three_way", g0[700:760].upper() + "N" + g1[900:950].upper() + "N" + genomes[labels[3]][0][400:470].upper()))
    fa.append(("long_multi_line", genomes[labels[4]][0][100:1300]))
    fa = [(n, (s[:k - 1] if n == "len_k_minus_1" else s)) for n, s in fa]
    with open(os.path.join(HERE, f"reads_k{k}.fa"), "w") as f:
        for i, (n, s) in enumerate(fa):
            width = 60 if (i % 3 == 0 or len(s) > 400) else 10 ** 9
            f.write(f">{n}\n{wrap(s, width)}\n")
    with open(os.path.join(HERE, f"reads_k{k}.fq"), "w") as f:
        for i in range(80):
            s = sample(int(rng.integers(70, 151)))
            if i % 17 == 0:
                s = s[:40] + "N" + s[41:]
            q = "".join(chr(33 + int(x)) for x in rng.integers(0, 41, len(s)))
            if i % 9 == 0:
                q = "@" + q[1:]          # quality line starting with '@'
            f.write(f"@fq{i}/1 extra\n{s}\n+\n{q}\n")
    # paired-end files (FASTQ), merged as seq1 + 'N' + seq2 by the caller
    for mate in (1, 2):
        with open(os.path.join(HERE, f"pairs_k{k}_{mate}.fq"), "w") as f:
            prng = np.random.default_rng(77 + mate)
            for i in range(40):
                s = sample(int(prng.integers(60, 120)))
                q = "I" * len(s)
                f.write(f"@pair{i}/{mate}\n{s}\n+\n{q}\n")


# light databases as cuCLARK-l builds them (CuCLARK_hh.hh:694-895); gap 4 is its default
LIGHT_GAP_CONFIGS = [("lightgap4_k27_u32", 27, 4, 4), ("lightgap5_k31_u64", 31, 8, 5), ("lightgap1_k20_u16", 20, 2, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also build the HTSIZE=1610612741 fixture")
    ap.add_argument("--tmp", default="/tmp/mic_golden")
    ap.add_argument("--only-light-gap", action="store_true",
                    help="only (re)build the cuCLARK-l style databases (non-overlapping k-blocks, every gap-th) from the "
                         "target files already in tests/golden/targets")
    args = ap.parse_args()
    os.makedirs(args.tmp, exist_ok=True)
    if args.only_light_gap:
        import golden_util as gu
        files = [(os.path.basename(fn), label) for fn, label in gu.target_files_and_labels()]
        for name, k, kb, gap in LIGHT_GAP_CONFIGS:
            _, n = build_db(REF_LIGHT, k, kb, files, args.tmp, name, gap=gap)
            print(f"{name}: {n} elements")
        return
    rng = np.random.default_rng(20241003)
    files, genomes = make_targets(rng)

    configs = [("light_k27_u32", REF_LIGHT, 27, 4), ("light_k31_u64", REF_LIGHT, 31, 8),
               ("light_k20_u16", REF_LIGHT, 20, 2), ("light_k32_u64", REF_LIGHT, 32, 8)]
    if args.full:
        configs.append(("full_k31_u32", REF_FULL, 31, 4))
    for name, k, kb, gap in LIGHT_GAP_CONFIGS:
        build_db(REF_LIGHT, k, kb, files, args.tmp, name, gap=gap)
    for name, binary, k, kb in configs:
        prefix, n = build_db(binary, k, kb, files, args.tmp, name)
        kmers = make_queries(np.random.default_rng(1000 + k), genomes, k)
        found, label = ref_query(binary, k, kb, prefix, kmers, args.tmp)
        extra = {}
        if name == "light_k27_u32":
            f3, l3 = ref_query(binary, k, kb, prefix, kmers, args.tmp, sampling=3, use_mmap=1)
            extra = dict(found_s3=f3, label_s3=l3)
        np.savez_compressed(os.path.join(HERE, f"queries_{name}.npz"), kmers=kmers, found=found, label=label, **extra)
        print(f"{name}: {n} elements, {int(found.sum())}/{kmers.size} queries found")

    for k in (27, 31):
        make_reads(np.random.default_rng(500 + k), genomes, k)

    # expected CSVs from the CPU restatement (pinned per k-mer by the queries above)
    import golden_util as gu
    gu.write_expected_csvs()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for the two host-side pieces of the reference that compile here next to its hash table and that
nothing pinned before round 3: the paired-end merge (mergePairedFiles, file.cc:205-268) and the k-mer codec
(getKmers / getReverse, kmersConversion.cc:39-68).  Produced by the REFERENCE's own code through
oracle/_ref/ref_table_light (oracle/ref_table_driver.cc: verbs `merge` and `codec`).

Run in the build container only (needs /root/reference and `make -C oracle`):   python tests/golden/make_golden_pairs.py

  pairs_k27_merged.fa, pairs_k31_merged.fa   what the reference writes for the committed pair files
  pairs_edge.json    small pairs of files covering its branches and exits: {name, f1, f2, rc, merged | null, stderr}
  codec_vectors.json {k, seq, fwd, rev} for seeded k-mers, k = 2 .. 32, mixed case
Data only: inputs and the reference's outputs.
"""
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_table_light")


def ref_merge(f1, f2):
    with tempfile.TemporaryDirectory() as d:
        p1, p2, out = (os.path.join(d, n) for n in ("a.fq", "b.fq", "m.fa"))
        open(p1, "wb").write(f1)
        open(p2, "wb").write(f2)
        r = subprocess.run([REF, "merge", p1, p2, out], capture_output=True)
        merged = open(out, "rb").read() if r.returncode == 0 and os.path.exists(out) else None
        return r.returncode, merged, r.stderr.decode(errors="replace")


def rec(i, seq, name=None, qual=None):
    name = name if name is not None else f"r{i}"
    return f"@{name}\n{seq}\n+\n{qual if qual is not None else 'I' * len(seq)}\n"


def edge_cases(rng):
    nt = lambda n: "".join(rng.choice(list("ACGT"), n))
    a = [nt(int(rng.integers(30, 60))) for _ in range(6)]
    b = [nt(int(rng.integers(30, 60))) for _ in range(6)]
    cases = []
    add = lambda name, f1, f2: cases.append((name, f1.encode(), f2.encode()))
    add("plain", "".join(rec(i, s) for i, s in enumerate(a)), "".join(rec(i, s) for i, s in enumerate(b)))
    add("mate_suffixes_and_comments",
        "".join(rec(i, s, f"read{i}/1 len={len(s)}") for i, s in enumerate(a)),
        "".join(rec(i, s, f"read{i}/2\tmate") for i, s in enumerate(b)))
    add("at_sign_inside_id", rec(0, a[0], "x@y z") + rec(1, a[1], "@double"), rec(0, b[0], "x@w") + rec(1, b[1], "@double/2"))
    add("crlf", "".join(rec(i, s) for i, s in enumerate(a[:3])).replace("\n", "\r\n"),
        "".join(rec(i, s) for i, s in enumerate(b[:3])).replace("\n", "\r\n"))
    add("second_file_shorter", "".join(rec(i, s) for i, s in enumerate(a)), "".join(rec(i, s) for i, s in enumerate(b[:4])))
    add("blank_lines_between_records", "\n".join(rec(i, s) for i, s in enumerate(a[:3])), "\n".join(rec(i, s) for i, s in enumerate(b[:3])))
    add("quality_line_starts_with_at", rec(0, a[0], qual="@" + "I" * (len(a[0]) - 1)) + rec(1, a[1]),
        rec(0, b[0], qual="@" + "F" * (len(b[0]) - 1)) + rec(1, b[1]))
    add("last_record_without_plus_and_quality", rec(0, a[0]) + f"@r1\n{a[1]}\n", rec(0, b[0]) + f"@r1\n{b[1]}\n")
    add("no_final_newline", (rec(0, a[0]) + rec(1, a[1]))[:-1], (rec(0, b[0]) + rec(1, b[1]))[:-1])
    add("lower_case_and_n", rec(0, a[0].lower()) + rec(1, a[1][:10] + "NN" + a[1][10:]), rec(0, b[0]) + rec(1, b[1].lower()))
    add("ids_differ", rec(0, a[0]) + rec(1, a[1], "other"), rec(0, b[0]) + rec(1, b[1]))
    add("fasta_input", f">r0\n{a[0]}\n", f">r0\n{b[0]}\n")
    add("different_formats", rec(0, a[0]), f">r0\n{b[0]}\n")
    add("header_without_sequence", rec(0, a[0]) + "@r1\n", rec(0, b[0]) + "@r1\n")
    add("misaligned_lines", rec(0, a[0]) + "junk\n" + rec(1, a[1]), rec(0, b[0]) + "@junk\n" + rec(1, b[1]))
    return cases


def main():
    rng = np.random.default_rng(20261004)
    for k in (27, 31):
        f1 = open(os.path.join(HERE, f"pairs_k{k}_1.fq"), "rb").read()
        f2 = open(os.path.join(HERE, f"pairs_k{k}_2.fq"), "rb").read()
        rc, merged, err = ref_merge(f1, f2)
        assert rc == 0 and merged, err
        open(os.path.join(HERE, f"pairs_k{k}_merged.fa"), "wb").write(merged)
    out = []
    for name, f1, f2 in edge_cases(rng):
        rc, merged, err = ref_merge(f1, f2)
        out.append(dict(name=name, f1=f1.decode("latin1"), f2=f2.decode("latin1"), rc=rc,
                        merged=None if merged is None else merged.decode("latin1"), stderr=err))
        print(f"{name:40s} rc={rc} merged={'-' if merged is None else len(merged)} {err.strip()[:70]}")
    json.dump(dict(cases=out), open(os.path.join(HERE, "pairs_edge.json"), "w"), indent=1)
    vec = []
    with tempfile.TemporaryDirectory() as d:
        for k in (2, 3, 8, 15, 16, 17, 20, 27, 31, 32):
            seqs = ["A" * k, "C" * k, "G" * k, "T" * k, ("ACGT" * 8)[:k], ("TTGCAA" * 6)[:k]]
            seqs += ["".join(rng.choice(list("ACGTacgt"), k)) for _ in range(40)]
            p = os.path.join(d, f"k{k}.txt")
            open(p, "w").write("\n".join(seqs) + "\n")
            r = subprocess.run([REF, "codec", str(k), p], capture_output=True, text=True, check=True)
            rows = [ln.split() for ln in r.stdout.splitlines()]
            assert len(rows) == len(seqs)
            vec += [dict(k=k, seq=s, fwd=int(a), rev=int(b)) for s, (a, b) in zip(seqs, rows)]
    json.dump(dict(vectors=vec), open(os.path.join(HERE, "codec_vectors.json"), "w"), indent=0)
    print(len(vec), "codec vectors")


if __name__ == "__main__":
    main()

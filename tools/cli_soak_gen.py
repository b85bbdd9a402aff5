#!/usr/bin/env python3
"""Input files for tools/cli_soak.sh (a process of its own: nothing of this runs inside the binary under test).
    cli_soak_gen.py db <dir> <seed>       random target genomes + targets.txt
    cli_soak_gen.py reads <dir> <seed>    one round of inputs sampled from those genomes: in.fa (sequences over several lines, some
                                          records with N / lower case / IUPAC / long names), in.fq, p1.fq + p2.fq, and gzip / block-gzip
                                          forms of the FASTQ files"""
import gzip
import os
import struct
import sys
import zlib

import numpy as np


def genomes(d, seed):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(d, "genomes"), exist_ok=True)
    shared = "".join(rng.choice(list("ACGT"), 3000))
    with open(os.path.join(d, "targets.txt"), "w") as t:
        for g in range(12):
            s = "".join(rng.choice(list("ACGT"), int(rng.integers(20000, 60000))))
            if g % 3 == 0:
                s = s[:5000] + shared + s[5000:]          # k-mers common to several targets are removed by the builder
            if g % 4 == 1:
                s = s[:9000] + "ACACACACAC" * 40 + s[9000:]   # a microsatellite: crowded minimizers
            p = os.path.join(d, "genomes", f"g{g}.fa")
            with open(p, "w") as f:
                f.write(f">genome{g}\n")
                for o in range(0, len(s), 70):
                    f.write(s[o:o + 70] + "\n")
            t.write(f"{p} T{g:02d}\n")


def bgzf(data, path, member):
    with open(path, "wb") as f:
        for o in range(0, len(data), member):
            blk = data[o:o + member]
            c = zlib.compressobj(1, zlib.DEFLATED, -15)
            body = c.compress(blk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk), len(blk)))
        f.write(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0")


def reads(d, seed):
    rng = np.random.default_rng(seed)
    gs = []
    for g in range(12):
        with open(os.path.join(d, "genomes", f"g{g}.fa")) as f:
            gs.append("".join(l.strip() for l in f if not l.startswith(">")))
    comp = str.maketrans("ACGT", "TGCA")
    n = int(rng.choice([1, 7, 300, 3000, 20000]))
    L = int(rng.choice([31, 50, 100, 150, 250, 600]))

    def seq():
        ln = int(rng.integers(max(1, L // 3), L + 1))
        if rng.random() < 0.75:
            g = gs[int(rng.integers(12))]
            o = int(rng.integers(0, len(g) - ln))
            s = g[o:o + ln]
            if rng.random() < 0.5:
                s = s[::-1].translate(comp)
        else:
            s = "".join(rng.choice(list("ACGT"), ln))
        s = list(s)
        for i in range(len(s)):
            r = rng.random()
            if r < 0.004:
                s[i] = "N"
            elif r < 0.006:
                s[i] = s[i].lower()
            elif r < 0.007:
                s[i] = "R"
            elif r < 0.017:
                s[i] = "ACGT"[int(rng.integers(4))]
        return "".join(s)
    fa, fq, p1, p2 = [], [], [], []
    for i in range(n):
        name = f"r{i}" + ("_" + "x" * int(rng.integers(30, 60)) if rng.random() < 0.02 else "") + (" desc" if rng.random() < 0.1 else "")
        s = seq()
        w = int(rng.choice([60, 70, 100000]))
        fa.append(f">{name}\n" + "\n".join(s[o:o + w] for o in range(0, len(s), w)) + "\n")
        fq.append(f"@{name}\n{s}\n+\n{'I' * len(s)}\n")
        a, b = seq(), seq()
        p1.append(f"@p{i}/1\n{a}\n+\n{'F' * len(a)}\n")
        p2.append(f"@p{i}/2\n{b}\n+\n{'F' * len(b)}\n")
    crlf = rng.random() < 0.1
    for name, recs in (("in.fa", fa), ("in.fq", fq), ("p1.fq", p1), ("p2.fq", p2)):
        text = "".join(recs)
        if crlf and name == "in.fa":
            text = text.replace("\n", "\r\n")
        with open(os.path.join(d, name), "w", newline="") as f:
            f.write(text)
    lvl = int(rng.integers(1, 10))
    for name in ("in.fq", "p1.fq", "p2.fq"):
        data = open(os.path.join(d, name), "rb").read()
        with open(os.path.join(d, name + ".gz"), "wb") as f:
            f.write(gzip.compress(data, lvl))
        bgzf(data, os.path.join(d, name + ".bgz"), int(rng.choice([300, 4000, 0xFF00])))


if __name__ == "__main__":
    {"db": genomes, "reads": reads}[sys.argv[1]](sys.argv[2], int(sys.argv[3]))

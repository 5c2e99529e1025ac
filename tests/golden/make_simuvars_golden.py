#!/usr/bin/env python3
"""Generates tests/golden/simuvars/: a small 3-record reference (lower-case runs, an N block, a "chrom" name, a header
with a description), an UCSC-style SNP table (both strands), a variation file (copy numbers 0..8 with every major-copy
shape, touching / overhanging CNV intervals, SNVs, insertions and deletions, het and homo, a deletion across a segment
end) -- and the haplotype FASTA that the compiled reference (`make -C oracle ref` -> oracle/_ref/scssim_ref simuvars)
writes for them.  The reference's simuvars never seeds rand() (src/scssim.cpp:33-38), so its output is a pure function
of these files.  Build container only: the GPU box sees the committed files, never the reference."""
import gzip
import hashlib
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "simuvars")
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "scssim_ref")

VARS = """# test variations
i\tchr20\t45100\ttcgagtcg\thomo
i\tchr20\t110100\ttcgagtc\thomo
i\tchr20\t120100\ttcgagt\thet
i\tchr20\t120150\tAAcc\thet
i\tchr20\t344100\ttcgagtcg\thet
i\tchr20\t70020\tGGGTTT\thomo
d\tchr20\t30100\t10\thomo
d\tchr20\t50100\t9\thet
d\tchr20\t95100\t8\thomo
d\tchr20\t120120\t8\thet
d\tchr20\t70030\t7\thomo
d\tchr20\t362000\t6\thet
d\tchr20\t199995\t12\thomo
s\tchr20\t20100\ta\tT\thomo
s\tchr20\t40100\tT\tG\thomo
s\tchr20\t85100\tA\tG\thet
s\tchr20\t90100\tG\tc\thomo
s\tchr20\t115100\tG\tC\thomo
s\tchr20\t116100\tc\tT\thet
s\tchr20\t251000\tg\tA\thet
s\tchr20\t253000\tC\tG\thet
s\tchr20\t355000\tC\tT\thomo
c\tchr20\t50000\t60000\t1\t1
c\tchr20\t100000\t145000\t3\t2
c\tchr20\t150000\t160000\t0\t0
c\tchr20\t175000\t200000\t4\t3
c\tchr20\t200000\t235000\t1\t1
c\tchr20\t240000\t270000\t2\t2
c\tchr20\t290000\t300000\t8\t4
c\tchr20\t310000\t320000\t5\t5
c\tchr20\t330000\t335000\t6\t1
c\tchr20\t340000\t350000\t7\t4
c\tchr20\t390000\t450000\t3\t3
c\tchr21\t1\t20000\t2\t1
c\tchr21\t50000\t60000\t8\t7
s\tchr21\t55000\tA\tC\thet
i\tchr21\t55010\tACGTACGT\thet
d\tchr21\t55100\t20\thet
"""


def main():
    rng = np.random.default_rng(42)

    def rec(n, nblock=0):
        s = np.frombuffer(b"ACGT", np.uint8)[rng.choice(4, size=n, p=[0.3, 0.2, 0.2, 0.3])].copy()
        if nblock:
            s[:nblock] = ord("N")
        s[rng.random(n) < 0.06] |= 0x20
        return s

    recs = [("chr20", rec(400000, 3000)), ("chr21 some description", rec(150000)), ("chrom5", rec(60000))]
    os.makedirs(OUT, exist_ok=True)
    td = tempfile.mkdtemp()
    ref = os.path.join(td, "ref.fa")
    with open(ref, "wb") as f:
        for name, s in recs:
            f.write(b">" + name.encode() + b"\n")
            for i in range(0, len(s), 60):
                f.write(s[i:i + 60].tobytes() + b"\n")
    seq20 = recs[0][1]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    snp = os.path.join(td, "snp.txt")
    with open(snp, "w") as f:
        for i, p in enumerate(sorted(rng.choice(np.arange(4000, 399000), size=300, replace=False))):
            r = chr(seq20[p - 1]).upper()
            alt = [b for b in "ACGT" if b != r][rng.integers(3)]
            strand = "+" if rng.random() < 0.7 else "-"
            obs = "/".join(sorted([r, alt])) if strand == "+" else "/".join(sorted([comp[r], comp[alt]]))
            f.write("rs%d\tchr20\t%d\t%s\t%s\t%s\n" % (i, p, obs, strand, r))
        for i, p in enumerate(sorted(rng.choice(np.arange(100, 149000), size=40, replace=False))):
            r = chr(recs[1][1][p - 1]).upper()
            alt = [b for b in "ACGT" if b != r][0]
            f.write("rt%d\tchr21\t%d\t%s/%s\t+\t%s\n" % (i, p, r, alt, r))
    var = os.path.join(td, "vars.txt")
    open(var, "w").write(VARS)
    cases = {"full": ["-s", snp, "-v", var], "snp_only": ["-s", snp], "plain": []}
    manifest = {}
    for name, extra in cases.items():
        out = os.path.join(td, name + ".fa")
        subprocess.check_call([REF_BIN, "simuvars", "-r", ref, "-o", out] + extra, stderr=subprocess.DEVNULL)
        manifest[name] = {"args": [a if a.startswith("-") else os.path.basename(a) for a in extra], "sha256": hashlib.sha256(open(out, "rb").read()).hexdigest(),
                          "bytes": os.path.getsize(out)}
    with gzip.GzipFile(os.path.join(OUT, "ref.fa.gz"), "wb", 9, mtime=0) as g:
        g.write(open(ref, "rb").read())
    with gzip.GzipFile(os.path.join(OUT, "expected_full.fa.gz"), "wb", 9, mtime=0) as g:
        g.write(open(os.path.join(td, "full.fa"), "rb").read())
    for fn in ("snp.txt", "vars.txt"):
        open(os.path.join(OUT, fn), "w").write(open(os.path.join(td, fn)).read())
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()

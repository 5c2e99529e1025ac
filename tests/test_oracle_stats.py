"""The counter-mode remaps (primer snapshot, binomial errors, det_log Poisson, polar-normal GC
factor, chunked weight sum) must leave the DISTRIBUTIONS of the reference untouched.  Both oracle modes run on the same
300 kb genome; summary statistics of ref mode (= the reference's streams) and counter mode must agree within sampling
noise (generous 5-sigma-style bounds so the test is not flaky)."""
import subprocess

import numpy as np
import pytest


def _dump(path):
    rows = []
    for line in open(path):
        f = line.rstrip("\n").split("\t")
        nerr = len(f[7].split(",")) if len(f) > 7 and f[7] else 0
        rows.append((int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), nerr))
    return np.array(rows, dtype=np.int64)


def _fastq_stats(path):
    lines = open(path, "rb").read().split(b"\n")
    seqs, quals = lines[1::4], lines[3::4]
    lens = np.array([len(s) for s in seqs if s])
    q = np.frombuffer(b"".join(quals), np.uint8).astype(np.float64) - 33
    return lens, q


@pytest.fixture(scope="module")
def runs(oracle_bin, models, tmp_path_factory):
    import os
    import sys
    from conftest import ROOT
    d = tmp_path_factory.mktemp("stats")
    fa = str(d / "simu.fa")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "300000", "--seed", "77", "--simu-out", fa])
    out = {}
    for mode, extra in (("ref", ["--rng", "ref", "--fixed-time", "1555555555"]), ("counter", ["--rng", "counter", "--seed", "12345", "-t", "4"])):
        pre = str(d / mode)
        subprocess.check_call([oracle_bin, "genreads", "-i", fa, "-m", models["Illumina_HiSeq2500"], "-c", "8", "-o", pre, "--dump", pre, "-q"] + extra)
        out[mode] = pre
    return out


def test_amplicon_statistics_agree(runs):
    s = {m: _dump(p + ".semis.tsv") for m, p in runs.items()}
    f = {m: _dump(p + ".fulls.tsv") for m, p in runs.items()}
    for tab, name in ((s, "semis"), (f, "fulls")):
        a, b = tab["ref"], tab["counter"]
        assert abs(len(a) - len(b)) < 0.12 * len(a), name + " count"      # run-to-run spread of the MALBAC growth process is ~4 %
        for col, what in ((2, "len"), (3, "gc"), (5, "errors")):
            ma, mb = a[:, col].mean(), b[:, col].mean()
            se = np.sqrt(a[:, col].var() / len(a) + b[:, col].var() / len(b))
            assert abs(ma - mb) < 6 * se + 1e-9, "%s mean %s: %.4f vs %.4f" % (name, what, ma, mb)
        # error-count histogram: binomial remap vs one Bernoulli per base
        ha = np.bincount(a[:, 5], minlength=8)[:8] / len(a)
        hb = np.bincount(b[:, 5], minlength=8)[:8] / len(b)
        assert np.abs(ha - hb).max() < 0.012, name + " error-count histogram"
    # GC weighting / primer budgets: mean semi budget of the last cycle
    assert abs(s["ref"][:, 4].mean() - s["counter"][:, 4].mean()) < 0.25


def test_read_statistics_agree(runs):
    la, qa = _fastq_stats(runs["ref"] + "_1.fq")
    lb, qb = _fastq_stats(runs["counter"] + "_1.fq")
    assert abs(len(la) - len(lb)) <= 2
    assert abs(la.mean() - lb.mean()) < 0.05 and abs((la != 125).mean() - (lb != 125).mean()) < 0.02
    assert abs(qa.mean() - qb.mean()) < 0.15
    # reads per amplicon (allocation): same mean / dispersion of the amplicon index gaps
    def idx(p):
        return np.array([int(l[1:l.index(b"#")]) for l in open(p + "_1.fq", "rb").read().split(b"\n")[0::4] if l])
    ia, ib = idx(runs["ref"]), idx(runs["counter"])
    assert abs(len(np.unique(ia)) - len(np.unique(ib))) < 0.05 * len(np.unique(ia))

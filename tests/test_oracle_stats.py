"""The counter-mode remaps (primer snapshot, binomial errors, det_log Poisson, polar-normal GC
factor, chunked weight sum) must leave the DISTRIBUTIONS of the reference untouched.  Both oracle modes run on the same
300 kb genome; summary statistics of ref mode (= the reference's streams) and counter mode must agree within sampling
noise (generous 5-sigma-style bounds so the test is not flaky)."""
import os
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


# ---- finer pins: the remaps of the read stage (7: per-read streams, 8: indel gaps, 9: alias qualities) on
# Profile::predict itself -- the same windows through the reference's streams and through counter mode -- and those of the
# amplification / planning stages (6: attach tries, insert sizes, allocation) on a larger job (400 kb, 40x).
def _chi2_two_sample(a, b, min_count=10):
    """two-sample chi-square of two count vectors over the cells that hold enough counts; returns (statistic, degrees of freedom)"""
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    keep = (a + b) >= min_count
    a, b = a[keep], b[keep]
    ka, kb = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    return float((((ka * a - kb * b) ** 2) / (a + b)).sum()), int(keep.sum()) - 1


def _assert_same_distribution(a, b, what, sigmas=6.0, min_count=10):
    chi2, df = _chi2_two_sample(a, b, min_count)
    assert df >= 1, what + ": nothing to compare"
    assert chi2 < df + sigmas * np.sqrt(2.0 * df) + 5, "%s: chi2 %.1f over %d degrees of freedom" % (what, chi2, df)


def _predict_batch(L_, h, mode, win, rd1, seed=None, first_uid=1 << 40):
    """Profile::predict of the oracle on a batch of windows: mode "ref" (the reference's two mt19937 streams) or "counter"; (bases, qualities, n')"""
    import ctypes
    count, n = win.shape
    stride = 2 * n + 64
    ob = np.zeros((count, stride), np.uint8); oq = np.zeros((count, stride), np.uint8); lens = np.zeros(count, np.int32)
    ptr = lambda x: x.ctypes.data_as(ctypes.c_void_p)
    if mode == "ref":
        rc = L_.scso_predict_ref_batch(h, ptr(win), n, count, ptr(rd1), 20240607 if seed is None else seed, ptr(ob), ptr(oq), ptr(lens))
    else:
        rc = L_.scso_predict_counter_batch(h, ptr(win), n, count, ptr(rd1), 777 if seed is None else seed, first_uid, ptr(ob), ptr(oq), ptr(lens))
    assert rc == 0
    return ob, oq, lens


@pytest.fixture(scope="module")
def predicted(oracle_lib, models):
    """60 000 random ACGT windows per model through Profile::predict, both mates: the reference's streams vs counter mode."""
    import ctypes
    L_ = oracle_lib
    L_.scso_profile_load.restype = ctypes.c_void_p
    L_.scso_profile_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    for f in (L_.scso_predict_ref_batch, L_.scso_predict_counter_batch):
        f.restype = ctypes.c_int
    L_.scso_predict_ref_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L_.scso_predict_counter_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L_.scso_profile_read_length.argtypes = [ctypes.c_void_p]
    out = {}
    for model in ("Illumina_HiSeq2500", "Illumina_HiSeqXTen"):
        h = L_.scso_profile_load(models[model].encode(), 1, 260)
        n = L_.scso_profile_read_length(h)
        rng = np.random.default_rng(99)
        count = 60000
        win = rng.choice(4, size=(count, n), p=[0.3, 0.2, 0.2, 0.3]).astype(np.uint8)
        rd1 = (np.arange(count) % 2 == 0).astype(np.uint8)
        res = {mode: _predict_batch(L_, h, mode, win, rd1) for mode in ("ref", "counter")}
        out[model] = (n, win, rd1, res)
    return out


def _predict_stats(n, win, rd1, res):
    """per mode: the statistics of a batch of predicted reads (see test_predict_distributions_agree_per_bin)"""
    code = np.full(256, 4, np.uint8); code[[65, 67, 71, 84]] = [0, 1, 2, 3]
    stats = {}
    for mode, (ob, oq, lens) in res.items():
        same = lens == n                                              # reads of unchanged length: position j came from window base j ...
        b = code[ob[same][:, :n]]; w = win[same]; q = oq[same][:, :n].astype(np.int64) - 33
        mates = rd1[same]
        sub = b != w
        framed = sub.sum(axis=1) <= 10                                # ... unless an insertion and a deletion cancelled: a shifted stretch, dozens of mismatches
        n_shifted = int((~framed).sum())
        b, w, q, mates, sub = b[framed], w[framed], q[framed], mates[framed], sub[framed]
        st = {"len_hist": np.bincount(lens - n + 64, minlength=160)[:160], "n_same": int(same.sum()), "n_shifted": n_shifted,
              "sub_per_bin": np.stack([sub[mates == m].sum(axis=0) for m in (1, 0)]),              # [mate][bin]
              "reads_per_mate": np.array([(mates == 1).sum(), (mates == 0).sum()]),
              "sub_matrix": np.bincount((w[sub].astype(np.int64) * 4 + b[sub]), minlength=16),
              "qual_hist": np.stack([np.bincount(q[:, lo:lo + 5].ravel(), minlength=94)[:94] for lo in range(0, n - 4, 5)]),    # groups of 5 bins
              "qual_of_subs": np.bincount(q[sub], minlength=94)[:94], "mean_q_per_bin": q.mean(axis=0)}
        stats[mode] = st
    return stats


def _check_qualities(a, c, what):
    """[REMAP 9] the alias rows: the quality histogram per group of 5 bins, the qualities of substituted bases, the mean per bin"""
    _assert_same_distribution(a["qual_hist"], c["qual_hist"], what + ": quality histogram per group of 5 bins")
    _assert_same_distribution(a["qual_of_subs"], c["qual_of_subs"], what + ": qualities of substituted bases (the off-diagonal rows)", min_count=6)
    assert np.abs(a["mean_q_per_bin"] - c["mean_q_per_bin"]).max() < 0.35, what + ": mean quality per bin"


@pytest.mark.parametrize("model", ["Illumina_HiSeq2500", "Illumina_HiSeqXTen"])
def test_predict_distributions_agree_per_bin(model, predicted):
    """[REMAP 7, 8, 9] Per position bin: the substitution rate and the substitution matrix, the quality histogram; per read: the length
    histogram (every insertion / deletion length shows as its own n' - n), the number of reads with any indel.  Chi-square of
    counter mode against the reference's streams over 7.5 M bases per model."""
    n, win, rd1, res = predicted[model]
    stats = _predict_stats(n, win, rd1, res)
    a, c = stats["ref"], stats["counter"]
    _assert_same_distribution(a["len_hist"], c["len_hist"], model + ": read-length histogram (indel kinds and lengths)")
    assert abs(a["n_same"] - c["n_same"]) < 6 * np.sqrt(a["n_same"] * (1 - a["n_same"] / 60000.0)) + 5
    assert abs(a["n_shifted"] - c["n_shifted"]) < 6 * np.sqrt(a["n_shifted"] + c["n_shifted"]) + 5, "reads whose indels cancel"
    for m in (0, 1):
        _assert_same_distribution(a["sub_per_bin"][m], c["sub_per_bin"][m], "%s: substitutions per bin, mate %d" % (model, m + 1), min_count=6)
        ra, rc = a["sub_per_bin"][m].sum() / (a["reads_per_mate"][m] * n), c["sub_per_bin"][m].sum() / (c["reads_per_mate"][m] * n)
        assert abs(ra - rc) < 6 * np.sqrt(ra / (a["reads_per_mate"][m] * n) * 2), "%s: substitution rate mate %d: %.5f vs %.5f" % (model, m + 1, ra, rc)
    _assert_same_distribution(a["sub_matrix"], c["sub_matrix"], model + ": substitution matrix")
    _check_qualities(a, c, model)


@pytest.fixture(scope="module")
def big_runs(oracle_bin, models, tmp_path_factory):
    import os
    import sys
    from conftest import ROOT
    d = tmp_path_factory.mktemp("stats_big")
    fa = str(d / "simu.fa")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "400000", "--seed", "78", "--simu-out", fa])
    out = {}
    for mode, extra in (("ref", ["--rng", "ref", "--fixed-time", "1666666666"]), ("counter", ["--rng", "counter", "--seed", "424242", "-t", "8"])):
        pre = str(d / mode)
        subprocess.check_call([oracle_bin, "genreads", "-i", fa, "-m", models["Illumina_HiSeq2500"], "-c", "40", "-o", pre, "--dump", pre, "-q"] + extra)
        out[mode] = pre
    return out


def test_attach_geometry_and_children_agree(big_runs):
    """[REMAP 6] geometric skip + closed-form feasible pair instead of try-by-try draws: the joint distribution of (spos, len) of the
    semi and of the full amplicons, and the number of amplicons a template ends up with (where the 50-try cap and the abort of a
    template's remaining primers show), ref vs counter."""
    for name, pbin in (("semis", 8000), ("fulls", 125)):
        tab = {m: _dump(p + "." + name + ".tsv") for m, p in big_runs.items()}
        h = {}
        for m, t in tab.items():
            h[m] = np.histogram2d(t[:, 1], t[:, 2], bins=[np.arange(0, t[:, 1].max() + pbin + 1, pbin) if name == "fulls" else 12, np.arange(1000, 2101, 100)])[0]
        if name == "semis":                                            # positions on fragments of different lengths: by relative position instead
            h = {m: np.histogram2d(t[:, 1] % 10000, t[:, 2], bins=[np.arange(0, 10001, 1000), np.arange(1000, 2101, 100)])[0] for m, t in tab.items()}
        _assert_same_distribution(h["ref"], h["counter"], name + ": (spos, len) joint histogram")
        _assert_same_distribution(np.bincount(tab["ref"][:, 2] - 1000, minlength=1001), np.bincount(tab["counter"][:, 2] - 1000, minlength=1001), name + ": length histogram", min_count=20)
        if name == "fulls":                                            # (a fragment has thousands of semi amplicons: nothing to histogram there)
            kids = {m: np.bincount(np.bincount(t[:, 0]), minlength=40)[:40] for m, t in tab.items()}     # semi amplicons with k children
            _assert_same_distribution(kids["ref"][1:], kids["counter"][1:], name + ": children per template")
        # and the counts themselves: the MALBAC growth process spreads ~2-4 % from run to run
        assert abs(len(tab["ref"]) - len(tab["counter"])) < 0.08 * len(tab["ref"]), name


def test_insert_sizes_and_allocation_agree(big_runs):
    """The pair planning (insert-size draws, positions) and the read allocation: insert-size histogram, mean and sigma; position of the pair
    inside its amplicon; reads per amplicon against the amplicon's length and GC content (GC factor: remap 4; sums: remap 5)."""
    rd = {m: np.loadtxt(p + ".reads.tsv", dtype=np.int64) for m, p in big_runs.items()}              # amplicon, fragCount, pos, isize, m1, m2
    ia, ic = rd["ref"][:, 3], rd["counter"][:, 3]
    assert abs(len(ia) - len(ic)) <= 2
    _assert_same_distribution(np.bincount(ia, minlength=700)[:700], np.bincount(ic, minlength=700)[:700], "insert-size histogram")
    assert abs(ia.mean() - ic.mean()) < 6 * ia.std() * np.sqrt(2.0 / len(ia)) and abs(ia.std() - ic.std()) < 6 * ia.std() / np.sqrt(len(ia))
    _assert_same_distribution(np.bincount(rd["ref"][:, 1], minlength=12)[:12], np.bincount(rd["counter"][:, 1], minlength=12)[:12], "attempt counter (rejected insert sizes)")
    fulls = {m: _dump(p + ".fulls.tsv") for m, p in big_runs.items()}
    relpos = {m: np.histogram(rd[m][:, 2] / np.maximum(1, fulls[m][rd[m][:, 0], 2] - rd[m][:, 3] + 1), bins=20, range=(0, 1))[0] for m in rd}
    _assert_same_distribution(relpos["ref"], relpos["counter"], "position of the pair inside its amplicon")
    per = {}
    for m in rd:
        rn = np.zeros(len(fulls[m]), np.int64)
        for line in open(big_runs[m] + ".readnum.tsv"):
            i, v = line.split(); rn[int(i)] = int(v)
        ln, gc = fulls[m][:, 2], fulls[m][:, 3] * 1000 // np.maximum(1, fulls[m][:, 2])      # GC content in per mille (a 1.5 kb window of a 40 % GC genome: 400 +- 13)
        def binned(key, edges):                                       # (mean, standard error, count) of the read numbers per bin of `key`
            rows = []
            for lo, hi in zip(edges[:-1], edges[1:]):
                v = rn[(key >= lo) & (key < hi)].astype(np.float64)
                rows.append((v.mean(), v.std() / np.sqrt(len(v)), len(v)) if len(v) >= 500 else (np.nan, np.nan, len(v)))
            return np.array(rows)
        per[m] = (binned(ln, np.arange(1000, 2101, 100)), binned(gc, np.arange(360, 441, 10)), np.bincount(np.minimum(rn, 12), minlength=13), rn.mean())
    for k, what in ((0, "length"), (1, "GC content")):
        A, C = per["ref"][k], per["counter"][k]
        ok = ~np.isnan(A[:, 0]) & ~np.isnan(C[:, 0])
        assert ok.sum() >= 4, what
        ra, rc = A[ok, 0] / per["ref"][3], C[ok, 0] / per["counter"][3]                  # relative to the job's mean: the growth process moves the amplicon count by a few %
        se = np.sqrt((A[ok, 1] / per["ref"][3]) ** 2 + (C[ok, 1] / per["counter"][3]) ** 2)
        assert (np.abs(ra - rc) < 6 * se + 0.02).all(), "reads per amplicon against its %s: %s vs %s (se %s)" % (what, ra, rc, se)
        assert ra.max() / ra.min() > (1.3 if k == 0 else 1.03), "the statistic must see the dependence it pins (%s)" % what
    ha, hc = per["ref"][2].astype(np.float64), per["counter"][2].astype(np.float64)
    assert np.abs(ha / ha.sum() - hc / hc.sum()).max() < 0.01, "reads-per-amplicon histogram"


# ---- the exhaustion regime: the primer stock is exact in both modes ---------------------------------------------------
# With the default -p / -r a primer type runs dry only on whole-genome inputs; a repeat-rich genome at -p 10000 -r 1e-8
# (same growth per cycle as the defaults: pool x gamma = 6.55 primers per template-length) exhausts the 8-mers of its
# low-complexity blocks within 1.5 Mb.  The reference decrements live (Malbac.cpp:91-103): a type is used exactly `stock`
# times, by the first `stock` attachments in list order that ask for it.  Counter mode keeps that (oracle: amplify_pass;
# HIP: the cut table, tests/test_gpu_parity.py::test_primer_exhaustion_*): the two modes differ in their draws only.
@pytest.fixture(scope="module")
def exhausted(oracle_bin, models, repeat_genome, tmp_path_factory):
    d = tmp_path_factory.mktemp("exhaust")
    fa = repeat_genome                                                   # conftest.write_repeat_genome: 25 % of it 2 kb runs of (A)n, (AC)n, (AG)n, (T)n, (GT)n
    out = {}
    for mode, extra in (("ref", ["--rng", "ref", "--fixed-time", "1555555555"]), ("counter", ["--rng", "counter", "--seed", "3", "-t", "4"])):
        pre = str(d / mode)
        subprocess.check_call([oracle_bin, "genreads", "-i", fa, "-m", models["Illumina_HiSeq2500"], "-c", "0.5", "-p", "10000", "-r", "1e-8",
                               "-o", pre, "--dump", pre, "-q"] + extra)
        out[mode] = np.loadtxt(pre + ".primers.tsv", dtype=np.int64)     # primer type, attachments, stock left
    return out


def test_primer_exhaustion_exact_in_both_modes(exhausted):
    stock = 10000
    ref, ctr = exhausted["ref"], exhausted["counter"]
    ex_r, ex_c = ref[ref[:, 2] == 0], ctr[ctr[:, 2] == 0]
    assert len(ex_r) >= 4, "the input must drive some primer types dry"
    assert set(ex_r[:, 0]) == set(ex_c[:, 0]), "the same primer types run dry in both modes"
    assert (ex_r[:, 1] == stock).all() and (ex_c[:, 1] == stock).all(), "an exhausted type is used exactly `stock` times"
    assert (ref[:, 1] + ref[:, 2] == stock).all() and (ctr[:, 1] + ctr[:, 2] == stock).all(), "stock left = stock - attachments, for every type"
    # the job as a whole: attachments (= amplicons made) and their spread over the other types
    assert abs(int(ctr[:, 1].sum()) - int(ref[:, 1].sum())) < 0.03 * ref[:, 1].sum()
    rest_r = ref[ref[:, 2] > 0][:, 1].astype(np.float64); rest_c = ctr[ctr[:, 2] > 0][:, 1].astype(np.float64)
    assert abs(rest_r.mean() - rest_c.mean()) < 0.03 * rest_r.mean()


def test_exhausted_pass_equals_the_plain_sequential_loop(oracle_bin, models, repeat_genome, tmp_path):
    """Counter mode settles a pass in which primer types run dry by evaluating again, one after the other, only the templates that ask
    for such a type (oracle: amplify_pass, steps 2 and 3).  SCSO_PLAIN_SEQUENTIAL_PASS=1 replaces that by the reference's loop to
    the letter -- every template, one worker, list order, live decrement (Malbac.cpp:91-103) --: same amplicons, same stock, same FASTQ."""
    import hashlib
    out = {}
    for mode, env in (("fast", {}), ("plain", {"SCSO_PLAIN_SEQUENTIAL_PASS": "1"})):
        pre = str(tmp_path / mode)
        subprocess.check_call([oracle_bin, "genreads", "-i", repeat_genome, "-m", models["Illumina_HiSeq2500"], "-c", "0.5", "-p", "10000", "-r", "1e-8",
                               "-o", pre, "--dump", pre, "-q", "--rng", "counter", "--seed", "77", "-t", "4"], env=dict(os.environ, **env))
        out[mode] = [hashlib.md5(open(pre + s, "rb").read()).hexdigest() for s in ("_1.fq", "_2.fq", ".primers.tsv", ".semis.tsv", ".fulls.tsv")]
    assert out["fast"] == out["plain"]
    prim = np.loadtxt(str(tmp_path / "fast") + ".primers.tsv", dtype=np.int64)
    assert (prim[:, 2] == 0).sum() >= 4


# ---- the gate calibrated: every remapped branch pinned on its own unit, and a 2 % bias in it is caught ---------------------
# The GPU is compared bit for bit with counter mode; what ties counter mode to the REFERENCE's distributions are the statistics of this
# file.  For each [REMAP] branch of oracle/scs_oracle.cpp there is (a) a check of the branch's own unit, counter mode against the
# reference's draws, on a sample large enough to see 2 %, and (b) test_a_biased_remap_is_caught: the same check must FAIL when the
# branch is made wrong by 2 % behind the oracle's test-only switch SCSO_TEST_BIAS (errors: the binomial error count's rate; attach: the
# probability that a try fits, i.e. the geometric gap; indel: the per-base event probability of the gap table; alias: a column's own
# share of its draws; gc: the sigma of the polar normal).  (Fragment.cpp:76-123, Amplicon.cpp:179-226, Profile.cpp:1503-1576.)
class _bias:
    """with _bias("indel"): ... -- the oracle's test-only switch for the calls inside (read by the library at every use)"""
    def __init__(self, branch):
        self.branch = branch

    def __enter__(self):
        if self.branch:
            os.environ["SCSO_TEST_BIAS"] = self.branch

    def __exit__(self, *a):
        os.environ.pop("SCSO_TEST_BIAS", None)


def _amp_run(oracle_bin, fa, prof, pre, extra, bias=None):
    env = dict(os.environ)
    env.pop("SCSO_TEST_BIAS", None)
    if bias:
        env["SCSO_TEST_BIAS"] = bias
    subprocess.check_call([oracle_bin, "genreads", "-i", fa, "-m", prof, "-c", "0.2", "-o", pre, "--dump", pre, "-q"] + extra, env=env)
    return _dump(pre + ".fulls.tsv")


@pytest.fixture(scope="module")
def amp_big(oracle_bin, models, tmp_path_factory):
    """1.5 Mb, amplification only (0.2x): 6 x 10^5 full amplicons per mode -- and the genome, for the biased run"""
    import sys
    from conftest import ROOT
    d = tmp_path_factory.mktemp("stats_amp")
    fa = str(d / "simu.fa")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "1500000", "--seed", "79", "--simu-out", fa])
    prof = models["Illumina_HiSeq2500"]
    return {"fa": fa, "prof": prof, "dir": d,
            "ref": _amp_run(oracle_bin, fa, prof, str(d / "ref"), ["--rng", "ref", "--fixed-time", "1555555555"]),
            "counter": _amp_run(oracle_bin, fa, prof, str(d / "counter"), ["--rng", "counter", "--seed", "5", "-t", "8"])}


def _check_error_counts(a, b):
    """[REMAP 2] errors per new amplicon: K ~ Binomial(l - 8, ber) in one draw against one Bernoulli(ber) per base (Fragment.cpp:100-104,
    Amplicon.cpp:203-207): the mean (5 sigma of two samples of 6 x 10^5) and the histogram"""
    ma, mb = a[:, 5].mean(), b[:, 5].mean()
    se = np.sqrt(a[:, 5].var() / len(a) + b[:, 5].var() / len(b))
    assert abs(ma - mb) < 5 * se, "errors per amplicon: %.5f vs %.5f (se %.5f)" % (ma, mb, se)
    _assert_same_distribution(np.bincount(a[:, 5], minlength=10)[:10], np.bincount(b[:, 5], minlength=10)[:10], "error-count histogram")


def test_error_counts_agree_on_many_amplicons(amp_big):
    assert len(amp_big["ref"]) > 500000 and len(amp_big["counter"]) > 500000
    _check_error_counts(amp_big["ref"], amp_big["counter"])


def _attach_tries(L_, length, count, counter, seed):
    import ctypes
    L_.scso_attach_tries_batch.restype = ctypes.c_int
    L_.scso_attach_tries_batch.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64] + [ctypes.c_void_p] * 3
    t = np.zeros(count, np.uint32); sp = np.zeros(count, np.uint32); al = np.zeros(count, np.uint32)
    assert L_.scso_attach_tries_batch(length, 1000, 2000, count, counter, seed, t.ctypes.data, sp.ctypes.data, al.ctypes.data) == 0
    return t, sp, al


def _check_attach_tries(L_, count=2000000):
    """[REMAP 6] one primer on an empty template: the number of tries until one fits (geometric gap, capped at 50 -- where the abort of a
    template's remaining primers comes from) and the (position, length) of the try that fits (closed form over the feasible pairs),
    against the reference's try-by-try draws (Fragment.cpp:76-82, Amplicon.cpp:179-185): a short semi amplicon (8 % of the tries fit), a
    long one, a fragment."""
    for length in (1500, 2600, 30000):
        r = _attach_tries(L_, length, count, 0, 5)
        c = _attach_tries(L_, length, count, 1, 6)
        _assert_same_distribution(np.bincount(r[0], minlength=52), np.bincount(c[0], minlength=52), "tries until a try fits, template of %d" % length)
        okr, okc = r[0] <= 50, c[0] <= 50
        e1, e2 = np.histogram_bin_edges(r[1][okr], 20), np.histogram_bin_edges(r[2][okr], 10)
        _assert_same_distribution(np.histogram2d(r[1][okr], r[2][okr], bins=[e1, e2])[0], np.histogram2d(c[1][okc], c[2][okc], bins=[e1, e2])[0],
                                  "(position, length) of the fitting try, template of %d" % length)
        assert (c[1][okc] + c[2][okc] <= length).all() and (c[1][okc] >= 27).all() and (c[2][okc] >= 1000).all() and (c[2][okc] <= 2000).all()


def test_attach_tries_agree(oracle_lib):
    _check_attach_tries(oracle_lib)


def _gc_factor(L_, h, gc, count, counter, seed):
    import ctypes
    L_.scso_gc_factor_batch.restype = ctypes.c_int
    L_.scso_gc_factor_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p]
    o = np.zeros(count)
    assert L_.scso_gc_factor_batch(h, gc, count, counter, seed, o.ctypes.data) == 0
    return o


def _load_profile(L_, path):
    import ctypes
    L_.scso_profile_load.restype = ctypes.c_void_p
    L_.scso_profile_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
    h = L_.scso_profile_load(path.encode(), 1, 260)
    assert h
    return h


def _check_gc_factor(L_, h, count=400000):
    """[REMAP 4] the GC factor of one GC percentage: keyed polar normal, resampled while negative, against the reference's minstd_rand0 +
    normal_distribution engine of that percentage (Profile.cpp:1405-1411,1503-1513): histogram, mean, sigma"""
    for gc in (30, 40, 55):
        r, c = _gc_factor(L_, h, gc, count, 0, 5), _gc_factor(L_, h, gc, count, 1, 6)
        assert r.min() >= 0 and c.min() >= 0
        e = np.histogram_bin_edges(r, 60)
        _assert_same_distribution(np.histogram(r, e)[0], np.histogram(c, e)[0], "GC factor at %d %%" % gc)
        assert abs(r.mean() - c.mean()) < 6 * r.std() * np.sqrt(2.0 / count) and abs(r.std() - c.std()) < 6 * r.std() / np.sqrt(count), "GC factor at %d %%: mean / sigma" % gc


def test_gc_factor_agrees(oracle_lib, models):
    _check_gc_factor(oracle_lib, _load_profile(oracle_lib, models["Illumina_HiSeq2500"]))


def _length_hist(L_, h, mode, total, threads=8, chunk=100000):
    """read-length histogram of `total` random windows through predict, `threads` batches at a time (ctypes releases the GIL)"""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    L_.scso_profile_read_length.argtypes = [ctypes.c_void_p]
    n = L_.scso_profile_read_length(h)

    def one(k):
        rng = np.random.default_rng(1000 + k)
        win = rng.choice(4, size=(chunk, n), p=[0.3, 0.2, 0.2, 0.3]).astype(np.uint8)
        rd1 = (np.arange(chunk) % 2 == 0).astype(np.uint8)
        _, _, lens = _predict_batch(L_, h, mode, win, rd1, seed=(31 + k) if mode == "ref" else 777, first_uid=(1 << 40) + k * chunk)
        return np.bincount(lens - n + 64, minlength=160)[:160]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return sum(ex.map(one, range(total // chunk))), n


@pytest.fixture(scope="module")
def ref_lengths(oracle_lib, models):
    """3.2 M reads through predict with the reference's streams: their length histogram (8 threads, ~10 s)"""
    h = _load_profile(oracle_lib, models["Illumina_HiSeq2500"])
    for f in (oracle_lib.scso_predict_ref_batch, oracle_lib.scso_predict_counter_batch):
        f.restype = __import__("ctypes").c_int
    import ctypes
    oracle_lib.scso_predict_ref_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    oracle_lib.scso_predict_counter_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    return _length_hist(oracle_lib, h, "ref", 3200000)


def _check_indel_rate(ref_hist, ctr_hist):
    """[REMAP 8] reads that keep their length (no event, or events that cancel) and the whole n' - n histogram: the gap table against one
    test per base (Profile.cpp:1552-1570, 1606-1622)"""
    total = int(ref_hist.sum())
    a, c = int(ref_hist[64]), int(ctr_hist[64])
    assert abs(a - c) < 6 * np.sqrt(2.0 * a * (1 - a / total)), "reads of unchanged length: %d vs %d of %d" % (a, c, total)
    _assert_same_distribution(ref_hist, ctr_hist, "read-length histogram")


def test_indel_rate_agrees_on_many_reads(oracle_lib, models, ref_lengths):
    h = _load_profile(oracle_lib, models["Illumina_HiSeq2500"])
    ctr, _ = _length_hist(oracle_lib, h, "counter", 3200000)
    _check_indel_rate(ref_lengths[0], ctr)


@pytest.mark.parametrize("branch", ["errors", "attach", "indel", "alias", "gc"])
def test_a_biased_remap_is_caught(branch, oracle_bin, oracle_lib, models, amp_big, predicted, ref_lengths):
    """The calibration: with ONE counter-mode branch off by 2 % (SCSO_TEST_BIAS, test-only, oracle only) the check that pins that branch must
    fail -- the unbiased run of the same check passes in the tests above."""
    prof = models["Illumina_HiSeq2500"]
    with pytest.raises(AssertionError):
        if branch == "errors":
            b = _amp_run(oracle_bin, amp_big["fa"], amp_big["prof"], str(amp_big["dir"] / "biased"), ["--rng", "counter", "--seed", "5", "-t", "8"], bias="errors")
            _check_error_counts(amp_big["ref"], b)
        elif branch == "attach":
            with _bias("attach"):
                _check_attach_tries(oracle_lib)
        elif branch == "gc":
            with _bias("gc"):
                _check_gc_factor(oracle_lib, _load_profile(oracle_lib, prof))
        elif branch == "indel":
            with _bias("indel"):
                h = _load_profile(oracle_lib, prof)                   # (the gap table is built when the model is loaded)
                ctr, _ = _length_hist(oracle_lib, h, "counter", 3200000)
            _check_indel_rate(ref_lengths[0], ctr)
        else:
            n, win, rd1, res = predicted["Illumina_HiSeq2500"]
            with _bias("alias"):
                h = _load_profile(oracle_lib, prof)                   # (the alias rows too)
                biased = _predict_batch(oracle_lib, h, "counter", win, rd1)
            st = _predict_stats(n, win, rd1, {"ref": res["ref"], "counter": biased})
            _check_qualities(st["ref"], st["counter"], "biased alias rows")

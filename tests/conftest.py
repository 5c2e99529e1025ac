import gzip
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


SEAMS_LIB = os.path.join(ROOT, "scssim_amd", "libscssim_hip_seams.so")
SEAMS_CLI = os.path.join(ROOT, "scssim_amd", "bin", "scssim_seams")


def seams_env(base=None, **knobs):
    """Environment of a child process that needs a test seam (scssim_amd/csrc/scs_seams.h: small batches, forced kernel variants,
    injected failures): the knobs + SCSSIM_HIP_LIB = the seams build of the library.  The product library reads none of them."""
    return dict(base if base is not None else os.environ, SCSSIM_HIP_LIB=SEAMS_LIB, **knobs)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_bin():
    """oracle/_build/scs_oracle -- the CPU restatement (test infrastructure)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return os.path.join(ROOT, "oracle", "_build", "scs_oracle")


@pytest.fixture(scope="session")
def oracle_lib():
    import ctypes
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libscs_oracle.so"))


def _gunzip_to(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())
    return dst


def write_repeat_genome(path, n=1500000, seed=11):
    """A repeat-rich diploid input: i.i.d. bases with 2 kb runs of (A)n, (AC)n, (AG)n, (T)n, (GT)n over 25 % of it.  At
    -p 10000 -r 1e-8 (the defaults' growth per cycle) the 8-mers of the runs exhaust their primer stock within 1.5 Mb."""
    import numpy as np
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.choice(4, size=n, p=[0.3, 0.2, 0.2, 0.3])].copy()
    motifs = [b"A", b"AC", b"AG", b"T", b"GT"]
    for pos in range(0, n - 8000, 8000):
        m = motifs[rng.integers(len(motifs))]
        seq[pos + 6000:pos + 8000] = np.frombuffer((m * 2000)[:2000], np.uint8)
    with open(path, "wb") as f:
        for hap in (1, 2):
            f.write(b">9_%d_%d\n" % (hap, n))
            f.write(np.concatenate([seq.reshape(-1, 100), np.full((n // 100, 1), 10, np.uint8)], axis=1).tobytes())
    return path


@pytest.fixture(scope="session")
def repeat_genome(tmp_path_factory):
    return write_repeat_genome(str(tmp_path_factory.mktemp("repeat") / "rep.fa"))


@pytest.fixture(scope="session")
def models(tmp_path_factory):
    """Unpacked copies of the shipped .profile data files (tests/golden/models/*.gz)."""
    d = tmp_path_factory.mktemp("models")
    out = {}
    for fn in sorted(os.listdir(os.path.join(GOLDEN, "models"))):
        name = fn[:-len(".profile.gz")]
        out[name] = _gunzip_to(os.path.join(GOLDEN, "models", fn), str(d / (name + ".profile")))
    return out


@pytest.fixture(scope="session")
def golden_inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("genomes")
    out = {}
    for fn in sorted(os.listdir(GOLDEN)):
        if fn.endswith(".simu.fa.gz"):
            name = fn[:-len(".simu.fa.gz")]
            out[name] = _gunzip_to(os.path.join(GOLDEN, fn), str(d / (name + ".fa")))
    return out

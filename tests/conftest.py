import gzip
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


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

"""simuvars: the oracle's restatement of Genome::loadAbers / saveSequence / generateSegment against the
haplotype FASTA written by the compiled reference (tests/golden/simuvars, made by make_simuvars_golden.py)."""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

from conftest import GOLDEN

SV = os.path.join(GOLDEN, "simuvars")
MANIFEST = json.load(open(os.path.join(SV, "manifest.json")))


@pytest.fixture(scope="module")
def sv_inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("sv")
    ref = str(d / "ref.fa")
    open(ref, "wb").write(gzip.open(os.path.join(SV, "ref.fa.gz")).read())
    return {"ref": ref, "snp.txt": os.path.join(SV, "snp.txt"), "vars.txt": os.path.join(SV, "vars.txt"), "dir": d}


@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_oracle_simuvars_matches_reference(case, oracle_bin, sv_inputs):
    out = str(sv_inputs["dir"] / (case + ".fa"))
    args = [sv_inputs.get(a, a) for a in MANIFEST[case]["args"]]
    subprocess.check_call([oracle_bin, "simuvars", "-r", sv_inputs["ref"], "-o", out] + args)
    data = open(out, "rb").read()
    assert len(data) == MANIFEST[case]["bytes"]
    assert hashlib.sha256(data).hexdigest() == MANIFEST[case]["sha256"]
    if case == "full":
        assert data == gzip.open(os.path.join(SV, "expected_full.fa.gz")).read()


def _fnv(data):
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("case", sorted(MANIFEST))
def test_product_planner_matches_reference(case, sv_inputs, oracle_bin):
    """The product's host planner (a rope of reference / literal pieces instead of std::string edits) folded over the host
    copy of the reference: the FASTA text it describes must be the compiled reference's, byte for byte (FNV-1a of the
    text; host-only seam scs_simuvars_probe)."""
    import scssim_amd
    args = dict(zip(MANIFEST[case]["args"][0::2], MANIFEST[case]["args"][1::2]))
    n, tot, h = scssim_amd.simuvars_probe(sv_inputs["ref"], sv_inputs.get(args.get("-s")), sv_inputs.get(args.get("-v")))
    out = str(sv_inputs["dir"] / (case + "_orc.fa"))
    subprocess.check_call([oracle_bin, "simuvars", "-r", sv_inputs["ref"], "-o", out] + [sv_inputs.get(a, a) for a in MANIFEST[case]["args"]])
    text = open(out, "rb").read()
    assert hashlib.sha256(text).hexdigest() == MANIFEST[case]["sha256"]
    assert n == 6 and h == _fnv(text)
    assert tot == sum(len(l) for l in text.split(b"\n") if l and not l.startswith(b">"))

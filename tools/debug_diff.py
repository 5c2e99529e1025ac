"""Debug helper (GPU box): the first full-pipeline case against the oracle, listing where the FASTQ text differs."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa
import scssim_amd
import numpy as np

def main():
    import __graft_entry__ as ge
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    oracle_bin = os.path.join(root, "oracle", "_build", "scs_oracle")
    tmp = tempfile.mkdtemp()
    from conftest import make_models, make_golden_inputs  # type: ignore

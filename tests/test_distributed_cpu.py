"""World-size-2 (and 3) runs on CPU over gloo: ONE genreads job sharded by fragment lineage must give, after the
k-way merge of the shards' pools, exactly the FASTQ of the unsharded job (shard invariance of the spec the GPU path
implements; exercises scssim_amd/dist.py: all-reduce / all-gatherv hooks and merge_fastq)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

from scssim_amd.dist import merge_fastq


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# the last case: the repeat-rich genome whose primer types run dry (the stock handed from shard to shard, oracle: amplify_pass)
@pytest.mark.parametrize("world,case,model,cov,layout", [(2, "g1_hiseq2500_pe", "Illumina_HiSeq2500", "3", "PE"),
                                                         (3, "g2_xten_pe_nblock", "Illumina_HiSeqXTen", "2", "PE"),
                                                         (2, "g3_hiseq2000_se", "Illumina_HiSeq2000", "2", "SE"),
                                                         (3, "repeat", "Illumina_HiSeq2500", "0.5", "PE")])
def test_sharded_oracle_job_is_shard_invariant(world, case, model, cov, layout, oracle_bin, oracle_lib, models, golden_inputs, repeat_genome, tmp_path):
    seed = "4242"
    whole = str(tmp_path / "whole")
    golden_inputs = dict(golden_inputs, repeat=repeat_genome)
    stock = ["-p", "10000", "-r", "1e-8"] if case == "repeat" else []
    subprocess.check_call([oracle_bin, "genreads", "-i", golden_inputs[case], "-m", models[model], "-c", cov, "-l", layout, "-o", whole,
                           "--rng", "counter", "--seed", seed, "-t", "2", "-q"] + stock + (["--dump", whole] if stock else []))
    if stock:
        import numpy as np
        prim = np.loadtxt(whole + ".primers.tsv", dtype=np.int64)
        assert (prim[:, 2] == 0).sum() >= 4 and (prim[:, 1] + prim[:, 2] == 10000).all(), "the case must drive primer types dry, exactly"
    port = str(_free_port())
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port, LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_oracle_worker.py"), golden_inputs[case], models[model],
                                       str(tmp_path / "shard"), cov, layout, seed] + stock[1::2], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    for suffix in (("_1.fq", "_2.fq") if layout == "PE" else (".fq",)):
        pools = [open(str(tmp_path / "shard") + ".r%d%s" % (r, suffix), "rb").read() for r in range(world)]
        assert all(len(p) > 0 for p in pools), "a shard produced nothing"
        assert merge_fastq(pools) == open(whole + suffix, "rb").read(), "sharded job differs from the unsharded one (%s)" % suffix
    assert "allreduce" in outs[0]

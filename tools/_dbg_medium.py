import os, sys, subprocess, gzip, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scssim_amd
td = tempfile.mkdtemp()
fa = os.path.join(td, "simu.fa")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_genome.py"), "--lengths", "7000000,5000000", "--seed", "31", "--n-block", "20000", "--simu-out", fa])
prof = os.path.join(td, "x.profile")
open(prof, "wb").write(gzip.open(os.path.join(ROOT, "tests", "golden", "models", "Illumina_HiSeqXTen.profile.gz")).read())
g = scssim_amd.GenReads(profile=prof, input_fasta=fa, coverage=float(sys.argv[1]), seed=8)
g.create_frags(); g.amplify(); g.allocate_reads(0)
print("allocated", g.stats()["full_amplicons"], flush=True)
mode = sys.argv[2]
if mode == "null":
    g.yield_reads(collect=False)
else:
    a, b = g.yield_reads()
    print(len(a), len(b))
print("done", g.stats()["pairs_written"], g.kernel_times()["k_reads"], flush=True)

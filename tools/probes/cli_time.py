"""The `scssim genreads` binary on a chr20-size record, three times, with its own stderr (GPU box helper)."""
import os, sys, time, subprocess, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
cd = tempfile.mkdtemp(prefix="clit_", dir="/dev/shm")
try:
    rng = np.random.default_rng(1)
    n20 = 63025520
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n20, p=[0.3, 0.2, 0.2, 0.3])
    fa = os.path.join(cd, "chr20.fa")
    bench.write_simu_fasta(fa, ["20_1_%d" % n20, "20_2_%d" % n20], [seq, seq])
    prof = bench.make_profile(cd)
    cli = os.path.join(ROOT, "scssim_amd", "bin", "scssim")
    for i in range(3):
        t = time.perf_counter()
        r = subprocess.run([cli, "genreads", "-i", fa, "-m", prof, "-c", "30", "-o", os.path.join(cd, "reads"), "--seed", "5"] + sys.argv[1:], capture_output=True, text=True)
        print("run %d: %.3f s rc %d" % (i, time.perf_counter() - t, r.returncode)); print(r.stderr[-1200:])
finally:
    shutil.rmtree(cd, ignore_errors=True)

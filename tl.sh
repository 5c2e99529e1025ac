set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tl -o t --output-format csv -- python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/tl.log 2>&1
python3 - <<'P'
import csv, glob, re
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
def short(n):
    m = re.match(r"(?:void )?(?:scs::)?(\w+(?:<[^>]*>)?)", n); return m.group(1)[:34] if m else n[:34]
# find step boundaries: k_amplify_init marks a job start
starts = [i for i, r in enumerate(rows) if "k_amplify_init" in r[2]]
i0, i1 = starts[-3], starts[-2]
step = rows[i0:i1]
t0 = step[0][0]
busy = sum(e - s for s, e, _ in step); wall = rows[i1][0] - t0
print("launches %d  busy %.1f us  wall %.1f us" % (len(step), busy / 1e3, wall / 1e3))
prev_end = t0
out = []
for s, e, n in step:
    out.append("%8.1f gap %6.1f dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(n)))
    prev_end = e
open("gpurun_out/timeline.txt", "w").write("\n".join(out) + "\n")
import collections
g = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = t0
for s, e, n in step:
    k = short(n); g[k][0] += 1; g[k][1] += (e - s) / 1e3; g[k][2] += (s - prev_end) / 1e3; prev_end = e
for k, v in sorted(g.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-36s n %3d  dur %7.1f  gap-before %7.1f" % (k, v[0], v[1], v[2]))
P

#!/usr/bin/env python3
"""Static instruction counts of one kernel of a `--save-temps -gline-tables-only` assembly listing, by source line.

    hipcc ... -gline-tables-only --save-temps -c scs_k_reads.hip      (in a scratch directory)
    tools/isa_lines.py scs_k_reads-hip-amdgcn-amd-amdhsa-gfx950.s _ZN3scs7k_readsILb1ELi64ELi1E [--file scs_k_reads.hip] [--top 40]

Every instruction is charged to the innermost line of the listed file that its .loc chain names (inlined lambdas and helpers count
where they are written).  Prints VALU / SALU / LDS / memory instructions per line and per basic block with the blocks' back edges
(loops), so that a trip count known from the source turns the table into instructions per wave."""
import re, sys, collections, argparse
ap = argparse.ArgumentParser(); ap.add_argument("asm"); ap.add_argument("prefix"); ap.add_argument("--file", default="scs_k_reads.hip"); ap.add_argument("--top", type=int, default=40)
ap.add_argument("--blocks", action="store_true")
a = ap.parse_args()
lines = open(a.asm, errors="replace").read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(a.prefix) and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = m.group(3) or m.group(2)
want = {k for k, v in files.items() if v.endswith(a.file)}
def kind(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "mem"
    return "other"
per_line = collections.defaultdict(collections.Counter); per_block = collections.OrderedDict(); cur_line = 0; blk = "entry"; per_block[blk] = collections.Counter(); order = {blk: 0}
edges = []
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith(";"): continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        if int(m.group(1)) in want: cur_line = int(m.group(2))
        continue
    m = re.match(r'(\.LBB\d+_\d+):', l)
    if m: blk = m.group(1); per_block[blk] = collections.Counter(); order[blk] = len(order); continue
    if l.startswith("."): continue
    op = l.split()[0]; k = kind(op)
    per_line[cur_line][k] += 1; per_block[blk][k] += 1
    if op.startswith("s_cbranch") or op == "s_branch": edges.append((blk, l.split()[1]))
tot = collections.Counter()
for c in per_line.values(): tot.update(c)
print("kernel total:", dict(tot))
print("\nback edges (loops): block -> target, instructions between")
for b, t in edges:
    if t in order and order[t] <= order[b]:
        span = collections.Counter()
        for k2, c in per_block.items():
            if order[t] <= order[k2] <= order[b]: span.update(c)
        print(f"  {b} -> {t}: {dict(span)}")
print(f"\ntop {a.top} source lines by VALU:")
for ln, c in sorted(per_line.items(), key=lambda kv: -kv[1]["valu"])[:a.top]:
    print(f"  line {ln:5d}: valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:4d} mem {c['mem']:4d}")
if a.blocks:
    for b, c in per_block.items(): print(b, dict(c))

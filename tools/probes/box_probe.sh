#!/bin/bash
# What the GPU box offers a FASTQ sink: cores, memory, file systems and their buffered-write rates (dd, 1 and 8 files).
out=gpurun_out/box_probe.log
{
echo "== nproc $(nproc); affinity $(taskset -pc $$ 2>/dev/null | sed 's/.*: //')"
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null
free -g
df -h /dev/shm /tmp "$PWD" 2>/dev/null
mount | grep -E ' /dev/shm | /tmp | / ' 
for d in /dev/shm /tmp; do
  t=$(mktemp -d -p $d probe.XXXX) || continue
  echo "== $d one file, 8 GB"
  ( time dd if=/dev/zero of=$t/a bs=16M count=512 2>&1 | tail -1 ) 2>&1 | grep -E 'copied|real'
  echo "== $d eight files in parallel, 4 GB each"
  s=$(date +%s.%N)
  for k in 1 2 3 4 5 6 7 8; do dd if=/dev/zero of=$t/p$k bs=16M count=256 2>/dev/null & done; wait
  e=$(date +%s.%N); echo "32 GB in $(echo "$e - $s" | bc) s"
  s=$(date +%s.%N); rm -rf $t; e=$(date +%s.%N); echo "rm of 40 GB: $(echo "$e - $s" | bc) s"
done
rocm-smi --showmeminfo vram 2>/dev/null | head -8
} > $out 2>&1
cat $out

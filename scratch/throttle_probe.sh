#!/bin/bash
# where does the process group hit its cgroup CPU quota?  nr_throttled / throttled_usec around runs of the bench
st() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | awk '{printf "%s ", $2}'; }
echo "start: $(st)"
python bench.py --timed-only --no-cpu-baseline > /dev/null 2>&1; echo "timed-only (resident, 4 in flight): $(st)"
python bench.py --no-cpu-baseline > /dev/null 2>&1; echo "full without cpu baseline (adds profile + host-witness legs): $(st)"
python bench.py > /dev/null 2>&1; echo "full default: $(st)"

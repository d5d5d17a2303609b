#!/bin/bash
# What the GPU box offers: CPU share, rocprofv3 counters that could separate HBM from Infinity Cache.
echo "nproc: $(nproc)"; python3 -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"
cat /sys/fs/cgroup/cpu.max 2>/dev/null || cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"
cd /tmp; export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --list-avail > "$GRAFT_REPO_ROOT/gpurun_out/avail.txt" 2>&1 || timeout -k 5 120 rocprofv3 -L > "$GRAFT_REPO_ROOT/gpurun_out/avail.txt" 2>&1
grep -c . "$GRAFT_REPO_ROOT/gpurun_out/avail.txt"
grep -i -E "MALL|DRAM|HBM|EA0_RDREQ|EA0_WRREQ|TCC_EA" "$GRAFT_REPO_ROOT/gpurun_out/avail.txt" | head -80

cd $GRAFT_REPO_ROOT
python scripts/single_proof_breakdown.py 2>&1 | tail -25
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; python -c "
import os; print(len(os.sched_getaffinity(0)), os.cpu_count())"

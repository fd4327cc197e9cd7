"""What the host of the GPU box really offers (cpu_baseline.cores must be the threads that can run, not os.cpu_count())."""
import os
print("os.cpu_count()", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "n/a")
os.system("lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ; nproc")

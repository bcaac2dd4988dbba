"""Prints what the host of a GPU box really grants: cores (affinity vs os.cpu_count vs cgroup quota), memory, CPU model.
Used to state `cpu_baseline.cores` honestly (os.cpu_count() is not the affinity mask)."""
import json
import os


def probe():
    out = {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            out[f] = open(f).read().strip()
        except OSError:
            pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                out["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        for line in open("/proc/meminfo"):
            if line.startswith(("MemTotal", "MemAvailable")):
                out[line.split(":")[0]] = line.split(":")[1].strip()
    except OSError:
        pass
    out["OMP_NUM_THREADS"] = os.environ.get("OMP_NUM_THREADS")
    out["OMP_PROC_BIND"] = os.environ.get("OMP_PROC_BIND")
    return out


def usable_cores():
    """min(affinity mask, cgroup CPU quota) -- the number of threads that can really run at once."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, q // per))
        except (OSError, ValueError):
            pass
    return n


if __name__ == "__main__":
    print(json.dumps(probe()))
    print("usable_cores", usable_cores())

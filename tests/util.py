"""Shared helpers of the parity tests."""
import os

import numpy as np


def rownorm_sign(R):
    """Flip rows so that the diagonal is non-negative (row-sign normalisation of R)."""
    R = np.array(R, copy=True)
    k = min(R.shape)
    for i in range(k):
        if R[i, i] < 0:
            R[i, :] = -R[i, :]
    return R


def r_rel_diff(R, Rref):
    """max |abs(R) - abs(Rref)| / max |Rref| over the upper trapezoid."""
    k = min(R.shape[0], Rref.shape[0])
    A = np.triu(np.abs(R[:k]))
    B = np.triu(np.abs(Rref[:k]))
    return float(np.max(np.abs(A - B)) / max(np.max(np.abs(B)), 1e-300))


def gram_resid(R, A):
    """|| triu(R)' triu(R) - A'A ||_1 / || A'A ||_1, the reference's dense-demo and
    check_r_factor metric (GPUQREngine/Demo/gpuqrengine_demo.cpp:94-100,
    SPQR/Tcov/qrtest.cpp:1218-1257)."""
    R = np.triu(R)
    G = A.T @ A
    return float(np.linalg.norm(R.T @ R - G, 1) / max(np.linalg.norm(G, 1), 1e-300))


def demo_rand_front(m, n, seed_state=None):
    """Entries rand()/1e8 as in gpuqrengine_demo.cpp:21-28; glibc rand() is replaced
    by a seeded numpy generator of the same range [0, 2^31) (the C library's
    generator is not reproducible across platforms)."""
    rng = np.random.default_rng(1 if seed_state is None else seed_state)
    return rng.integers(0, 2**31, size=(m, n)).astype(np.float64) / 1e8


def stair_front(m, n, rng, aggression=4):
    """Staircase-shaped front like randfill_stair (gpuqrengine_demo.cpp:32-44)."""
    F = rng.uniform(-1, 1, size=(m, n))
    last = 0
    for j in range(n):
        last = min(m - 1, last + int(rng.integers(0, aggression + 1)))
        F[last + 1:, j] = 0.0
    return F


def gpu_state_report(trace_prefix=None, world=0):
    """For the message of a failed child-process test: what this (parent) process sees of the device, who
    else holds memory on it, and where the ranks were (their MGPU_TRACE files)."""
    import subprocess
    out = []
    try:
        import torch
        free, total = torch.cuda.mem_get_info()
        out.append("parent sees %.2f GiB free of %.2f GiB; torch reserved %.2f GiB" % (
            free / 2**30, total / 2**30, torch.cuda.memory_reserved() / 2**30))
    except Exception as ex:
        out.append("mem_get_info failed: %r" % (ex,))
    try:
        keep = ("VmRSS", "VmHWM", "VmLck", "VmPin", "VmSize", "Threads")
        out.append("parent /proc/self/status: " + ", ".join(
            ln.strip().replace("\t", " ") for ln in open("/proc/self/status") if ln.startswith(keep)))
        keep = ("MemTotal", "MemFree", "MemAvailable", "Unevictable", "Mlocked", "Shmem:", "Committed_AS", "CommitLimit")
        out.append("/proc/meminfo: " + ", ".join(
            " ".join(ln.split()) for ln in open("/proc/meminfo") if ln.startswith(keep)))
        out.append("open fds of the parent: %d" % len(os.listdir("/proc/self/fd")))
    except OSError as ex:
        out.append("proc files: %r" % (ex,))
    try:
        # what the kernel driver knows: processes with a compute context, their queues and memory
        base = "/sys/class/kfd/kfd/proc"
        procs = sorted(os.listdir(base))
        lines = []
        for pid in procs:
            d = os.path.join(base, pid)
            nq = len(os.listdir(os.path.join(d, "queues"))) if os.path.isdir(os.path.join(d, "queues")) else -1
            extra = []
            for f in sorted(os.listdir(d)):
                if f.startswith(("vram_", "sdma_")):
                    try:
                        extra.append("%s=%s" % (f, open(os.path.join(d, f)).read().strip()))
                    except OSError:
                        pass
            lines.append("pid %s (self=%s): %d queues %s" % (pid, pid == str(os.getpid()), nq, " ".join(extra)))
        out.append("kfd processes: %d\n  " % len(procs) + "\n  ".join(lines))
    except OSError as ex:
        out.append("kfd sysfs: %r" % (ex,))
    for cmd in (["rocm-smi", "--showmeminfo", "vram"], ["sh", "-c", "dmesg 2>&1 | tail -40"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            out.append("$ %s\n%s" % (" ".join(cmd), (r.stdout + r.stderr)[-1500:]))
        except Exception as ex:
            out.append("%s failed: %r" % (cmd[0], ex))
    try:
        # what this process has mapped, by backing file, and its largest mappings
        groups, big = {}, []
        for ln in open("/proc/self/maps"):
            f = ln.split()
            lo, hi = (int(x, 16) for x in f[0].split("-"))
            name = f[5] if len(f) > 5 else "[anon]"
            key = name if name.startswith(("/dev/", "[")) else ("libs/files" if name.startswith("/") else name)
            groups[key] = groups.get(key, 0) + hi - lo
            big.append((hi - lo, f[0], f[1], name))
        out.append("mappings by backing: " + ", ".join("%s %.2f GiB" % (k, v / 2**30) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])[:8]))
        out.append("largest mappings: " + "; ".join("%.2f GiB %s %s %s" % (sz / 2**30, a, pr, nm) for sz, a, pr, nm in sorted(big, reverse=True)[:12]))
    except Exception as ex:
        out.append("maps: %r" % (ex,))
    out.append("parent attributes:\n" + process_attrs())
    for r in range(world):
        try:
            out.append("-- rank %d --\n%s" % (r, open("%s.%d" % (trace_prefix, r)).read()))
        except OSError:
            out.append("-- rank %d: no trace --" % r)
    return "\n".join(out)


def process_attrs():
    """What a child process inherits from this one besides the environment: CPU affinity, memory
    policy, signal mask and dispositions, resource limits."""
    import ctypes
    import resource
    import signal
    out = []
    try:
        out.append("affinity: %d cpus %s" % (len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8]))
        out.append("blocked signals: %s" % sorted(int(x) for x in signal.pthread_sigmask(signal.SIG_BLOCK, [])))
        keep = ("SigBlk", "SigIgn", "SigCgt", "Cpus_allowed_list", "Mems_allowed_list", "NoNewPrivs", "THP_enabled", "Seccomp:")
        out.append(", ".join(" ".join(ln.split()) for ln in open("/proc/self/status") if ln.startswith(keep)))
        libc = ctypes.CDLL(None, use_errno=True)
        mode = ctypes.c_int(-1)
        mask = (ctypes.c_ulong * 16)()
        rc = libc.syscall(239, ctypes.byref(mode), mask, ctypes.c_ulong(1024), None, ctypes.c_ulong(0))   # get_mempolicy
        out.append("mempolicy: rc %d mode %d nodemask %s" % (rc, mode.value, [hex(x) for x in mask if x]))
        names = [n for n in dir(resource) if n.startswith("RLIMIT_")]
        out.append("rlimits: " + ", ".join("%s=%s" % (n[7:], resource.getrlimit(getattr(resource, n))) for n in names))
        out.append("personality: %s" % open("/proc/self/personality").read().strip())
    except Exception as ex:
        out.append("process_attrs: %r" % (ex,))
    return "\n".join(out)


def fresh_process_attrs():
    """preexec_fn for child processes that must look like processes started from a shell: all CPUs, the
    default memory policy, no blocked signals, whatever this (long-lived, library-laden) parent's calling
    thread has ended up with."""
    import ctypes
    import signal
    try:
        os.sched_setaffinity(0, range(os.cpu_count() or 1))
    except OSError:
        pass
    try:
        signal.pthread_sigmask(signal.SIG_SETMASK, [])
        ctypes.CDLL(None).syscall(238, 0, None, ctypes.c_ulong(0))     # set_mempolicy (MPOL_DEFAULT)
    except Exception:
        pass

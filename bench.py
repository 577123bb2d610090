#!/usr/bin/env python3
"""Benchmark of the numeric QR hot path on MI355X.

Workload (N = 1): BASELINE.json configs[1], the GPUQREngine stand-alone batch of random dense
fronts: every (fm, fn) in {96,192,384,768,1536,3072,4096} x {32,64,128,256,512,1024,2048}, 12 of
each (588 fronts, 3.96 GB of FP64), uniform(-1,1) entries, seed 1.  A step is one call of the
engine's front-batch entry point on the whole batch with the fronts already resident in HBM
(device buffers handed over through the sparse overload, R left on the device).  The metric is
the reference's canonical flop count for dense QR, 2n^2(m - n/3) (m >= n) or 2m^2(n - m/3)
(GPUQREngine/Demo/gpuqrengine_demo.cpp:203-212), divided by wall time.

With --gpus N > 1 every rank factorizes its own copy of the batch (independent fronts shard with
no exchange): weak scaling, value = N * flops / max-over-ranks time.

--workload sparse2d:K / sparse3d:K time the sparse path (symbolic analysis excluded) on the
stacked grid least-squares problems instead.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FM = [96, 192, 384, 768, 1536, 3072, 4096]
FN = [32, 64, 128, 256, 512, 1024, 2048]
COPIES = 12
PEAK_FP64_MFMA_TFLOPS = 78.6      # MI355X FP64 matrix peak = half the FP32-input MFMA rate (157.3 TF)
PEAK_HBM_GBS = 8000.0
KERNEL_NAMES = ["k_factorize<4, 8>", "k_factorize<4, 16>", "k_factorize<8, 16>", "k_factorize<16, 16>",
                "k_apply<2>", "k_apply<4>", "k_apply<8>", "k_apply<16>", "k_assemble",
                "k_factorize_all", "k_apply_small", "k_apply_all"]


def canon_flops(m, n):
    return 2.0 * n * n * (m - n / 3.0) if m >= n else 2.0 * m * m * (n - m / 3.0)


def dense_shapes(copies=COPIES):
    return [(m, n) for m in FM for n in FN for _ in range(copies)]


def cpu_baseline_dense():
    """The C oracle (a port, not the reference binary) on the host cores, on a bounded sample of the
    same workload: one front of every one of the 49 shapes (1/12 of the batch), OpenMP over fronts."""
    from oracle import oracle
    rng = np.random.default_rng(1)
    shapes = [(m, n) for m in FM for n in FN]
    mats = [rng.uniform(-1, 1, size=s) for s in shapes]
    cores = min(16, os.cpu_count() or 1)
    sec = oracle.front_qr_batch(mats, cores)
    fl = sum(canon_flops(*s) for s in shapes)
    return {"value": fl / sec / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": "one front of each of the %d shapes = 1/12 of the batch (%.1f GFLOP, %.1f s)"
                      % (len(shapes), fl / 1e9, sec)}


def cpu_baseline_sparse(kind, k):
    """The C oracle's whole-tree multifrontal QR (a port; serial over fronts) on a bounded sample of the
    same generator: the grid shrunk until the run takes seconds."""
    import time as _t
    from oracle import oracle
    from suitesparse_amd import problems, spqr
    ks = min(k, 640 if kind == "sparse2d" else 36)      # about 10-20 s on one core
    shape = (ks, ks) if kind == "sparse2d" else (ks, ks, ks)
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=8), relax=20, gpuMemorySize=200 * 2**30)
    Sx = spqr.s_values(sym, A)
    t0 = _t.perf_counter()
    oracle.multifrontal(sym, Sx)
    sec = _t.perf_counter() - t0
    return {"value": sym.flopcount / sec / 1e9, "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": "%s grid %s of the same generator (%.2f GFLOP, %.1f s)"
                      % (kind, "x".join(map(str, shape)), sym.flopcount / 1e9, sec)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dense")
    ap.add_argument("--copies", type=int, default=COPIES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import suitesparse_amd as ss
    from suitesparse_amd import _lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (the ranks
    # then share devices); the driver's runs use RCCL, one GPU per rank
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    L = _lib.lib()
    K, W = args.steps, args.warmup

    if args.workload == "dense":
        shapes = dense_shapes(args.copies)
        total = sum(m * n for m, n in shapes)
        flops = sum(canon_flops(m, n) for m, n in shapes)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1 + rank)
        src = torch.rand(total, dtype=torch.float64, device=dev, generator=gen) * 2 - 1
        free, _ = torch.cuda.mem_get_info()
        nbuf = max(1, min(K + W, int(free * 0.8) // (total * 8)))
        bufs = [torch.empty_like(src) for _ in range(nbuf)]

        def make_fronts(buf):
            fronts, off = [], 0
            for k, (m, n) in enumerate(shapes):
                f = ss.Front(k, ss.EMPTY, m, n)
                f.gpuF = buf.data_ptr() + off * 8
                off += m * n
                fronts.append(f)
            return fronts

        front_sets = [ss.FrontArray(make_fronts(b)) for b in bufs]     # the Front[] a C caller would keep

        def prepare(i):
            bufs[i % nbuf].copy_(src)

        def step(i, stats=None):
            code = ss.GPUQREngine(0, front_sets[i % nbuf], len(shapes), stats=stats)
            assert code == ss.QRENGINE_SUCCESS, code

        workload = "dense-front-batch %dx{%s}x{%s}" % (args.copies, ",".join(map(str, FM)), ",".join(map(str, FN)))
        extra = {"fronts": len(shapes), "bytes": total * 8}
    else:
        from suitesparse_amd import problems, spqr
        kind, k = args.workload.split(":")
        k = int(k)
        shape = (k, k) if kind == "sparse2d" else (k, k, k)
        A = problems.stacked_ls(shape)
        Q = problems.nested_dissection(shape, leaf=8)
        sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=200 * 2**30)
        flops = sym.flopcount
        workload = "%s grid %s stacked LS, ND order, m=%d n=%d nf=%d" % (kind, "x".join(map(str, shape)),
                                                                          A.shape[0], A.shape[1], sym.nf)
        extra = {"fronts": sym.nf, "stages": sym.numStages, "nnzR": sym.rnz}

        def prepare(i):
            pass

        from suitesparse_amd import multigpu

        def step(i, stats=None):
            if world > 1:
                # subtrees on all ranks, contribution blocks of the cut to rank 0 over RCCL
                _, info = multigpu.factorize_distributed(sym, A, gather_R=False)
                if stats is not None:
                    stats.kernelTime = info["phase1"]["kernel_ms"] + info.get("phase2", {}).get("kernel_ms", 0.0)
                    stats.numLaunches = info["phase1"]["launches"] + info.get("phase2", {}).get("launches", 0)
                return
            res = spqr.factorize(sym, A, want_R=False)      # R stays as packed blocks; CSC conversion is not on the path
            if stats is not None:
                stats.kernelTime, stats.numLaunches, stats.flopsActual = (
                    res.gpuKernelTime, res.gpuNumKernelLaunches, res.gpuFlops)
        nbuf = 1

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        prepare(i)
        step(i)
    for i in range(min(K, nbuf)):
        prepare(W + i)
    kernel_ms = 0.0
    launches = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        if i >= nbuf:
            prepare(W + i)          # not enough memory for one buffer per step: restore inside the timed region
        st = ss.QREngineStats()
        step(W + i, st)
        kernel_ms += st.kernelTime
        launches = st.numLaunches
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one more step with per-kernel HIP-event timing for the roofline of the dominant kernel
    roofline = None
    per_kernel = {}
    if rank == 0:
        import ctypes as C
        L.gpuqr_profile_reset()
        L.gpuqr_profile_enable(1)
        prepare(0)
        step(0)
        L.gpuqr_profile_enable(0)
        L.gpuqr_profile_get.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_double), C.POINTER(C.c_double)]
        best = None
        for kid, name in enumerate(KERNEL_NAMES):
            ms, fl, by = C.c_double(0), C.c_double(0), C.c_double(0)
            ln, tk = C.c_int64(0), C.c_int64(0)
            L.gpuqr_profile_get(kid, C.byref(ms), C.byref(ln), C.byref(tk), C.byref(fl), C.byref(by))
            if ln.value == 0:
                continue
            per_kernel[name] = {"ms": round(ms.value, 3), "launches": ln.value, "tasks": tk.value,
                                "gflops": round(fl.value / 1e9, 2), "gbytes": round(by.value / 1e9, 3)}
            if best is None or ms.value > best[1]:
                best = (name, ms.value, ln.value, fl.value, by.value)
        name, ms, ln, fl, by = best
        # HBM bytes per launch from the hardware counters: collected with rocprofv3 --pmc in separate
        # passes of this same command (a profiler cannot run inside the timed process) and committed
        traffic = None
        try:
            pmc = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")))
            if args.workload == "dense" and args.copies == COPIES and name in pmc["kernels"]:
                traffic = pmc["kernels"][name]["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
        avg_s = ms / 1e3 / ln
        tflops = fl / ln / avg_s / 1e12
        gbs = by / ln / avg_s / 1e9
        # the 32-column block reflector does about 8 flop per byte of A it must move, below the
        # chip's FP64 ridge (78.6 TF / 8 TB/s = 9.8): HBM bounds it
        if name.startswith("k_apply") or name.startswith("k_assemble"):
            roofline = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": gbs / PEAK_HBM_GBS, "traffic": traffic,
                        "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)"
                        if traffic is not None else None,
                        "avg_launch_us": avg_s * 1e6,
                        "launches": ln, "algorithmic_bytes_per_launch": by / ln,
                        "mfma_tflops": tflops, "mfma_frac": tflops / PEAK_FP64_MFMA_TFLOPS}
        else:
            roofline = {"kernel": name, "bound": "mfma", "achieved": tflops, "peak": PEAK_FP64_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                        "avg_launch_us": avg_s * 1e6, "launches": ln}

    # the assembly kernels (S entries and contribution blocks) against the same HBM peak
    roofline_assembly = None
    if rank == 0 and "k_assemble" in per_kernel and per_kernel["k_assemble"]["ms"] > 0:
        ka = per_kernel["k_assemble"]
        gbs_a = ka["gbytes"] / (ka["ms"] / 1e3)
        roofline_assembly = {"kernel": "k_assemble", "bound": "hbm", "achieved": gbs_a, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": gbs_a / PEAK_HBM_GBS, "traffic": None,
                             "avg_launch_us": ka["ms"] * 1e3 / ka["launches"], "launches": ka["launches"]}

    if rank == 0:
        strong = args.workload != "dense"          # the sparse problems are one fixed job split over the ranks
        value = (1 if strong else world) * flops * K / elapsed / 1e9
        out = {
            "metric": "numeric QR factor GFLOP/s", "value": value, "unit": "GFLOP/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": dict({"workload": workload}, **extra),
            "kernel_ms_per_step": kernel_ms / K, "launches_per_step": launches,
            "pct_fp64_mfma_peak": 100.0 * value / world / 1e3 / PEAK_FP64_MFMA_TFLOPS,
            "roofline": roofline, "per_kernel": per_kernel,
        }
        if roofline_assembly is not None:
            out["roofline_assembly"] = roofline_assembly
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = (cpu_baseline_dense() if args.workload == "dense"
                                   else cpu_baseline_sparse(*[f(x) for f, x in zip((str, int), args.workload.split(":"))]))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()                  # rank 0 is still profiling and printing while the others arrive here
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

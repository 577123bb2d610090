#!/usr/bin/env python3
"""Benchmark of the numeric sparse QR hot path on MI355X.

Headline workload = BASELINE.json configs[3]: numeric multifrontal QR of the 3D 7-point grid
least-squares matrix A = [L ; sqrt(0.1) I] on a 128^3 grid (m = 4,194,304, n = 2,097,152), columns in
the reference's METIS order, fronts as SPQR's analysis makes them (CHOLMOD's relaxed supernodes), no
tolerance.  A step is one numeric factorization through the SPQR glue's C entry point (S built on the
device from A's values, front assembly, panel factorization, block-reflector updates, R packed and
copied to the host stack).  The metric is the reference's canonical flop count of the symbolic analysis
(SPQR/Source/spqr_analyze.cpp:888-902, = cc->SPQR_flopcount) divided by wall time.

The column order is an INPUT of the path (the reference computes it in spqr_1colamd before the
numeric phase).  It comes from the reference's own METIS binding, run on the host through the
reference build that also provides the CPU baseline (oracle/_ref); without that build the geometric
nested-dissection order of suitesparse_amd.problems is used and the line says so.

--gpus N > 1: the SAME problem on N ranks (strong scaling).  The assembly tree is mapped onto the
ranks (suitesparse_amd/multigpu.py), every rank factorizes its fronts, contribution blocks cross ranks
device to device over RCCL.  Without a launcher (WORLD_SIZE unset) the N ranks are started here.

Which number is which: "value" = re-factorization on one symbolic object, as a solver inside an
iteration does (pattern work and the engine's schedule reused, "refactorization": true);
"value_first_factorization" = the first call on a fresh symbolic object, the region the reference
times as SPQR_factorize_time (SPQR/Source/spqr_1factor.cpp:628-638: assembly maps, S, the numeric
factorization; here also the device workspaces, which this fork allocates at symbolic time).

"verified": after the timed steps and outside them, the R of the last one is checked the way the
reference's test suite checks an R factor (check_r_factor, SPQR/Tcov/qrtest.cpp:1218-1257, bar 1e-10),
through a random vector: max |R'(R y) - (AE)'((AE) y)| / max |(AE)'((AE) y)|.

Extras at N = 1 (same JSON line, key "extras"): the 2D 1024^2 grid (configs[2]), the dense front batch
(configs[1]) and the 96^3 grid of the earlier rounds' headline.
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
                "k_factorize_all", "k_apply_small", "k_apply_all", "k_round_all"]
HEADLINE = "sparse3d:128"
# device memory declared to the staging (cc->gpuMemorySize); BENCH_DEVICE_GIB overrides it, e.g. when ranks of a
# rehearsal share one device
DECLARED_DEVICE_MEMORY = int(float(os.environ.get("BENCH_DEVICE_GIB", "200")) * 2**30)
T_START = time.perf_counter()


def log(msg):
    """Progress on stderr (the JSON line is the only thing on stdout): long phases say where they are."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %6.1f s] %s\n" % (time.perf_counter() - T_START, msg))
        sys.stderr.flush()


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


def reference_available():
    try:
        from oracle import ref
        return ref.available()
    except Exception:
        return False


def column_order(cx, A, shape, want):
    """The column order the factorization is given: the reference's METIS order of A'A (rank 0 runs the
    reference's binding on the host and hands the permutation to the other ranks), or the geometric
    nested dissection when the reference build is absent or --ordering nd asks for it."""
    import torch
    import torch.distributed as dist
    from suitesparse_amd import problems
    n = A.shape[1]
    use_ref = (want == "metis") and reference_available()
    t0 = time.perf_counter()
    if not use_ref:
        return problems.nested_dissection(shape, leaf=8), "geometric nested dissection (suitesparse_amd.problems)", 0.0
    if cx.rank == 0:
        from oracle import ref
        # (a child process by default: METIS takes and frees GBs of heap; BENCH_METIS_INPROCESS=1 runs it here,
        # for runs under a profiler that should see one process only)
        order = ref.metis_order if os.environ.get("BENCH_METIS_INPROCESS") else ref.metis_order_child
        Q = np.ascontiguousarray(order(A), dtype=np.int64)
    else:
        Q = np.zeros(n, dtype=np.int64)
    if cx.world > 1:
        t = torch.from_numpy(Q).to(cx.dev if cx.backend == "nccl" else "cpu")
        dist.broadcast(t, src=0)
        Q = t.cpu().numpy()
    return Q, "METIS on A'A through the reference's cholmod_l_metis (SPQR_ORDERING_METIS)", time.perf_counter() - t0


def cpu_baseline_sparse(kind, k):
    """The REFERENCE's own CPU path (SuiteSparseQR -> spqr_1factor -> spqr_front on MKL's threaded
    LAPACK/BLAS, built from the reference's sources into oracle/_ref) timed on this host's cores, on a
    bounded sample of the same generator and ordering: the reference's own SPQR_factorize_time and
    flop count.  Without that build: the LAPACK-per-front port of the oracle."""
    from suitesparse_amd import problems
    ks = min(k, 1024 if kind == "sparse2d" else 48)      # (3D 48^3: 1.6 TFLOP, 10-20 s of the reference on 16 cores)
    shape = (ks, ks) if kind == "sparse2d" else (ks, ks, ks)
    A = problems.stacked_ls(shape)
    if reference_available():
        from oracle import ref
        cores = ref.threads()
        log("CPU baseline: the reference's SuiteSparseQR on %s, %d threads ..." % ("x".join(map(str, shape)), cores))
        r = ref.spqr(A, ordering=ref.ORDERING_METIS, tol=ref.NO_TOL, symbolic=True, want_R=False)
        return {"value": r.flopcount / r.factorize_time / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "reference",
                "sample": "%s grid %s of the same generator, METIS order: the reference's SuiteSparseQR on the host "
                          "(SPQR_factorize_time %.2f s, SPQR_analyze_time %.2f s, SPQR_flopcount %.2f GFLOP, %d fronts)"
                          % (kind, "x".join(map(str, shape)), r.factorize_time, r.analyze_time, r.flopcount / 1e9, r.nf)}
    from oracle import oracle
    from suitesparse_amd import spqr
    ks = min(ks, 640 if kind == "sparse2d" else 40)
    shape = (ks, ks) if kind == "sparse2d" else (ks, ks, ks)
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=8), relax=20, gpuMemorySize=DECLARED_DEVICE_MEMORY)
    Sx = spqr.s_values(sym, A)
    try:
        from threadpoolctl import threadpool_info
        cores = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
    except ImportError:
        cores = 1
    _, t_qr, sec = oracle.multifrontal_lapack(sym, Sx, keep_R=False)
    return {"value": sym.flopcount / sec / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": "%s grid %s of the same generator, LAPACK dgeqrf per front on threaded BLAS (%.2f GFLOP, %.1f s)"
                      % (kind, "x".join(map(str, shape)), sym.flopcount / 1e9, sec)}


class Ctx:
    pass


def read_profile(L):
    """Per-kernel totals of the profiled engine calls since the last reset."""
    import ctypes as C
    L.gpuqr_profile_get.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
    per_kernel, best = {}, None
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
    return per_kernel, best


def pmc_traffic(kernel, workload_tag):
    """HBM bytes per launch of `kernel` from the committed counter summary of this workload, or None."""
    for fname in ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fname)))
            for entry in (pmc if isinstance(pmc, list) else [pmc]):
                if entry.get("workload_tag", "dense") == workload_tag and kernel in entry["kernels"]:
                    return (entry["kernels"][kernel]["traffic_bytes_per_launch"],
                            "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)" % fname)
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def isolated_assembly():
    """The assembly kernel alone (one launch on a 16,384^2 block, inputs resident), from the committed output
    of scripts/ubench/assemble_bench.hip: the figure to hold against the chip's peak; what a step's
    own launches reach beside the update kernel is `achieved`."""
    import re
    path = os.path.join(ROOT, "profiles", "r03_ubench_assemble.txt")
    try:
        out = {}
        for ln in open(path):
            if ln.startswith("device-to-device copy") and "device_copy_GBps" not in out:
                out["device_copy_GBps"] = float(re.search(r"([\d.]+) TB/s", ln).group(1)) * 1e3
            m = re.search(r"([\d.]+) TB/s = ([\d.]+) of 8 TB/s", ln)
            if not m:
                continue
            if ln.startswith("pack-assembly (runs in the maps) 16384") and "pack_assembly_GBps" not in out:
                out["pack_assembly_GBps"] = float(m.group(1)) * 1e3
                out["pack_assembly_frac"] = float(m.group(2))
            elif ln.startswith("pack-assembly (scattered maps)"):
                out["pack_assembly_scattered_maps_GBps"] = float(m.group(1)) * 1e3
            elif ln.startswith("S-assembly"):
                out["s_assembly_GBps"] = float(m.group(1)) * 1e3
        if out:
            out["source"] = "profiles/r03_ubench_assemble.txt (scripts/ubench/assemble_bench.hip, one launch each)"
            return out
    except OSError:
        pass
    return None


def rooflines(per_kernel, best, workload_tag):
    """`roofline` of the kernel with the largest total time, `roofline_assembly` of k_assemble."""
    roofline = roofline_assembly = None
    total_ms = sum(v["ms"] for v in per_kernel.values()) or 1.0
    if best is not None:
        name, ms, ln, fl, by = best
        # HBM bytes per launch from the hardware counters: collected with rocprofv3 --pmc in separate
        # passes of this same command (a profiler cannot run inside the timed process) and committed
        traffic, traffic_src = pmc_traffic(name, workload_tag)
        avg_s = ms / 1e3 / ln
        tflops = fl / ln / avg_s / 1e12
        gbs = by / ln / avg_s / 1e9
        # The block-reflector update moves 16 bytes per entry of the trailing matrix for about 4 w flops
        # (w reflectors): 8 flop/B at w = 32, 16 at w = 64, against the chip's FP64 ridge of 78.6 TF / 8 TB/s
        # = 9.8.  The panel kernels issue no MFMA and are bound by latency; they are priced against the
        # memory roof too (their bytes are what they must move) and their share of the time is what counts.
        roofline = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": avg_s * 1e6, "launches": ln, "algorithmic_bytes_per_launch": by / ln,
                    "time_share": ms / total_ms}
        if name.startswith("k_apply") or name.startswith("k_round"):
            roofline.update({"mfma_tflops": tflops, "mfma_frac": tflops / PEAK_FP64_MFMA_TFLOPS,
                             "flop_per_byte": fl / max(by, 1.0)})
        elif name.startswith("k_factorize"):
            roofline["note"] = "panel kernel: latency-bound (one barrier per column), no MFMA; see time_share"
    if "k_assemble" in per_kernel and per_kernel["k_assemble"]["ms"] > 0:
        ka = per_kernel["k_assemble"]
        gbs_a = ka["gbytes"] / (ka["ms"] / 1e3)
        traffic_a, _ = pmc_traffic("k_assemble", workload_tag)
        roofline_assembly = {"kernel": "k_assemble", "bound": "hbm", "achieved": gbs_a, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": gbs_a / PEAK_HBM_GBS, "traffic": traffic_a,
                             "avg_launch_us": ka["ms"] * 1e3 / ka["launches"], "launches": ka["launches"],
                             "algorithmic_bytes_per_launch": ka["gbytes"] * 1e9 / ka["launches"],
                             "time_share": ka["ms"] / total_ms,
                             "note": "launches beside the updates (contribution blocks travel 32 rows at a time); "
                                     "the kernel alone on large blocks: scripts/ubench/assemble_bench.hip, "
                                     "profiles/r03_ubench_assemble.txt",
                             "isolated": isolated_assembly()}
    return roofline, roofline_assembly


def time_steps(cx, step, prepare, K, W, nbuf=1):
    """W untimed steps, then K timed ones between barriers; max over ranks.  Returns (seconds, stats)."""
    import torch
    import torch.distributed as dist
    import suitesparse_amd as ss

    def barrier():
        torch.cuda.synchronize()
        if cx.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(W):
        prepare(i)
        step(i)
        log("warm-up step %d of %d done" % (i + 1, W))
    for i in range(min(K, nbuf)):
        prepare(W + i)
    kernel_ms, launches = 0.0, 0
    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        if i >= nbuf:
            prepare(W + i)          # not enough memory for one buffer per step: restore inside the timed region
        st = ss.QREngineStats()
        step(W + i, st, last=(i == K - 1))
        kernel_ms += st.kernelTime
        launches = st.numLaunches
        log("timed step %d of %d done" % (i + 1, K))
    barrier()
    elapsed = time.perf_counter() - t0
    if cx.world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cx.dev if cx.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, kernel_ms / max(K, 1), launches


def profile_step(cx, step, prepare):
    """One more step with HIP events around every launch (on the streams the launches run on)."""
    if cx.rank == 0:
        cx.L.gpuqr_profile_reset()
        cx.L.gpuqr_profile_enable(1)
    prepare(0)
    step(0)
    if cx.rank == 0:
        cx.L.gpuqr_profile_enable(0)
        return read_profile(cx.L)
    return {}, None


def run_dense(cx, K, W, copies=COPIES):
    import torch
    import suitesparse_amd as ss
    shapes = dense_shapes(copies)
    total = sum(m * n for m, n in shapes)
    flops = sum(canon_flops(m, n) for m, n in shapes)
    gen = torch.Generator(device=cx.dev)
    gen.manual_seed(1 + cx.rank)
    src = torch.rand(total, dtype=torch.float64, device=cx.dev, generator=gen) * 2 - 1
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

    def step(i, stats=None, last=False):
        code = ss.GPUQREngine(0, front_sets[i % nbuf], len(shapes), stats=stats)
        assert code == ss.QRENGINE_SUCCESS, code

    elapsed, kernel_ms, launches = time_steps(cx, step, prepare, K, W, nbuf)
    # the check of the reference's dense demo (||triu(R)'triu(R) - A'A||_1 / ||A'A||_1 <= 1e-12,
    # GPUQREngine/Demo/gpuqrengine_demo.cpp:94-100) on the largest front of every width, from the last step
    lastbuf = bufs[(W + K - 1) % nbuf]
    worst, off = 0.0, 0
    for (m, n) in shapes:
        if m == FM[-1]:
            Fr = lastbuf[off:off + m * n].view(m, n)
            Ar = src[off:off + m * n].view(m, n)
            R = torch.triu(Fr[:min(m, n)])
            G = Ar.T @ Ar
            worst = max(worst, float((R.T @ R - G).abs().sum(0).max() / G.abs().sum(0).max()))
        off += m * n
    verified = {"check": "||triu(R)'triu(R) - A'A||_1 / ||A'A||_1 on every 4096-row front of the last step "
                         "(gpuqrengine_demo.cpp:94-100, bar 1e-12)", "value": worst, "passed": worst <= 1e-12}
    per_kernel, best = profile_step(cx, step, prepare)
    workload = "dense-front-batch %dx{%s}x{%s}" % (copies, ",".join(map(str, FM)), ",".join(map(str, FN)))
    return dict(workload=workload, tag="dense", flops=flops, elapsed=elapsed, kernel_ms=kernel_ms, launches=launches,
                per_kernel=per_kernel, best=best, scaling="weak", verified=verified,
                config={"fronts": len(shapes), "bytes": total * 8})


def run_sparse(cx, kind, k, K, W, ordering="metis", profile=True):
    """Time the sparse numeric factorization: the cold first call (it is also the first warm-up), W-1
    more warm-ups, K timed steps, the check of the last R, one profiled step."""
    import torch
    from suitesparse_amd import multigpu, problems, spqr
    t0 = time.perf_counter()
    shape = (k, k) if kind == "sparse2d" else (k, k, k)
    A = problems.stacked_ls(shape)
    log("%s %d: matrix generated, ordering ..." % (kind, k))
    Q, order_name, order_s = column_order(cx, A, shape, ordering)
    log("column order in %.1f s (%s), analysis ..." % (order_s, order_name.split(" ")[0]))
    t1 = time.perf_counter()
    # relax = -1: the fronts SPQR's own analysis makes for this column order (bit-identical arrays)
    sym = spqr.analyze(A, Q, relax=-1, gpuMemorySize=DECLARED_DEVICE_MEMORY)
    analysis_s = time.perf_counter() - t1
    workload = "%s grid %s stacked LS (lambda 0.1), no tolerance, m=%d n=%d" % (kind, "x".join(map(str, shape)), A.shape[0], A.shape[1])
    config = {"ordering": order_name, "analysis": "SPQR-compatible (CHOLMOD relaxed supernodes, spqr_amd_analyze relax=-1)",
              "fronts": sym.nf, "stages": sym.numStages, "nnzR": sym.rnz, "m": A.shape[0], "n": A.shape[1],
              "canonical_gflop": sym.flopcount / 1e9, "declared_device_GiB": DECLARED_DEVICE_MEMORY / 2**30,
              "largest_front": [int(sym.Fm.max()), int(sym.maxfn)]}
    mapping = {}
    for parts in sorted({2, 4, 8, cx.world} - {1}):
        owner, level = multigpu.map_tree(sym, parts)
        rep = multigpu.mapping_report(sym, owner, level, parts)
        mapping[str(parts)] = {"top_flops_share": round(rep["top_flops_share"], 4),
                               "speedup_bound": round(rep["speedup_bound"], 3)}
    csc = multigpu._Csc(A)
    last_info, kept = {}, {}

    def prepare(i):
        pass

    def step(i, stats=None, last=False):
        if cx.world > 1:
            _, info = multigpu.factorize_distributed(sym, csc, gather_R=False)
            last_info.update(info)
            if stats is not None:
                stats.kernelTime = sum(c["kernel_ms"] for c in info["calls"])
                stats.numLaunches = sum(c["launches"] for c in info["calls"])
            return
        # R stays as packed blocks on the host stack; the CSC conversion is not on the path.  The last
        # timed step keeps its numeric object for the check that follows the timed region.
        res = spqr.factorize(sym, csc, want_R=False, keep_num=last)
        if last:
            kept["res"] = res
        if stats is not None:
            stats.kernelTime, stats.numLaunches, stats.flopsActual = (
                res.gpuKernelTime, res.gpuNumKernelLaunches, res.gpuFlops)

    # the first call: pattern work (maps, S permutation), workspace allocation, planning and upload
    torch.cuda.synchronize()
    t0c = time.perf_counter()
    step(0)
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0c) * 1e3
    log("analysis %.1f s (%d fronts, %d stages, %.1f TFLOP); first factorization %.2f s" % (
        analysis_s, sym.nf, sym.numStages, sym.flopcount / 1e12, cold_ms / 1e3))
    elapsed, kernel_ms, launches = time_steps(cx, step, prepare, K, max(W - 1, 0))
    verified = None
    if "res" in kept:
        tv = time.perf_counter()
        res = kept.pop("res")
        err = spqr.gram_probe(sym, res, A, seed=1)
        verified = {"check": "max|R'(R y) - (AE)'((AE) y)| / max|(AE)'((AE) y)|, y random: check_r_factor "
                             "(SPQR/Tcov/qrtest.cpp:1218-1257, bar 1e-10) through a vector, on the R of the last timed step",
                    "value": err, "passed": bool(err <= 1e-10 and res.rank == sym.n), "rank": res.rank, "n": int(sym.n),
                    "seconds": None}
        res.free()
        verified["seconds"] = round(time.perf_counter() - tv, 2)
        log("R of the last step checked: %.2e (%s)" % (err, "passed" if verified["passed"] else "FAILED"))
    elif cx.world > 1:
        verified = {"check": "not run: R is spread over the ranks (tests/test_multigpu_cpu.py and the -m gpu rank tests "
                             "compare the distributed R with the single-process one bit for bit)", "value": None, "passed": None}
    per_kernel, best = profile_step(cx, step, prepare) if profile else ({}, None)
    log("profiled step done")
    out = dict(workload=workload, tag="%s:%d" % (kind, k), flops=sym.flopcount, elapsed=elapsed, kernel_ms=kernel_ms,
               launches=launches, per_kernel=per_kernel, best=best, scaling="strong", config=config,
               cold_ms=cold_ms, analysis_s=analysis_s, ordering_s=order_s, mapping=mapping, verified=verified,
               setup_s=time.perf_counter() - t0)
    if cx.world > 1:
        out["multi"] = {"levels": last_info.get("levels"), "cut_blocks": last_info.get("n_cut"),
                        "cut_bytes": last_info.get("cut_bytes"),
                        "cut_bytes_rectangles": last_info.get("cut_bytes_rectangles"),
                        "rank0_calls": last_info.get("calls")}
    return out


def spawn_ranks(n):
    """bench.py --gpus N without a launcher: start the N ranks through torch.distributed.run, before this
    process has touched a GPU, and relay rank 0's line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line)
    else:
        sys.stdout.write(proc.stdout)
    return proc.returncode if line is None or proc.returncode != 0 else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=HEADLINE, help="sparse3d:K | sparse2d:K | dense")
    ap.add_argument("--ordering", default="metis", choices=["metis", "nd"])
    ap.add_argument("--copies", type=int, default=COPIES)
    ap.add_argument("--extras", default="auto", choices=["auto", "none", "all"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="leave out the extra step with events around every launch")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # The ranks of one node share the host's page-locked memory: the driver's per-process limit for R
    # stacks (48 GiB) is divided among them (a stack beyond it is pageable, with its end page-locked).
    nranks = int(os.environ.get("WORLD_SIZE", "1"))
    if nranks > 1:
        os.environ.setdefault("GPUQR_PIN_LIMIT_GIB", "%.3f" % (48.0 / nranks))

    import torch
    import torch.distributed as dist

    import suitesparse_amd as ss  # noqa: F401
    from suitesparse_amd import _lib

    cx = Ctx()
    cx.rank = int(os.environ.get("RANK", "0"))
    cx.world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if cx.world > 1:
        sys.stderr.write("[bench] rank %d of %d started (pid %d)\n" % (cx.rank, cx.world, os.getpid()))
        sys.stderr.flush()
    # BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks (the ranks
    # then share devices); the driver's runs use RCCL, one GPU per rank
    cx.backend = os.environ.get("BENCH_BACKEND", "nccl")
    if cx.backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    cx.dev = torch.device("cuda", local)
    if cx.world > 1:
        if cx.backend == "nccl":
            dist.init_process_group("nccl", device_id=cx.dev)
        else:
            dist.init_process_group(cx.backend)
    cx.L = _lib.lib()
    K, W = args.steps, args.warmup

    def run(workload, K, W, profile=True):
        if workload == "dense":
            return run_dense(cx, K, W, args.copies)
        kind, k = workload.split(":")
        return run_sparse(cx, kind, int(k), K, W, args.ordering, profile and not args.no_profile)

    res = run(args.workload, K, W)

    if cx.rank == 0:
        strong = res["scaling"] == "strong"
        value = (1 if strong else cx.world) * res["flops"] * K / res["elapsed"] / 1e9
        roofline, roofline_assembly = rooflines(res["per_kernel"], res["best"], res["tag"])
        out = {
            "metric": "numeric QR factor GFLOP/s", "value": value, "unit": "GFLOP/s",
            "n_gpus": cx.world, "steps": K, "warmup": W, "ms_per_step": res["elapsed"] / K * 1e3,
            "higher_is_better": True, "scaling": res["scaling"], "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": dict({"workload": res["workload"]}, **res["config"]),
            "kernel_ms_per_step": res["kernel_ms"], "launches_per_step": res["launches"],
            "pct_fp64_mfma_peak": 100.0 * value / cx.world / 1e3 / PEAK_FP64_MFMA_TFLOPS,
            "roofline": roofline, "per_kernel": res["per_kernel"], "verified": res.get("verified"),
        }
        if roofline_assembly is not None:
            out["roofline_assembly"] = roofline_assembly
        if "cold_ms" in res:
            out["refactorization"] = True
            out["cold_first_call_ms"] = res["cold_ms"]
            out["value_first_factorization"] = res["flops"] / (res["cold_ms"] / 1e3) / 1e9
            out["analysis_s"] = res["analysis_s"]
            out["ordering_s"] = res["ordering_s"]
            out["mapping_bounds"] = res["mapping"]
            if cx.world > 1:
                me = res["mapping"].get(str(cx.world), {})
                out["top_flops_share"] = me.get("top_flops_share")
                out["speedup_bound"] = me.get("speedup_bound")
                out["multi"] = res.get("multi")
            else:
                out["top_flops_share"], out["speedup_bound"] = 0.0, 1.0
        # ---- the other configurations, one short run each (N = 1 only) -----------------------------
        if cx.world == 1 and args.extras != "none" and args.workload == HEADLINE:
            extras = {}
            plan = [("sparse2d:1024", 5, 2, 330.0), ("dense", 5, 2, 360.0), ("sparse3d:96", 1, 1, 300.0)]
            for wl, k_, w_, latest_start in plan:
                spent = time.perf_counter() - T_START
                if args.extras == "auto" and spent > latest_start:
                    extras[wl] = {"skipped": "bench already %.0f s old" % spent}
                    continue
                try:
                    cx.L.spqrgpu_amd_finalize()      # give the previous problem's device workspaces back
                    torch.cuda.empty_cache()
                    r = run(wl, k_, w_)
                    v = r["flops"] * k_ / r["elapsed"] / 1e9
                    e = {"workload": r["workload"], "config": r["config"], "value": v, "unit": "GFLOP/s", "steps": k_,
                         "ms_per_step": r["elapsed"] / k_ * 1e3, "kernel_ms_per_step": r["kernel_ms"],
                         "launches_per_step": r["launches"],
                         "pct_fp64_mfma_peak": 100.0 * v / 1e3 / PEAK_FP64_MFMA_TFLOPS, "verified": r.get("verified")}
                    rl, rla = rooflines(r["per_kernel"], r["best"], r["tag"])
                    if rl is not None:
                        e["roofline"] = rl
                    if rla is not None:
                        e["roofline_assembly"] = rla
                    if "cold_ms" in r:
                        e["cold_first_call_ms"] = r["cold_ms"]
                        e["value_first_factorization"] = r["flops"] / (r["cold_ms"] / 1e3) / 1e9
                    if wl == "sparse2d:1024" and not args.no_cpu_baseline:
                        e["cpu_baseline"] = cpu_baseline_sparse("sparse2d", 1024)
                    extras[wl] = e
                except Exception as ex:          # an extra must never take the headline down with it
                    extras[wl] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            out["extras"] = extras
        if cx.world == 1 and not args.no_cpu_baseline:
            cx.L.spqrgpu_amd_finalize()
            if args.workload == "dense":
                out["cpu_baseline"] = cpu_baseline_dense()
            else:
                kind, k = args.workload.split(":")
                out["cpu_baseline"] = cpu_baseline_sparse(kind, int(k))
        out["bench_seconds"] = round(time.perf_counter() - T_START, 1)
        print(json.dumps(out))
    if cx.world > 1:
        dist.barrier()                  # rank 0 is still printing while the others arrive here
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

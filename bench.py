#!/usr/bin/env python3
"""Benchmark of the numeric sparse QR hot path on MI355X.

Headline workload: numeric multifrontal QR of the 3D 7-point grid least-squares matrix
A = [L ; sqrt(0.1) I] on a 96^3 grid (m = 1,769,472, n = 884,736; BASELINE.json configs[3] is the
same generator at 128^3, whose 45 s steps do not fit a run of 25 of them; it is carried as a one-call
extra).  A step is one numeric factorization through the SPQR glue's C entry point (S built on the
device from A's values, front assembly, panel factorization, block-reflector updates, R packed and
copied to the host stack).  The metric is the reference's canonical flop count of the symbolic analysis
(SPQR/Source/spqr_analyze.cpp:888-902) divided by wall time.

--gpus N > 1: the SAME problem on N ranks (strong scaling).  The assembly tree is mapped onto the
ranks (suitesparse_amd/multigpu.py), every rank factorizes its fronts, contribution blocks cross ranks
device to device over RCCL.

The timed steps re-factorize on one symbolic object, as a solver inside an iteration does: pattern
work and the engine's schedule are reused ("refactorization": true); the first call, which includes
them, is reported beside it ("cold_first_call_ms").

Extras at N = 1 (same JSON line, key "extras"): 128^3 (the cold call and one re-factorization), the 2D 1024^2 grid
(BASELINE.json configs[2]) and the dense front batch (configs[1]).
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
HEADLINE = "sparse3d:96"
# device memory declared to the staging (cc->gpuMemorySize); BENCH_DEVICE_GIB overrides it, e.g. when ranks of a
# rehearsal share one device
DECLARED_DEVICE_MEMORY = int(float(os.environ.get("BENCH_DEVICE_GIB", "200")) * 2**30)
T_START = time.perf_counter()


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
    """The CPU multifrontal path of the oracle on a bounded sample of the same generator (the grid
    shrunk until the run takes 10-30 s): fronts in postorder, each factorized by LAPACK's blocked
    Householder QR (dgeqrf = dlarfg/dlarft/dlarfb, what SPQR's CPU path calls) on the host's threaded
    BLAS, all cores; 2D grids (small fronts, nothing for a threaded BLAS to do) use the scalar C port."""
    from oracle import oracle
    from suitesparse_amd import problems, spqr
    ks = min(k, 640 if kind == "sparse2d" else 40)
    shape = (ks, ks) if kind == "sparse2d" else (ks, ks, ks)
    A = problems.stacked_ls(shape)
    sym = spqr.analyze(A, problems.nested_dissection(shape, leaf=8), relax=20, gpuMemorySize=DECLARED_DEVICE_MEMORY)
    Sx = spqr.s_values(sym, A)
    if kind == "sparse2d":
        t0 = time.perf_counter()
        oracle.multifrontal(sym, Sx)
        sec = time.perf_counter() - t0
        cores, how = 1, "scalar C port"
    else:
        try:
            from threadpoolctl import threadpool_info
            cores = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
        except ImportError:
            cores = 1
        _, t_qr, sec = oracle.multifrontal_lapack(sym, Sx, keep_R=False)
        how = "LAPACK dgeqrf per front on threaded BLAS (%.1f s of it in dgeqrf)" % t_qr
    return {"value": sym.flopcount / sec / 1e9, "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": "%s grid %s of the same generator, %s (%.2f GFLOP, %.1f s)"
                      % (kind, "x".join(map(str, shape)), how, sym.flopcount / 1e9, sec)}


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
    for fname in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", fname)))
            if pmc.get("workload_tag", "dense") == workload_tag and kernel in pmc["kernels"]:
                return (pmc["kernels"][kernel]["traffic_bytes_per_launch"],
                        "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, bytes per launch)" % fname)
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def rooflines(per_kernel, best, workload_tag):
    """`roofline` of the kernel with the largest total time, `roofline_assembly` of k_assemble."""
    roofline = roofline_assembly = None
    if best is not None:
        name, ms, ln, fl, by = best
        # HBM bytes per launch from the hardware counters: collected with rocprofv3 --pmc in separate
        # passes of this same command (a profiler cannot run inside the timed process) and committed
        traffic, traffic_src = pmc_traffic(name, workload_tag)
        avg_s = ms / 1e3 / ln
        tflops = fl / ln / avg_s / 1e12
        gbs = by / ln / avg_s / 1e9
        # the 32-column block reflector does about 8 flop per byte of A it must move, below the
        # chip's FP64 ridge (78.6 TF / 8 TB/s = 9.8): HBM bounds it
        if name.startswith("k_apply") or name.startswith("k_assemble"):
            roofline = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                        "avg_launch_us": avg_s * 1e6, "launches": ln, "algorithmic_bytes_per_launch": by / ln,
                        "mfma_tflops": tflops, "mfma_frac": tflops / PEAK_FP64_MFMA_TFLOPS}
        else:
            roofline = {"kernel": name, "bound": "mfma", "achieved": tflops, "peak": PEAK_FP64_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": None,
                        "avg_launch_us": avg_s * 1e6, "launches": ln}
    if "k_assemble" in per_kernel and per_kernel["k_assemble"]["ms"] > 0:
        ka = per_kernel["k_assemble"]
        gbs_a = ka["gbytes"] / (ka["ms"] / 1e3)
        traffic_a, _ = pmc_traffic("k_assemble", workload_tag)
        roofline_assembly = {"kernel": "k_assemble", "bound": "hbm", "achieved": gbs_a, "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": gbs_a / PEAK_HBM_GBS, "traffic": traffic_a,
                             "avg_launch_us": ka["ms"] * 1e3 / ka["launches"], "launches": ka["launches"],
                             "algorithmic_bytes_per_launch": ka["gbytes"] * 1e9 / ka["launches"],
                             "note": "many small launches beside the updates (contribution blocks travel 32 rows at a "
                                     "time); their sum is 3 % of the kernel time at 96^3, where whole-front "
                                     "hand-over (GPUQR_NO_PIPELINE=1) moves the same bytes in 353 launches at "
                                     "2.28 TB/s and makes the step 12 % longer"}
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
    for i in range(min(K, nbuf)):
        prepare(W + i)
    kernel_ms, launches = 0.0, 0
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

    def step(i, stats=None):
        code = ss.GPUQREngine(0, front_sets[i % nbuf], len(shapes), stats=stats)
        assert code == ss.QRENGINE_SUCCESS, code

    elapsed, kernel_ms, launches = time_steps(cx, step, prepare, K, W, nbuf)
    per_kernel, best = profile_step(cx, step, prepare)
    workload = "dense-front-batch %dx{%s}x{%s}" % (copies, ",".join(map(str, FM)), ",".join(map(str, FN)))
    return dict(workload=workload, tag="dense", flops=flops, elapsed=elapsed, kernel_ms=kernel_ms, launches=launches,
                per_kernel=per_kernel, best=best, scaling="weak",
                config={"fronts": len(shapes), "bytes": total * 8})


def run_sparse(cx, kind, k, K, W, single_call=False, profile=True):
    """Time the sparse numeric factorization; single_call: one call in all, the cold one; profile=False
    leaves out the extra step with events around every launch."""
    import torch
    from suitesparse_amd import multigpu, problems, spqr
    t0 = time.perf_counter()
    shape = (k, k) if kind == "sparse2d" else (k, k, k)
    A = problems.stacked_ls(shape)
    Q = problems.nested_dissection(shape, leaf=8)
    sym = spqr.analyze(A, Q, relax=20, gpuMemorySize=DECLARED_DEVICE_MEMORY)
    analysis_s = time.perf_counter() - t0
    workload = "%s grid %s stacked LS, geometric ND order, m=%d n=%d nf=%d" % (
        kind, "x".join(map(str, shape)), A.shape[0], A.shape[1], sym.nf)
    config = {"fronts": sym.nf, "stages": sym.numStages, "nnzR": sym.rnz, "m": A.shape[0], "n": A.shape[1],
              "canonical_gflop": sym.flopcount / 1e9}
    mapping = {}
    for parts in sorted({2, 4, 8, cx.world} - {1}):
        owner, level = multigpu.map_tree(sym, parts)
        rep = multigpu.mapping_report(sym, owner, level, parts)
        mapping[str(parts)] = {"top_flops_share": round(rep["top_flops_share"], 4),
                               "speedup_bound": round(rep["speedup_bound"], 3)}
    csc = multigpu._Csc(A)
    last_info = {}

    def prepare(i):
        pass

    def step(i, stats=None):
        if cx.world > 1:
            _, info = multigpu.factorize_distributed(sym, csc, gather_R=False)
            last_info.update(info)
            if stats is not None:
                stats.kernelTime = sum(c["kernel_ms"] for c in info["calls"])
                stats.numLaunches = sum(c["launches"] for c in info["calls"])
            return
        res = spqr.factorize(sym, csc, want_R=False)      # R stays as packed blocks; CSC conversion is not on the path
        if stats is not None:
            stats.kernelTime, stats.numLaunches, stats.flopsActual = (
                res.gpuKernelTime, res.gpuNumKernelLaunches, res.gpuFlops)

    # the first call: pattern work (maps, S permutation), workspace allocation, planning and upload
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(0)
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0) * 1e3
    if single_call:
        return dict(workload=workload, tag=kind, flops=sym.flopcount, elapsed=cold_ms / 1e3, kernel_ms=0.0, launches=0,
                    per_kernel={}, best=None, scaling="strong", config=config, cold_ms=cold_ms,
                    analysis_s=analysis_s, mapping=mapping, steps=1)
    elapsed, kernel_ms, launches = time_steps(cx, step, prepare, K, max(W - 1, 0))
    per_kernel, best = profile_step(cx, step, prepare) if profile else ({}, None)
    out = dict(workload=workload, tag="%s:%d" % (kind, k), flops=sym.flopcount, elapsed=elapsed, kernel_ms=kernel_ms,
               launches=launches, per_kernel=per_kernel, best=best, scaling="strong", config=config,
               cold_ms=cold_ms, analysis_s=analysis_s, mapping=mapping)
    if cx.world > 1:
        out["multi"] = {"levels": last_info.get("levels"), "cut_blocks": last_info.get("n_cut"),
                        "cut_bytes": last_info.get("cut_bytes"),
                        "rank0_calls": last_info.get("calls")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=HEADLINE, help="sparse3d:K | sparse2d:K | dense")
    ap.add_argument("--copies", type=int, default=COPIES)
    ap.add_argument("--extras", default="auto", choices=["auto", "none", "all"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import suitesparse_amd as ss  # noqa: F401
    from suitesparse_amd import _lib

    cx = Ctx()
    cx.rank = int(os.environ.get("RANK", "0"))
    cx.world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    def run(workload, K, W, single_call=False, profile=True):
        if workload == "dense":
            return run_dense(cx, K, W, args.copies)
        kind, k = workload.split(":")
        return run_sparse(cx, kind, int(k), K, W, single_call, profile)

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
            "roofline": roofline, "per_kernel": res["per_kernel"],
        }
        if roofline_assembly is not None:
            out["roofline_assembly"] = roofline_assembly
        if "cold_ms" in res:
            out["refactorization"] = True
            out["cold_first_call_ms"] = res["cold_ms"]
            out["analysis_s"] = res["analysis_s"]
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
            # (config 4 at full size: the cold call, then one timed re-factorization; no profiled step, it is 33 s a call)
            plan = [("sparse3d:128", 1, 1, False, 300.0), ("sparse2d:1024", 5, 2, False, 420.0), ("dense", 5, 2, False, 450.0)]
            for wl, k_, w_, single, latest_start in plan:
                spent = time.perf_counter() - T_START
                if args.extras == "auto" and spent > latest_start:
                    extras[wl] = {"skipped": "bench already %.0f s old" % spent}
                    continue
                try:
                    cx.L.spqrgpu_amd_finalize()      # give the previous problem's device workspaces back
                    torch.cuda.empty_cache()
                    r = run(wl, k_, w_, single, profile=(wl != "sparse3d:128"))
                    steps = r.get("steps", k_)
                    v = r["flops"] * steps / r["elapsed"] / 1e9
                    e = {"workload": r["workload"], "value": v, "unit": "GFLOP/s", "steps": steps,
                         "ms_per_step": r["elapsed"] / steps * 1e3,
                         "pct_fp64_mfma_peak": 100.0 * v / 1e3 / PEAK_FP64_MFMA_TFLOPS}
                    if single:
                        e["note"] = "one call, the cold one (pattern work, allocation and planning included)"
                    else:
                        rl, rla = rooflines(r["per_kernel"], r["best"], r["tag"])
                        if rl is not None:
                            e["roofline"] = rl
                        if rla is not None:
                            e["roofline_assembly"] = rla
                        if "cold_ms" in r:
                            e["cold_first_call_ms"] = r["cold_ms"]
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
        print(json.dumps(out))
    if cx.world > 1:
        dist.barrier()                  # rank 0 is still printing while the others arrive here
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Quick timing probe: device-resident dense fronts through the sparse entry point."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import suitesparse_amd as ss

def canon(m, n):
    return 2*n*n*(m - n/3) if m >= n else 2*m*m*(n - m/3)

def run(shapes, reps=3):
    dev = torch.device("cuda:0")
    tot = sum(m*n for m, n in shapes)
    src = torch.rand(tot, dtype=torch.float64, device=dev) * 2 - 1
    buf = torch.empty_like(src)
    fronts, off = [], 0
    for k, (m, n) in enumerate(shapes):
        f = ss.Front(k, ss.EMPTY, m, n)
        f.gpuF = buf.data_ptr() + off * 8
        off += m*n
        fronts.append(f)
    flops = sum(canon(m, n) for m, n in shapes)
    best = None
    for r in range(reps):
        buf.copy_(src); torch.cuda.synchronize()
        st = ss.QREngineStats()
        t0 = time.perf_counter()
        code = ss.GPUQREngine(0, fronts, len(fronts), stats=st)
        t1 = time.perf_counter()
        assert code == 0
        best = (t1-t0, st.kernelTime/1e3, st.numLaunches, st.flopsActual) if best is None or t1-t0 < best[0] else best
    w, k, nl, fa = best
    print(f"{len(shapes):4d} x {shapes[0]}: wall {w*1e3:8.2f} ms  kernels {k*1e3:8.2f} ms  launches {nl:5d}  "
          f"canon {flops/w/1e9:8.1f} GF/s (kernel-only {flops/k/1e9:8.1f})  actual/canon {fa/flops:.2f}", flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        m, n, cnt = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
        run([(m, n)]*cnt, reps=int(sys.argv[4]) if len(sys.argv) > 4 else 3)
        sys.exit(0)
    run([(96, 32)]*4096)
    run([(384, 128)]*1024)
    run([(1536, 512)]*64)
    run([(4096, 2048)]*1)
    run([(4096, 2048)]*16)
    run([(4096, 512)]*64)
    run([(8192, 4096)]*4)

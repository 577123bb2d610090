// Does a hipGraph make dependent launches cheaper?  The same two chains as launch_gap.cpp -- tiny kernels
// back to back on one stream, and alternating between two streams with an event per hop -- captured into
// a graph once and replayed: time per kernel of a replay, measured on the device (events around
// hipGraphLaunch) and on the host.
//   hipcc --offload-arch=gfx950 -O2 scripts/ubench/graph_gap.cpp -o scripts/ubench/graph_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>
__global__ void k_tiny (int *p) { if (threadIdx.x == 0) p [blockIdx.x] += 1 ; }
static double now_ms ( ) { return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now ( ).time_since_epoch ( )).count ( ) ; }
#define CK(x) do { hipError_t e_ = (x) ; if (e_ != hipSuccess) { printf ("%s: %s\n", #x, hipGetErrorString (e_)) ; return 1 ; } } while (0)
int main ( )
{
    const int n = 1000 ;
    int *d ; CK (hipMalloc (&d, 4096)) ; CK (hipMemset (d, 0, 4096)) ;
    hipStream_t s [2] ;
    for (int i = 0 ; i < 2 ; i++) CK (hipStreamCreateWithFlags (&s [i], hipStreamNonBlocking)) ;
    std::vector<hipEvent_t> ev (n) ;
    for (auto &e : ev) CK (hipEventCreateWithFlags (&e, hipEventDisableTiming)) ;
    for (int variant = 0 ; variant < 2 ; variant++)      // 0: one stream, 1: ping-pong between two
    {
        hipGraph_t graph ; hipGraphExec_t exec ;
        CK (hipStreamBeginCapture (s [0], hipStreamCaptureModeGlobal)) ;
        for (int i = 0 ; i < n ; i++)
        {
            hipStream_t st = variant ? s [i & 1] : s [0] ;
            if (variant && i > 0) CK (hipStreamWaitEvent (st, ev [i-1], 0)) ;
            hipLaunchKernelGGL (k_tiny, dim3 (4), dim3 (64), 0, st, d) ;
            if (variant) CK (hipEventRecord (ev [i], st)) ;
        }
        if (variant) CK (hipStreamWaitEvent (s [0], ev [n-1], 0)) ;       // join the other stream back
        CK (hipStreamEndCapture (s [0], &graph)) ;
        CK (hipGraphInstantiate (&exec, graph, NULL, NULL, 0)) ;
        hipEvent_t d0, d1 ; CK (hipEventCreate (&d0)) ; CK (hipEventCreate (&d1)) ;
        for (int rep = 0 ; rep < 3 ; rep++)
        {
            CK (hipDeviceSynchronize ( )) ;
            CK (hipEventRecord (d0, s [0])) ;
            double t0 = now_ms ( ) ;
            CK (hipGraphLaunch (exec, s [0])) ;
            double t1 = now_ms ( ) ;
            CK (hipEventRecord (d1, s [0])) ;
            CK (hipDeviceSynchronize ( )) ;
            float ms = 0 ; CK (hipEventElapsedTime (&ms, d0, d1)) ;
            if (rep == 2) printf ("graph, %s: host %.2f us per kernel to launch, device %.2f us per kernel\n",
                variant ? "two streams, one hop per kernel" : "one stream", (t1 - t0) * 1e3 / n, ms * 1e3 / n) ;
        }
        CK (hipGraphExecDestroy (exec)) ; CK (hipGraphDestroy (graph)) ;
    }
    // Two chains side by side that lean on each other, as the panel and the update streams of the engine
    // do: A(i+1) waits for A(i) and B(i), B(i+1) waits for B(i) and A(i).  Streams with events, then the
    // same thing captured into a graph.
    {
        const int steps = 500 ;
        std::vector<hipEvent_t> ea (steps), eb (steps) ;
        for (auto &e : ea) CK (hipEventCreateWithFlags (&e, hipEventDisableTiming)) ;
        for (auto &e : eb) CK (hipEventCreateWithFlags (&e, hipEventDisableTiming)) ;
        auto enqueue = [&] ( ) -> hipError_t
        {
            for (int i = 0 ; i < steps ; i++)
            {
                if (i > 0) { hipError_t e = hipStreamWaitEvent (s [0], eb [i-1], 0) ; if (e != hipSuccess) return e ; }
                hipLaunchKernelGGL (k_tiny, dim3 (4), dim3 (64), 0, s [0], d) ;
                { hipError_t e = hipEventRecord (ea [i], s [0]) ; if (e != hipSuccess) return e ; }
                if (i > 0) { hipError_t e = hipStreamWaitEvent (s [1], ea [i-1], 0) ; if (e != hipSuccess) return e ; }
                else { hipError_t e = hipStreamWaitEvent (s [1], ea [0], 0) ; if (e != hipSuccess) return e ; }    // (fork for the capture)
                hipLaunchKernelGGL (k_tiny, dim3 (4), dim3 (64), 0, s [1], d + 64) ;
                { hipError_t e = hipEventRecord (eb [i], s [1]) ; if (e != hipSuccess) return e ; }
            }
            return hipStreamWaitEvent (s [0], eb [steps-1], 0) ;
        } ;
        hipEvent_t d0, d1 ; CK (hipEventCreate (&d0)) ; CK (hipEventCreate (&d1)) ;
        for (int rep = 0 ; rep < 3 ; rep++)
        {
            CK (hipDeviceSynchronize ( )) ;
            CK (hipEventRecord (d0, s [0])) ;
            double t0 = now_ms ( ) ;
            CK (enqueue ( )) ;
            double t1 = now_ms ( ) ;
            CK (hipEventRecord (d1, s [0])) ;
            CK (hipDeviceSynchronize ( )) ;
            float ms = 0 ; CK (hipEventElapsedTime (&ms, d0, d1)) ;
            if (rep == 2) printf ("two leaning chains, streams and events: host %.2f us per step, wall %.2f us per step (two kernels a step)\n",
                (t1 - t0) * 1e3 / steps, ms * 1e3 / steps) ;
        }
        hipGraph_t graph ; hipGraphExec_t exec ;
        CK (hipStreamBeginCapture (s [0], hipStreamCaptureModeGlobal)) ;
        CK (enqueue ( )) ;
        CK (hipStreamEndCapture (s [0], &graph)) ;
        CK (hipGraphInstantiate (&exec, graph, NULL, NULL, 0)) ;
        for (int rep = 0 ; rep < 3 ; rep++)
        {
            CK (hipDeviceSynchronize ( )) ;
            CK (hipEventRecord (d0, s [0])) ;
            double t0 = now_ms ( ) ;
            CK (hipGraphLaunch (exec, s [0])) ;
            double t1 = now_ms ( ) ;
            CK (hipEventRecord (d1, s [0])) ;
            CK (hipDeviceSynchronize ( )) ;
            float ms = 0 ; CK (hipEventElapsedTime (&ms, d0, d1)) ;
            if (rep == 2) printf ("two leaning chains, graph: host %.2f us per step to launch, device %.2f us per step\n",
                (t1 - t0) * 1e3 / steps, ms * 1e3 / steps) ;
        }
        CK (hipGraphExecDestroy (exec)) ; CK (hipGraphDestroy (graph)) ;
    }
    return 0 ;
}

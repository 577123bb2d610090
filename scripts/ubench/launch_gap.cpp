// Time per dependent launch of a tiny kernel: (a) back to back on one stream, (b) with an untimed event
// recorded after every launch, (c) alternating between two streams, each launch waiting for the event
// of the previous one, (d) as (b) and (e) as (c) but with the event attached to the launch itself
// (hipExtLaunchKernelGGL's stop event: the kernel's own completion signal, no marker packet).   hipcc --offload-arch=gfx950 -O2 scripts/ubench/launch_gap.cpp -o scripts/ubench/launch_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#include <chrono>
__global__ void k_tiny (int *p) { if (threadIdx.x == 0) p [blockIdx.x] += 1 ; }
// holds the queue for about `cycles` shader clocks, so that what the host enqueues meanwhile is timed
// on the device alone
__global__ void k_hold (long long cycles, int *p)
{
    const long long t0 = __builtin_amdgcn_s_memtime ( ) ;
    while (__builtin_amdgcn_s_memtime ( ) - t0 < cycles) __builtin_amdgcn_s_sleep (16) ;
    if (p == NULL) return ;
}
static double now_ms ( ) { return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now ( ).time_since_epoch ( )).count ( ) ; }
int main ( )
{
    const int n = 2000 ;
    int *d ; (void) hipMalloc (&d, 4096) ; (void) hipMemset (d, 0, 4096) ;
    hipStream_t s [2] ;
    for (int i = 0 ; i < 2 ; i++) (void) hipStreamCreateWithFlags (&s [i], hipStreamNonBlocking) ;
    std::vector<hipEvent_t> ev (n) ;
    for (auto &e : ev) (void) hipEventCreateWithFlags (&e, hipEventDisableTiming) ;
    for (int variant = 0 ; variant < 5 ; variant++)
    {
        for (int rep = 0 ; rep < 2 ; rep++)
        {
            (void) hipDeviceSynchronize ( ) ;
            // device-side time: a hold kernel on stream 0 (100 M clocks at 100 MHz would be far too long: the
            // counter runs at about 100 MHz here, 4 M ticks = 40 ms), then the chain behind it, timed by events
            hipEvent_t d0, d1 ;
            (void) hipEventCreate (&d0) ; (void) hipEventCreate (&d1) ;
            hipLaunchKernelGGL (k_hold, dim3 (1), dim3 (64), 0, s [0], 4000000LL, (int *) NULL) ;
            (void) hipEventRecord (d0, s [0]) ;
            double t0 = now_ms ( ) ;
            for (int i = 0 ; i < n ; i++)
            {
                const bool two = (variant == 2 || variant == 4) ;
                hipStream_t st = two ? s [i & 1] : s [0] ;
                if (two && i > 0) (void) hipStreamWaitEvent (st, ev [i-1], 0) ;
                if (variant >= 3) hipExtLaunchKernelGGL (k_tiny, dim3 (4), dim3 (64), 0, st, NULL, ev [i], 0, d) ;
                else hipLaunchKernelGGL (k_tiny, dim3 (4), dim3 (64), 0, st, d) ;
                if (variant == 1 || variant == 2) (void) hipEventRecord (ev [i], st) ;
            }
            double t1 = now_ms ( ) ;
            const bool two_streams = (variant == 2 || variant == 4) ;
            hipStream_t last = two_streams ? s [(n - 1) & 1] : s [0] ;
            (void) hipEventRecord (d1, last) ;
            (void) hipDeviceSynchronize ( ) ;
            float dev_ms = 0 ;
            (void) hipEventElapsedTime (&dev_ms, d0, d1) ;
            if (rep == 1) printf ("variant %d: host %.2f us per launch, device %.2f us per launch (chain queued behind a hold kernel)\n",
                variant, (t1 - t0) * 1e3 / n, dev_ms * 1e3 / n) ;
            (void) hipEventDestroy (d0) ; (void) hipEventDestroy (d1) ;
        }
    }
    int h [4] ;
    (void) hipMemcpy (h, d, sizeof (h), hipMemcpyDeviceToHost) ;
    printf ("increments seen %d %d %d %d of %d (a lost one means two dependent launches overlapped), last error %s\n",
        h [0], h [1], h [2], h [3], 5 * 2 * n, hipGetErrorName (hipGetLastError ( ))) ;
    return 0 ;
}

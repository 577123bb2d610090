// Isolated timing of the panel factorize kernel: one launch of ntask panels of K rows x 32 columns.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I suitesparse_amd/csrc scripts/ubench/factor_bench.hip -o scripts/ubench/factor_bench
#include "qr_kernels.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

int main (int argc, char **argv)
{
    int ntask = argc > 1 ? atoi (argv [1]) : 256 ;
    int K     = argc > 2 ? atoi (argv [2]) : 256 ;
    int ld    = argc > 3 ? atoi (argv [3]) : 2048 ;
    int reps  = argc > 4 ? atoi (argv [4]) : 5 ;
    size_t rows = (size_t) ntask * K ;
    double *F, *VT ;
    hipMalloc (&F, rows * ld * 8) ;
    hipMalloc (&VT, (size_t) ntask * 1024 * 8) ;
    std::vector<double> h (rows * ld) ;
    for (size_t i = 0 ; i < h.size ( ) ; i++) h [i] = (double) rand ( ) / RAND_MAX - 0.5 ;
    std::vector<QrTask> tasks (ntask) ;
    std::vector<QrRef> refs (ntask) ;
    for (int t = 0 ; t < ntask ; t++)
    {
        QrTask &q = tasks [t] ;
        q = QrTask ( ) ;
        q.type = QR_TASK_FACTORIZE ; q.tier = K <= 32 ? 0 : K <= 64 ? 1 : K <= 128 ? 2 : 3 ;
        QrPanelTask &p = q.u.panel ;
        p.F = F ; p.VT = VT + (size_t) t * 1024 ; p.ld = ld ; p.col0 = 0 ; p.w = 32 ; p.K = K ;
        p.nseg = 1 ; p.seg_row [0] = t * K ; p.seg_n [0] = K ;
        refs [t].task = t ; refs [t].blk = 0 ;
    }
    QrTask *d_tasks ; QrRef *d_refs ;
    hipMalloc (&d_tasks, tasks.size ( ) * sizeof (QrTask)) ;
    hipMalloc (&d_refs, refs.size ( ) * sizeof (QrRef)) ;
    hipMemcpy (d_tasks, tasks.data ( ), tasks.size ( ) * sizeof (QrTask), hipMemcpyHostToDevice) ;
    hipMemcpy (d_refs, refs.data ( ), refs.size ( ) * sizeof (QrRef), hipMemcpyHostToDevice) ;
    if (qr_kernels_init ( ) != hipSuccess) { printf ("init failed\n") ; return 1 ; }
    hipEvent_t e0, e1 ; hipEventCreate (&e0) ; hipEventCreate (&e1) ;
    float best = 1e30f ;
    for (int r = 0 ; r < reps + 1 ; r++)
    {
        hipMemcpy (F, h.data ( ), h.size ( ) * 8, hipMemcpyHostToDevice) ;
        hipEventRecord (e0) ;
        hipError_t e = qr_launch_tasks (tasks [0].tier, d_tasks, d_refs, ntask, 0) ;
        hipEventRecord (e1) ; hipEventSynchronize (e1) ;
        if (e != hipSuccess || hipGetLastError ( ) != hipSuccess) { printf ("launch failed\n") ; return 1 ; }
        float ms ; hipEventElapsedTime (&ms, e0, e1) ;
        if (r > 0 && ms < best) best = ms ;
    }
    printf ("factorize tasks %d K %d ld %d: %.1f us per launch, %.2f us per task per CU slot\n", ntask, K, ld,
        best * 1e3, best * 1e3 / ((ntask + 255) / 256)) ;
    return 0 ;
}

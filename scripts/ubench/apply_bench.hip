// Isolated timing of the apply kernel: one launch of k_apply<16> over a tall matrix cut into
// 256-row tasks, every task split into column blocks.  Build variants with -D flags.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I suitesparse_amd/csrc scripts/ubench/apply_bench.hip -o scripts/ubench/apply_bench
// -DNOMFMA: timing-only build, the matrix instructions are replaced by one add that keeps the operands
// alive; -DNT_SLABS: the trailing matrix is loaded and stored with the non-temporal hint.
#ifdef NOMFMA
typedef double v4f64_ub __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) ((c) + (v4f64_ub) {(a) * (b), 0.0, 0.0, 0.0})
#endif
#ifdef NT_SLABS
#define SLAB_LOAD(p) __builtin_nontemporal_load (p)
#define SLAB_STORE(v, p) __builtin_nontemporal_store ((v), (p))
#endif
#include "qr_kernels.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

int main (int argc, char **argv)
{
    int ntask = argc > 1 ? atoi (argv [1]) : 64 ;       // 256-row tasks
    int ncols = argc > 2 ? atoi (argv [2]) : 4096 ;     // trailing columns
    int cb    = argc > 3 ? atoi (argv [3]) : 128 ;      // columns per workgroup
    int reps  = argc > 4 ? atoi (argv [4]) : 5 ;
    int K     = argc > 5 ? atoi (argv [5]) : 256 ;
    int ld = ncols + 32 ;
    size_t rows = (size_t) ntask * K ;
    double *F, *VT ;
    hipMalloc (&F, rows * ld * 8) ;
    hipMalloc (&VT, (size_t) ntask * 1024 * 8) ;
    std::vector<double> h (rows * ld) ;
    for (size_t i = 0 ; i < h.size ( ) ; i++) h [i] = (double) rand ( ) / RAND_MAX - 0.5 ;
    hipMemcpy (F, h.data ( ), h.size ( ) * 8, hipMemcpyHostToDevice) ;
    std::vector<double> hvt ((size_t) ntask * 1024) ;
    for (size_t i = 0 ; i < hvt.size ( ) ; i++) hvt [i] = 0.01 * ((double) rand ( ) / RAND_MAX - 0.5) ;
    hipMemcpy (VT, hvt.data ( ), hvt.size ( ) * 8, hipMemcpyHostToDevice) ;

    int nblk = (ncols + cb - 1) / cb ;
    std::vector<QrTask> tasks (ntask) ;
    std::vector<QrRef> refs ;
    for (int t = 0 ; t < ntask ; t++)
    {
        QrTask &q = tasks [t] ;
        q = QrTask ( ) ;
        q.type = QR_TASK_APPLY ; q.tier = 3 ;
        QrPanelTask &p = q.u.panel ;
        p.F = F ; p.VT = VT + (size_t) t * 1024 ; p.ld = ld ; p.col0 = 0 ; p.w = 32 ; p.K = K ;
        p.nseg = 1 ; p.seg_row [0] = t * K ; p.seg_n [0] = K ;
        p.c_begin = 32 ; p.c_end = ld ; p.cb_cols = cb ;
        for (int b = 0 ; b < nblk ; b++) { QrRef r ; r.task = t ; r.blk = b ; refs.push_back (r) ; }
    }
    QrTask *d_tasks ; QrRef *d_refs ;
    hipMalloc (&d_tasks, tasks.size ( ) * sizeof (QrTask)) ;
    hipMalloc (&d_refs, refs.size ( ) * sizeof (QrRef)) ;
    hipMemcpy (d_tasks, tasks.data ( ), tasks.size ( ) * sizeof (QrTask), hipMemcpyHostToDevice) ;
    hipMemcpy (d_refs, refs.data ( ), refs.size ( ) * sizeof (QrRef), hipMemcpyHostToDevice) ;
    if (qr_kernels_init ( ) != hipSuccess) { printf ("init failed\n") ; return 1 ; }
    int kid = 4 + (K <= 32 ? 0 : K <= 64 ? 1 : K <= 128 ? 2 : 3) ;
    hipEvent_t e0, e1 ; hipEventCreate (&e0) ; hipEventCreate (&e1) ;
    float best = 1e30f ;
    for (int r = 0 ; r < reps + 1 ; r++)
    {
        hipEventRecord (e0) ;
        hipError_t e = qr_launch_tasks (kid, d_tasks, d_refs, (int) refs.size ( ), 0) ;
        hipEventRecord (e1) ; hipEventSynchronize (e1) ;
        if (e != hipSuccess || hipGetLastError ( ) != hipSuccess) { printf ("launch failed\n") ; return 1 ; }
        float ms ; hipEventElapsedTime (&ms, e0, e1) ;
        if (r > 0 && ms < best) best = ms ;
    }
    double flops = (double) ntask * ((4.0 * K * 32 + 2.0 * 32 * 32) * ncols) ;
    double bytes = (double) rows * ncols * 16 ;
    printf ("tasks %d K %d cols %d cb %d: %d workgroups  %.3f ms  %.2f TF/s  %.2f TB/s\n", ntask, K, ncols, cb,
        (int) refs.size ( ), best, flops / best / 1e9, bytes / best / 1e9) ;
    return 0 ;
}

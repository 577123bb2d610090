// Microbenchmark: which slab shape streams fastest?  Every wave moves 4096 doubles of a row-major
// FP64 matrix through registers (load all, add 1, store all) with 16 bytes per lane; LPR lanes
// share a row, so one instruction covers 64/LPR rows x 16*LPR bytes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) double gdouble ;
typedef double d2 __attribute__((ext_vector_type(2))) ;
typedef __attribute__((address_space(1))) d2 gd2 ;

template <int LPR>
__global__ void __launch_bounds__(256, 2) k (double *Fp, int ld, int ncols, int nrows)
{
    // slab of a wave: ROWS x COLS, COLS = 2*LPR, ROWS = 4096 / COLS
    constexpr int COLS = 2 * LPR, ROWS = 4096 / COLS, RPI = 64 / LPR ;      // rows per instruction
    gdouble *F = (gdouble *) Fp ;
    const int lane = threadIdx.x & 63 ;
    const long wave = (long) blockIdx.x * 4 + (threadIdx.x >> 6) ;
    const long slabs_per_rowblock = ncols / COLS ;
    const long rb = wave / slabs_per_rowblock, cbk = wave % slabs_per_rowblock ;
    if (rb * ROWS >= nrows) return ;
    const size_t row0 = (size_t) rb * ROWS, col0 = (size_t) cbk * COLS ;
    const int r = lane / LPR, j = lane % LPR ;
    d2 a [32] ;
    #pragma unroll
    for (int i = 0 ; i < 32 ; i++) a [i] = *(gd2 *) (F + (row0 + RPI*i + r) * ld + col0 + 2*j) ;
    #pragma unroll
    for (int i = 0 ; i < 32 ; i++) { d2 v = a [i] ; v.x += 1.0 ; v.y += 1.0 ; *(gd2 *) (F + (row0 + RPI*i + r) * ld + col0 + 2*j) = v ; }
}

template <int LPR> void run (double *F, int ld, int ncols, int nrows)
{
    hipEvent_t e0, e1 ; hipEventCreate (&e0) ; hipEventCreate (&e1) ;
    long nwaves = (long) nrows * ncols / 4096 ;
    float best = 1e30f ;
    for (int r = 0 ; r < 6 ; r++)
    {
        hipEventRecord (e0) ;
        hipLaunchKernelGGL (k<LPR>, dim3 ((unsigned) ((nwaves + 3) / 4)), dim3 (256), 0, 0, F, ld, ncols, nrows) ;
        hipEventRecord (e1) ; hipEventSynchronize (e1) ;
        float ms ; hipEventElapsedTime (&ms, e0, e1) ;
        if (r > 0 && ms < best) best = ms ;
    }
    printf ("%3d lanes per row (%4d B x %2d rows per instruction, slab %4d rows x %3d cols): %.3f ms  %.2f TB/s\n", LPR, 16*LPR, 64/LPR,
        4096 / (2*LPR), 2*LPR, best, (double) nrows * ncols * 16 / best / 1e9) ;
}

int main (int argc, char **argv)
{
    int nrows = argc > 1 ? atoi (argv [1]) : 16384 ;
    int ncols = argc > 2 ? atoi (argv [2]) : 4096 ;
    int ld = ncols + (argc > 3 ? atoi (argv [3]) : 32) ;
    double *F ; hipMalloc (&F, (size_t) nrows * ld * 8) ; hipMemset (F, 0, (size_t) nrows * ld * 8) ;
    printf ("%d x %d, ld %d\n", nrows, ncols, ld) ;
    run<8> (F, ld, ncols, nrows) ; run<16> (F, ld, ncols, nrows) ; run<32> (F, ld, ncols, nrows) ; run<64> (F, ld, ncols, nrows) ;
    return 0 ;
}

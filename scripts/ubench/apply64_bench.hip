// Prototype, timing and a spot check only: the block update with 64 reflectors per pass instead of 32,
//   A <- A - V (T' (V' A)),   V 256 x 64, T 64 x 64 upper triangular,
// to see what the doubled arithmetic intensity (about 18 flop per byte of A) buys on this part.
// One workgroup of 8 waves per (256-row block, 256-column block); V (row stride 66 doubles) and the
// three nonzero 32 x 32 blocks of T fill the LDS of a compute unit, so one workgroup is resident;
// each wave keeps a 256 x 16 slab in registers as in apply_block<16>.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I suitesparse_amd/csrc scripts/ubench/apply64_bench.hip -o scripts/ubench/apply64_bench
//   apply64_bench [row blocks] [columns] [reps]
#include "qr_kernels.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define V64_LD 66
#define T64_LD 34
#define A64_THREADS 512
#define A64_LDS (256 * V64_LD * 8 + 3 * 32 * T64_LD * 8)

__global__ void __launch_bounds__ (A64_THREADS, 1)
k_apply64 (double *Fp, const double *Vp, const double *Tp, int ld, int ncols)
{
    extern __shared__ __attribute__((aligned(16))) char smem [] ;
    double (*Vs)[V64_LD] = reinterpret_cast<double (*)[V64_LD]>(smem) ;                          // [256][66]
    double (*T11)[T64_LD] = reinterpret_cast<double (*)[T64_LD]>(smem + 256 * V64_LD * 8) ;       // T(0:32, 0:32)
    double (*T12)[T64_LD] = T11 + 32 ;                                                           // T(0:32, 32:64)
    double (*T22)[T64_LD] = T11 + 64 ;                                                           // T(32:64, 32:64)
    constexpr int MT = 16 ;
    const int tid = threadIdx.x ;
    const int rb = blockIdx.y ;
    gdouble *F = (gdouble *) Fp + (size_t) rb * 256 * ld ;
    gcdouble *V = (gcdouble *) Vp + (size_t) rb * 256 * 64 ;
    gcdouble *T = (gcdouble *) Tp + (size_t) rb * 64 * 64 ;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane (tid >> 6) ;
    const int lr = lane >> 4, lc = lane & 15 ;
    const int l8 = lane >> 3 ;
    const int j2 = (lane & 7) * 2 ;
    const int pi_lc = 8*(lc >> 3) + 2*(lc & 3) + ((lc >> 2) & 1) ;
    const int c_begin = blockIdx.x * 256 ;
    const int c_end = (c_begin + 256 < ncols) ? (c_begin + 256) : ncols ;
    const int nslab = (c_end - c_begin + 15) >> 4 ;

    d2 A [MT][2] ;
    {
        const int slab0 = (wave < nslab) ? wave : 0 ;
        const int col = c_begin + slab0*16 + j2 ;
        const int lcol = (col < c_end - 2) ? col : (c_end - 2) ;
        #pragma unroll
        for (int mt = 0 ; mt < MT ; mt++)
        {
            #pragma unroll
            for (int h = 0 ; h < 2 ; h++)
            {
                const int r = mt*16 + 8*h + l8 ;
                A [mt][h] = *((gd2 *) (F + (size_t) r * ld + lcol)) ;
            }
        }
    }
    typedef __attribute__((address_space(1))) const d2 gcd2 ;
    for (int idx = tid ; idx < 256*32 ; idx += A64_THREADS)
    {
        const int r = idx >> 5, c = (idx & 31) * 2 ;
        const d2 val = *(gcd2 *) (V + r*64 + c) ;
        Vs [r][c] = val.x ; Vs [r][c + 1] = val.y ;
    }
    for (int idx = tid ; idx < 64*32 ; idx += A64_THREADS)
    {
        const int r = idx >> 5, c = (idx & 31) * 2 ;
        const d2 tv = *(gcd2 *) (T + r*64 + c) ;
        const double x = (c >= r) ? tv.x : 0.0, y = (c + 1 >= r) ? tv.y : 0.0 ;
        if (r < 32 && c < 32)        { T11 [r][c] = x ; T11 [r][c+1] = y ; }
        else if (r < 32)             { T12 [r][c-32] = x ; T12 [r][c-31] = y ; }
        else if (c >= 32)            { T22 [r-32][c-32] = x ; T22 [r-32][c-31] = y ; }
    }
    __syncthreads ( ) ;

    for (int slab = wave ; slab < nslab ; slab += 8)
    {
        const int col = c_begin + slab*16 + j2 ;
        const bool v0 = col < c_end, v1 = col + 1 < c_end ;
        const int ncol = (slab + 8 < nslab) ? (col + 128) : col ;
        const int nlcol = (ncol < c_end - 2) ? ncol : (c_end - 2) ;

        // step 1: W = V' A, four 16-row tiles of W
        v4f64 W0 = {0,0,0,0}, W1 = {0,0,0,0}, W2 = {0,0,0,0}, W3 = {0,0,0,0} ;
        #pragma unroll
        for (int mt = 0 ; mt < MT ; mt++)
        {
            #pragma unroll
            for (int h = 0 ; h < 2 ; h++)
            {
                d2 v = A [mt][h] ;
                v.x = v1 ? v.x : v.y ;
                A [mt][h] = pair_exchange (v) ;
                #pragma unroll
                for (int e = 0 ; e < 2 ; e++)
                {
                    const int r = mt*16 + 8*h + 2*lr + e ;
                    const double b = A [mt][h][e] ;
                    W0 = MFMA64 (Vs [r][lc],      b, W0) ;
                    W1 = MFMA64 (Vs [r][16 + lc], b, W1) ;
                    W2 = MFMA64 (Vs [r][32 + lc], b, W2) ;
                    W3 = MFMA64 (Vs [r][48 + lc], b, W3) ;
                }
            }
        }

        // step 2: X = T' W, T upper triangular in three 32 x 32 blocks
        v4f64 X0 = {0,0,0,0}, X1 = {0,0,0,0}, X2 = {0,0,0,0}, X3 = {0,0,0,0} ;
        #pragma unroll
        for (int q = 0 ; q < 4 ; q++)
        {
            X0 = MFMA64 (T11 [4*q + lr][lc],           W0 [q], X0) ;
            X1 = MFMA64 (T11 [4*q + lr][16 + lc],      W0 [q], X1) ;
            X1 = MFMA64 (T11 [16 + 4*q + lr][16 + lc], W1 [q], X1) ;
            X2 = MFMA64 (T12 [4*q + lr][lc],           W0 [q], X2) ;
            X2 = MFMA64 (T12 [16 + 4*q + lr][lc],      W1 [q], X2) ;
            X3 = MFMA64 (T12 [4*q + lr][16 + lc],      W0 [q], X3) ;
            X3 = MFMA64 (T12 [16 + 4*q + lr][16 + lc], W1 [q], X3) ;
            X2 = MFMA64 (T22 [4*q + lr][lc],           W2 [q], X2) ;
            X3 = MFMA64 (T22 [4*q + lr][16 + lc],      W2 [q], X3) ;
            X3 = MFMA64 (T22 [16 + 4*q + lr][16 + lc], W3 [q], X3) ;
        }
        X0 = -X0 ; X1 = -X1 ; X2 = -X2 ; X3 = -X3 ;

        // step 3: A += V (-X), tile by tile; store the tile, fetch its successor
        int l8_st = l8 ;
        asm volatile ("" : "+v" (l8_st)) ;
        #pragma unroll
        for (int mt = 0 ; mt < MT ; mt++)
        {
            v4f64 acc = {A [mt][0][0], A [mt][0][1], A [mt][1][0], A [mt][1][1]} ;
            #pragma unroll
            for (int q = 0 ; q < 4 ; q++) acc = MFMA64 (Vs [mt*16 + pi_lc][4*q + lr],      X0 [q], acc) ;
            #pragma unroll
            for (int q = 0 ; q < 4 ; q++) acc = MFMA64 (Vs [mt*16 + pi_lc][16 + 4*q + lr], X1 [q], acc) ;
            #pragma unroll
            for (int q = 0 ; q < 4 ; q++) acc = MFMA64 (Vs [mt*16 + pi_lc][32 + 4*q + lr], X2 [q], acc) ;
            #pragma unroll
            for (int q = 0 ; q < 4 ; q++) acc = MFMA64 (Vs [mt*16 + pi_lc][48 + 4*q + lr], X3 [q], acc) ;
            #pragma unroll
            for (int h = 0 ; h < 2 ; h++)
            {
                const int r = mt*16 + 8*h + l8_st ;
                d2 o = {acc [2*h], acc [2*h + 1]} ;
                o = pair_exchange (o) ;
                gdouble *rowp = F + (size_t) r * ld ;
                if (v1)      *((gd2 *) (rowp + col)) = o ;
                else if (v0) rowp [col] = o.x ;
                A [mt][h] = *((gd2 *) (rowp + nlcol)) ;
            }
        }
    }
}

int main (int argc, char **argv)
{
    int nrb   = argc > 1 ? atoi (argv [1]) : 64 ;
    int ncols = argc > 2 ? atoi (argv [2]) : 8192 ;
    int reps  = argc > 3 ? atoi (argv [3]) : 5 ;
    const int ld = ncols ;
    const size_t rows = (size_t) nrb * 256 ;
    double *F, *V, *T ;
    hipMalloc (&F, rows * ld * 8) ; hipMalloc (&V, rows * 64 * 8) ; hipMalloc (&T, (size_t) nrb * 4096 * 8) ;
    std::vector<double> h (rows * ld), hv (rows * 64), ht ((size_t) nrb * 4096) ;
    for (auto &x : h)  x = (double) rand ( ) / RAND_MAX - 0.5 ;
    for (auto &x : hv) x = 0.1 * ((double) rand ( ) / RAND_MAX - 0.5) ;
    for (auto &x : ht) x = 0.1 * ((double) rand ( ) / RAND_MAX - 0.5) ;
    hipMemcpy (F, h.data ( ), h.size ( ) * 8, hipMemcpyHostToDevice) ;
    hipMemcpy (V, hv.data ( ), hv.size ( ) * 8, hipMemcpyHostToDevice) ;
    hipMemcpy (T, ht.data ( ), ht.size ( ) * 8, hipMemcpyHostToDevice) ;
    if (hipFuncSetAttribute ((const void *) k_apply64, hipFuncAttributeMaxDynamicSharedMemorySize, A64_LDS) != hipSuccess)
    { printf ("LDS opt-in failed (%d bytes)\n", A64_LDS) ; return 1 ; }
    dim3 grid ((ncols + 255) / 256, nrb) ;
    hipEvent_t e0, e1 ; hipEventCreate (&e0) ; hipEventCreate (&e1) ;

    // spot check of the first launch: block row 0, columns 0..15 and the last 16, against the host
    hipLaunchKernelGGL (k_apply64, grid, dim3 (A64_THREADS), A64_LDS, 0, F, V, T, ld, ncols) ;
    if (hipDeviceSynchronize ( ) != hipSuccess) { printf ("launch failed\n") ; return 1 ; }
    {
        std::vector<double> got (256 * (size_t) ld) ;
        hipMemcpy (got.data ( ), F, got.size ( ) * 8, hipMemcpyDeviceToHost) ;
        double maxerr = 0 ;
        for (int pass = 0 ; pass < 2 ; pass++)
        {
            const int c0 = pass ? ncols - 16 : 0 ;
            for (int j = c0 ; j < c0 + 16 ; j++)
            {
                double W [64], X [64] ;
                for (int k = 0 ; k < 64 ; k++) { W [k] = 0 ; for (int r = 0 ; r < 256 ; r++) W [k] += hv [r*64 + k] * h [(size_t) r*ld + j] ; }
                for (int k = 0 ; k < 64 ; k++) { X [k] = 0 ; for (int i = 0 ; i <= k ; i++) X [k] += ht [i*64 + k] * W [i] ; }
                for (int r = 0 ; r < 256 ; r++)
                {
                    double ref = h [(size_t) r*ld + j] ;
                    for (int k = 0 ; k < 64 ; k++) ref -= hv [r*64 + k] * X [k] ;
                    maxerr = fmax (maxerr, fabs (ref - got [(size_t) r*ld + j])) ;
                }
            }
        }
        printf ("spot check: max abs error %.3e\n", maxerr) ;
        if (!(maxerr < 1e-12)) return 1 ;
    }

    float best = 1e30f ;
    for (int r = 0 ; r < reps ; r++)
    {
        hipEventRecord (e0) ;
        hipLaunchKernelGGL (k_apply64, grid, dim3 (A64_THREADS), A64_LDS, 0, F, V, T, ld, ncols) ;
        hipEventRecord (e1) ; hipEventSynchronize (e1) ;
        float ms ; hipEventElapsedTime (&ms, e0, e1) ;
        if (ms < best) best = ms ;
    }
    const double slabs = (double) nrb * ncols / 16.0 ;
    const double flops = slabs * (256.0 + 40.0 + 256.0) * 2048.0 ;      // MFMAs issued per slab x flops each
    const double bytes = (double) rows * ncols * 16.0 ;
    printf ("apply64 %d x %d: %.3f ms, %.1f TF/s issued, %.2f TB/s of A, equals %.1f TF/s of 32-wide work\n",
        (int) rows, ncols, best, flops / best / 1e9, bytes / best / 1e9,
        2.0 * slabs * (128.0 + 8.0 + 128.0) * 2048.0 / best / 1e9) ;
    return 0 ;
}

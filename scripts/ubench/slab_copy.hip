// Microbenchmark: streaming 256-row x 16-column slabs of a row-major FP64 matrix through
// registers (load slab, add 1, store slab), with 8-byte or 16-byte accesses per lane.
//   X2: lane (r4 = l/16, c = l%16) moves rows 4i+r4, one double     (4 rows x 128 B per instruction)
//   X4: lane (r8 = l/8,  j = l%8)  moves rows 8i+r8, two doubles    (8 rows x 128 B per instruction)
// and two patterns that say what the memory system gives when the layout is not in the way:
//   ROW: a wave moves whole 1 KB row segments of the same 256 x 128 block (1 row x 1 KB per instruction)
//   LIN: the whole array as one stream, 16 B per lane, 4 KB per workgroup and instruction
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) double gdouble ;
typedef double d2 __attribute__((ext_vector_type(2))) ;
typedef __attribute__((address_space(1))) d2 gd2 ;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k (double *Fp, int ld, int nslab_per_wg, int blocks_per_task)
{
    gdouble *F = (gdouble *) Fp ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6 ;
    const int task = blockIdx.x / blocks_per_task, blk = blockIdx.x % blocks_per_task ;
    const size_t row0 = (size_t) task * 256 ;
    for (int s = wave ; s < nslab_per_wg ; s += 4)
    {
        const int col0 = (blk * nslab_per_wg + s) * 16 ;
        if (MODE == 8)
        {
            // the workgroup's block is 256 rows x (nslab_per_wg * 16) columns; this variant needs 128 columns
            if (s != wave) break ;
            const int colb = blk * nslab_per_wg * 16 ;
            for (int half = 0 ; half < 2 ; half++)
            {
                d2 a [32] ;
                #pragma unroll
                for (int i = 0 ; i < 32 ; i++) a [i] = *(gd2 *) (F + (row0 + half * 128 + 4*i + wave) * ld + colb + 2*lane) ;
                #pragma unroll
                for (int i = 0 ; i < 32 ; i++) { d2 v = a [i] ; v.x += 1.0 ; v.y += 1.0 ; *(gd2 *) (F + (row0 + half * 128 + 4*i + wave) * ld + colb + 2*lane) = v ; }
            }
        }
        else if (MODE == 2)
        {
            double a [64] ;
            const int r4 = lane >> 4, c = lane & 15 ;
            #pragma unroll
            for (int i = 0 ; i < 64 ; i++) a [i] = F [(row0 + 4*i + r4) * ld + col0 + c] ;
            #pragma unroll
            for (int i = 0 ; i < 64 ; i++) F [(row0 + 4*i + r4) * ld + col0 + c] = a [i] + 1.0 ;
        }
        else if (MODE == 16 || MODE == 17 || MODE == 18)
        {
            // x4 with non-temporal loads (16, 18) and stores (17, 18)
            d2 a [32] ;
            const int r8 = lane >> 3, j = lane & 7 ;
            #pragma unroll
            for (int i = 0 ; i < 32 ; i++)
            {
                gd2 *q = (gd2 *) (F + (row0 + 8*i + r8) * ld + col0 + 2*j) ;
                a [i] = (MODE != 17) ? __builtin_nontemporal_load (q) : *q ;
            }
            #pragma unroll
            for (int i = 0 ; i < 32 ; i++)
            {
                d2 v = a [i] ; v.x += 1.0 ; v.y += 1.0 ;
                gd2 *q = (gd2 *) (F + (row0 + 8*i + r8) * ld + col0 + 2*j) ;
                if (MODE != 16) __builtin_nontemporal_store (v, q) ; else *q = v ;
            }
        }
        else
        {
            d2 a [32] ;
            const int r8 = lane >> 3, j = lane & 7 ;
            #pragma unroll
            for (int i = 0 ; i < 32 ; i++) a [i] = *(gd2 *) (F + (row0 + 8*i + r8) * ld + col0 + 2*j) ;
            #pragma unroll
            for (int i = 0 ; i < 32 ; i++) { d2 v = a [i] ; v.x += 1.0 ; v.y += 1.0 ; *(gd2 *) (F + (row0 + 8*i + r8) * ld + col0 + 2*j) = v ; }
        }
    }
}

__global__ void __launch_bounds__(256, 2) k_linear (double *Fp, size_t n2)
{
    gd2 *F = (gd2 *) Fp ;
    // each workgroup streams 32 pieces of 4 KB that lie 4 KB * gridDim apart
    for (size_t base = (size_t) blockIdx.x * 256 + threadIdx.x ; base < n2 ; base += (size_t) gridDim.x * 256 * 8)
    {
        d2 a [8] ;
        #pragma unroll
        for (int i = 0 ; i < 8 ; i++) { size_t q = base + (size_t) i * gridDim.x * 256 ; if (q < n2) a [i] = F [q] ; }
        #pragma unroll
        for (int i = 0 ; i < 8 ; i++) { size_t q = base + (size_t) i * gridDim.x * 256 ; if (q < n2) { d2 v = a [i] ; v.x += 1.0 ; v.y += 1.0 ; F [q] = v ; } }
    }
}

int main (int argc, char **argv)
{
    int ntask = argc > 1 ? atoi (argv [1]) : 64 ;
    int ncols = argc > 2 ? atoi (argv [2]) : 4096 ;
    int cb = argc > 3 ? atoi (argv [3]) : 128 ;
    int ld = ncols + 32 ;
    size_t rows = (size_t) ntask * 256 ;
    double *F ; hipMalloc (&F, rows * ld * 8) ; hipMemset (F, 0, rows * ld * 8) ;
    hipEvent_t e0, e1 ; hipEventCreate (&e0) ; hipEventCreate (&e1) ;
    for (int mode = 2 ; mode <= 4 ; mode += 2)
    {
        float best = 1e30f ;
        for (int r = 0 ; r < 6 ; r++)
        {
            int bpt = ncols / cb ;
            hipEventRecord (e0) ;
            if (mode == 2) hipLaunchKernelGGL (k<2>, dim3 (ntask * bpt), dim3 (256), 0, 0, F, ld, cb / 16, bpt) ;
            else           hipLaunchKernelGGL (k<4>, dim3 (ntask * bpt), dim3 (256), 0, 0, F, ld, cb / 16, bpt) ;
            hipEventRecord (e1) ; hipEventSynchronize (e1) ;
            float ms ; hipEventElapsedTime (&ms, e0, e1) ;
            if (r > 0 && ms < best) best = ms ;
        }
        printf ("x%d: tasks %d cols %d cb %d  %.3f ms  %.2f TB/s\n", mode, ntask, ncols, cb, best, (double) rows * ncols * 16 / best / 1e9) ;
    }
    if (cb == 128)
    {
        float best = 1e30f ;
        int bpt = ncols / cb ;
        for (int r = 0 ; r < 6 ; r++)
        {
            hipEventRecord (e0) ;
            hipLaunchKernelGGL (k<8>, dim3 (ntask * bpt), dim3 (256), 0, 0, F, ld, cb / 16, bpt) ;
            hipEventRecord (e1) ; hipEventSynchronize (e1) ;
            float ms ; hipEventElapsedTime (&ms, e0, e1) ;
            if (r > 0 && ms < best) best = ms ;
        }
        printf ("row: tasks %d cols %d cb %d  %.3f ms  %.2f TB/s\n", ntask, ncols, cb, best, (double) rows * ncols * 16 / best / 1e9) ;
    }
    for (int mode = 16 ; mode <= 18 ; mode++)
    {
        float best = 1e30f ;
        int bpt = ncols / cb ;
        for (int r = 0 ; r < 6 ; r++)
        {
            hipEventRecord (e0) ;
            if (mode == 16)      hipLaunchKernelGGL (k<16>, dim3 (ntask * bpt), dim3 (256), 0, 0, F, ld, cb / 16, bpt) ;
            else if (mode == 17) hipLaunchKernelGGL (k<17>, dim3 (ntask * bpt), dim3 (256), 0, 0, F, ld, cb / 16, bpt) ;
            else                 hipLaunchKernelGGL (k<18>, dim3 (ntask * bpt), dim3 (256), 0, 0, F, ld, cb / 16, bpt) ;
            hipEventRecord (e1) ; hipEventSynchronize (e1) ;
            float ms ; hipEventElapsedTime (&ms, e0, e1) ;
            if (r > 0 && ms < best) best = ms ;
        }
        printf ("x4 non-temporal %s: tasks %d cols %d cb %d  %.3f ms  %.2f TB/s\n", mode == 16 ? "loads" : (mode == 17 ? "stores" : "loads and stores"),
            ntask, ncols, cb, best, (double) rows * ncols * 16 / best / 1e9) ;
    }
    for (int wgs = 2048 ; wgs <= 32768 ; wgs *= 4)
    {
        float best = 1e30f ;
        const size_t n2 = rows * ld / 2 ;
        for (int r = 0 ; r < 6 ; r++)
        {
            hipEventRecord (e0) ;
            hipLaunchKernelGGL (k_linear, dim3 (wgs), dim3 (256), 0, 0, F, n2) ;
            hipEventRecord (e1) ; hipEventSynchronize (e1) ;
            float ms ; hipEventElapsedTime (&ms, e0, e1) ;
            if (r > 0 && ms < best) best = ms ;
        }
        printf ("linear, %d workgroups: %.3f ms  %.2f TB/s\n", wgs, best, (double) n2 * 32 / best / 1e9) ;
    }
    {
        // the runtime's own device-to-device copy of the same array (read + write counted)
        double *G ; hipMalloc (&G, rows * ld * 8) ;
        float best = 1e30f ;
        for (int r = 0 ; r < 4 ; r++)
        {
            hipEventRecord (e0) ;
            hipMemcpyAsync (G, F, rows * ld * 8, hipMemcpyDeviceToDevice, 0) ;
            hipEventRecord (e1) ; hipEventSynchronize (e1) ;
            float ms ; hipEventElapsedTime (&ms, e0, e1) ;
            if (r > 0 && ms < best) best = ms ;
        }
        printf ("hipMemcpy device to device: %.3f ms  %.2f TB/s\n", best, (double) rows * ld * 16 / best / 1e9) ;
        hipFree (G) ;
    }
    return 0 ;
}

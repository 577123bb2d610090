// Are the hardware seeds of 1/sqrt(x) and 1/x, and the Newton steps the panel kernel builds on them,
// exactly scale-invariant under x -> 16 x (values -> 4 x values)?  Counts the inputs for which
// rsq (16 x) != rsq (x) / 4, rcp (16 x) != rcp (x) / 16, and for which the refined square root differs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k (const double *x, int n, unsigned long long *bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x ;
    if (i >= n) return ;
    const double a = x [i], b = 16.0 * a ;
    const double ya = __builtin_amdgcn_rsq (a), yb = __builtin_amdgcn_rsq (b) ;
    const double ra = __builtin_amdgcn_rcp (a), rb = __builtin_amdgcn_rcp (b) ;
    if (yb * 4.0 != ya) atomicAdd (&bad [0], 1ull) ;
    if (rb * 16.0 != ra) atomicAdd (&bad [1], 1ull) ;
    auto root = [] (double n2, double y)
    {
        y = fma (0.5 * y, fma (-n2 * y, y, 1.0), y) ;
        y = fma (0.5 * y, fma (-n2 * y, y, 1.0), y) ;
        double s = n2 * y ;
        return fma (fma (-s, s, n2), 0.5 * y, s) ;
    } ;
    if (root (b, yb) != 4.0 * root (a, ya)) atomicAdd (&bad [2], 1ull) ;
}

int main ( )
{
    const int n = 1 << 24 ;
    double *h = (double *) malloc (n * sizeof (double)), *d ;
    srand (1) ;
    for (int i = 0 ; i < n ; i++) h [i] = (1.0 + rand ( ) / (double) RAND_MAX * 40.0) * ((i & 1) ? 1.0 : 1e-3) ;
    unsigned long long *bad, hb [3] = {0, 0, 0} ;
    if (hipMalloc (&d, n * sizeof (double)) != hipSuccess || hipMalloc (&bad, sizeof (hb)) != hipSuccess) return 1 ;
    (void) hipMemcpy (d, h, n * sizeof (double), hipMemcpyHostToDevice) ;
    (void) hipMemcpy (bad, hb, sizeof (hb), hipMemcpyHostToDevice) ;
    hipLaunchKernelGGL (k, dim3 (n / 256), dim3 (256), 0, 0, d, n, bad) ;
    (void) hipMemcpy (hb, bad, sizeof (hb), hipMemcpyDeviceToHost) ;
    printf ("%d inputs: rsq seed not scale-invariant for %llu, rcp seed for %llu, refined square root for %llu\n", n, hb [0], hb [1], hb [2]) ;
    return 0 ;
}

// The reference's dense demo, restated against the shim header: the four
// experiments of GPUQREngine/Demo/gpuqrengine_demo.cpp:305-308 with its
// srand(1), rand()/1e8 fill (:21-28,283) and the check its MATLAB script makes,
// || triu(R)'triu(R) - A'A || / || A'A || <= 1e-12 (:94-100), done here in C++.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "GPUQREngine.hpp"

static double check (const double *A, const double *R, Int m, Int n)
{
    Int r = m < n ? m : n ;
    double num = 0, den = 0 ;
    for (Int j = 0 ; j < n ; j++)
    {
        double cn = 0, cd = 0 ;
        for (Int i = 0 ; i < n ; i++)
        {
            double g = 0, h = 0 ;
            for (Int k = 0 ; k < m ; k++) g += A [k*n+i] * A [k*n+j] ;
            Int top = (i < j ? i : j) ; if (top >= r) top = r - 1 ;
            for (Int k = 0 ; k <= top ; k++) h += R [k*n+i] * R [k*n+j] ;
            cn += fabs (h - g) ; cd += fabs (g) ;
        }
        if (cn > num) num = cn ;
        if (cd > den) den = cd ;
    }
    return num / den ;
}

static int experiment (int count, Int m, Int n)
{
    std::vector<std::vector<double> > A (count), R (count) ;
    Front *fronts = (Front *) malloc (count * sizeof (Front)) ;
    for (int f = 0 ; f < count ; f++)
    {
        A [f].resize (m*n) ; R [f].assign ((m < n ? m : n) * n, 0.0) ;
        for (Int i = 0 ; i < m*n ; i++) A [f][i] = rand ( ) / 1e8 ;
        Front *front = new (&fronts [f]) Front (f, EMPTY, m, n, n) ;
        front->F = A [f].data ( ) ;
        front->cpuR = R [f].data ( ) ;
    }
    QREngineStats stats ;
    QREngineResultCode code = GPUQREngine (0, fronts, count, &stats) ;
    int bad = (code != QRENGINE_SUCCESS) ;
    for (int f = 0 ; f < count && !bad ; f++)
    {
        double err = check (A [f].data ( ), R [f].data ( ), m, n) ;
        printf ("%d x (%ld x %ld) front %d: err %.2e launches %ld flops %ld\n", count, m, n, f, err, stats.numLaunches, stats.flopsActual) ;
        if (!(err <= 1e-12)) bad = 1 ;
    }
    free (fronts) ;
    return bad ;
}

int main (int argc, char **argv)
{
    if (argc > 1) { printf ("compiled and linked\n") ; return 0 ; }     // CPU-only link check
    srand (1) ;
    int bad = 0 ;
    bad |= experiment (2, 8, 8) ;
    bad |= experiment (2, 12, 8) ;
    bad |= experiment (2, 64, 32) ;
    bad |= experiment (1, 100, 200) ;
    printf (bad ? "FAILED\n" : "all dense demo experiments pass\n") ;
    return bad ;
}

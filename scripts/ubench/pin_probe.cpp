// How much host memory the box pins: hipHostMalloc of growing sizes, then 16 GB chunks until refusal.
//   hipcc -O2 scripts/ubench/pin_probe.cpp -o scripts/ubench/pin_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>
int main ()
{
    size_t gb = 1ull << 30 ;
    size_t sizes [] = {32, 48, 64, 80, 96} ;
    for (size_t s : sizes)
    {
        void *p = NULL ;
        auto t0 = std::chrono::steady_clock::now () ;
        hipError_t e = hipHostMalloc (&p, s * gb, hipHostMallocDefault) ;
        double ms = std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t0).count () ;
        printf ("single %zu GB: %s (%.0f ms)\n", s, hipGetErrorName (e), ms) ; fflush (stdout) ;
        if (e == hipSuccess) hipHostFree (p) ; else (void) hipGetLastError () ;
    }
    std::vector<void *> ps ;
    for (int i = 0 ; i < 8 ; i++)
    {
        void *p = NULL ;
        hipError_t e = hipHostMalloc (&p, 16 * gb, hipHostMallocDefault) ;
        printf ("chunk %d of 16 GB: %s\n", i, hipGetErrorName (e)) ; fflush (stdout) ;
        if (e != hipSuccess) { (void) hipGetLastError () ; break ; }
        ps.push_back (p) ;
    }
    for (void *p : ps) hipHostFree (p) ;
    return 0 ;
}

// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0)
{
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0,0,0,0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int i = 0; i < NACC; i++)
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC> void run(int blocks_per_cu, int threads)
{
    int nb = 256 * blocks_per_cu;
    double *out; hipMalloc(&out, (size_t)nb * threads * 8);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(nb), dim3(threads), 0, 0, out, 10, 1.0, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(nb), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nmfma = (double)nb * (threads / 64) * iters * NACC;
    double tf = nmfma * 2048 / (ms * 1e-3) / 1e12;
    double per_simd = (double)blocks_per_cu * (threads / 64) / 4.0;   // waves per SIMD
    double cyc = ms * 1e-3 * 2.4e9 / (iters * NACC * per_simd);
    printf("NACC %d  blocks/CU %d threads %d (%.1f waves/SIMD): %.2f ms  %.1f TF  ~%.1f cycles@2.4GHz per MFMA per SIMD\n",
           NACC, blocks_per_cu, threads, per_simd, ms, tf, cyc);
    hipFree(out);
}

int main()
{
    run<1>(1, 256); run<2>(1, 256); run<4>(1, 256); run<8>(1, 256);
    run<1>(2, 256); run<4>(2, 256); run<8>(2, 256); run<4>(4, 256);
    return 0;
}

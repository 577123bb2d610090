// Microbenchmark: issue rate of v_fma_f64 on gfx950, alone and beside v_mfma_f64_16x16x4_f64 on the
// same SIMDs (are the two FP64 pipes one pipe?).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));

// MODE 0: NACC independent FMA chains per lane.  MODE 1: MFMA only.  MODE 2: even waves FMA, odd waves MFMA.
template <int NACC, int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, double a0, double b0)
{
    const int wave = threadIdx.x >> 6;
    double a = a0 + 1e-9 * threadIdx.x, b = b0 - 1e-9 * threadIdx.x;
    double s = 0;
    if (MODE == 0 || (MODE == 2 && (wave & 1) == 0))
    {
        double acc[NACC];
        for (int i = 0; i < NACC; i++) acc[i] = i;
        for (int it = 0; it < iters; it++) {
            #pragma unroll
            for (int u = 0; u < 8; u++) {
                #pragma unroll
                for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(a, acc[i], b);
            }
        }
        for (int i = 0; i < NACC; i++) s += acc[i];
    }
    else
    {
        v4f64 acc[NACC];
        for (int i = 0; i < NACC; i++) acc[i] = (v4f64){0,0,0,0};
        for (int it = 0; it < iters; it++) {
            #pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE> void run(int blocks_per_cu)
{
    const int threads = 256, nb = 256 * blocks_per_cu, iters = 4000;
    double *out; hipMalloc(&out, (size_t)nb * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(nb), dim3(threads), 0, 0, out, 10, 0.999, 2.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(nb), dim3(threads), 0, 0, out, iters, 0.999, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)nb * 4;
    const double fma_waves = MODE == 0 ? waves : (MODE == 2 ? waves / 2 : 0), mfma_waves = waves - fma_waves;
    const double fl_fma = fma_waves * iters * 8.0 * NACC * 64 * 2, fl_mfma = mfma_waves * iters * (double)NACC * 2048;
    printf("mode %d NACC %d blocks/CU %d: %.2f ms  v_fma_f64 %.1f TF + mfma %.1f TF = %.1f TF\n", MODE, NACC, blocks_per_cu, ms,
           fl_fma / (ms * 1e-3) / 1e12, fl_mfma / (ms * 1e-3) / 1e12, (fl_fma + fl_mfma) / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main()
{
    run<8, 0>(1); run<8, 0>(2); run<8, 0>(4);
    run<8, 1>(2); run<8, 1>(4);
    run<8, 2>(2); run<8, 2>(4);
    return 0;
}

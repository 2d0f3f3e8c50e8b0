// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate of an MI355X (no memory traffic).
// Build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_f64_peak mfma_f64_peak.hip ; run: ./mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int wgs_per_cu, int n_cu) {
    const int grid = wgs_per_cu * n_cu, iters = 4000;
    double* out;
    hipMalloc(&out, sizeof(double) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 100, 1.0, 2.0);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NACC><<<grid, 256>>>(out, iters, 1.0, 2.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2048.0 * NACC * iters * 4.0 * grid;
        printf("acc=%d wg/cu=%d  %.3f ms  %.1f TFLOP/s\n", NACC, wgs_per_cu, ms, flop / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<4>(1, p.multiProcessorCount);
    run<16>(1, p.multiProcessorCount);
    run<16>(2, p.multiProcessorCount);
    run<8>(4, p.multiProcessorCount);
    run<4>(8, p.multiProcessorCount);
    run<8>(3, p.multiProcessorCount);
    run<8>(1, 64);      // a quarter of the chip: is the ceiling a power/clock limit?
    return 0;
}

// Micro-benchmark: v_mfma_f32_16x16x4_f32 with VGPR accumulators in NCHAIN dependent chains
// (pass 1 of the attend path uses 4), at 1 and 4 workgroups (waves per SIMD) per CU.
// Build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_f32_chains mfma_f32_chains.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NCHAIN, bool AGPR>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f4 acc[NCHAIN];
    for (int i = 0; i < NCHAIN; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / NCHAIN; ++rep)
#pragma unroll
            for (int i = 0; i < NCHAIN; ++i) {
                if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+a"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
    }
    float s = 0;
    for (int i = 0; i < NCHAIN; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCHAIN, bool AGPR>
void run(int wgs_per_cu, int n_cu) {
    const int grid = wgs_per_cu * n_cu, iters = 1000;
    float* out;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NCHAIN, AGPR><<<grid, 256>>>(out, 50, 1.0f, 2.0f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NCHAIN, AGPR><<<grid, 256>>>(out, iters, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2048.0 * 64 * iters * 4.0 * grid;
    printf("chains=%2d acc=%s wg/cu=%d  %.3f ms  %.1f%% of 157.3 TFLOP/s\n", NCHAIN, AGPR ? "agpr" : "vgpr",
           wgs_per_cu, best, flop / best / 1e9 / 1.573);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n = p.multiProcessorCount;
    run<4, false>(1, n); run<4, true>(1, n); run<8, false>(1, n); run<16, false>(1, n);
    run<2, false>(1, n); run<1, false>(1, n); run<1, false>(4, n);
    run<4, false>(4, n); run<4, true>(4, n); run<2, false>(4, n); run<16, false>(4, n);
    return 0;
}

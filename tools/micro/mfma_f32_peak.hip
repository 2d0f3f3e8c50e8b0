// Micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate of an MI355X (no memory traffic),
// one workgroup of 4 waves per CU (one wave per SIMD, the occupancy of the attend kernels).
// Build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_f32_peak mfma_f32_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, bool NOP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (NOP) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+a"(acc[i]) : "v"(a), "v"(b));
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool NOP>
void run(int wgs_per_cu, int n_cu) {
    const int grid = wgs_per_cu * n_cu, iters = 4000;
    float* out;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, NOP><<<grid, 256>>>(out, 100, 1.0f, 2.0f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<NACC, NOP><<<grid, 256>>>(out, iters, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2048.0 * NACC * iters * 4.0 * grid;
    printf("acc=%d asm+nop=%d wg/cu=%d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", NACC, (int)NOP,
           wgs_per_cu, best, flop / best / 1e9, flop / best / 1e9 / 1.573);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
    run<16, false>(1, p.multiProcessorCount);
    run<16, true>(1, p.multiProcessorCount);
    run<64, true>(1, p.multiProcessorCount);
    run<16, true>(2, p.multiProcessorCount);
    run<16, false>(4, p.multiProcessorCount);
    return 0;
}

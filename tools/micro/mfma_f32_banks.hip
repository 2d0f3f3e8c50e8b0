// Micro-benchmark: does the VGPR bank of the A / B operands change the issue rate of
// v_mfma_f32_16x16x4_f32 (AGPR accumulators, one wave per SIMD)?
// Build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_f32_banks mfma_f32_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

#define MF(A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, " A ", " B ", %0\n\ts_nop 1" : "+a"(acc[i]))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0) {
    f4 acc[64];
    for (int i = 0; i < 64; ++i) acc[i] = f4{0, 0, 0, 0};
    asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, %0\n\tv_mov_b32 v102, %0\n\tv_mov_b32 v103, %0\n\t"
                 "v_mov_b32 v104, %0\n\tv_mov_b32 v105, %0\n\tv_mov_b32 v106, %0\n\tv_mov_b32 v107, %0\n\t"
                 "v_mov_b32 v108, %0"
                 :: "v"(a0 + threadIdx.x) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (MODE == 0) MF("v100", "v101");            // banks 0,1
            else if (MODE == 1) MF("v100", "v104");       // same bank
            else if (MODE == 2) MF("v100", "v100");       // same register
            else if (MODE == 3) {                          // like the kernel: A fixed, B cycles x,y,z,w
                switch (i & 3) {
                    case 0: MF("v108", "v100"); break;
                    case 1: MF("v108", "v101"); break;
                    case 2: MF("v108", "v102"); break;
                    default: MF("v108", "v103"); break;
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int n_cu) {
    const int grid = n_cu, iters = 2000;
    float* out;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, 50, 1.0f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<grid, 256>>>(out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2048.0 * 64 * iters * 4.0 * grid;
    printf("mode=%d  %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", MODE, best, flop / best / 1e9, flop / best / 1e9 / 1.573);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    run<0>(p.multiProcessorCount);
    run<1>(p.multiProcessorCount);
    run<2>(p.multiProcessorCount);
    run<3>(p.multiProcessorCount);
    return 0;
}

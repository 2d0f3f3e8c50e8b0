// Micro-benchmark: the MFMA operand pattern of pass 1 (QK^T): 4 dependent chains with VGPR
// accumulators, A operand from a rotating set of registers, B operand from 64 distinct registers
// (the query fragments), 64 MFMAs per "tile", no memory traffic.
// Build: hipcc -w --offload-arch=gfx950 -O3 -o qk mfma_f32_qk_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool AGPR, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k(float* out, int iters, const float* in) {
    f4 q[16];
    for (int s = 0; s < 16; ++s) q[s] = *reinterpret_cast<const f4*>(in + 4 * ((threadIdx.x + s * 64) & 1023));
    f4 ka[4];
    for (int s = 0; s < 4; ++s) ka[s] = *reinterpret_cast<const f4*>(in + 4 * ((threadIdx.x * 3 + s) & 1023));
    for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(q[s]));
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(ka[s]));
    f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f4 kk = ka[s & 3];
            if (AGPR) {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+a"(a0) : "v"(kk.x), "v"(q[s].x));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+a"(a1) : "v"(kk.y), "v"(q[s].y));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+a"(a2) : "v"(kk.z), "v"(q[s].z));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+a"(a3) : "v"(kk.w), "v"(q[s].w));
            } else {
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(a0) : "v"(kk.x), "v"(q[s].x));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(a1) : "v"(kk.y), "v"(q[s].y));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(a2) : "v"(kk.z), "v"(q[s].z));
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(a3) : "v"(kk.w), "v"(q[s].w));
            }
        }
    }
    asm volatile("s_nop 15" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

template <bool AGPR, int W>
void run(int n_cu, const float* in) {
    const int grid = W * n_cu, iters = 2000;
    float* out;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<AGPR, W><<<grid, 256>>>(out, 50, in);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        k<AGPR, W><<<grid, 256>>>(out, iters, in);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = 2048.0 * 64 * iters * 4.0 * grid;
    printf("acc=%s waves/SIMD=%d  %.3f ms  %.1f%% of 157.3 TFLOP/s\n", AGPR ? "agpr" : "vgpr", W, best,
           flop / best / 1e9 / 1.573);
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* in;
    hipMalloc(&in, 4096 * sizeof(float));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.0f - 0.5f;   // "real" data
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int n = p.multiProcessorCount;
    run<false, 1>(n, in); run<false, 1>(n, in); run<true, 1>(n, in); run<false, 2>(n, in); run<false, 4>(n, in); run<true, 4>(n, in);
    return 0;
}

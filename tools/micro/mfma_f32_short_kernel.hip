// Micro-benchmark: what a SHORT kernel (tens of microseconds, as the small-batch top-k scan) gets
// from a dependent v_mfma_f32_16x16x4_f32 chain written with the compiler builtin (accumulator in
// AGPRs or VGPRs as hipcc chooses), alone and with VALU work interleaved: shader cycles per MFMA
// (s_memtime) and the clock the chip actually holds (s_memtime / s_memrealtime, 100 MHz).
// Build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_f32_short_kernel mfma_f32_short_kernel.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int VALU>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* stamps, int tiles, float a0, float b0) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = a0 + threadIdx.x + i; b[i] = b0 - threadIdx.x * i; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = -1e30f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f4 tot = {0, 0, 0, 0};
    for (int t = 0; t < tiles; ++t) {
        f4 c = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[(s + j) & 15], c, 0, 0, 0);
        }
        if (VALU) {   // a serial bubble insertion per value, like the top-k lists
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = tot[r];
                const bool ins = x > v[7];
                v[7] = ins ? x : v[7];
#pragma unroll
                for (int i = 7; i > 0; --i) {
                    const bool up = v[i] > v[i - 1];
                    const float hv = up ? v[i] : v[i - 1], lv = up ? v[i - 1] : v[i];
                    v[i - 1] = hv; v[i] = lv;
                }
            }
            if (VALU == 2) {
#pragma unroll
                for (int m = 0; m < 64; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
                }
            }
        }
        tot += c;
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(a[i]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = tot[0] + tot[1] + tot[2] + tot[3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int VALU>
void run(int tiles, int n_cu, const char* what) {
    float* out; unsigned long long* st;
    hipMalloc(&out, sizeof(float) * n_cu * 256);
    hipMalloc(&st, 16 * n_cu);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) k<VALU><<<n_cu, 256>>>(out, st, tiles, 1.0f, 2.0f);
    hipDeviceSynchronize();
    float sum = 0;
    const int reps = 30;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        k<VALU><<<n_cu, 256>>>(out, st, tiles, 1.0f, 2.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms;
    }
    unsigned long long h[4];
    hipMemcpy(h, st, 32, hipMemcpyDeviceToHost);
    printf("%-34s tiles=%4d  %7.1f us/launch  %6.1f cycles/MFMA  clock %.2f GHz\n", what, tiles,
           sum / reps * 1e3, (double)h[0] / (tiles * 64.0), (double)h[0] / (double)h[1] * 0.1);
    hipFree(out); hipFree(st);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n = p.multiProcessorCount;
    for (int tiles : {7, 28, 2000}) {
        run<0>(tiles, n, "MFMA chain only");
        run<1>(tiles, n, "MFMA chain + bubble (compiler)");
        run<2>(tiles, n, "MFMA chain + bubble (1 MFMA:2 VALU)");
    }
    return 0;
}

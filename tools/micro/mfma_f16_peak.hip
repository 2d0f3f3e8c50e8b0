// Micro-benchmark: the v_mfma_f32_16x16x32_f16 rate an MI355X SUSTAINS, operands in registers (no memory
// traffic, no LDS), on CONSTANT operands and on RANDOM operands - the ceiling DESIGN.md 3.5 argues the
// batch top-k's product loop against ("~1.25 PFLOP/s sustained on random data": the chip lowers its clock
// under the switching activity of random operands; MI355X_MICROARCH.md 'DVFS give-back').
// The loop is the batch top-k's shape: a wave holds NQ query fragments (B operands) and walks key
// fragments (A operands, NK distinct ones in registers, re-used round robin), one accumulator set per
// query group - every MFMA sees another (A, B) pair, so with random fragments the multiplier inputs toggle
// between instructions as they do in the real loop; with constant fragments nothing toggles.
// Build: hipcc -w --offload-arch=gfx950 -O3 -o mfma_f16_peak mfma_f16_peak.hip ; run: ./mfma_f16_peak
// (under rocprofv3 --kernel-trace --stats the kernels' durations are the same numbers: profiles/r06/)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// RANDOM = 1: fragments of pseudo-random fp16 values in [-1, 1) (every lane, every element another
// value); RANDOM = 0: every element 1/16.
template <int RANDOM, int NQ, int NK>
__global__ __launch_bounds__(256) void f16_loop(float* out, int iters) {
    h8 q[NQ], k[NK];
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            q[i][e] = RANDOM ? (_Float16)((float)(int)(hash32(t * 64 + i * 8 + e) >> 16) * (1.0f / 32768.f) - 1.0f) : (_Float16)0.0625f;
#pragma unroll
    for (int i = 0; i < NK; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e)
            k[i][e] = RANDOM ? (_Float16)((float)(int)(hash32(0x9e3779b9u + t * 64 + i * 8 + e) >> 16) * (1.0f / 32768.f) - 1.0f) : (_Float16)0.0625f;
    f4 acc[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < NK; ++j)
#pragma unroll
            for (int i = 0; i < NQ; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k[j], q[i], acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[t] = s;
}

template <int RANDOM, int NQ, int NK>
void run(const char* what, int wgs_per_cu, int n_cu, int iters, int reps) {
    const int grid = wgs_per_cu * n_cu;
    float* out;
    hipMalloc(&out, sizeof(float) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // pre-heat: the chip needs tens of ms of continuous work to hold its clock (tools/clock_ramp.py)
    for (int i = 0; i < 3; ++i) f16_loop<RANDOM, NQ, NK><<<grid, 256>>>(out, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        f16_loop<RANDOM, NQ, NK><<<grid, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 16 * 16 * 32 * (double)NQ * NK * iters * 4.0 * grid;      // 4 waves per workgroup
        printf("%-8s acc=%d keyfrags=%d wg/cu=%d  %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)\n", what, NQ, NK, wgs_per_cu, ms,
               flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
    }
    hipFree(out);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cu = p.multiProcessorCount;
    printf("%s CUs=%d clock=%d kHz  v_mfma_f32_16x16x32_f16, operands in registers, ~25 ms per launch\n", p.gcnArchName, cu, p.clockRate);
    // 4 query groups x 8 key fragments per wave (the batch top-k's wave: 4 groups of 16 queries), 1 and 2 workgroups per CU
    run<0, 4, 8>("constant", 1, cu, 96000, 4);
    run<1, 4, 8>("random", 1, cu, 96000, 4);
    run<0, 4, 8>("constant", 2, cu, 48000, 4);
    run<1, 4, 8>("random", 2, cu, 48000, 4);
    run<0, 8, 4>("constant", 2, cu, 48000, 3);
    run<1, 8, 4>("random", 2, cu, 48000, 3);
    return 0;
}

// Micro-benchmark around the batch top-k's product loop (range_amd/csrc/topk_gemm.h: pass A, the sampled
// group maxima): the library's own kernel against variants of its loop, to find what keeps it at 1.2 PFLOP/s
// when a register-only loop of the same MFMAs sustains 1.85-1.89 (tools/micro/mfma_f16_peak.hip).
//   NOLDS      the key fragments are read from LDS once, before the loop (ring, barriers, consume unchanged)
//   NOBAR      no phase barrier (results invalid)
//   NOCONSUME  the tile's values are not looked at
//   OCC1       one workgroup per CU (one wave per SIMD): is the second wave of a SIMD worth anything?
//   PIPE       the NEXT tile's fragments are read under the current tile's MFMAs (two fragment sets)
//   FASTMAX    the tile's maxima as two asm v_max3 per group instead of fmaxf() (which hipcc compiles to a
//              canonicalising v_max_f32 x, x, x per MFMA result first: 28 vector instructions per tile, not 8).
//              TIMING ONLY here: the copy pads no MFMA-result hazard for its asm; the library's kernel orders its
//              asm behind the younger tile's first accumulator (topk_gemm.h: consume).
// Round 6, one MI355X (profiles/r06/tg_loop.log): library pass A 96.4 us -> 89.8 with FASTMAX + a branch-free
// phase loop; what is left against the register-only loop's 1.85 PFLOP/s is the fragment reads (NOLDS) and the
// barrier, which cost as much with the reads issued a tile ahead (PIPE): not latency.
// Build: hipcc -w --offload-arch=gfx950 -O3 -I range_amd/csrc -o tg_loop tools/micro/tg_loop.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../range_amd/csrc/topk_gemm.h"

using namespace range_hip;

template <int OCC, int PIPE, int NOLDS, int NOBAR, int NOCONSUME, int FASTMAX = 0, int GQ = TG_GQ>
__global__ __launch_bounds__(TG_WAVES * 64, OCC) void tg_var(TopkGemmArgs a) {
    constexpr int TG_GQ = GQ;               // (shadows the library's: query groups of 16 per wave)
    constexpr int TG_QBLOCK = TG_WAVES * GQ * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int split = (int)blockIdx.x / a.n_qblocks, qb = (int)blockIdx.x - split * a.n_qblocks;
    const int t0 = (int)(((int64_t)split * a.n_blocks) / a.n_splits);
    const int t1 = (int)(((int64_t)(split + 1) * a.n_blocks) / a.n_splits);
    const int stride = a.tile_stride;
    const int b1 = (t1 - t0 + stride - 1) / stride;
    const int n_phase = (b1 + TG_KT - 1) / TG_KT;
    const int64_t q0 = (int64_t)qb * TG_QBLOCK + wave * (TG_GQ * 16);
    ts_u32x4 qf[TG_GQ][8];
    {
        const ts_u32x4* qsrc = reinterpret_cast<const ts_u32x4*>(a.qfrag);
        const int64_t n_groups = (a.B + 15) / 16;
#pragma unroll
        for (int gi = 0; gi < TG_GQ; ++gi) {
            int64_t grp = (q0 >> 4) + gi;
            grp = grp < n_groups ? grp : n_groups - 1;
#pragma unroll
            for (int c = 0; c < 8; ++c) qf[gi][c] = qsrc[(grp * 8 + c) * 64 + lane];
        }
    }
    float mx[TG_GQ][2];
#pragma unroll
    for (int gi = 0; gi < TG_GQ; ++gi) {
        mx[gi][0] = mx[gi][1] = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(qf[gi][c]));
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const char* kb = reinterpret_cast<const char*>(a.keys_f16);
    const int last = b1 - 1;
    auto issue = [&](int p) __attribute__((always_inline)) {
        const int piece0 = wave * TG_DMA_PER_WAVE, tsel = piece0 >> 3, poff = (piece0 & 7) * 1024;
        const int tile = min(p * TG_KT + tsel, last);
        const char* src = kb + ((int64_t)t0 + (int64_t)tile * stride) * TSB_TILE_BYTES + poff;
        const uint32_t dst = lds0 + ((p % TG_SLOTS) * TG_KT + tsel) * TSB_TILE_BYTES + poff;
        dma_group_begin(dst);
#pragma unroll
        for (int i4 = 0; i4 < TG_DMA_PER_WAVE; ++i4) dma_b128_q(src, (uint32_t)(lane << 4), i4);
    };
    issue(0);
    issue(1);
    auto read_frags = [&](const char* kt, ts_u32x4 (&kf)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 8; ++c) kf[c] = *reinterpret_cast<const ts_u32x4*>(kt + c * 1024);
    };
    auto mfmas = [&](ts_u32x4 (&kf)[8], f32x4 (&acc)[TG_GQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int gi = 0; gi < TG_GQ; ++gi) {
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 8; ++c)
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ts_f16x8, kf[c]),
                                                            __builtin_bit_cast(ts_f16x8, qf[gi][c]), c0, 0, 0, 0);
            acc[gi] = c0;
        }
    };
    auto consume = [&](f32x4 (&acc)[TG_GQ], int par) __attribute__((always_inline)) {
        if (NOCONSUME) {
#pragma unroll
            for (int gi = 0; gi < TG_GQ; ++gi) asm volatile("" :: "v"(acc[gi]));
            return;
        }
#pragma unroll
        for (int gi = 0; gi < TG_GQ; ++gi) {
            if (FASTMAX) {
                // two v_max3 per group, written as asm: fmaxf() makes hipcc canonicalise every MFMA result first
                // (v_max_f32 x, x, x: 4 + 3 + 1 vector instructions per group instead of 2)
                float t;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(acc[gi][0]), "v"(acc[gi][1]), "v"(acc[gi][2]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx[gi][par]) : "v"(t), "v"(acc[gi][3]), "v"(mx[gi][par]));
            } else {
                const float m4 = fmaxf(fmaxf(acc[gi][0], acc[gi][1]), fmaxf(acc[gi][2], acc[gi][3]));
                mx[gi][par] = fmaxf(mx[gi][par], m4);
            }
        }
    };
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8 * TG_GQ; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
    };
    f32x4 accA[TG_GQ], accB[TG_GQ];
    ts_u32x4 kf0[8], kf1[8];
    bool have_b = false;
    if (NOLDS) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(smem + lane * 16, kf0);
        read_frags(smem + TSB_TILE_BYTES + lane * 16, kf1);
#pragma unroll
        for (int c = 0; c < 8; ++c) { asm volatile("" : "+v"(kf0[c])); asm volatile("" : "+v"(kf1[c])); }
    }
    if (PIPE) {
        // prologue: phase 0 has to land before its first fragments can be read
        asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        read_frags(smem + lane * 16, kf0);
    }
    for (int p = 0; p < n_phase; ++p) {
        if (!PIPE) {
            if (!NOBAR) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            issue(p + 2);
            const char* slot = smem + (p % TG_SLOTS) * TG_KT * TSB_TILE_BYTES + lane * 16;
            if (!NOLDS) read_frags(slot, kf0);
            mfmas(kf0, accA);
            if (have_b) consume(accB, 1);
            interleave();
            if (!NOLDS) read_frags(slot + TSB_TILE_BYTES, kf1);
            mfmas(kf1, accB);
            consume(accA, 0);
            interleave();
            have_b = true;
        } else {
            // tile 2p's fragments are in kf0 (read one tile ago); tile 2p + 1's are read under tile 2p's MFMAs,
            // tile 2p + 2's (the NEXT phase: behind its barrier) under tile 2p + 1's
            const char* slot = smem + (p % TG_SLOTS) * TG_KT * TSB_TILE_BYTES + lane * 16;
            read_frags(slot + TSB_TILE_BYTES, kf1);
            mfmas(kf0, accA);
            if (have_b) consume(accB, 1);
            interleave();
            // phase p + 1 must have landed (and phase p - 1 must be read by all) before its first tile is read
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            issue(p + 2);
            const char* nslot = smem + ((p + 1) % TG_SLOTS) * TG_KT * TSB_TILE_BYTES + lane * 16;
            read_frags(nslot, kf0);
            mfmas(kf1, accB);
            consume(accA, 0);
            interleave();
            have_b = true;
        }
    }
    if (have_b) consume(accB, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int gi = 0; gi < TG_GQ; ++gi) {
        const int64_t q = q0 + gi * 16 + j;
        if (q < a.B) {
#pragma unroll
            for (int par = 0; par < 2; ++par) a.gmax[((int64_t)(split * 2 + par) * a.B + q) * 4 + g] = mx[gi][par];
        }
    }
}

static TopkGemmArgs ga;
static dim3 ggrid;

template <typename K>
static void timeit(const char* name, K kern, int lds, double flop) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(kern, ggrid, dim3(TG_WAVES * 64), lds, 0, ga);     // pre-heat
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, ggrid, dim3(TG_WAVES * 64), lds, 0, ga);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 50;
        best = ms < best ? ms : best;
        sum += ms;
    }
    printf("%-34s %7.1f us (best %7.1f)  %7.1f TFLOP/s  err=%s\n", name, sum / 5 * 1e3, best * 1e3, flop / (sum / 5) / 1e9,
           hipGetErrorString(hipGetLastError()));
}

int main(int argc, char** argv) {
    const int64_t B = 10000, N = 100000;
    const int stride = argc > 1 ? atoi(argv[1]) : 4;       // 4: pass A's sample; 1: a full pass
    const int n_blocks = (int)((N + 15) / 16);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    // random fp16 fragments (values in [-1, 1) x 2^13-ish scale does not matter for timing; use small randoms)
    const size_t key_words = (size_t)n_blocks * 8 * 64 * 4, q_words = (size_t)((B + 15) / 16) * 8 * 64 * 4;
    std::vector<uint32_t> hk(key_words), hq(q_words);
    uint32_t s = 12345u;
    auto rnd16 = [&]() { s = s * 1664525u + 1013904223u; const float f = (float)(s >> 8) * (1.0f / 8388608.f) - 1.0f; _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return (uint32_t)u; };
    for (auto& w : hk) w = rnd16() | (rnd16() << 16);
    for (auto& w : hq) w = rnd16() | (rnd16() << 16);
    void *dk, *dq;
    float* gmax;
    ga = TopkGemmArgs{};
    ga.n_qblocks = (int32_t)((B + TG_QBLOCK - 1) / TG_QBLOCK);
    ga.n_splits = std::max(4, std::min(std::min(TG_WG_PER_CU * p.multiProcessorCount / ga.n_qblocks, n_blocks / 8), 64));
    hipMalloc(&dk, key_words * 4); hipMalloc(&dq, q_words * 4); hipMalloc(&gmax, (size_t)ga.n_splits * 2 * B * 4 * 4);
    hipMemcpy(dk, hk.data(), key_words * 4, hipMemcpyHostToDevice);
    hipMemcpy(dq, hq.data(), q_words * 4, hipMemcpyHostToDevice);
    ga.keys_f16 = dk; ga.qfrag = dq; ga.B = B; ga.n_valid = N; ga.n_blocks = n_blocks; ga.gmax = gmax; ga.tile_stride = stride;
    ggrid = dim3((unsigned)(ga.n_qblocks * ga.n_splits));
    const double flop = 2.0 * 256 * B * (double)N / stride;
    printf("%s CUs=%d  B=%lld N=%lld stride=%d  grid=%u (qblocks %d x splits %d)\n", p.gcnArchName, p.multiProcessorCount,
           (long long)B, (long long)N, stride, ggrid.x, ga.n_qblocks, ga.n_splits);
    timeit("library topk_gemm_kernel<0>", topk_gemm_kernel<0>, TG_LDS_BYTES, flop);
    timeit("copy of its loop", tg_var<2, 0, 0, 0, 0>, TG_LDS_BYTES, flop);
    timeit("NOCONSUME", tg_var<2, 0, 0, 0, 1>, TG_LDS_BYTES, flop);
    timeit("NOBAR", tg_var<2, 0, 0, 1, 0>, TG_LDS_BYTES, flop);
    timeit("NOLDS (fragments read once)", tg_var<2, 0, 1, 0, 0>, TG_LDS_BYTES, flop);
    timeit("NOLDS + NOBAR + NOCONSUME", tg_var<2, 0, 1, 1, 1>, TG_LDS_BYTES, flop);
    timeit("OCC1 (one workgroup per CU)", tg_var<1, 0, 0, 0, 0>, 96 * 1024, flop);
    timeit("PIPE (next tile's fragments early)", tg_var<2, 1, 0, 0, 0>, TG_LDS_BYTES, flop);
    timeit("PIPE, OCC1", tg_var<1, 1, 0, 0, 0>, 96 * 1024, flop);
    timeit("FASTMAX (asm v_max3 pair)", tg_var<2, 0, 0, 0, 0, 1>, TG_LDS_BYTES, flop);
    timeit("FASTMAX + PIPE", tg_var<2, 1, 0, 0, 0, 1>, TG_LDS_BYTES, flop);
    timeit("FASTMAX + NOLDS", tg_var<2, 0, 1, 0, 0, 1>, TG_LDS_BYTES, flop);
    timeit("FASTMAX + NOBAR", tg_var<2, 0, 0, 1, 0, 1>, TG_LDS_BYTES, flop);
    // 8 query groups per wave (512 queries per workgroup, one workgroup per CU: half the fragment reads, DMA
    // pieces and barriers per MFMA; 256 registers of query fragments)
    {
        const TopkGemmArgs keep = ga;
        const dim3 keepg = ggrid;
        ga.n_qblocks = (int32_t)((B + 511) / 512);
        ga.n_splits = std::max(4, std::min(std::min(p.multiProcessorCount / ga.n_qblocks, n_blocks / 8), 64));
        ggrid = dim3((unsigned)(ga.n_qblocks * ga.n_splits));
        printf("GQ=8: grid=%u (qblocks %d x splits %d)\n", ggrid.x, ga.n_qblocks, ga.n_splits);
        timeit("GQ8 OCC1", tg_var<1, 0, 0, 0, 0, 0, 8>, TG_LDS_BYTES, flop);
        timeit("GQ8 OCC1 FASTMAX", tg_var<1, 0, 0, 0, 0, 1, 8>, TG_LDS_BYTES, flop);
        timeit("GQ8 OCC1 FASTMAX PIPE", tg_var<1, 1, 0, 0, 0, 1, 8>, TG_LDS_BYTES, flop);
        timeit("GQ8 OCC1 FASTMAX NOLDS", tg_var<1, 0, 1, 0, 0, 1, 8>, TG_LDS_BYTES, flop);
        ga = keep;
        ggrid = keepg;
    }
    return 0;
}

// Batch-scale top-k (range_topk_stream for batches beyond the one-launch regime: B > 256).
//
// The streaming scan of topk_stream.h is built for a handful of queries: 16 or 32 of them per pass
// over the keys, every value pushed through per-lane candidate lists.  At BASELINE's batch (10 000
// queries against range_db_large) that is 313 passes re-streaming a bank that sits in the Infinity
// Cache, bound by list maintenance and pass seams (3.9 ms, round 4) - while the arithmetic,
// 2 x 256 x 10^9 FLOP, is 0.2 ms of 16-bit MFMA.  Here the scan is GEMM-shaped and list-free:
//
//   pass A  topk_gemm_kernel<0>  approximate similarities S~ = K~ Q~^T (fp16 operands scaled by powers
//           of two, f32 accumulation: v_mfma_f32_16x16x32_f16) of a SAMPLE of the bank - every TG_SAMPLE-th
//           16-row tile - of which only the MAXIMUM per (query, row group) is kept: one v_max per
//           value.  Row groups = bank splits x 2 tile parities x 4 accumulator lane groups:
//           disjoint row sets, at least 16 of them.
//   theta   per query the 16th largest group maximum: at least 16 rows have S~ >= theta~ (a lower
//           bound from the sample: looser than the whole bank's by about log2(TG_SAMPLE) ranks'
//           worth of rows, at 1 / TG_SAMPLE of a pass)
//   pass B  topk_gemm_kernel<1>  the products of ALL rows; every row with S~ >= theta~ - 2 eps is
//           appended, with its S~, to a candidate list (one compare per tile and lane; the append is
//           a rare branch).  A lane owns the list of its (query, bank split, accumulator lane group)
//           outright - TG_CAP_L slots and a counter in a register - so an append is one plain
//           store: no atomic, no wait (with one list per query filled through returning atomics
//           the waves spent 54 % of their cycles waiting)
//   rerank  topk_gemm_rerank_kernel: v16 = the 16th largest S~ of the query's candidates (the 16
//           largest S~ of the whole bank are among them); the candidates with S~ >= v16 - 2 eps get
//           their FLOAT32 similarity by the fmaf chain every float32 kernel of this library computes
//           (topk_exact_dot) and are ranked by (value, lower row first): values and indices are
//           those of the float32 scan, bit for bit.
//
// Operands: the keys as fp16 in the MFMA fragment order of the streaming scan's bf16 copy
// (keyfrag_f16_kernel: built on the first call that needs it, 512 B per row), scaled by the power of two
// that puts the largest key norm in [2^13, 2^14); a query scaled by the power of two that puts its
// largest element there.  Both scalings are exact, keep every element that matters in fp16's normal
// range (an element that falls below 2^-14 after scaling is below 2^-27 of the operand's largest: its
// loss is 2^-23 relative), and make S~ a per-query multiple of the similarity - thresholds, candidates'
// values and eps of a query live in that query's scale.  fp16 instead of bf16 (same MFMA rate): eps is
// a quarter, and both the appends of pass B and the re-rank's row gathers scale with the width of
// the 2-eps band (10^4 x 10^5: bf16 operands 0.84 ms, fp16 0.77 ms).  The queries'
// fragments are written once per call (qfrag_f16_kernel, 5 us), not by every workgroup of both passes.
//
// Exactness: |S~ - S| <= eps = TG_EPS_REL |q| max|k| for every pair.  At least 16 rows have
// S~ >= v16, hence S >= v16 - eps: the 16th best exact value is >= v16 - eps and every member of the
// exact top 16 has S~ >= v16 - 2 eps (>= theta~ - 2 eps: it was appended).  A query whose lists
// overflow (a bank of near-duplicates; a sample that missed the query's neighbourhood) is recomputed
// by brute force (topk_gemm_brute_kernel): slower, never wrong.
//
// Decomposition: workgroup = 4 waves = 256 queries (wave w: 4 groups of 16 queries, their fp16
// fragments in 128 registers for the whole kernel) x one bank split; key tiles (16 rows = 8 KB in
// the fragment order keyfrag_kernel wrote) arrive by LDS-DMA in phases of 2 tiles through a 3-slot
// ring shared by the 4 waves, one barrier per phase; per tile and wave 4 x 8 MFMAs.  Two
// workgroups per CU (48 KB of LDS, <= 256 registers): one's MFMAs cover the other's barrier.
#pragma once
#include "topk_stream.h"

namespace range_hip {

constexpr int TG_GQ = 4;                     // query groups (of 16) per wave
#ifndef RANGE_TG_WAVES
#define RANGE_TG_WAVES 4                     // waves per workgroup (4: two workgroups per CU; 8: one)
#endif
constexpr int TG_WAVES = RANGE_TG_WAVES;
constexpr int TG_WG_PER_CU = 8 / TG_WAVES;
constexpr int TG_QBLOCK = TG_WAVES * TG_GQ * 16;    // queries per workgroup
constexpr int TG_DMA_PER_WAVE = 16 / TG_WAVES;      // 1 KB pieces of a phase's 2 x 8 KB each wave moves
constexpr int TG_KT = 2;                     // key tiles per phase
constexpr int TG_SLOTS = 3;
constexpr int TG_LDS_BYTES = TG_SLOTS * TG_KT * TSB_TILE_BYTES;   // 48 KB
constexpr int TG_SAMPLE = 5;                 // pass A looks at every 5th tile (round 6, behind the faster passes: 3: 710, 4: 698, 5: 689, 6: 688 us per 10 000-query call)
constexpr int TG_CAP = 1024;                 // candidates of a query the re-rank takes in (all lists together)
constexpr int TG_CAP_L = 32;                 // slots per (query, bank split, lane group) list
constexpr int TG_MAX_GROUPS = 512;            // row groups per query: splits (<= 64) x 2 tile parities x 4 lane groups
constexpr int TG_CAP_X = 256;                // candidates whose float32 similarity the re-rank evaluates
// eps / (|q| |k|): key rounding 2^-11 + query rounding 2^-11 + their product 2^-22 = 0.00097680, elements
// lost below fp16's normal range 2 x 2^-23, MFMA accumulation (256 terms, f32) 1.6e-5, the float32
// chain's own rounding 1.5e-5: 0.0010080
constexpr float TG_EPS_REL = 0.00105f;
typedef _Float16 ts_f16x8 __attribute__((ext_vector_type(8)));

// two floats -> packed fp16, round to nearest even: gfx950's v_cvt_pk_f16_f32 (bitwise the (_Float16) cast
// on 2^24 pairs incl. exact ties and subnormal results: tools/micro/cvt_pk_f16_rne.hip)
__device__ __forceinline__ uint32_t tg_cvt_pk_f16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the power of two that puts a positive float's value in [2^13, 2^14) (1 for zero / denormal input)
__device__ __forceinline__ float tg_scale_to_2p13(float mx) {
    const uint32_t E = (__float_as_uint(mx) >> 23) & 0xFFu;
    if (E == 0u) return 1.0f;
    const uint32_t f = 267u - E;                  // 2^(13 - (E - 127))
    return __uint_as_float((f > 254u ? 254u : f) << 23);
}

// queries (B x 256 f32) -> fp16 B-operand fragments of v_mfma_f32_16x16x32_f16, each query scaled by the power
// of two that puts its largest element in [2^13, 2^14) (qscale[q]; rows past B: copies of the last query):
// group grp of 16 queries, chunk c, lane (j, kg): Q[16 grp + j][32 c + 8 kg + 0..7].  One workgroup per group.
__global__ __launch_bounds__(256) void qfrag_f16_kernel(const float* __restrict__ ehat, int64_t B,
                                                        ts_u32x4* __restrict__ out, float* __restrict__ qscale) {
    __shared__ float sh_mx[16][17];
    const int t = threadIdx.x, j = t & 15, part = t >> 4;            // 16 parts of 16 elements per query
    const int64_t grp = blockIdx.x;
    const int64_t q = grp * 16 + j < B ? grp * 16 + j : B - 1;
    const f32x4* row = reinterpret_cast<const f32x4*>(ehat + q * KEY_DIM + 16 * part);
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 v = row[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    sh_mx[j][part] = m;
    __syncthreads();
    if (part == 0) {
        float mm = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) mm = fmaxf(mm, sh_mx[j][i]);
        const float sc = tg_scale_to_2p13(mm);
        sh_mx[j][16] = sc;
        if (grp * 16 + j < B) qscale[grp * 16 + j] = sc;
    }
    __syncthreads();
    for (int item = t; item < 8 * 64; item += 256) {                  // (chunk, lane) of this group
        const int c = item >> 6, ln = item & 63, jj = ln & 15, kg = ln >> 4;
        const int64_t qq = grp * 16 + jj < B ? grp * 16 + jj : B - 1;
        const float sc = sh_mx[jj][16];
        const f32x4* src = reinterpret_cast<const f32x4*>(ehat + qq * KEY_DIM + 32 * c + 8 * kg);
        const f32x4 v0 = src[0] * sc, v1 = src[1] * sc;
        ts_u32x4 o;
        o[0] = tg_cvt_pk_f16(v0.x, v0.y); o[1] = tg_cvt_pk_f16(v0.z, v0.w);
        o[2] = tg_cvt_pk_f16(v1.x, v1.y); o[3] = tg_cvt_pk_f16(v1.z, v1.w);
        out[(grp * 8 + c) * 64 + ln] = o;
    }
}

// keys (n_alloc rows x 256 f32) x scale -> fp16 A-operand fragments, the layout of keyfrag_kernel
__global__ __launch_bounds__(256) void keyfrag_f16_kernel(const float* __restrict__ keys, int64_t n_alloc,
                                                          int64_t n_tiles, float scale, ts_u32x4* __restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_tiles * 8 * 64) return;
    const int lane = (int)(id & 63), c = (int)((id >> 6) & 7);
    const int64_t t = id >> 9;
    const int64_t row = t * 16 + pi_row(lane & 15);
    ts_u32x4 o = {0u, 0u, 0u, 0u};
    if (row < n_alloc) {
        const f32x4* src = reinterpret_cast<const f32x4*>(keys + row * KEY_DIM + 32 * c + 8 * (lane >> 4));
        const f32x4 a = src[0] * scale, b = src[1] * scale;
        o[0] = tg_cvt_pk_f16(a.x, a.y); o[1] = tg_cvt_pk_f16(a.z, a.w);
        o[2] = tg_cvt_pk_f16(b.x, b.y); o[3] = tg_cvt_pk_f16(b.z, b.w);
    }
    out[id] = o;
}

struct TopkGemmArgs {
    const void* keys_f16;       // (n_tiles, 8 chunks, 64 lanes, 8) fp16 x key_scale (keyfrag_f16_kernel)
    const float* keys;          // (n_pad, 256) f32 (rerank)
    const float* ehat;          // (B, 256)
    const void* qfrag;          // (ceil(B/16), 8 chunks, 64 lanes, 8) fp16: the queries as B-operand fragments, scaled (qfrag_f16_kernel)
    int64_t B;
    int64_t n_valid;
    int32_t n_blocks;
    int32_t n_qblocks;          // ceil(B / TG_QBLOCK)
    int32_t n_splits;
    float* gmax;                // pass A out: (n_splits * 2, B, 4) group maxima of the approximate similarity
    const float* theta;         // pass B in: (B, 2) candidate threshold on the approximate value (2 eps below theta~), 2 eps
    uint32_t* cnt;              // (B, n_splits * 4) candidates each list was offered (may exceed TG_CAP_L: overflow)
    uint2* cand;                // (B, n_splits * 4, TG_CAP_L) candidates: (row, bits of S~)
    int32_t tile_stride;        // pass A: TG_SAMPLE (a sample of the tiles); pass B: 1
    uint32_t* ovf;              // (B) rerank -> brute force: 1 = a list or the query's total overflowed
    int32_t k;
    int64_t row_offset;
    float* oval;                // (B, k)
    int64_t* oidx;              // (B, k)
    int32_t* exact_count;       // queries recomputed by brute force (optional)
};

// Pass B's append, for the lanes whose value reaches their threshold: (row, bits of S~) as ONE 8-byte
// store to slot min(count, TG_CAP_L - 1) of the lane's own list, count + 1 - under an exec mask, skipped
// as a whole when no lane of the wave has a hit for this accumulator register.  (Measured, ~200
// candidates per query: pass B without hits 440 us; the rare branch and its per-group ballots +17; the
// append's arithmetic +73 and its stores +73 as first written - four unconditional masked blocks per
// group, two 4-byte stores per hit.)  The store is not counted by hipcc: every hand-counted vmcnt wait of the kernel only gets
// stricter by younger operations (the counter retires in issue order).
__device__ __forceinline__ void tg_append(float val, float th, uint32_t& count, uint32_t list_off, uint32_t row,
                                          const void* cand) {
    uint32_t t;
    unsigned long long sv;
    const unsigned long long pair = ((unsigned long long)__float_as_uint(val) << 32) | row;   // (row, bits of S~)
    asm volatile(
        "v_cmp_ge_f32 vcc, %[val], %[th]\n\t"
        "s_and_saveexec_b64 %[sv], vcc\n\t"
        "s_cbranch_execz .Ltg_none_%=\n\t"          // (of a group's four values usually one reaches the threshold)
        "v_min_u32 %[t], %[capm1], %[cnt]\n\t"
        "v_add_u32 %[cnt], 1, %[cnt]\n\t"
        "v_lshl_add_u32 %[t], %[t], 3, %[off]\n\t"
#ifndef RANGE_EXP_TG_NOSTORE     // (timing experiment: the append without its store)
        "global_store_dwordx2 %[t], %[pair], %[cand]\n\t"
#endif
        ".Ltg_none_%=:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [t] "=&v"(t), [sv] "=&s"(sv), [cnt] "+v"(count)
        : [val] "v"(val), [th] "v"(th), [off] "v"(list_off), [pair] "v"(pair), [cand] "s"(cand), [capm1] "n"(TG_CAP_L - 1)
        : "vcc", "memory");
}

template <int MODE>
__global__ __launch_bounds__(TG_WAVES * 64, 2) void topk_gemm_kernel(TopkGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    // consecutive workgroups take the query blocks of ONE split: its key tiles are shared in L2
    const int split = (int)blockIdx.x / a.n_qblocks, qb = (int)blockIdx.x - split * a.n_qblocks;
    // the split's tiles, or (pass A) every tile_stride-th of them: "tile" counts the visited ones
    const int t0 = (int)(((int64_t)split * a.n_blocks) / a.n_splits);
    const int t1 = (int)(((int64_t)(split + 1) * a.n_blocks) / a.n_splits);
    const int stride = a.tile_stride;
    const int b0 = 0, b1 = (t1 - t0 + stride - 1) / stride;
    const int n_phase = (b1 - b0 + TG_KT - 1) / TG_KT;
    const int64_t q0 = (int64_t)qb * TG_QBLOCK + wave * (TG_GQ * 16);

    // B operand: lane (n = query j, kg = g) holds Q[j][32 c + 8 g + 0..7] of chunk c as fp16 - converted and
    // scaled once per call by qfrag_f16_kernel (every workgroup of both passes converting its 256 queries
    // itself cost pass A 20 % of its time): 16 bytes per lane and chunk, coalesced
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
    float th[TG_GQ];            // pass B: this lane's query's threshold per group
    float mx[TG_GQ][2];         // pass A: running maxima per group and tile parity
    uint32_t nc[TG_GQ];         // pass B: candidates this lane's list of the group's query was offered
    bool qok[TG_GQ];
#pragma unroll
    for (int gi = 0; gi < TG_GQ; ++gi) {
        const int64_t q = q0 + gi * 16 + j;
        qok[gi] = q < a.B;
        // (a lane whose query does not exist never reaches its threshold)
        th[gi] = MODE == 1 ? (q < a.B ? a.theta[2 * q] : INFINITY) : 0.f;
        mx[gi][0] = mx[gi][1] = -INFINITY;
        nc[gi] = 0u;
    }
    // (ordinary loads: hipcc's waits for them end here, in front of the hand-counted ring)
#pragma unroll
    for (int gi = 0; gi < TG_GQ; ++gi) {
#pragma unroll
        for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(qf[gi][c]));
        asm volatile("" : "+v"(th[gi]));
    }

    // ring: phase p = tiles b0 + TG_KT p .. ; wave w moves half (w & 1) of tile (w >> 1): 4 x 1 KB
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const char* kb = reinterpret_cast<const char*>(a.keys_f16);
    const int last = b1 - 1;
    auto issue = [&](int p) __attribute__((always_inline)) {
        // (wave w moves pieces [w D, (w + 1) D) of the phase's 16, D = 16 / waves: a run inside one tile)
        const int piece0 = wave * TG_DMA_PER_WAVE, tsel = piece0 >> 3, poff = (piece0 & 7) * 1024;
        const int tile = min(b0 + p * TG_KT + tsel, last);      // (past the split's end: re-read its last tile)
        const char* src = kb + ((int64_t)t0 + (int64_t)tile * stride) * TSB_TILE_BYTES + poff;
        const uint32_t dst = lds0 + ((p % TG_SLOTS) * TG_KT + tsel) * TSB_TILE_BYTES + poff;
        dma_group_begin(dst);
#pragma unroll
        for (int i4 = 0; i4 < TG_DMA_PER_WAVE; ++i4) dma_b128_q(src, (uint32_t)(lane << 4), i4);
    };
    issue(0);
    issue(1);

    uint32_t prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = (uint32_t)pi_row(4 * g + r);
    const uint32_t n_valid32 = (uint32_t)a.n_valid;
    // byte offset of this lane's candidate list per group (the host keeps all lists below 4 GB)
    uint32_t list_off[TG_GQ];
#pragma unroll
    for (int gi = 0; gi < TG_GQ; ++gi)
        list_off[gi] = (uint32_t)(((((q0 + gi * 16 + j) * a.n_splits + split) * 4 + g) * TG_CAP_L) * sizeof(uint2));

    // Only the bank's last tile can hold pad rows (zero keys: similarity 0, which must neither raise a
    // group maximum nor become a candidate): it is kept out of the loop and handled behind it.
    const bool tail_partial = stride == 1 ? (t1 == a.n_blocks && (a.n_valid % BLK) != 0)
                                          : ((t0 + (b1 - 1) * stride) == a.n_blocks - 1 && (a.n_valid % BLK) != 0);
    const int b1_main = b1 - (tail_partial ? 1 : 0);

    // a tile's 4 x 8 MFMAs (its fragments from LDS or, for the tail tile, straight from memory)
    auto mfma_tile = [&](const char* kt, f32x4 (&acc)[TG_GQ]) __attribute__((always_inline)) {
        ts_u32x4 kf[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) kf[c] = *reinterpret_cast<const ts_u32x4*>(kt + c * 1024);
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
    // what a tile's 4 x 4 approximate values per lane are used for.  The common path is branch-free
    // vector work (it sits in the same basic block as the NEXT tile's MFMAs, interleaved with them):
    // pass A one v_max3 pair per group, pass B one compare per group; the append is ONE rare,
    // wave-uniform branch per tile behind them.
    // `dep`: an accumulator of the tile whose MFMAs were issued AFTER this tile's (its first query group's).  The
    // maxima below are inline asm: hipcc pads no hazard for them (MFMA D -> VALU read: up to 12 wait states),
    // so every statement that reads this tile's accumulators names `dep` as an operand too - it cannot be
    // placed before the 8 MFMAs that produce `dep` have been issued, 128 cycles behind this tile's last MFMA.
    // fresh = true (the odd last tile, the tail tile: no younger tile): explicit wait states instead.
    auto consume = [&](f32x4 (&acc)[TG_GQ], f32x4& dep, int tile, int par, bool masked, bool fresh) __attribute__((always_inline)) {
        const uint32_t row0 = (uint32_t)(t0 + tile * stride) * BLK;
        if (fresh) {
            static_assert(TG_GQ == 4, "the wait statement names the four accumulators");
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        }
        if (masked) {
#pragma unroll
            for (int gi = 0; gi < TG_GQ; ++gi)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[gi][r] = row0 + prow[r] < n_valid32 ? acc[gi][r] : -INFINITY;
        }
        // (the maxima as asm v_max3 / v_max: through fmaxf() hipcc first canonicalises every MFMA result - v_max_f32
        // x, x, x - which made it 8 vector instructions per group where 2 (pass A) or 3 (pass B) do: with 28 instead
        // of 8 per tile the "one MFMA, two vector instructions" pattern below ran out of MFMA gaps half way through
        // the tile; tools/micro/tg_loop.hip: pass A 93 -> 86 us, a full pass 361 -> 331.  NaN operands - a NaN query -
        // are ignored by the instructions as by fmaxf: such a lane never raises a maximum and never has a hit)
        bool hit = false;
        float m4[TG_GQ];
#pragma unroll
        for (int gi = 0; gi < TG_GQ; ++gi) {
            float t;
            asm("v_max3_f32 %0, %2, %3, %4" : "=v"(t), "+v"(dep) : "v"(acc[gi][0]), "v"(acc[gi][1]), "v"(acc[gi][2]));
            if (MODE == 0) {
                asm("v_max3_f32 %0, %2, %3, %4" : "=v"(mx[gi][par]), "+v"(dep) : "v"(t), "v"(acc[gi][3]), "v"(mx[gi][par]));
            } else {
                asm("v_max_f32_e32 %0, %2, %3" : "=v"(m4[gi]), "+v"(dep) : "v"(t), "v"(acc[gi][3]));
                hit = hit || m4[gi] >= th[gi];
            }
        }
        if (MODE == 1 && __builtin_amdgcn_ballot_w64(hit) != 0ull) {
#pragma unroll
            for (int gi = 0; gi < TG_GQ; ++gi) {
                if (__builtin_amdgcn_ballot_w64(m4[gi] >= th[gi]) == 0ull) continue;      // (wave-uniform)
#ifndef RANGE_EXP_TG_NOAPPEND    // (timing experiment: the rare branch entered, nothing appended)
#pragma unroll
                for (int r = 0; r < 4; ++r) tg_append(acc[gi][r], th[gi], nc[gi], list_off[gi], row0 + prow[r], a.cand);
#else
                asm volatile("s_nop 0" ::: "memory");
#endif
            }
        }
    };
    // one MFMA, two vector instructions, ... : the order the 32 MFMAs of a tile and the vector work of
    // the previous tile's values are issued in
    auto interleave = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8 * TG_GQ; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
    };

    // two tiles per phase: tile 2p -> accA (parity 0), tile 2p + 1 -> accB (parity 1); a tile's values are
    // consumed one tile later, beside the next tile's MFMAs
    f32x4 accA[TG_GQ], accB[TG_GQ];
    bool have_b = false;
    const int n_phase_main = (b1_main + TG_KT - 1) / TG_KT;
    // phase p has landed when at most the 4 operations of phase p + 1 are outstanding (own share; the barrier
    // makes it everybody's); every wave is then also done reading phase p - 1, whose slot phase p + 2 takes
    auto phase_begin = [&](int p) __attribute__((always_inline)) {
        if (TG_DMA_PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        issue(p + 2);
    };
    // The phases that hold two main tiles run in ONE basic block each, the first of them peeled (no earlier
    // tile to consume): with the per-tile conditions inside the loop (is there a second tile? was there an
    // earlier one?) the compiler kept flags in scalar registers and branched between the tiles, and the vector
    // work of tile t - 1 could not be placed beside the MFMAs of tile t across those branches.
    const int n_full = b1_main / TG_KT;
    int p = 0;
    if (n_full > 0) {
        phase_begin(0);
        const char* slot = smem + lane * 16;
        mfma_tile(slot, accA);
        mfma_tile(slot + TSB_TILE_BYTES, accB);
        consume(accA, accB[0], 0, 0, false, false);
        interleave();
        for (p = 1; p < n_full; ++p) {
            phase_begin(p);
            slot = smem + (p % TG_SLOTS) * TG_KT * TSB_TILE_BYTES + lane * 16;
            mfma_tile(slot, accA);
            consume(accB, accA[0], 2 * p - 1, 1, false, false);
            interleave();
            mfma_tile(slot + TSB_TILE_BYTES, accB);
            consume(accA, accB[0], 2 * p, 0, false, false);
            interleave();
        }
        have_b = true;
    }
    // what is left: a phase with ONE main tile (an odd count), and / or a phase that holds only the tail tile
    for (; p < n_phase; ++p) {
        phase_begin(p);
        if (p >= n_phase_main) continue;                     // (a last phase that held only the tail tile)
        const char* slot = smem + (p % TG_SLOTS) * TG_KT * TSB_TILE_BYTES + lane * 16;
        mfma_tile(slot, accA);
        if (have_b) consume(accB, accA[0], 2 * p - 1, 1, false, false);
        consume(accA, accA[0], 2 * p, 0, false, true);
        have_b = false;
    }
    if (have_b) consume(accB, accB[0], 2 * n_phase_main - 1, 1, false, true);
    if (tail_partial) {
        const int tile = b1 - 1;
        mfma_tile(kb + ((int64_t)t0 + (int64_t)tile * stride) * TSB_TILE_BYTES + lane * 16, accA);
        consume(accA, accA[0], tile, tile & 1, true, true);
    }
    // the clamped prefetches of the last phases are still in flight into this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int gi = 0; gi < TG_GQ; ++gi) {
        const int64_t q = q0 + gi * 16 + j;
        if (q < a.B) {
            if (MODE == 0) {
#pragma unroll
                for (int par = 0; par < 2; ++par)
                    a.gmax[((int64_t)(split * 2 + par) * a.B + q) * 4 + g] = mx[gi][par];
            } else {
                a.cnt[(q * a.n_splits + split) * 4 + g] = nc[gi];
            }
        }
    }
}

// theta[q] = {(the 16th largest of the query's n_parts x 4 group maxima) - 2 eps, 2 eps}, eps = eps_rel |q| max|k|
// in the query's scale of S~ (eps_ks = eps_rel x max|k| x the keys' scale; the norm is taken of the SCALED
// query: neither factor can under- or overflow, whatever the magnitudes of bank and queries)
// (with 1e-4 of slack for the rounding of the norm and of the subtraction);
// -inf when fewer than 16 groups saw a row (every row is then a candidate), and for a query whose float32
// products themselves leave the normal range (|q| max|k| below 2^-90 or above 2^100, the zero query
// included: the error bound of the float32 chain does not hold there) - its lists overflow and it is
// answered by brute force.  One wave per query.
__global__ __launch_bounds__(256) void topk_gemm_threshold_kernel(const float* __restrict__ gmax, int n_parts, int64_t B,
                                                                  const float* __restrict__ ehat, float eps_ks,
                                                                  float log2_kmax, const float* __restrict__ qscale,
                                                                  float* __restrict__ theta) {
    // the 16th largest by counting: every lane ranks its own values against all of the query's (LDS
    // broadcasts) - a third of the time of pushing them through the scan's candidate lists
    __shared__ float sh_g[4][TG_MAX_GROUPS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t q = (int64_t)blockIdx.x * 4 + w;
    if (q >= B) return;                               // (whole waves; no workgroup barrier below)
    const int total = n_parts * 4;
    // (every global load of the wave up front: one round trip, not three)
    const f32x4 v = *reinterpret_cast<const f32x4*>(ehat + q * KEY_DIM + 4 * lane);
    const float qs = qscale[q];
    for (int e = lane; e < total; e += 64) sh_g[w][e] = gmax[((int64_t)(e >> 2) * B + q) * 4 + (e & 3)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float t16 = -INFINITY;
    for (int e = lane; e < total; e += 64) {
        const float mine = sh_g[w][e];
        int rank = 0;
        for (int f = 0; f < total; ++f) {
            const float o = sh_g[w][f];
            rank += (o > mine || (o == mine && f < e)) ? 1 : 0;
        }
        if (rank == MAX_TOPK - 1) t16 = mine;          // (exactly one entry has this rank; -inf: fewer than 16 groups saw a row)
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) t16 = fmaxf(t16, __shfl_xor(t16, off));
    const f32x4 vs = v * qs;                            // (exact: a power of two; largest element in [2^13, 2^14))
    float sq = vs.x * vs.x + vs.y * vs.y + vs.z * vs.z + vs.w * vs.w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) sq += __shfl_xor(sq, off);
    if (lane == 0) {
        // the margin carries 1e-4 of slack for the rounding of this subtraction and of the norm
        const float eps2 = 2.0f * eps_ks * sqrtf(sq) * 1.0001f;
        const float l2 = 0.5f * log2f(sq) - log2f(qs) + log2_kmax;       // log2(|q| max|k|); -inf for a zero query or bank
        const bool normal = l2 >= -90.0f && l2 <= 100.0f;
        theta[2 * q] = (normal && t16 > -INFINITY) ? t16 - eps2 : -INFINITY;
        theta[2 * q + 1] = normal ? eps2 : 0.0f;
    }
}

// topk_exact_dot (topk_stream.h) - the same products in the same order, the same float - with the key
// row's loads issued in two bursts of 32 instead of sixteen dependent groups of four: a thread that
// walks a row of its own pays the memory latency twice, not sixteen times
__device__ __forceinline__ float topk_exact_dot_burst(const float* __restrict__ kr, const float* sh_q) {
    float acc = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 kc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) kc[i] = *reinterpret_cast<const f32x4*>(kr + 128 * h + 4 * i);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                    acc = __builtin_fmaf(kc[4 * s8 + gg][c], sh_q[128 * h + 16 * s8 + 4 * gg + c], acc);
            }
        }
    }
    return acc;
}

// The candidates of one query -> its top k: one workgroup per query.
//   1. the lists' lengths (n_splits x 4 of them), a prefix sum, the entries compacted into LDS as
//      64-bit keys (ordered bits of S~, ~row);
//   2. v16 = the 16th largest S~ (rank by counting);
//   3. the entries with S~ >= v16 - 2 eps (at most TG_CAP_X): float32 similarity by the fmaf chain
//      (one thread per entry), ranked by counting; ranks 0..k-1 are the result.
// A query with a list that overflowed, more than TG_CAP candidates or more than TG_CAP_X in step 3 is
// handed to topk_gemm_brute_kernel.
constexpr int TG_RR_LDS = KEY_DIM * 4 + TG_CAP * 8 + 256 * 4 + TG_CAP_X * 8 + 32;
__global__ __launch_bounds__(256, 2) void topk_gemm_rerank_kernel(TopkGemmArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[TG_RR_LDS];
    float* sh_q = reinterpret_cast<float*>(lds);
    unsigned long long* ka = reinterpret_cast<unsigned long long*>(lds + KEY_DIM * 4);     // [TG_CAP] approximate keys
    uint32_t* sh_c = reinterpret_cast<uint32_t*>(ka + TG_CAP);                            // [256] list lengths
    unsigned long long* kx = reinterpret_cast<unsigned long long*>(sh_c + 256);           // [TG_CAP_X] exact keys
    int* sh_i = reinterpret_cast<int*>(kx + TG_CAP_X);          // [0] overflow, [1] entries of step 3, [2..3] v16 key
    const int64_t q = blockIdx.x;
    const int t = threadIdx.x;
    const int R = a.n_splits * 4;                                    // lists of this query (<= 256)
    const uint32_t c = t < R ? a.cnt[q * R + t] : 0u;
    sh_c[t] = c;
    sh_q[t] = a.ehat[q * KEY_DIM + t];
    if (t < 4) sh_i[t] = 0;
    __syncthreads();
    if (c > (uint32_t)TG_CAP_L) sh_i[0] = 1;
    uint32_t off = 0, total = 0;
    for (int u = 0; u < R; ++u) {
        const uint32_t cu = sh_c[u];
        off += u < t ? cu : 0u;
        total += cu;
    }
    __syncthreads();
    if (sh_i[0] != 0 || total > (uint32_t)TG_CAP) {
        if (t == 0) a.ovf[q] = 1u;
        return;
    }
    // (all threads walk the lists' slots side by side: the loads are independent and a list's entries
    // sit next to each other; one thread per list walking its entries paid a memory latency per entry)
    sh_c[t] = t < R ? ((c << 16) | off) : 0u;                        // (both <= TG_CAP: 11 bits each)
    __syncthreads();
    for (int e = t; e < R * TG_CAP_L; e += 256) {
        const int r = e / TG_CAP_L, i = e - r * TG_CAP_L;
        const uint32_t co = sh_c[r];
        if ((uint32_t)i < (co >> 16)) {
            const uint2 v = a.cand[(q * R + r) * TG_CAP_L + i];
            ka[(co & 0xFFFFu) + i] = topk_key(__uint_as_float(v.y), v.x);
        }
    }
    __syncthreads();
    const int n = (int)total;
    // v16: the entry of rank min(16, n) - 1 among the approximate keys
    const int want = (n < MAX_TOPK ? n : MAX_TOPK) - 1;
    for (int e = t; e < n; e += 256) {
        const unsigned long long key = ka[e];
        int r = 0;
        for (int u = 0; u < n; ++u) r += ka[u] > key ? 1 : 0;
        if (r == want) *reinterpret_cast<unsigned long long*>(sh_i + 2) = key;
    }
    __syncthreads();
    const float eps2 = a.theta[2 * q + 1];
    const float lo = n > 0 ? topk_key_val(*reinterpret_cast<unsigned long long*>(sh_i + 2)) - eps2 : -INFINITY;
    for (int e = t; e < n; e += 256) {
        const unsigned long long key = ka[e];
        if (topk_key_val(key) >= lo) {
            const int pos = atomicAdd(sh_i + 1, 1);
            if (pos < TG_CAP_X) kx[pos] = key;
        }
    }
    __syncthreads();
    const int nx = sh_i[1];
    if (nx > TG_CAP_X) {
        if (t == 0) a.ovf[q] = 1u;
        return;
    }
    if (t == 0) a.ovf[q] = 0u;
    unsigned long long key = 0ull;
    if (t < nx) {
        const uint32_t row = topk_key_row(kx[t]);
        key = topk_key(topk_exact_dot_burst(a.keys + (int64_t)row * KEY_DIM, sh_q), row);
    }
    __syncthreads();
    kx[t] = key;
    __syncthreads();
    if (t < nx) {
        int r = 0;
        for (int u = 0; u < nx; ++u) r += kx[u] > key ? 1 : 0;      // (keys are unique: the row is part of the key)
        if (r < a.k) {
            a.oval[q * a.k + r] = topk_key_val(key);
            a.oidx[q * a.k + r] = (int64_t)topk_key_row(key) + a.row_offset;
        }
    }
    if (t >= nx && t < a.k) {                                       // (a bank with fewer than k rows)
        a.oval[q * a.k + t] = -INFINITY;
        a.oidx[q * a.k + t] = -1;
    }
}

// more candidates than the lists hold (a bank of near-duplicates, or fewer than 16 row groups): every
// row's float32 similarity (topk_stream.h: topk_brute_force).  A few workgroups walk the queries'
// flags; normally none is set.
__global__ __launch_bounds__(256) void topk_gemm_brute_kernel(TopkGemmArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[KEY_DIM * 4 + 16 * MAX_TOPK * 8 + MAX_TOPK * 8];
    float* sh_q = reinterpret_cast<float*>(lds);
    unsigned long long* sh = reinterpret_cast<unsigned long long*>(lds + KEY_DIM * 4);
    unsigned long long* res = sh + 16 * MAX_TOPK;
    __shared__ uint32_t sh_set[256];
    const int t = threadIdx.x;
    // workgroup b owns the queries b, b + grid, b + 2 grid, ... (neighbours that overflow together are
    // spread over the workgroups) and reads their flags 256 at a time (one load per thread: a dependent
    // load per query made the empty walk 12 us)
    const int64_t G = gridDim.x;
    const int64_t mine = (a.B - blockIdx.x + G - 1) / G;             // queries of this workgroup
    for (int64_t base = 0; base < mine; base += 256) {
        const uint32_t set = (base + t < mine) ? a.ovf[blockIdx.x + G * (base + t)] : 0u;
        if (!__syncthreads_or((int)set)) continue;                   // (uniform over the workgroup)
        sh_set[t] = set;
        __syncthreads();
        for (int i = 0; i < 256 && base + i < mine; ++i) {
            if (sh_set[i] == 0u) continue;
            const int64_t q = blockIdx.x + G * (base + i);
            __syncthreads();
            sh_q[t] = a.ehat[q * KEY_DIM + t];
            __syncthreads();
            topk_brute_force(a.keys, a.n_valid, sh_q, sh, res);
            if (t < a.k) {
                const unsigned long long m = res[t];
                a.oval[q * a.k + t] = m ? topk_key_val(m) : -INFINITY;
                a.oidx[q * a.k + t] = m ? (int64_t)topk_key_row(m) + a.row_offset : (int64_t)-1;
            }
            if (t == 0 && a.exact_count) atomicAdd(a.exact_count, 1);
        }
        __syncthreads();
    }
}

}  // namespace range_hip

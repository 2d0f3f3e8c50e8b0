// Small-batch top-k: the HBM-streaming form of the keys scan (range_topk_stream).
//
// For a handful of queries the scan is bound by streaming the N x 1 KB key rows, not by MFMA, and
// the 64-queries-per-workgroup decomposition of pass 1 wastes the machine.  Here the grid is
// PERSISTENT (one workgroup per CU, launched once): every WAVE streams its own 16-row key tiles
// through a wave-private LDS ring of two tiles filled by LDS-DMA - no workgroup barrier in the
// loop - against G groups of 16 queries held in registers (G = 1 or 2 groups share one pass
// over the keys), and several passes run back to back in the one launch with the ring kept full
// across the pass boundary.  (This file holds two forms of the scan: the float32 one described
// here, and the bf16-key prefilter built on it - further down - which is what range_topk_stream
// runs by default.)
//
// What makes the stream the only thing that takes time:
//  * the K fragments of a tile are read into registers at once (16 ds_read_b128), so the ring
//    slot is free - and its refill is on its way - BEFORE the tile's MFMAs and list work: two
//    tiles (32 KB) per wave are in flight nearly all the time;
//  * the MFMAs are compiler builtins here (no 256-accumulator register pressure as in pass 2), so
//    hipcc pads their hazards and interleaves the list maintenance of the PREVIOUS tile's values
//    into the MFMA shadow of the current one (software pipeline of depth one);
//  * the per-lane candidate lists are SHORT (L = 4 values per lane and group instead of 16): a
//    wave sees only N / 16 / n_waves tiles (6 for range_db_large), i.e. ~24 values per lane, so
//    16-deep lists never saturate and every value costs a full insertion.  Exactness is kept by
//    bookkeeping: every lane tracks the largest value it ever let go (dmax); the final merge
//    (topk_merge_kernel) compares the largest dmax of a query with the k-th value it found, and
//    only if some dropped value could have belonged to the top-k - 5 of a query's best 16 rows in
//    the few rows ONE lane sees, or exact ties - recomputes that query by brute force in the same
//    kernel (bit-identical dot products: an MFMA chain is an fmaf chain in a fixed order).
//
// Dot products are the same single dependent MFMA chain, in the same k order, as in the scan
// kernels of attend_kernels.h: every kernel that forms a similarity gets the same float.
#pragma once
#include "attend_kernels.h"

namespace range_hip {

struct TopkStreamArgs {
    const float* keys;          // (n_pad,256)
    const float* ehat;          // (B,256)
    unsigned long long* cand;   // (n_groups, 16 queries, n_waves, L) keys, sorted descending, 0 = empty
    float* dmax;                // (n_groups, 16 queries, n_waves) largest value dropped on the way
    int64_t B;
    int64_t n_valid;
    int32_t n_blocks;
    int32_t n_groups;           // ceil(B / 16)
    unsigned long long* stamps; // RANGE_EXP_TS_STAMPS builds: 8 s_memrealtime stamps per wave
    const void* keys_bf16;      // prefilter form: (n_tiles, 8 chunks, 64 lanes, 8) bf16, see keyfrag_kernel
};

#ifdef RANGE_EXP_TS_STAMPS   // tuning only: where a wave's time goes (100 MHz real-time counter)
#define RANGE_TS_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(size_t)w_id * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RANGE_TS_NOW() __builtin_amdgcn_s_memrealtime()
#define RANGE_TS_ADD(acc, t0) (acc) += __builtin_amdgcn_s_memrealtime() - (t0)
#else
#define RANGE_TS_STAMP(i) do { } while (0)
#define RANGE_TS_NOW() 0ull
#define RANGE_TS_ADD(acc, t0) do { } while (0)
#endif

#ifndef RANGE_TOPKS_VALU_PER_MFMA
#define RANGE_TOPKS_VALU_PER_MFMA 4
#endif
constexpr int TOPKS_SG = 4;         // groups whose lists a wave carries through consecutive passes
constexpr int TOPKS_LDS_BYTES = 8 * BLK * KEY_DIM * 4;          // 128 KB of key tiles per workgroup

// Sorted (descending) list of the L best (value, row) a lane has met, plus the largest value it
// has let go.  push() is branch-free: the new value replaces the last entry if it is larger and
// bubbles up; whichever of the two does not stay goes into dmax.  Rows arrive in increasing
// order and the comparisons are strict, so among equal values the lower row stays ahead.
template <int L>
struct ShortList {
    float v[L];
    uint32_t row[L];
    float dmax;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < L; ++i) { v[i] = -INFINITY; row[i] = 0xFFFFFFFFu; }
        dmax = -INFINITY;
    }
    __device__ __forceinline__ void push(float x, uint32_t r) {
        const float last = v[L - 1];
        dmax = fmaxf(dmax, fminf(x, last));
        const bool ins = x > last;
        v[L - 1] = ins ? x : last;
        row[L - 1] = ins ? r : row[L - 1];
#pragma unroll
        for (int i = L - 1; i > 0; --i) {
            const bool up = v[i] > v[i - 1];
            const float hv = up ? v[i] : v[i - 1], lv = up ? v[i - 1] : v[i];
            const uint32_t hr = up ? row[i] : row[i - 1], lr = up ? row[i - 1] : row[i];
            v[i - 1] = hv; v[i] = lv; row[i - 1] = hr; row[i] = lr;
        }
    }
};

// The 4 lanes (j, g = 0..3) of a query each hold a sorted list of L keys: afterwards every one of
// them holds the sorted L best of the union, and `drop` the largest key that did not make it.
// Two rounds (lane ^ 16, lane ^ 32) of a bitonic merge: c[i] = max(a[i], b[L-1-i]) are the L
// largest of two descending lists and form a bitonic sequence, which log2(L) stages of
// compare-exchanges sort.  No serial extraction loop: ~300 instructions for L = 8.
template <int L>
__device__ __forceinline__ void merge4_short(unsigned long long (&k)[L], unsigned long long& drop) {
    static_assert(L == 4 || L == 8 || L == 16, "power-of-two list");
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        unsigned long long p[L];
#pragma unroll
        for (int i = 0; i < L; ++i) p[i] = shfl_xor_u64(k[L - 1 - i], off);
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const bool up = k[i] > p[i];
            const unsigned long long lo = up ? p[i] : k[i];
            k[i] = up ? k[i] : p[i];
            drop = lo > drop ? lo : drop;
        }
#pragma unroll
        for (int d = L / 2; d >= 1; d >>= 1) {
#pragma unroll
            for (int i = 0; i < L; ++i) {
                if ((i & d) == 0) {
                    const unsigned long long a = k[i], b = k[i + d];
                    k[i] = a > b ? a : b;
                    k[i + d] = a > b ? b : a;
                }
            }
        }
    }
}

// NW waves per workgroup, each with a ring of DEPTH tiles (NW * DEPTH * 16 KB of LDS = 128 KB).
// Per-wave stamps (RANGE_EXP_TS_STAMPS) show that with 4 waves x 2 tiles a wave never waits for a
// tile once the first has landed: over a pass it spends 7.9 us in arithmetic, 2.4 us issuing
// LDS-DMA and 0 us waiting.  8 waves x 1 tile (two waves per SIMD, K fragments to registers and the
// slot refilled before the arithmetic) was measured for the 1-group kernel: 21.4 us instead of 21.6
// at 16 queries, but 37.3 / 70 us instead of 36.2 / 63.6 at 2 / 4 passes (2048 waves leave 3.05
// tiles per wave and pass: the last-tile imbalance grows) and a merge over twice the lists - so
// every instantiation is 4 x 2.
template <int G, int L, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64, NW / 4) void topk_stream_kernel(TopkStreamArgs a) {
    static_assert(TOPKS_SG % G == 0, "groups per pass must divide the supergroup");
    static_assert(NW * DEPTH == 8 && (DEPTH == 1 || DEPTH == 2), "128 KB of key tiles per workgroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t KT_BYTES = BLK * KEY_DIM * 4;
    constexpr int PPS = TOPKS_SG / G;                          // passes per supergroup
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem) + wave * DEPTH * KT_BYTES;
    const char* my = smem + wave * DEPTH * KT_BYTES;

    const int n_waves = gridDim.x * NW;
    // (wave-major ids: the waves that get one tile more than the rest - the first n_blocks mod
    // n_waves ids - are then spread one per CU instead of filling whole workgroups)
    const int w_id = wave * gridDim.x + blockIdx.x;
    // this wave's tiles: w_id, w_id + n_waves, ... once per pass (T per pass, the last one of a
    // pass possibly past the bank: fetched as the bank's last tile, never consumed).  (Dealing a
    // workgroup's tiles to its waves through a counter in LDS was measured: the spread of the
    // waves' finishing times is between XCDs and CUs, not inside a workgroup, and did not shrink.)
    const int T = (a.n_blocks + n_waves - 1) / n_waves;        // tiles per wave and pass
    const int n_pass = (a.n_groups + G - 1) / G;               // passes over the keys, all supergroups
    const int n_sg = (a.n_groups + TOPKS_SG - 1) / TOPKS_SG;
    const int total = n_pass * T;                              // this wave's tile sequence
    const int last = a.n_blocks - 1;
    RANGE_TS_STAMP(0);
    // one tile = 16 rows = 16 DMA instructions (4 groups of 4 rows, swizzled source: chunk c of
    // row R lands at chunk position c ^ R, which makes the ds_read_b128 below conflict-free).
    // Sequence positions past the end fetch the bank's last tile again - never consumed - so
    // that every wait below is a constant.
    auto issue_seq = [&](int k) __attribute__((always_inline)) {
        const int i = k < total ? k % T : T - 1;
        const int tile = w_id + i * n_waves;
        const float* src = a.keys + (int64_t)(tile < last ? tile : last) * BLK * KEY_DIM;
        const uint32_t dst = lds0 + (k % DEPTH) * KT_BYTES;
#pragma unroll
        for (int gr = 0; gr < 4; ++gr) {
            dma_group_begin(dst + gr * 4096);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                // (non-temporal: a key tile is read by exactly one wave per pass - measured 1 us
                // per 16-query launch and 3 us per four passes faster than the default policy)
                dma_b128_q_nt(src + gr * 4 * KEY_DIM, (uint32_t)((lane ^ (4 * gr + i4)) << 4), i4);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue_seq(d);
    RANGE_TS_STAMP(1);

    KAddr kaddr;
    kaddr.init(lane);
    uint32_t prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = (uint32_t)pi_row(4 * g + r);
    const uint32_t n_valid32 = (uint32_t)a.n_valid;

    int k = 0;                                                 // position in the tile sequence
    unsigned long long ts_wait = 0, ts_issue = 0, ts_comp = 0; // (stamp builds: time in each part of the loop)
    for (int sg = 0; sg < n_sg; ++sg) {
        // the lists of a supergroup's 4 query groups live in registers through its passes and
        // are merged once, at its end: no merge work at a pass boundary
        ShortList<L> lists[TOPKS_SG];
#pragma unroll
        for (int gi = 0; gi < TOPKS_SG; ++gi) lists[gi].init();
#pragma unroll
        for (int ps = 0; ps < PPS; ++ps) {
            const int grp0 = (sg * PPS + ps) * G;                         // first group of this pass
            if (grp0 < a.n_groups) {
                // query fragments of this pass's groups: lane (j, g) holds Q[j][16 s + 4 g .. +3]
                f32x4 qf[G][16];
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    const int grp = min(grp0 + gi, a.n_groups - 1);
                    const int64_t q = (int64_t)grp * 16 + j;
                    const f32x4* rowp =
                        reinterpret_cast<const f32x4*>(a.ehat + (q < a.B ? q : a.B - 1) * KEY_DIM);
#pragma unroll
                    for (int s = 0; s < 16; ++s) qf[gi][s] = rowp[4 * s + g];
                }
                // (ordinary loads that hipcc counts: "using" them here puts its wait for them in
                // front of the tile loop - at their first use inside it, it would be a vmcnt(0)
                // that drains the hand-counted LDS-DMA ring every iteration)
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(qf[gi][s]));
                }
                // values of the previous tile, pushed while the current tile's MFMAs run (the
                // first round pushes -inf: a no-op)
                f32x4 prev[G];
#pragma unroll
                for (int gi = 0; gi < G; ++gi) prev[gi] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                uint32_t prev_row0 = 0;
                auto push_prev = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t row = prev_row0 + prow[r];
                        // (pad rows exist in the bank's last tile only; a compare is cheaper than a branch)
                        const float x = row < n_valid32 ? prev[gi][r] : -INFINITY;
                        lists[ps * G + gi].push(x, row);
                    }
                };

                for (int i = 0; i < T; ++i, ++k) {
                    const int tile = w_id + i * n_waves;
                    const unsigned long long ts0 = RANGE_TS_NOW();
#ifndef RANGE_EXP_TS_NODMA
                    // tile k has landed when at most the 16 operations of each younger tile are outstanding
                    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    RANGE_TS_ADD(ts_wait, ts0);
                    if (k == 0) RANGE_TS_STAMP(2);            // first tile landed
                    const char* kt = my + (k % DEPTH) * KT_BYTES;
                    f32x4 kf[16];
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        kf[s] = *reinterpret_cast<const f32x4*>(kt + kaddr.b[s & 3] + 256 * (s >> 2));
                    // the slot is free once these reads have returned: refill it before the arithmetic
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(kf[s]));
                    const unsigned long long ts1 = RANGE_TS_NOW();
#ifndef RANGE_EXP_TS_NODMA   // timing experiment only (results invalid): no stream, compute only
                    issue_seq(k + DEPTH);
#endif
                    RANGE_TS_ADD(ts_issue, ts1);
                    if (tile >= a.n_blocks) continue;      // (the ragged last round of a pass)
                    const unsigned long long ts2 = RANGE_TS_NOW();
                    f32x4 acc[G];
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        f32x4 c = {0.f, 0.f, 0.f, 0.f};
#ifdef RANGE_EXP_TS_NOMFMA   // timing experiment only (results invalid)
                        c = kf[gi] + kf[gi + 4] + qf[gi][3];
#else
#pragma unroll
                        for (int s = 0; s < 16; ++s) {
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].x, qf[gi][s].x, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].y, qf[gi][s].y, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].z, qf[gi][s].z, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].w, qf[gi][s].w, c, 0, 0, 0);
                        }
#endif
                        acc[gi] = c;
                        // list maintenance of the PREVIOUS tile's values of this group:
                        // independent of the chain above, placed into its shadow
#ifdef RANGE_EXP_TS_NOPUSH   // timing experiment only (results invalid)
                        lists[ps * G + gi].v[0] += prev[gi][0] + prev[gi][1] + prev[gi][2] + prev[gi][3];
#else
                        push_prev(gi);
#endif
#if RANGE_TOPKS_VALU_PER_MFMA > 0
#pragma unroll
                        for (int m = 0; m < 64; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);    // 1 MFMA
                            __builtin_amdgcn_sched_group_barrier(0x2, RANGE_TOPKS_VALU_PER_MFMA, 0);
                        }
#endif
                    }
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) prev[gi] = acc[gi];
                    prev_row0 = (uint32_t)tile * BLK;
#ifdef RANGE_EXP_TS_STAMPS
                    asm volatile("" :: "v"(prev[0]));      // (the tile's results are in registers here)
#endif
                    RANGE_TS_ADD(ts_comp, ts2);
                }
#pragma unroll
                for (int gi = 0; gi < G; ++gi) push_prev(gi);
            }
        }
        RANGE_TS_STAMP(4);   // all tiles of the supergroup consumed
        // end of the supergroup: per group, the 4 lanes of a query merge their lists (bitonic,
        // shuffles only) and lane group 0 writes the wave's L best and its dmax.  No barrier,
        // no LDS; the ring keeps streaming the next supergroup's first tiles meanwhile.
#pragma unroll
        for (int gi = 0; gi < TOPKS_SG; ++gi) {
            const int grp = sg * TOPKS_SG + gi;
            if (grp < a.n_groups) {
                unsigned long long kk[L];
#pragma unroll
                for (int i = 0; i < L; ++i)
                    kk[i] = lists[gi].row[i] != 0xFFFFFFFFu ? topk_key(lists[gi].v[i], lists[gi].row[i]) : 0ull;
                unsigned long long drop = 0ull;
                merge4_short<L>(kk, drop);
                float dm = lists[gi].dmax;
                dm = fmaxf(dm, __shfl_xor(dm, 16));
                dm = fmaxf(dm, __shfl_xor(dm, 32));
                if (drop != 0ull) dm = fmaxf(dm, topk_key_val(drop));
                if (g == 0) {
                    const int64_t at = ((int64_t)grp * 16 + j) * n_waves + w_id;
                    unsigned long long* o = a.cand + at * L;
#pragma unroll
                    for (int i = 0; i < L; i += 2)
                        *reinterpret_cast<ulonglong2*>(o + i) = make_ulonglong2(kk[i], kk[i + 1]);
                    a.dmax[at] = dm;
                }
            }
        }
    }
    RANGE_TS_STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped prefetches past the end
    RANGE_TS_STAMP(6);
#ifdef RANGE_EXP_TS_STAMPS
    if (lane == 0 && a.stamps) {   // sums over the loop: waiting for tiles / issuing LDS-DMA / arithmetic
        a.stamps[(size_t)w_id * 8 + 3] = ts_wait;
        a.stamps[(size_t)w_id * 8 + 7] = (ts_issue << 32) | (ts_comp & 0xFFFFFFFFull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Prefilter on bf16 keys (the default of range_topk_stream; results identical to the float32 scan).
//
// The float32 scan above is not waiting for HBM at 16 queries: a wave spends its time in the 64
// float32 MFMAs per tile and per query group.  Here the scan reads a bf16 copy of the keys (half
// the bytes) and forms APPROXIMATE similarities with 16 bf16 MFMAs per tile and group - the query
// to 16 significant bits (two bf16 planes), the key rounded to bf16 - so
//     |approx - exact| <= (2^-9 + 2^-17) |q| |k|   (+ 3e-5 of accumulation)  =: eps.
// A group's query operand is 64 registers; one or two groups share a pass over the keys.
// The lists, their dmax bookkeeping and the candidate layout are those of the float32 scan, on
// approximate values.  topk_merge_kernel then takes every candidate within 2 eps of the k-th best
// approximate value, recomputes ITS similarity with the float32 fmaf chain (= the MFMA chain of the
// float32 kernels, bit for bit) and ranks those: a row of the true top k cannot be missing (its
// approximate value is within eps of its exact one, and the k-th best approximate value within eps
// of the k-th best exact one) unless a list dropped it - which the dmax check, widened by the same
// 2 eps, detects and answers with the brute-force path as before.

typedef __bf16 ts_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t ts_u32x4 __attribute__((ext_vector_type(4)));
constexpr int TSB_TILE_BYTES = 8 * 1024;          // 16 rows x 256 bf16 in fragment order
constexpr int TSB_DEPTH = 4;                      // ring slots per wave (32 KB in flight per wave, as above)
// eps / (|q| |k|): key rounding 2^-9 + query planes 2^-17 = 0.0019608, bf16 MFMA accumulation
// (512 terms) 3.1e-5, the float32 chain's own rounding (256 terms) 1.5e-5: 0.0020066 in the worst case
constexpr float TSB_EPS_REL = 0.0021f;

__device__ __forceinline__ uint32_t ts_cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// keys (n_alloc rows x 256 f32) -> bf16 A-operand fragments of v_mfma_f32_16x16x32_bf16: tile t,
// chunk c (32 dims), lane (m, kg): the 8 values K[16 t + pi_row(m)][32 c + 8 kg + 0..7], RNE.
// (pi_row: the row order of the float32 tiles, so that the list code sees the same rows.)
__global__ __launch_bounds__(256) void keyfrag_kernel(const float* __restrict__ keys, int64_t n_alloc,
                                                      int64_t n_tiles, ts_u32x4* __restrict__ out) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_tiles * 8 * 64) return;
    const int lane = (int)(id & 63), c = (int)((id >> 6) & 7);
    const int64_t t = id >> 9;
    const int64_t row = t * 16 + pi_row(lane & 15);
    ts_u32x4 o = {0u, 0u, 0u, 0u};
    if (row < n_alloc) {
        const f32x4* src = reinterpret_cast<const f32x4*>(keys + row * KEY_DIM + 32 * c + 8 * (lane >> 4));
        const f32x4 a = src[0], b = src[1];
        o[0] = ts_cvt_pk_bf16(a.x, a.y); o[1] = ts_cvt_pk_bf16(a.z, a.w);
        o[2] = ts_cvt_pk_bf16(b.x, b.y); o[3] = ts_cvt_pk_bf16(b.z, b.w);
    }
    out[id] = o;
}

template <int G, int L>
__global__ __launch_bounds__(256, 1) void topk_stream_bf16_kernel(TopkStreamArgs a) {
    static_assert(TOPKS_SG % G == 0, "groups per pass must divide the supergroup");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PPS = TOPKS_SG / G;
    constexpr int DEPTH = TSB_DEPTH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem) + wave * DEPTH * TSB_TILE_BYTES;
    const char* my = smem + wave * DEPTH * TSB_TILE_BYTES + lane * 16;
    const int n_waves = gridDim.x * 4;
    const int w_id = wave * gridDim.x + blockIdx.x;
    const int T = (a.n_blocks + n_waves - 1) / n_waves;
    const int n_pass = (a.n_groups + G - 1) / G;
    const int n_sg = (a.n_groups + TOPKS_SG - 1) / TOPKS_SG;
    const int total = n_pass * T;
    const int last = a.n_blocks - 1;
    const char* kb = reinterpret_cast<const char*>(a.keys_bf16);
    // one tile = 8 KB, contiguous in fragment order: 8 LDS-DMA operations (two groups of four)
    auto issue_seq = [&](int k) __attribute__((always_inline)) {
        const int i = k < total ? k % T : T - 1;
        const int tile = w_id + i * n_waves;
        const char* src = kb + (int64_t)(tile < last ? tile : last) * TSB_TILE_BYTES;
        const uint32_t dst = lds0 + (k % DEPTH) * TSB_TILE_BYTES;
#pragma unroll
        for (int gr = 0; gr < 2; ++gr) {
            dma_group_begin(dst + gr * 4096);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) dma_b128_q_nt(src + gr * 4096, (uint32_t)(lane << 4), i4);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue_seq(d);

    uint32_t prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = (uint32_t)pi_row(4 * g + r);
    const uint32_t n_valid32 = (uint32_t)a.n_valid;

    int k = 0;
    for (int sg = 0; sg < n_sg; ++sg) {
        ShortList<L> lists[TOPKS_SG];
#pragma unroll
        for (int gi = 0; gi < TOPKS_SG; ++gi) lists[gi].init();
#pragma unroll
        for (int ps = 0; ps < PPS; ++ps) {
            const int grp0 = (sg * PPS + ps) * G;
            if (grp0 < a.n_groups) {
                // B operand: lane (n = query j, kg = g) holds Q[j][32 c + 8 g + 0..7], chunk c, as
                // two bf16 planes (q = q_h + q_m to 2^-17)
                ts_u32x4 qh[G][8], qm[G][8];
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    const int grp = min(grp0 + gi, a.n_groups - 1);
                    const int64_t q = (int64_t)grp * 16 + j;
                    const f32x4* rowp =
                        reinterpret_cast<const f32x4*>(a.ehat + (q < a.B ? q : a.B - 1) * KEY_DIM);
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x4 v0 = rowp[8 * c + 2 * g], v1 = rowp[8 * c + 2 * g + 1];
                        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float x = v[2 * e], y = v[2 * e + 1];
                            const uint32_t h = ts_cvt_pk_bf16(x, y);
                            const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xFFFF0000u);
                            qh[gi][c][e] = h; qm[gi][c][e] = ts_cvt_pk_bf16(rx, ry);
                        }
                    }
                }
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(qh[gi][c]), "+v"(qm[gi][c]));
                }
                f32x4 prev[G];
#pragma unroll
                for (int gi = 0; gi < G; ++gi) prev[gi] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                uint32_t prev_row0 = 0;
                auto push_prev = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint32_t row = prev_row0 + prow[r];
                        const float x = row < n_valid32 ? prev[gi][r] : -INFINITY;
                        lists[ps * G + gi].push(x, row);
                    }
                };
                for (int i = 0; i < T; ++i, ++k) {
                    const int tile = w_id + i * n_waves;
                    // tile k has landed when at most the 8 operations of each younger tile are outstanding
                    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                    const char* kt = my + (k % DEPTH) * TSB_TILE_BYTES;
                    ts_u32x4 kf[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) kf[c] = *reinterpret_cast<const ts_u32x4*>(kt + c * 1024);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(kf[c]));
                    issue_seq(k + DEPTH);
                    if (tile >= a.n_blocks) continue;
                    f32x4 acc[G];
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        f32x4 c0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int c = 0; c < 8; ++c) {      // small terms first
                            const ts_bf16x8 kk = __builtin_bit_cast(ts_bf16x8, kf[c]);
                            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kk, __builtin_bit_cast(ts_bf16x8, qm[gi][c]), c0, 0, 0, 0);
                        }
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            const ts_bf16x8 kk = __builtin_bit_cast(ts_bf16x8, kf[c]);
                            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kk, __builtin_bit_cast(ts_bf16x8, qh[gi][c]), c0, 0, 0, 0);
                        }
                        acc[gi] = c0;
                        push_prev(gi);     // the previous tile's values, in the shadow of this chain
                    }
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) prev[gi] = acc[gi];
                    prev_row0 = (uint32_t)tile * BLK;
                }
#pragma unroll
                for (int gi = 0; gi < G; ++gi) push_prev(gi);
            }
        }
#pragma unroll
        for (int gi = 0; gi < TOPKS_SG; ++gi) {
            const int grp = sg * TOPKS_SG + gi;
            if (grp < a.n_groups) {
                unsigned long long kk[L];
#pragma unroll
                for (int i = 0; i < L; ++i)
                    kk[i] = lists[gi].row[i] != 0xFFFFFFFFu ? topk_key(lists[gi].v[i], lists[gi].row[i]) : 0ull;
                unsigned long long drop = 0ull;
                merge4_short<L>(kk, drop);
                float dm = lists[gi].dmax;
                dm = fmaxf(dm, __shfl_xor(dm, 16));
                dm = fmaxf(dm, __shfl_xor(dm, 32));
                if (drop != 0ull) dm = fmaxf(dm, topk_key_val(drop));
                if (g == 0) {
                    const int64_t at = ((int64_t)grp * 16 + j) * n_waves + w_id;
                    unsigned long long* o = a.cand + at * L;
#pragma unroll
                    for (int i = 0; i < L; i += 2)
                        *reinterpret_cast<ulonglong2*>(o + i) = make_ulonglong2(kk[i], kk[i + 1]);
                    a.dmax[at] = dm;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// One workgroup per query.  Thread p owns the sorted list (L keys) of stream wave p (<= 1024).
//  1. a lower bound T of the query's 16th best value: inside each wave of this kernel, the 16th
//     largest list head (rank by counting over the wave's 64 heads); T = the largest of the waves'
//     bounds.  Entries below T cannot be in the top 16.
//  2. the survivors (>= T; typically a few hundred of the 8192 entries) are compacted into LDS
//     and ranked by counting; ranks 0..k-1 are the result, already in order (keys are unique).
//  3. exactness of the short lists: if the largest value any lane or wave let go reaches the
//     k-th value found (or the survivors overflow their buffer, or `force_exact`), the query is
//     recomputed by brute force over all rows, each thread walking its rows with the SAME fmaf
//     chain as the MFMA (k order: for s, for component, for lane group) and full 16-deep lists.
//     exact_count (optional) counts the queries that took that path.
//  Prefilter form (eps_rel > 0: the candidates carry APPROXIMATE values from bf16 keys, within
//  eps = eps_rel |q| kmax of the float32 similarity): the threshold of step 1 is lowered by 2 eps,
//  step 2 ranks the survivors by approximate value to find the k-th best approximate value v_k,
//  step 3 widens the check to dall >= v_k - 2 eps, and then every survivor >= v_k - 2 eps gets its
//  float32 similarity by the fmaf chain of the brute-force path (one thread per survivor, 1 KB of
//  key row each) and the survivors are ranked again by those: the result is that of the float32 scan.
constexpr int TOPKM_CAP = 2048;    // survivor buffer (keys)

// the similarity every float32 kernel computes: acc = fmaf(K[16 s + 4 g + c], Q[16 s + 4 g + c], acc)
// in the order s = 0..15, c = 0..3, g = 0..3 of the MFMA chain
__device__ __forceinline__ float topk_exact_dot(const float* __restrict__ kr, const float* sh_q) {
    float acc = 0.f;
    for (int s = 0; s < 16; ++s) {
        f32x4 kc[4];
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) kc[gg] = *reinterpret_cast<const f32x4*>(kr + 16 * s + 4 * gg);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) acc = __builtin_fmaf(kc[gg][c], sh_q[16 * s + 4 * gg + c], acc);
        }
    }
    return acc;
}
__device__ __forceinline__ uint32_t topk_ordered_bits(float v) { return (uint32_t)(topk_key(v, 0u) >> 32); }

// The brute-force path of topk_merge_kernel: every row's float32 similarity, full 16-deep lists,
// wave merges through `sh` (16 x MAX_TOPK keys), result in res[0..MAX_TOPK).  Not inlined: its
// 16-deep lists would cost the common path of the kernel (128 registers at 1024 threads) spills.
__device__ __attribute__((noinline)) void topk_brute_force(const float* __restrict__ keys, int64_t n_valid,
                                                           const float* sh_q, unsigned long long* sh,
                                                           unsigned long long* res) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
    KeyList X;
    X.init();
    for (int64_t row = threadIdx.x; row < n_valid; row += blockDim.x)
        X.push(topk_key(topk_exact_dot(keys + row * KEY_DIM, sh_q), (uint32_t)row));
    merge_wave(X);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < MAX_TOPK; ++i) sh[wave * MAX_TOPK + i] = X.k[i];
    }
    __syncthreads();
    if (wave == 0) {
        KeyList M;
        M.init();
        if (lane < n_wv) {
#pragma unroll
            for (int i = 0; i < MAX_TOPK; ++i) M.k[i] = sh[lane * MAX_TOPK + i];
        }
        merge_wave(M);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < MAX_TOPK; ++i) res[i] = M.k[i];
        }
    }
    __syncthreads();
}

template <int L>
__global__ __launch_bounds__(1024) void topk_merge_kernel(const unsigned long long* cand, const float* dmax,
                                                          int n_parts, int64_t B, int k, int64_t row_offset,
                                                          const float* keys, const float* ehat, int64_t n_valid,
                                                          int force_exact, int* exact_count,
                                                          float eps_rel, float kmax,
                                                          float* oval, int64_t* oidx) {
    __shared__ unsigned long long surv[TOPKM_CAP];
    __shared__ unsigned long long sh[16 * MAX_TOPK];
    __shared__ unsigned long long res[MAX_TOPK];
    __shared__ uint32_t sh_top[64];
    __shared__ float sh_d[16];
    __shared__ float sh_n2[16];
    __shared__ unsigned long long surv2[TOPKM_CAP];      // prefilter form: the survivors' float32 keys
    __shared__ float sh_q[KEY_DIM];
    __shared__ int sh_cnt, sh_flag;
    const int n_wv = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = blockIdx.x;
    const int64_t group = q >> 4;
    const int p = threadIdx.x;
    unsigned long long kk[L];
    float dm = -INFINITY;
#pragma unroll
    for (int i = 0; i < L; ++i) kk[i] = 0ull;
    if (p < n_parts) {
        const int64_t at = (group * 16 + (q & 15)) * n_parts + p;   // contiguous over the threads
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(cand + at * L);
#pragma unroll
        for (int i = 0; i < L; i += 2) { const ulonglong2 t = src[i / 2]; kk[i] = t.x; kk[i + 1] = t.y; }
        dm = dmax[at];
    }
    if (threadIdx.x == 0) { sh_cnt = 0; sh_flag = 0; }
    if (threadIdx.x < MAX_TOPK) res[threadIdx.x] = 0ull;
    // prefilter form: the query (for the float32 similarities of the survivors) and its norm
    // (its partial sums of squares travel with step 1's barrier)
    float eps2 = 0.f;
    if (eps_rel > 0.f) {
        float sq = 0.f;
        for (int e = threadIdx.x; e < KEY_DIM; e += blockDim.x) { const float v = ehat[q * KEY_DIM + e]; sh_q[e] = v; sq += v * v; }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) sq += __shfl_xor(sq, off);
        if (lane == 0) sh_n2[wave] = sq;
    }
    // ---- 1. lower bound T of the 16th best value (ordered value bits only): every wave hands in
    //      its K largest list heads (n_wv * K >= 32 values, each the head of a different list, so
    //      sixteen candidates are >= the 16th largest of them); with 1024 lists that is close to
    //      the 20th-25th largest head overall and a few dozen entries survive it
    const uint32_t head = (uint32_t)(kk[0] >> 32);       // 0 = empty list
    const int K = max(2, (32 + n_wv - 1) / n_wv);        // n_wv * K in [32, 47]
    {
        uint32_t h = head;
        for (int r = 0; r < K; ++r) {
            uint32_t m = h;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_xor(m, off); m = o > m ? o : m; }
            if (lane == 0) sh_top[wave * K + r] = m;
            const unsigned long long holders = __ballot(h == m && m != 0u);
            if (holders != 0ull && lane == __ffsll((long long)holders) - 1) h = 0u;   // one holder leaves
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) dm = fmaxf(dm, __shfl_xor(dm, off));
    if (lane == 0) sh_d[wave] = dm;
    __syncthreads();
    uint32_t T = 0u;
    {
        const int nv = n_wv * K;                         // <= 47
        const uint32_t v = lane < nv ? sh_top[lane] : 0u;
        int rank = 0;                                    // unique ranks: ties by lane
        for (int i = 0; i < nv; ++i) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)v, i);
            rank += (o > v || (o == v && i < lane)) ? 1 : 0;
        }
        const unsigned long long at15 = __ballot(lane < nv && rank == 15);
        if (at15 != 0ull) T = (uint32_t)__builtin_amdgcn_readlane((int)v, __ffsll((long long)at15) - 1);
    }
    float dall = -INFINITY;
    for (int w = 0; w < n_wv; ++w) dall = fmaxf(dall, sh_d[w]);
    if (eps_rel > 0.f) {
        float n2 = 0.f;
        for (int w = 0; w < n_wv; ++w) n2 += sh_n2[w];
        // (a bound, not a result: 1 % over the norm covers its rounding)
        eps2 = 2.f * eps_rel * 1.01f * sqrtf(n2) * kmax;
    }
    if (eps2 > 0.f && T != 0u) T = topk_ordered_bits(topk_key_val((unsigned long long)T << 32) - eps2);
    // ---- 2. survivors
    // (one LDS atomic per wave and list position: the lanes of a wave take consecutive places)
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const bool keep = kk[i] != 0ull && (uint32_t)(kk[i] >> 32) >= T;
        const unsigned long long m = __ballot(keep);
        if (m != 0ull) {                                       // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&sh_cnt, __popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            const int at = base + __popcll(m & ((1ull << lane) - 1ull));
            if (keep && at < TOPKM_CAP) surv[at] = kk[i];
        }
    }
    __syncthreads();
    const int S = sh_cnt;
    if (S <= TOPKM_CAP) {
        for (int t = threadIdx.x; t < S; t += blockDim.x) {
            const unsigned long long key = surv[t];
            int r = 0;
            for (int u = 0; u < S; ++u) r += surv[u] > key ? 1 : 0;
            if (r < MAX_TOPK) res[r] = key;
        }
    }
    __syncthreads();
    // ---- 3. exactness
    if (threadIdx.x == 0) {
        const unsigned long long kth = res[k - 1];
        // (nothing can have been dropped while fewer than k rows exist)
        const bool unsafe = force_exact || S > TOPKM_CAP || (kth != 0ull && dall >= topk_key_val(kth) - eps2) ||
                            (kth == 0ull && dall > -INFINITY);
        sh_flag = unsafe ? 1 : 0;
        if (unsafe && exact_count) atomicAdd(exact_count, 1);
    }
    __syncthreads();
    if (!sh_flag && eps2 > 0.f) {
        // ---- 3b. prefilter form: float32 similarities of the survivors within 2 eps of the k-th
        //      best approximate value, ranked again (keys stay unique: the row is part of the key)
        const unsigned long long kth = res[k - 1];
        const float vmin = kth != 0ull ? topk_key_val(kth) - eps2 : -INFINITY;
        __syncthreads();                                     // (every thread has read res)
        if (threadIdx.x < MAX_TOPK) res[threadIdx.x] = 0ull;
        // 16 lanes per survivor: lane s loads chunk s of the key row (one round trip for the whole
        // row), then the chain runs chunk by chunk in the kernels' order, its value handed from lane
        // to lane (every lane computes on its own chunk each step; only lane `step`'s result counts)
        {
            const int sub = lane & 15;
            f32x4 qc[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) qc[gg] = *reinterpret_cast<const f32x4*>(sh_q + 16 * sub + 4 * gg);
            for (int t0 = 0; t0 < S; t0 += blockDim.x >> 4) {
                const int t = t0 + (threadIdx.x >> 4);
                const unsigned long long key = t < S ? surv[t] : 0ull;
                const bool live = key != 0ull && topk_key_val(key) >= vmin;
                const uint32_t row = live ? topk_key_row(key) : 0u;
                const float* kr = keys + (int64_t)row * KEY_DIM + 16 * sub;
                f32x4 kc[4];
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) kc[gg] = *reinterpret_cast<const f32x4*>(kr + 4 * gg);
                float acc = 0.f;
                for (int step = 0; step < 16; ++step) {
                    float v = acc;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) v = __builtin_fmaf(kc[gg][c], qc[gg][c], v);
                    }
                    acc = __shfl(v, (lane & 48) | step);
                }
                if (t < S && sub == 0) surv2[t] = live ? topk_key(acc, row) : 0ull;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < S; t += blockDim.x) {
            const unsigned long long key = surv2[t];
            if (key == 0ull) continue;
            int r = 0;
            for (int u = 0; u < S; ++u) r += surv2[u] > key ? 1 : 0;
            if (r < MAX_TOPK) res[r] = key;
        }
        __syncthreads();
    }
    if (sh_flag) {
        for (int e = threadIdx.x; e < KEY_DIM; e += blockDim.x) sh_q[e] = ehat[q * KEY_DIM + e];
        __syncthreads();
        topk_brute_force(keys, n_valid, sh_q, sh, res);
    }
    if (threadIdx.x < k) {
        const unsigned long long mm = res[threadIdx.x];
        oval[q * k + threadIdx.x] = mm ? topk_key_val(mm) : -INFINITY;
        oidx[q * k + threadIdx.x] = mm ? (int64_t)topk_key_row(mm) + row_offset : (int64_t)-1;
    }
}

}  // namespace range_hip

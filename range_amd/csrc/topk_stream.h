// Small-batch top-k: the HBM-streaming form of the keys scan (range_topk_stream).
//
// For a handful of queries the scan is bound by streaming the N x 1 KB key rows, not by MFMA, and
// the 64-queries-per-workgroup decomposition of pass 1 wastes the machine.  Here the grid is
// PERSISTENT (one workgroup per CU, launched once): every WAVE streams its own 16-row key tiles
// through a wave-private LDS ring of two tiles filled by LDS-DMA - no workgroup barrier in the
// loop - against G groups of 16 queries held in registers (G = 1, 2 or 4 groups share one pass
// over the keys), and several passes run back to back in the one launch with the ring kept full
// across the pass boundary.
//
// What makes the stream the only thing that takes time:
//  * the K fragments of a tile are read into registers at once (16 ds_read_b128), so the ring
//    slot is free - and its refill is on its way - BEFORE the tile's MFMAs and list work: two
//    tiles (32 KB) per wave are in flight nearly all the time;
//  * the MFMAs are compiler builtins here (no 256-accumulator register pressure as in pass 2), so
//    hipcc pads their hazards and interleaves the list maintenance of the PREVIOUS tile's values
//    into the MFMA shadow of the current one (software pipeline of depth one);
//  * the per-lane candidate lists are SHORT (L = 8 values per lane and group instead of 16): a
//    wave sees only N / 16 / n_waves tiles (6 for range_db_large), i.e. ~24 values per lane, so
//    16-deep lists never saturate and every value costs a full insertion.  Exactness is kept by
//    bookkeeping: every lane tracks the largest value it ever let go (dmax); the final merge
//    (topk_merge_kernel) compares the largest dmax of a query with the k-th value it found, and
//    only if some dropped value could have belonged to the top-k - 9 of a query's best 16 rows in
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
    unsigned long long* cand;   // (n_groups, n_wg, 16 queries, 16) keys, sorted descending, 0 = empty
    float* dmax;                // (n_groups, n_wg, 16 queries) largest value a lane list dropped
    int64_t B;
    int64_t n_valid;
    int32_t n_blocks;
    int32_t n_groups;           // ceil(B / 16)
};

#ifndef RANGE_TOPKS_VALU_PER_MFMA
#define RANGE_TOPKS_VALU_PER_MFMA 4
#endif
constexpr int TOPKS_DEPTH = 2;                                       // ring slots per wave
constexpr int TOPKS_SCRATCH_BYTES = 4 * 16 * MAX_TOPK * 8 + 4 * 16 * 4;   // cross-wave merge
constexpr int TOPKS_LDS_BYTES = 4 * TOPKS_DEPTH * BLK * KEY_DIM * 4 + TOPKS_SCRATCH_BYTES;

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
    __device__ __forceinline__ void to_keys(KeyList& K) const {
#pragma unroll
        for (int i = 0; i < MAX_TOPK; ++i)
            K.k[i] = (i < L && row[i < L ? i : 0] != 0xFFFFFFFFu) ? topk_key(v[i < L ? i : 0], row[i < L ? i : 0]) : 0ull;
    }
};

template <int G, int L>
__global__ __launch_bounds__(256, 1) void topk_stream_kernel(TopkStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr uint32_t KT_BYTES = BLK * KEY_DIM * 4;
    constexpr int DEPTH = TOPKS_DEPTH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem) + wave * DEPTH * KT_BYTES;
    const char* my = smem + wave * DEPTH * KT_BYTES;
    unsigned long long* sh_keys = reinterpret_cast<unsigned long long*>(smem + 4 * DEPTH * KT_BYTES);
    float* sh_dmax = reinterpret_cast<float*>(smem + 4 * DEPTH * KT_BYTES + 4 * 16 * MAX_TOPK * 8);

    const int n_waves = gridDim.x * 4;
    const int w_id = blockIdx.x * 4 + wave;
    const int T = (a.n_blocks + n_waves - 1) / n_waves;        // tiles per wave and pass
    const int n_pass = (a.n_groups + G - 1) / G;
    const int total = n_pass * T;                              // this wave's tile sequence
    const int last = a.n_blocks - 1;
    // one tile = 16 rows = 16 DMA instructions (4 groups of 4 rows, swizzled source: chunk c of
    // row R lands at chunk position c ^ R, which makes the ds_read_b128 below conflict-free).
    // Sequence positions past the end (and tiles past the bank: the ragged last round) fetch the
    // bank's last tile again - never consumed - so that every wait below is a constant.
    auto issue_seq = [&](int k) __attribute__((always_inline)) {
        const int i = k < total ? k % T : T - 1;
        const int tile = w_id + i * n_waves;
        const float* src = a.keys + (int64_t)(tile < last ? tile : last) * BLK * KEY_DIM;
        const uint32_t dst = lds0 + (k & (DEPTH - 1)) * KT_BYTES;
#pragma unroll
        for (int gr = 0; gr < 4; ++gr) {
            dma_group_begin(dst + gr * 4096);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4)
                dma_b128_q(src + gr * 4 * KEY_DIM, (uint32_t)((lane ^ (4 * gr + i4)) << 4), i4);
        }
    };
    issue_seq(0);
    issue_seq(1);

    KAddr kaddr;
    kaddr.init(lane);
    uint32_t prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = (uint32_t)pi_row(4 * g + r);

    int k = 0;
    for (int pass = 0; pass < n_pass; ++pass) {
        // query fragments of this pass's groups: lane (j, g) holds Q[j][16 s + 4 g .. +3]
        f32x4 qf[G][16];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int grp = min(pass * G + gi, a.n_groups - 1);
            const int64_t q = (int64_t)grp * 16 + j;
            const f32x4* rowp = reinterpret_cast<const f32x4*>(a.ehat + (q < a.B ? q : a.B - 1) * KEY_DIM);
#pragma unroll
            for (int s = 0; s < 16; ++s) qf[gi][s] = rowp[4 * s + g];
        }
        // (ordinary loads that hipcc counts: "using" them here puts its wait for them in front of
        // the tile loop - at their first use inside it, it would be a vmcnt(0) that drains the
        // hand-counted LDS-DMA ring every iteration)
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
#pragma unroll
            for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(qf[gi][s]));
        }
        ShortList<L> lists[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) lists[gi].init();
        // values of the previous tile, pushed while the current tile's MFMAs run (the first
        // round pushes -inf: a no-op)
        f32x4 prev[G];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) prev[gi] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        uint32_t prev_row0 = 0;
        const uint32_t n_valid32 = (uint32_t)a.n_valid;
        auto push_prev = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t row = prev_row0 + prow[r];
                // (pad rows exist in the bank's last tile only; the compare is cheaper than a branch)
                const float x = row < n_valid32 ? prev[gi][r] : -INFINITY;
                lists[gi].push(x, row);
            }
        };

        for (int i = 0; i < T; ++i, ++k) {
            const int tile = w_id + i * n_waves;
            // tile k has landed when at most the 16 operations of tile k+1 are outstanding
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            const char* kt = my + (k & (DEPTH - 1)) * KT_BYTES;
            f32x4 kf[16];
#pragma unroll
            for (int s = 0; s < 16; ++s)
                kf[s] = *reinterpret_cast<const f32x4*>(kt + kaddr.b[s & 3] + 256 * (s >> 2));
            // the slot is free once these reads have returned: refill it before the arithmetic
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(kf[s]));
            issue_seq(k + 2);
            if (tile < a.n_blocks) {
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
                    // list maintenance of the PREVIOUS tile's values of this group: independent of
                    // the chain above, placed into its shadow (about 4 VALU instructions per MFMA)
#ifdef RANGE_EXP_TS_NOPUSH   // timing experiment only (results invalid)
                    lists[gi].v[0] += prev[gi][0] + prev[gi][1] + prev[gi][2] + prev[gi][3];
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
            }
        }
#pragma unroll
        for (int gi = 0; gi < G; ++gi) push_prev(gi);
        // per group: merge the 4 lane lists of a query, then the 4 waves through LDS; one sorted
        // list of 16 and one dmax per (group, workgroup, query) go to HBM.  The ring keeps
        // streaming the next pass's first tiles meanwhile.
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            const int grp = pass * G + gi;
            KeyList Lk;
            lists[gi].to_keys(Lk);
            merge_lane_groups(Lk);
            float dm = lists[gi].dmax;
            dm = fmaxf(dm, __shfl_xor(dm, 16));
            dm = fmaxf(dm, __shfl_xor(dm, 32));
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < MAX_TOPK; ++i) sh_keys[(wave * 16 + j) * MAX_TOPK + i] = Lk.k[i];
                sh_dmax[wave * 16 + j] = dm;
            }
            __syncthreads();
            if (wave == 0 && grp < a.n_groups) {
                KeyList M;
#pragma unroll
                for (int i = 0; i < MAX_TOPK; ++i) M.k[i] = sh_keys[(g * 16 + j) * MAX_TOPK + i];
                merge_lane_groups(M);
                float d4 = sh_dmax[g * 16 + j];
                d4 = fmaxf(d4, __shfl_xor(d4, 16));
                d4 = fmaxf(d4, __shfl_xor(d4, 32));
                if (g == 0) {
                    const int64_t at = ((int64_t)grp * gridDim.x + blockIdx.x) * 16 + j;
                    unsigned long long* o = a.cand + at * MAX_TOPK;
#pragma unroll
                    for (int i = 0; i < MAX_TOPK; ++i) o[i] = M.k[i];
                    a.dmax[at] = d4;
                }
            }
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped prefetches past the end
}

// One workgroup per query: thread p owns the sorted candidate list of stream workgroup p
// (<= 512); every wave reduces its 64 lists, wave 0 the per-wave results.  Then the exactness
// check of the short lists: if the largest value any lane let go reaches the k-th value found
// (or `force_exact`), the query is recomputed by brute force over all rows, each thread walking
// its rows with the SAME fmaf chain as the MFMA (k order: for s, for component, for lane group),
// with full 16-deep lists.  exact_count (optional) counts the queries that took that path.
__global__ __launch_bounds__(512) void topk_merge_kernel(const unsigned long long* cand, const float* dmax,
                                                         int n_parts, int64_t B, int k, int64_t row_offset,
                                                         const float* keys, const float* ehat, int64_t n_valid,
                                                         int force_exact, int* exact_count,
                                                         float* oval, int64_t* oidx) {
    __shared__ unsigned long long sh[8 * MAX_TOPK];
    __shared__ float sh_d[8];
    __shared__ float sh_q[KEY_DIM];
    __shared__ int sh_flag;
    const int n_wv = blockDim.x >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q = blockIdx.x;
    const int64_t group = q >> 4;
    const int p = threadIdx.x;
    KeyList L;
    L.init();
    float dm = -INFINITY;
    if (p < n_parts) {
        const int64_t at = (group * n_parts + p) * 16 + (q & 15);
        const unsigned long long* src = cand + at * MAX_TOPK;
#pragma unroll
        for (int i = 0; i < MAX_TOPK; ++i) L.k[i] = src[i];
        dm = dmax[at];
    }
    auto reduce = [&](KeyList& X) __attribute__((always_inline)) {   // result in wave 0, every lane
        merge_wave(X);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < MAX_TOPK; ++i) sh[wave * MAX_TOPK + i] = X.k[i];
        }
        __syncthreads();
        KeyList M;
        M.init();
        if (wave == 0) {
            if (lane < n_wv) {
#pragma unroll
                for (int i = 0; i < MAX_TOPK; ++i) M.k[i] = sh[lane * MAX_TOPK + i];
            }
            merge_wave(M);
        }
        __syncthreads();
        return M;
    };
    KeyList M = reduce(L);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) dm = fmaxf(dm, __shfl_xor(dm, off));
    if (lane == 0) sh_d[wave] = dm;
    __syncthreads();
    if (threadIdx.x == 0) {
        float d = sh_d[0];
        for (int w = 1; w < n_wv; ++w) d = fmaxf(d, sh_d[w]);
        // the k-th value found (nothing can have been dropped while fewer than k rows exist)
        const unsigned long long kth = M.k[k - 1];
        const bool unsafe = force_exact || (kth != 0ull && d >= topk_key_val(kth)) ||
                            (kth == 0ull && d > -INFINITY);
        sh_flag = unsafe ? 1 : 0;
        if (unsafe && exact_count) atomicAdd(exact_count, 1);
    }
    __syncthreads();
    if (sh_flag) {
        // brute force, bit-identical similarities: acc = fmaf(K[row][16 s + 4 g + c], Q[..], acc)
        // in the order s = 0..15, c = 0..3, g = 0..3 of the MFMA chain (qk order of the kernels)
        for (int e = threadIdx.x; e < KEY_DIM; e += blockDim.x) sh_q[e] = ehat[q * KEY_DIM + e];
        __syncthreads();
        KeyList X;
        X.init();
        for (int64_t row = threadIdx.x; row < n_valid; row += blockDim.x) {
            const float* kr = keys + row * KEY_DIM;
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
            X.push(topk_key(acc, (uint32_t)row));
        }
        M = reduce(X);
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < MAX_TOPK; ++i) {
            if (i < k) {
                const unsigned long long mm = M.k[i];
                oval[q * k + i] = mm ? topk_key_val(mm) : -INFINITY;
                oidx[q * k + i] = mm ? (int64_t)topk_key_row(mm) + row_offset : (int64_t)-1;
            }
        }
    }
}

}  // namespace range_hip

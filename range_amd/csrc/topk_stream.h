// Small-batch top-k: the HBM-streaming form of the keys scan (range_topk_stream).
//
// For a handful of queries the scan is bound by streaming the N x 1 KB key rows, not by MFMA, and
// the 64-queries-per-workgroup decomposition of pass 1 wastes the machine.  Here the grid is
// PERSISTENT (one workgroup per CU, launched once): every WAVE streams its own 16-row key tiles
// through a wave-private LDS ring filled by LDS-DMA - no workgroup barrier in the tile loop -
// against G groups of 16 queries held in registers (G = 1 or 2 groups share one pass over the
// keys), and several passes run back to back in the one launch with the ring kept full across the
// pass boundary.  (This file holds two forms of the scan: the float32 one, and the bf16-key
// prefilter built on it - further down - which is what range_topk_stream runs by default.)
//
// What makes the stream the only thing that takes time:
//  * the K fragments of a tile are read into registers at once, so the ring slot is free - and
//    its refill is on its way - BEFORE the tile's MFMAs and list work;
//  * the MFMAs are compiler builtins here (no 256-accumulator register pressure as in pass 2), so
//    hipcc pads their hazards and interleaves the list maintenance of the PREVIOUS tile's values
//    into the MFMA shadow of the current one (software pipeline of depth one);
//  * the per-lane candidate lists are SHORT (4 values per lane and group instead of 16): a wave
//    sees only N / 16 / n_waves tiles (6 for range_db_large), i.e. ~24 values per lane, so 16-deep
//    lists never saturate and every value costs a full insertion.  Exactness is kept by
//    bookkeeping: every lane tracks the largest value it ever let go (dmax), and so does every
//    merge on the way up; the final merge compares the largest dmax of a query with the k-th
//    value it found, and only if some dropped value could have belonged to the top-k - 5 of a
//    query's best 16 rows in the few rows ONE lane sees, or exact ties - recomputes that query by
//    brute force (bit-identical dot products: an MFMA chain is an fmaf chain in a fixed order).
//
// The way up, all inside the one launch (round 3; it used to be a second kernel of one 1024-thread
// workgroup per query over 4096 list entries, as long as the scan itself):
//   lane lists (4) -> the 4 lanes of a query: bitonic merge over shuffles -> wave list (8)
//   -> LDS -> the 4 waves of the workgroup: the same merge -> workgroup list (8) -> HBM,
//   written through (sc1), then one agent-scope ticket per workgroup;
//   the LAST min(B, n_wg) workgroups to take a ticket wait until all have (they are the ones
//   that wait least) and merge ONE query each from its n_wg x 8 entries (topk_merge_query): rank
//   by counting against the 16th largest list head, exactness check, float32 re-rank in the
//   prefilter form.  A batch with more queries than workgroups runs the same merge as a second
//   launch (topk_merge_kernel).
//
// Dot products are the same single dependent MFMA chain, in the same k order, as in the scan
// kernels of attend_kernels.h: every kernel that forms a similarity gets the same float.
#pragma once
#include "async_err.h"
#include "attend_kernels.h"

namespace range_hip {

struct TopkStreamArgs {
    const float* keys;          // (n_pad,256)
    const float* ehat;          // (B,256)
    unsigned long long* cand;   // (n_groups * 16 queries, TOPKS_WL, n_wg) keys: entry i of every workgroup's list
                                // (sorted descending, 0 = empty) side by side, so that the merge reads coalesced
    float* dmax;                // (n_groups * 16 queries, n_wg) largest value dropped on the way
    int64_t B;
    int64_t n_valid;
    int32_t n_blocks;
    int32_t n_groups;           // ceil(B / 16)
    unsigned long long* stamps; // RANGE_EXP_TS_STAMPS builds: 8 s_memrealtime stamps per wave
    const void* keys_bf16;      // prefilter form: (n_tiles, 8 chunks, 64 lanes, 8) bf16, see keyfrag_kernel
    // ---- the merge (topk_merge_query), as the tail of the same launch when `fused`
    uint32_t* sync;             // TOPKS_SYNC_WORDS words (topks_tail): 8 arrival counters that only ever count up
    uint32_t sync_base[8];      // what the counters read when this launch starts (the host adds every fused
                                // launch's arrivals: nothing is zeroed, nothing a give-up could leave behind)
    uint32_t* err;              // host-mapped word (or null): set when a merging workgroup's bounded wait gave up
    int32_t debug_giveup;       // test hook: the merging workgroups behave as if their wait had expired
    int32_t fused;              // 1: the last min(B, n_wg) workgroups to arrive merge one query each
    int32_t k;
    int64_t row_offset;
    int32_t force_exact;
    int32_t* exact_count;
    float eps_rel, kmax;        // prefilter form: error bound of the approximate values (0: exact values)
    float* oval;                // (B,k)
    int64_t* oidx;              // (B,k)
};

#ifdef RANGE_EXP_TS_STAMPS   // tuning only: where a wave's time goes (100 MHz real-time counter)
#define RANGE_TS_STAMP(i) do { if (lane == 0 && a.stamps) a.stamps[(size_t)w_id * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RANGE_TS_NOW() __builtin_amdgcn_s_memrealtime()
#define RANGE_TS_ADD(acc, t0) (acc) += __builtin_amdgcn_s_memrealtime() - (t0)
// tail stamps: 16 per workgroup behind the waves' stamps (thread 0 of the workgroup)
#define RANGE_TT_STAMP(i) do { if (threadIdx.x == 0 && a.stamps) a.stamps[(size_t)gridDim.x * 4 * 8 + (size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RANGE_TS_STAMP(i) do { } while (0)
#define RANGE_TS_NOW() 0ull
#define RANGE_TS_ADD(acc, t0) do { } while (0)
#define RANGE_TT_STAMP(i) do { } while (0)
#endif

#ifndef RANGE_TOPKS_VALU_PER_MFMA
#define RANGE_TOPKS_VALU_PER_MFMA 4
#endif
#ifdef RANGE_EXP_TS_TEMPORAL
#define TS_DMAQ dma_b128_q
#else
#define TS_DMAQ dma_b128_q_nt
#endif
constexpr int TOPKS_SG = 4;         // groups whose lists a wave carries through consecutive passes
constexpr int TOPKS_WL = 8;         // entries of a wave's and of a workgroup's list of one query
constexpr int TOPKS_RING_BYTES = 8 * BLK * KEY_DIM * 4;          // 128 KB of key tiles per workgroup
// behind the ring: the waves' lists of a supergroup (4 groups x 4 waves x 16 queries x 8 keys),
// their dmax, and the control words of the tail
constexpr int TOPKS_XL_BYTES = TOPKS_SG * 4 * 16 * TOPKS_WL * 8;
constexpr int TOPKS_XD_BYTES = TOPKS_SG * 4 * 16 * 4;
constexpr int TOPKS_LDS_BYTES = TOPKS_RING_BYTES + TOPKS_XL_BYTES + TOPKS_XD_BYTES + 64;
constexpr uint32_t TOPKS_SPIN_LIMIT = 1u << 21;   // polls (~1 us each) before a merging workgroup gives up

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
// them holds the sorted L best of the union.  Two rounds (lane ^ 16, lane ^ 32) of a bitonic
// merge: c[i] = max(a[i], b[L-1-i]) are the L largest of two descending lists and form a bitonic
// sequence, which log2(L) stages of compare-exchanges sort.  `drop` receives the largest key this
// LANE saw leave; the two lanes of a pair see the same pairs, but the first round of lanes (2,3)
// is invisible to lanes (0,1): the caller reduces `drop` over the 4 lanes (merge4_drop_max).
template <int L, int OFF>
__device__ __forceinline__ void merge4_short_round(unsigned long long (&k)[L], unsigned long long& drop) {
    unsigned long long p[L];
#pragma unroll
    for (int i = 0; i < L; ++i) p[i] = lane_xor_u64<OFF>(k[L - 1 - i]);
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
template <int L>
__device__ __forceinline__ void merge4_short(unsigned long long (&k)[L], unsigned long long& drop) {
    static_assert(L == 4 || L == 8 || L == 16, "power-of-two list");
    merge4_short_round<L, 16>(k, drop);
    merge4_short_round<L, 32>(k, drop);
}
// compare-exchange: afterwards a >= b
__device__ __forceinline__ void topk_cmpx(unsigned long long& a, unsigned long long& b) {
    const unsigned long long hi = a > b ? a : b, lo = a > b ? b : a;
    a = hi; b = lo;
}
// The 4 lanes (j, g) of a query each hold a sorted list of FOUR keys in k[0..3]: afterwards every
// one of them holds the sorted 8 best of the 16 in k[0..7].  Round 1 (lane ^ 16): two lists of 4 make
// a bitonic sequence of 8 - nothing is dropped, no padding moved around (the general merge4_short<8>
// on zero-padded lists costs 1.6x the instructions); round 2 (lane ^ 32): the 8 largest of two
// sorted lists of 8, as in merge4_short.
__device__ __forceinline__ void merge4_lists4_top8(unsigned long long (&k)[8], unsigned long long& drop) {
#pragma unroll
    for (int i = 0; i < 4; ++i) k[4 + i] = lane_xor_u64<16>(k[3 - i]);          // [own descending | partner ascending]
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if ((i & d) == 0) topk_cmpx(k[i], k[i + d]);
        }
    }
    unsigned long long p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = lane_xor_u64<32>(k[7 - i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool up = k[i] > p[i];
        const unsigned long long lo = up ? p[i] : k[i];
        k[i] = up ? k[i] : p[i];
        drop = lo > drop ? lo : drop;
    }
#pragma unroll
    for (int d = 4; d >= 1; d >>= 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if ((i & d) == 0) topk_cmpx(k[i], k[i + d]);
        }
    }
}
// dmax of the merged list: the lanes' own dmax, and everything the merge dropped in ANY of the 4 lanes
__device__ __forceinline__ float merge4_drop_max(float dm, unsigned long long drop) {
    if (drop != 0ull) dm = fmaxf(dm, topk_key_val(drop));
    dm = fmaxf(dm, lane_xor16(dm));
    dm = fmaxf(dm, lane_xor32(dm));
    return dm;
}

// Agent-scope relaxed accesses (global_load / global_store ... sc1: past this CU's L1 and written
// through this XCD's L2): everything one workgroup hands to another inside the launch goes through
// these, and only these (guide: cdna_hip_programming.md, Guideline 16 - "every load sc1" form).
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(const_cast<unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(const_cast<uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The first pass's query operand, loaded BEHIND THE COMPILER'S BACK: vector-memory results return
// in issue order, so a load hipcc counts, issued after the ring's first requests, makes hipcc wait
// with vmcnt(0) - for the whole ring (measured: the first tile's arithmetic started 6 us after the
// kernel, when all four tiles of every wave had landed).  These loads are issued FIRST, the ring's
// requests behind them, and topk_qwait<N> waits until at most the N younger operations (the
// ring's) are outstanding: the query is in registers one round trip after the kernel starts.
// (guide 5.7, form (ii): loads, then a wait statement naming every destination read-write.  The
// destinations are ACCUMULATOR registers: with vector-register destinations hipcc parked each
// loaded value in an accumulator register at once - a copy of data that had not landed yet; the
// generated code is checked for copies between the loads and the wait by tests/test_host_cpu.py.)
// PAIRS = false: d[i] = 16 bytes at p + 64 i; PAIRS = true: d[2 c], d[2 c + 1] = 32 bytes at p + 128 c
template <int I, int N, bool PAIRS>
struct TopkQLoad {
    static __device__ __forceinline__ void run(f32x4 (&d)[N], const char* p) {
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2"
                     : "=a"(d[I]) : "v"(p), "i"(PAIRS ? (I >> 1) * 128 + (I & 1) * 16 : I * 64) : "memory");
        TopkQLoad<I + 1, N, PAIRS>::run(d, p);
    }
};
template <int N, bool PAIRS>
struct TopkQLoad<N, N, PAIRS> {
    static __device__ __forceinline__ void run(f32x4 (&)[N], const char*) {}
};
template <int VMCNT>
__device__ __forceinline__ void topk_qwait(f32x4 (&d)[16]) {
    asm volatile("s_waitcnt vmcnt(%16)"
                 : "+a"(d[0]), "+a"(d[1]), "+a"(d[2]), "+a"(d[3]), "+a"(d[4]), "+a"(d[5]), "+a"(d[6]), "+a"(d[7]),
                   "+a"(d[8]), "+a"(d[9]), "+a"(d[10]), "+a"(d[11]), "+a"(d[12]), "+a"(d[13]), "+a"(d[14]), "+a"(d[15])
                 : "i"(VMCNT) : "memory");
}

// End of a supergroup (its 4 query groups have seen all their passes): per group, the 4 lanes of
// a query merge their lists (bitonic, shuffles only) into the wave's 8 best; through LDS, wave w
// then merges the 4 waves' lists of group w into the workgroup's 8 best and writes them (sc1).
// Two workgroup barriers, outside the tile loop; the ring keeps streaming the next supergroup's
// first tiles meanwhile (its LDS is not touched here).
template <int L>
__device__ __forceinline__ void topks_publish(ShortList<L> (&lists)[TOPKS_SG], int sg, const TopkStreamArgs& a,
                                              char* smem, int lane, int wave, bool more,
                                              unsigned long long* ts = nullptr) {
    static_assert(L == 4 && TOPKS_WL == 8, "merge4_lists4_top8");
    // (stamp builds: ts[0] += lane merges + LDS stores, ts[1] += the wait at the barrier, ts[2] += the
    // workgroup merge, its stores and the second barrier)
    const unsigned long long ts_p0 = RANGE_TS_NOW();
    unsigned long long* wl = reinterpret_cast<unsigned long long*>(smem + TOPKS_RING_BYTES);
    float* wd = reinterpret_cast<float*>(smem + TOPKS_RING_BYTES + TOPKS_XL_BYTES);
    const int g = lane >> 4, j = lane & 15;
#pragma unroll
    for (int gi = 0; gi < TOPKS_SG; ++gi) {
        if (sg * TOPKS_SG + gi < a.n_groups) {
            unsigned long long kk[TOPKS_WL];
#pragma unroll
            for (int i = 0; i < TOPKS_WL; ++i) kk[i] = 0ull;
#pragma unroll
            for (int i = 0; i < L; ++i)
                kk[i] = lists[gi].row[i] != 0xFFFFFFFFu ? topk_key(lists[gi].v[i], lists[gi].row[i]) : 0ull;
            unsigned long long drop = 0ull;
            merge4_lists4_top8(kk, drop);
            const float dm = merge4_drop_max(lists[gi].dmax, drop);
            if (g == 0) {
                ulonglong2* o = reinterpret_cast<ulonglong2*>(wl + ((gi * 4 + wave) * 16 + j) * TOPKS_WL);
#pragma unroll
                for (int i = 0; i < TOPKS_WL; i += 2) o[i / 2] = make_ulonglong2(kk[i], kk[i + 1]);
                wd[(gi * 4 + wave) * 16 + j] = dm;
            }
        }
    }
    const unsigned long long ts_p1 = RANGE_TS_NOW();
    __syncthreads();
    const unsigned long long ts_p2 = RANGE_TS_NOW();
    const int grp = sg * TOPKS_SG + wave;
    if (grp < a.n_groups) {
        const ulonglong2* src = reinterpret_cast<const ulonglong2*>(wl + ((wave * 4 + g) * 16 + j) * TOPKS_WL);
        unsigned long long kk[TOPKS_WL];
#pragma unroll
        for (int i = 0; i < TOPKS_WL; i += 2) { const ulonglong2 t = src[i / 2]; kk[i] = t.x; kk[i + 1] = t.y; }
        unsigned long long drop = 0ull;
        merge4_short<TOPKS_WL>(kk, drop);
        const float dm = merge4_drop_max(wd[(wave * 4 + g) * 16 + j], drop);
        if (g == 0) {
            const int64_t qq = (int64_t)grp * 16 + j;
#pragma unroll
            for (int i = 0; i < TOPKS_WL; ++i) st_agent(a.cand + (qq * TOPKS_WL + i) * gridDim.x + blockIdx.x, kk[i]);
            st_agent(reinterpret_cast<uint32_t*>(a.dmax + qq * gridDim.x + blockIdx.x), __float_as_uint(dm));
        }
    }
    if (more) __syncthreads();       // (the list area is written again at the end of the next supergroup)
#ifdef RANGE_EXP_TS_STAMPS
    if (ts) { ts[0] += ts_p1 - ts_p0; ts[1] += ts_p2 - ts_p1; ts[2] += RANGE_TS_NOW() - ts_p2; }
#else
    (void)ts; (void)ts_p0; (void)ts_p1; (void)ts_p2;
#endif
}

template <int L>
__device__ void topk_merge_query(char* lds, int64_t q, const TopkStreamArgs& a, int n_parts);

template <int L>
__device__ void topk_merge_prefetch(char* lds, int64_t q, const TopkStreamArgs& a);

// End of the stream kernels.  Every workgroup takes a ticket once all its waves' list stores have
// completed; the last min(B, n_wg) to do so - per shard, below - wait for the rest and merge one
// query each.
//  * hand-off (guide, Guideline 16 / MI355X_MICROARCH.md visibility table, first row): payload
//    stores sc1, every storing wave waits vmcnt(0), workgroup barrier, ONE lane adds to a counter
//    (agent scope); the consumer polls the counters with sc1 loads, the other waves pass a
//    barrier that wave then joins, every load of the payload is sc1.  No fence on either side.
//  * the arrivals are counted in 8 SHARDS (blockIdx % 8, each counter on a line of its own): 256
//    adds to one word take ~3 us to drain (guide: fanin), 32 per word a fraction of that.  The
//    workgroups of shard s that arrive last merge the queries q = s (mod 8); a merging workgroup
//    polls all 8 counters (8 lanes, one load each).
//  * all workgroups of the grid are resident (one per CU, grid <= CUs), every one takes its
//    ticket BEFORE it may wait, and those that do not merge exit: the wait ends.  The spin is
//    bounded all the same; a workgroup that gives up marks its query's results invalid
//    (index -1, NaN) and sets the context's host-mapped error word: the host reports it at
//    its next entry or synchronising exit and runs the merge as a second launch from then on.
//  * the counters only ever count up: the host knows what they read when a launch starts
//    (a.sync_base: the sum of all earlier fused launches' arrivals, modulo 2^32) and tickets and
//    waits are differences to that base.  Nothing is zeroed - neither by the host in front of a
//    launch nor by the last workgroup to leave - so a workgroup that gave up leaves nothing behind
//    that a later launch could trip over (every workgroup of its launch still took its ticket).
//  * a merging workgroup loads its query (topk_merge_prefetch) before it waits.
constexpr int TOPKS_SYNC_STRIDE = 64;                          // words between shard counters (256 B)
constexpr int TOPKS_SYNC_SCRATCH = 8 * TOPKS_SYNC_STRIDE;      // host scratch behind the counters (the key-norm reduction)
constexpr int TOPKS_SYNC_WORDS = TOPKS_SYNC_SCRATCH + 16;
__device__ __forceinline__ void topks_tail(const TopkStreamArgs& a, char* smem) {
    RANGE_TT_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // list stores done; the clamped prefetches past the end landed
    RANGE_TT_STAMP(1);
    if (!a.fused) return;
    int* ctl = reinterpret_cast<int*>(smem + TOPKS_RING_BYTES + TOPKS_XL_BYTES + TOPKS_XD_BYTES);
    __syncthreads();
    const int n_wg = (int)gridDim.x;
    const int n_workers = a.B < (int64_t)n_wg ? (int)a.B : n_wg;
    const int shard = (int)(blockIdx.x & 7);
    const int shard_size = (n_wg - shard + 7) >> 3, shard_workers = (n_workers - shard + 7) >> 3;
    if (threadIdx.x == 0)
        ctl[0] = (int)(__hip_atomic_fetch_add(a.sync + TOPKS_SYNC_STRIDE * shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) -
                       a.sync_base[shard]);
    __syncthreads();
    RANGE_TT_STAMP(2);
    const int local = ctl[0] - (shard_size - shard_workers);
    if (local < 0) return;
    const int64_t q = shard + 8 * local;
    if (q >= a.B) return;                  // (cannot happen while host and counters agree; never index past the batch)
    topk_merge_prefetch<TOPKS_WL>(smem, q, a);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const uint32_t need = lane < 8 ? (uint32_t)((n_wg - lane + 7) >> 3) : 0u;
        const uint32_t base = lane < 8 ? a.sync_base[lane] : 0u;
        int ok = 1;
        for (uint32_t spins = 0;; ++spins) {
            const uint32_t v = lane < 8 ? ld_agent(a.sync + TOPKS_SYNC_STRIDE * lane) - base : 0u;
            if (__ballot(v >= need) == ~0ull) break;
            if (spins > TOPKS_SPIN_LIMIT) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (a.debug_giveup) ok = 0;
        if (lane == 0) {
            ctl[1] = ok;
            if (!ok && a.err) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    RANGE_TT_STAMP(3);
    if (!ctl[1]) {
        if ((int)threadIdx.x < a.k) {
            a.oval[q * a.k + threadIdx.x] = __builtin_nanf("");
            a.oidx[q * a.k + threadIdx.x] = (int64_t)-1;
        }
        return;
    }
#ifdef RANGE_EXP_TS_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    topk_merge_query<TOPKS_WL>(smem, q, a, n_wg);
    RANGE_TT_STAMP(15);
#ifdef RANGE_EXP_TS_STAMPS   // shader clocks the merge took (slot 14): against stamps 3 -> 15 it gives the clock rate
    if (threadIdx.x == 0 && a.stamps)
        a.stamps[(size_t)gridDim.x * 4 * 8 + (size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memtime() - clk0;
#endif
}

// NW waves per workgroup, each with a ring of DEPTH tiles (NW * DEPTH * 16 KB of LDS = 128 KB).
// Per-wave stamps (RANGE_EXP_TS_STAMPS) show that with 4 waves x 2 tiles a wave never waits for a
// tile once the first has landed: over a pass it spends 7.9 us in arithmetic, 2.4 us issuing
// LDS-DMA and 0 us waiting.  8 waves x 1 tile (two waves per SIMD) was measured for the 1-group
// kernel: 21.4 us instead of 21.6 at 16 queries, but 37.3 / 70 us instead of 36.2 / 63.6 at 2 / 4
// passes - so the instantiation is 4 x 2 (and the workgroup-level merge counts on 4 waves).
template <int G, int L, int NW, int DEPTH>
__global__ __launch_bounds__(NW * 64, NW / 4) void topk_stream_kernel(TopkStreamArgs a) {
    static_assert(TOPKS_SG % G == 0, "groups per pass must divide the supergroup");
    static_assert(NW == 4 && DEPTH == 2, "128 KB of key tiles per workgroup, one wave per group in the workgroup merge");
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
                TS_DMAQ(src + gr * 4 * KEY_DIM, (uint32_t)((lane ^ (4 * gr + i4)) << 4), i4);
        }
    };
    // the first pass's query fragments first (topk_qwait below), the ring's first requests behind them
    f32x4 qf0[G][16];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const int grp = min(gi, a.n_groups - 1);
        const int64_t q = (int64_t)grp * 16 + j;
        TopkQLoad<0, 16, false>::run(qf0[gi], reinterpret_cast<const char*>(a.ehat + (q < a.B ? q : a.B - 1) * KEY_DIM + 4 * g));
    }
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
                if (ps == 0 && sg == 0) {
                    // (issued in front of the ring's DEPTH x 16 requests, and of the later groups' loads)
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        if (gi == 0 && G == 2) topk_qwait<DEPTH * 16 + 16>(qf0[0]);
                        else topk_qwait<DEPTH * 16>(qf0[gi]);
#pragma unroll
                        for (int s = 0; s < 16; ++s) qf[gi][s] = qf0[gi][s];
                    }
                } else {
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        const int grp = min(grp0 + gi, a.n_groups - 1);
                        const int64_t q = (int64_t)grp * 16 + j;
                        const f32x4* rowp =
                            reinterpret_cast<const f32x4*>(a.ehat + (q < a.B ? q : a.B - 1) * KEY_DIM);
#pragma unroll
                        for (int s = 0; s < 16; ++s) qf[gi][s] = rowp[4 * s + g];
                    }
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
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
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
        topks_publish<L>(lists, sg, a, smem, lane, wave, sg + 1 < n_sg);
    }
    RANGE_TS_STAMP(5);
#ifdef RANGE_EXP_TS_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RANGE_TS_STAMP(6);
    if (lane == 0 && a.stamps) {   // sums over the loop: waiting for tiles / issuing LDS-DMA / arithmetic
        a.stamps[(size_t)w_id * 8 + 3] = ts_wait;
        a.stamps[(size_t)w_id * 8 + 7] = (ts_issue << 32) | (ts_comp & 0xFFFFFFFFull);
    }
#endif
    topks_tail(a, smem);
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
// approximate values.  The merge then takes every candidate within 2 eps of the k-th best
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

// largest squared row norm of the keys (the bits of a non-negative float order like an integer):
// the error bound of the prefilter scales with it, and the constant-shift softmax of pass 1
// needs it <= 1.  One wave per row.
__global__ __launch_bounds__(256) void key_norm_kernel(const float* __restrict__ keys, int64_t n_rows,
                                                       uint32_t* __restrict__ n2max_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(keys + row * KEY_DIM + 4 * lane);
    float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    if (lane == 0) atomicMax(n2max_bits, __float_as_uint(s));
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
    RANGE_TS_STAMP(0);
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
            for (int i4 = 0; i4 < 4; ++i4) TS_DMAQ(src + gr * 4096, (uint32_t)(lane << 4), i4);
        }
    };
    // the first pass's query rows first (topk_qwait below), the ring's first requests behind them:
    // lane (j, g) holds Q[j][32 c + 8 g + 0..7] of chunk c in qraw[2 c], qraw[2 c + 1]
    f32x4 qraw0[G][16];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const int grp = min(gi, a.n_groups - 1);
        const int64_t q = (int64_t)grp * 16 + j;
        const char* pb = reinterpret_cast<const char*>(a.ehat + (q < a.B ? q : a.B - 1) * KEY_DIM + 8 * g);
        TopkQLoad<0, 16, true>::run(qraw0[gi], pb);
    }
    RANGE_TS_STAMP(1);       // (stamp builds: the query loads are out)
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue_seq(d);

    uint32_t prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = (uint32_t)pi_row(4 * g + r);
    const uint32_t n_valid32 = (uint32_t)a.n_valid;
    unsigned long long ts_pub[3] = {0ull, 0ull, 0ull};       // (stamp builds: where the publishes' time goes)

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
                    f32x4 qraw[16];
                    if (ps == 0 && sg == 0) {
                        // (issued in front of the ring's DEPTH x 8 requests, and of the later group's loads)
                        if (gi == 0 && G == 2) topk_qwait<DEPTH * 8 + 16>(qraw0[0]);
                        else topk_qwait<DEPTH * 8>(qraw0[gi]);
#pragma unroll
                        for (int i = 0; i < 16; ++i) qraw[i] = qraw0[gi][i];
                    } else {
                        const int grp = min(grp0 + gi, a.n_groups - 1);
                        const int64_t q = (int64_t)grp * 16 + j;
                        const f32x4* rowp =
                            reinterpret_cast<const f32x4*>(a.ehat + (q < a.B ? q : a.B - 1) * KEY_DIM);
#pragma unroll
                        for (int c = 0; c < 8; ++c) { qraw[2 * c] = rowp[8 * c + 2 * g]; qraw[2 * c + 1] = rowp[8 * c + 2 * g + 1]; }
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const f32x4 v0 = qraw[2 * c], v1 = qraw[2 * c + 1];
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
                if (k == 0) RANGE_TS_STAMP(6);   // (stamp builds: the first pass's query operand is in registers)
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
                    if (k == 0) RANGE_TS_STAMP(2);            // first tile landed
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
        RANGE_TS_STAMP(4);   // all tiles of the supergroup consumed
        topks_publish<L>(lists, sg, a, smem, lane, wave, sg + 1 < n_sg, ts_pub);
    }
    RANGE_TS_STAMP(5);
#ifdef RANGE_EXP_TS_STAMPS
    if (lane == 0 && a.stamps) {   // sums over the supergroups' publishes: lane merges / barrier wait / workgroup merge
        a.stamps[(size_t)w_id * 8 + 3] = ts_pub[1];
        a.stamps[(size_t)w_id * 8 + 7] = (ts_pub[0] << 32) | (ts_pub[2] & 0xFFFFFFFFull);
    }
#endif
    topks_tail(a, smem);
}

// ------------------------------------------------------------------------------------------------
// The merge of one query: 256 threads, thread p owns the sorted list (L keys) of stream
// workgroup p (n_parts <= 256).  Runs as the tail of the stream kernels (topks_tail) or as
// topk_merge_kernel (one workgroup per query) when the batch has more queries than workgroups.
//  1. a lower bound T of the query's 16th best value: the 16th largest list head (rank by
//     counting over the 256 heads in LDS) - each head is the maximum of a different row set, so
//     sixteen candidates are >= T.  Entries below T cannot be in the top 16.
//  2. the survivors (>= T; a few dozen of the 2048 entries) are compacted into LDS and ranked by
//     counting; ranks 0..k-1 are the result, already in order (keys are unique).
//  3. exactness of the short lists: if the largest value any lane, wave or workgroup let go
//     reaches the k-th value found (or `force_exact`), the query is recomputed by brute force
//     over all rows, each thread walking its rows with the SAME fmaf chain as the MFMA (k order:
//     for s, for component, for lane group) and full 16-deep lists.  exact_count (optional)
//     counts the queries that took that path.
//  Prefilter form (eps_rel > 0: the candidates carry APPROXIMATE values from bf16 keys, within
//  eps = eps_rel |q| kmax of the float32 similarity): the threshold of step 1 is lowered by 2 eps,
//  step 2 ranks the survivors by approximate value to find the k-th best approximate value v_k,
//  step 3 widens the check to dall >= v_k - 2 eps, and then every survivor >= v_k - 2 eps gets its
//  float32 similarity by the fmaf chain of the brute-force path (16 lanes per survivor, 1 KB of
//  key row each) and the survivors are ranked again by those: the result is that of the float32 scan.
constexpr int TOPKM_CAP = 256 * TOPKS_WL;    // every entry of every list
constexpr int TOPKM_OFF_SURV2 = TOPKM_CAP * 8;
constexpr int TOPKM_OFF_SH = 2 * TOPKM_CAP * 8;              // brute force: 16 x MAX_TOPK keys
constexpr int TOPKM_OFF_RES = TOPKM_OFF_SH + 16 * MAX_TOPK * 8;
constexpr int TOPKM_OFF_HEAD = TOPKM_OFF_RES + MAX_TOPK * 8;
constexpr int TOPKM_OFF_Q = TOPKM_OFF_HEAD + 256 * 4;
constexpr int TOPKM_OFF_F = TOPKM_OFF_Q + KEY_DIM * 4;
constexpr int TOPKM_OFF_I = TOPKM_OFF_F + 8 * 4;
constexpr int TOPKM_OFF_X = TOPKM_OFF_I + 16;                // transpose area of topk_exact_values: 64 padded key rows
constexpr int TOPKM_LDS_BYTES = TOPKM_OFF_X + 64 * 4 * (256 + 16);
static_assert(TOPKM_LDS_BYTES <= TOPKS_RING_BYTES, "the tail's scratch fits the drained ring");

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

// The brute-force path of the merge: every row's float32 similarity, full 16-deep lists, wave
// merges through `sh` (16 x MAX_TOPK keys), result in res[0..MAX_TOPK).  Inlined: a call would
// give the stream kernels a stack (scratch memory set up at every dispatch).
__device__ __forceinline__ void topk_brute_force(const float* __restrict__ keys, int64_t n_valid,
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

// value of lane l-1 of this lane's row of 16 (lane 0 of a row: 0.0f) - one DPP move
__device__ __forceinline__ float topk_row_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
}

// largest value of the wave, in every lane: two quad permutes and two mirrors inside each row of 16
// (DPP operands of v_max_u32), then the four row results through scalar registers
__device__ __forceinline__ uint32_t topk_wave_umax(uint32_t v) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); v = t > v ? t : v;    // quad_perm [1,0,3,2]
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); v = t > v ? t : v;    // quad_perm [2,3,0,1]
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); v = t > v ? t : v;   // row_half_mirror
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false); v = t > v ? t : v;   // row_mirror
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}

struct TopkMergeLds {
    unsigned long long *surv, *surv2, *sh, *res;
    uint32_t* sh_head;
    float *sh_q, *sh_d, *sh_n2;
    int* sh_i;
    __device__ __forceinline__ explicit TopkMergeLds(char* lds)
        : surv(reinterpret_cast<unsigned long long*>(lds)),
          surv2(reinterpret_cast<unsigned long long*>(lds + TOPKM_OFF_SURV2)),
          sh(reinterpret_cast<unsigned long long*>(lds + TOPKM_OFF_SH)),
          res(reinterpret_cast<unsigned long long*>(lds + TOPKM_OFF_RES)),
          sh_head(reinterpret_cast<uint32_t*>(lds + TOPKM_OFF_HEAD)),
          sh_q(reinterpret_cast<float*>(lds + TOPKM_OFF_Q)),
          sh_d(reinterpret_cast<float*>(lds + TOPKM_OFF_F)),          // [4] dmax per wave (ordered bits)
          sh_n2(reinterpret_cast<float*>(lds + TOPKM_OFF_F) + 4),     // [4] the waves' shares of |q| (norms, not squares)
          sh_i(reinterpret_cast<int*>(lds + TOPKM_OFF_I)) {}          // survivor count, flag
};

// what the merge of query q needs and nobody else writes: the query itself (float32 similarities
// of the survivors / brute force) and its norm.  In the fused tail this runs BEFORE the wait for
// the other workgroups.  256 threads.
template <int L>
__device__ void topk_merge_prefetch(char* lds, int64_t q, const TopkStreamArgs& a) {
    TopkMergeLds m(lds);
    const int p = threadIdx.x;                                        // blockDim.x == 256 == KEY_DIM
    const float v = a.ehat[q * KEY_DIM + p];
    m.sh_q[p] = v;
    // the wave's share of |q|, taken of the elements times the power of two that brings the wave's largest
    // near 1 (a query of norm 1e-25 squared in float32 is 0 - and a zero error bound made the merge hand out
    // the APPROXIMATE values as if they were exact; found by tests/test_gpu_topk_gemm.py in round 5)
    float mx = fabsf(v);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const uint32_t mE = (__float_as_uint(mx) >> 23) & 0xFFu;
    const float up = (mE >= 1u && mE <= 253u) ? __uint_as_float((254u - mE) << 23) : 1.0f;     // 2^(127 - E)
    const float vs = v * up;
    float sq = vs * vs;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) sq += __shfl_xor(sq, off);
    if ((p & 63) == 0) m.sh_n2[p >> 6] = sqrtf(sq) / up;              // (a NORM per wave, not a square)
    if (p == 0) { m.sh_i[0] = 0; m.sh_i[1] = 0; }
    if (p < MAX_TOPK) m.res[p] = 0ull;
}

// ranks (by counting) of the S keys of `src`; ranks 0..15 land in res[] in order (keys are unique)
__device__ __forceinline__ void topk_rank_into(const unsigned long long* src, int S, unsigned long long* res) {
    for (int t = threadIdx.x; t < S; t += 256) {
        const unsigned long long key = src[t];
        if (key == 0ull) continue;
        int r = 0;
        for (int u = 0; u < S; ++u) r += src[u] > key ? 1 : 0;
        if (r < MAX_TOPK) res[r] = key;
    }
}

// float32 similarities of the survivors whose (approximate) value is >= vmin: surv2[t] = key of
// (exact value, row), 0 for the others.  4 lanes per survivor (64 at a time).
//  * loads: 16 instructions, the quad of a survivor reading 64 contiguous bytes of its key row in
//    each (with one lane reading a contiguous quarter row the 64 lanes of an instruction hit 64
//    different cache lines: measured 2.7 us for this step, most of it the tag lookups);
//  * through LDS (rows of 4 quarters, each padded by 16 B so that both the writes above and the
//    reads below are bank-conflict free) every lane s then holds dims 64 s .. 64 s + 63;
//  * the chain runs chunk by chunk in the kernels' order, its value handed from a lane to the
//    next by a DPP row shift (every lane computes on its own four chunks at every step; at step
//    s only lane s has the right input, and lane 3 ends with the result).  The 256 fmaf of a
//    similarity are one dependent chain whatever the split.
constexpr int TOPKM_XROW = 4 * (256 + 16);                   // a key row in the transpose area
__device__ __forceinline__ void topk_exact_values(const TopkStreamArgs& a, const float* sh_q, char* xarea,
                                                  const unsigned long long* surv, unsigned long long* surv2, int S, float vmin) {
    const int p = threadIdx.x, sub = p & 3, cand = p >> 2;
    f32x4 qc[4][4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) qc[h][gg] = *reinterpret_cast<const f32x4*>(sh_q + 64 * sub + 16 * h + 4 * gg);
    }
    char* xrow = xarea + cand * TOPKM_XROW;
    for (int t0 = 0; t0 < (S > 0 ? S : 1); t0 += 64) {            // (at least one round: surv2 is always written)
        const int t = t0 + cand;
        const unsigned long long key = t < S ? surv[t] : 0ull;
        const bool live = key != 0ull && topk_key_val(key) >= vmin;
        const uint32_t row = live ? topk_key_row(key) : 0u;
        const float* kr = a.keys + (int64_t)row * KEY_DIM + 4 * sub;
        f32x4 kc[4][4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) kc[h][gg] = *reinterpret_cast<const f32x4*>(kr + 16 * (4 * h + gg));
        }
        // (a candidate's area row is written and read by its own 4 lanes only - one wave, whose LDS
        // operations execute in order: no barrier between rounds or between the writes and the reads)
        // piece (4 h + gg) of the row: quarter h, 64 bytes gg, this lane's 16 of them
#pragma unroll
        for (int h = 0; h < 4; ++h) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) *reinterpret_cast<f32x4*>(xrow + h * 272 + gg * 64 + sub * 16) = kc[h][gg];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 4; ++h) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) kc[h][gg] = *reinterpret_cast<const f32x4*>(xrow + sub * 272 + h * 64 + gg * 16);
        }
        float acc = 0.f, v = 0.f;
        for (int step = 0; step < 4; ++step) {
            v = acc;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) v = __builtin_fmaf(kc[h][gg][c], qc[h][gg][c], v);
                }
            }
            acc = topk_row_shr1(v);
        }
        if (sub == 3) surv2[t] = live ? topk_key(v, row) : 0ull;     // (t < TOPKM_CAP; entries past S: 0)
    }
}

// sum over the 4 lanes of a quad, in every lane of it (two DPP adds)
__device__ __forceinline__ int topk_sum4(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);     // quad_perm [2,3,0,1]
    return v;
}

// (after topk_merge_prefetch of the same query, and a workgroup barrier)
// Every instruction of this function is on the call's critical path and runs once: a wave64
// instruction takes >= 4 cycles, so ~500 of them are a microsecond (measured: a thread-serial rank
// over 256 heads alone took 7 us).  The usual case - up to 32 candidates reach the ranking - is
// therefore a short straight path: DPP reductions, ONE LDS atomic per wave, 8 lanes per candidate
// for the float32 chain and for its rank, results written by the lanes that hold them; four
// workgroup barriers and two global round trips (the lists, the candidates' key rows).
template <int L>
__device__ void topk_merge_query(char* lds, int64_t q, const TopkStreamArgs& a, int n_parts) {
    TopkMergeLds m(lds);
    const int p = threadIdx.x;                                        // blockDim.x == 256
    const int lane = p & 63, wave = p >> 6;
    const int k = a.k;
    unsigned long long kk[L];
    float dm = -INFINITY;
#pragma unroll
    for (int i = 0; i < L; ++i) kk[i] = 0ull;
    if (p < n_parts) {
#pragma unroll
        for (int i = 0; i < L; ++i) kk[i] = ld_agent(a.cand + (q * L + i) * n_parts + p);   // contiguous over the threads
        dm = __uint_as_float(ld_agent(reinterpret_cast<const uint32_t*>(a.dmax + q * n_parts + p)));
    }
    // ---- 1. a lower bound T of the 16th best value, from the list heads (each the maximum of a
    //      different row set): every wave finds its R largest heads - R rounds of a DPP
    //      max-reduction, one holder leaving per round.  R = 4 when all four waves hold lists: T =
    //      the smallest of the waves' 4th largest heads (sixteen heads are >= it; about the
    //      20th-25th largest head overall).  Banks so small that fewer workgroups streamed them:
    //      R = 16 and T = the 16th largest of the values handed in.
    const int R = n_parts > 192 ? 4 : MAX_TOPK;
    {
        uint32_t h = (uint32_t)(kk[0] >> 32);            // 0 = empty list
        uint32_t mx = 0u;
        for (int r = 0; r < R; ++r) {
            mx = topk_wave_umax(h);
            if (R != 4 && lane == 0) m.sh_head[wave * MAX_TOPK + r] = mx;
            const unsigned long long holders = __ballot(h == mx && mx != 0u);
            if (holders != 0ull && lane == __ffsll((long long)holders) - 1) h = 0u;   // one holder leaves
        }
        const uint32_t dmx = topk_wave_umax(topk_ordered_bits(dm));
        if (lane == 0) {
            if (R == 4) m.sh_head[wave] = mx;            // the wave's 4th largest head
            m.sh_d[wave] = topk_key_val((unsigned long long)dmx << 32);
        }
    }
    __syncthreads();
    RANGE_TT_STAMP(4);
    uint32_t T;
    if (R == 4) {
        const uint32_t t01 = m.sh_head[0] < m.sh_head[1] ? m.sh_head[0] : m.sh_head[1];
        const uint32_t t23 = m.sh_head[2] < m.sh_head[3] ? m.sh_head[2] : m.sh_head[3];
        T = t01 < t23 ? t01 : t23;
    } else {
        T = 0u;
        const uint32_t v = m.sh_head[lane];              // 4 x 16 values: one per lane
        int rank = 0;                                    // unique ranks: ties by lane
        for (int i = 0; i < 64; ++i) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)v, i);
            rank += (o > v || (o == v && i < lane)) ? 1 : 0;
        }
        const unsigned long long at15 = __ballot(rank == MAX_TOPK - 1);
        if (at15 != 0ull) T = (uint32_t)__builtin_amdgcn_readlane((int)v, __ffsll((long long)at15) - 1);
    }
    const float dall = fmaxf(fmaxf(m.sh_d[0], m.sh_d[1]), fmaxf(m.sh_d[2], m.sh_d[3]));
    float eps2 = 0.f;
    const bool approx = a.eps_rel > 0.f;        // prefilter form: the candidates' values are approximate
    if (approx) {          // (a bound, not a result: 1 % over the norm covers its rounding)
        const float nm = fmaxf(fmaxf(m.sh_n2[0], m.sh_n2[1]), fmaxf(m.sh_n2[2], m.sh_n2[3]));
        float r2 = 0.f;
        for (int i = 0; i < 4; ++i) { const float r = nm > 0.f ? m.sh_n2[i] / nm : 0.f; r2 += r * r; }
        eps2 = 2.f * a.eps_rel * 1.01f * (nm * sqrtf(r2)) * a.kmax;
    }
    if (eps2 > 0.f && T != 0u) T = topk_ordered_bits(topk_key_val((unsigned long long)T << 32) - eps2);
    // ---- 2. survivors: the entries >= T - the first c of a thread's sorted list - compacted into
    //      LDS; list position by list position while any lane still has one, the lanes of a wave
    //      taking consecutive places behind ONE LDS atomic per wave
    {
        int c = 0;
#pragma unroll
        for (int i = 0; i < L; ++i) c += (kk[i] != 0ull && (uint32_t)(kk[i] >> 32) >= T) ? 1 : 0;
        unsigned long long mask[L];
        int total = 0;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            mask[i] = __ballot(c > i);
            total += __popcll(mask[i]);
        }
        if (total != 0) {                                      // wave-uniform
            int base = 0;
            if (lane == 0) base = atomicAdd(&m.sh_i[0], total);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int i = 0; i < L; ++i) {
                if (mask[i] == 0ull) break;                    // (uniform; the lists are sorted: later masks are empty too)
                if (c > i)                                     // (at < TOPKM_CAP: every entry has a place)
                    m.surv[base + __popcll(mask[i] & ((1ull << lane) - 1ull))] = kk[i];
                base += __popcll(mask[i]);
            }
        }
    }
    __syncthreads();
    RANGE_TT_STAMP(5);
    const int S = m.sh_i[0];
    // ---- 3. ranking and exactness of the short lists: if the largest value any lane, wave or
    //      workgroup let go could belong to the top k, the query goes to the brute-force path.
    bool unsafe;
    if (S <= 128) {
        // the usual case.  4 lanes per candidate, 64 candidates per round (two rounds beyond 64):
        // its float32 similarity (prefilter form: the candidates carry approximate values), then
        // its rank among all of them by counting, each lane of the quad against a quarter of
        // them; the quad that holds rank r writes result r.  The check compares the largest
        // dropped (approximate) value with the k-th exact one, 2 eps apart.
        const int sub = p & 3;
        const int n64 = S > 64 ? 2 : 1;
        if (approx) topk_exact_values(a, m.sh_q, lds + TOPKM_OFF_X, m.surv, m.surv2, S, -INFINITY);
        else if (sub == 3) {
            for (int rr = 0; rr < n64; ++rr) { const int t = 64 * rr + (p >> 2); m.surv2[t] = t < S ? m.surv[t] : 0ull; }
        }
        __syncthreads();
        RANGE_TT_STAMP(6);
        for (int rr = 0; rr < n64; ++rr) {
            const unsigned long long mine = m.surv2[64 * rr + (p >> 2)];      // (0 past S)
            int r = 0;
            const ulonglong2* o = reinterpret_cast<const ulonglong2*>(m.surv2 + 16 * n64 * sub);
            for (int i = 0; i < 8 * n64; ++i) {
                const ulonglong2 oo = o[i];
                r += (oo.x > mine ? 1 : 0) + (oo.y > mine ? 1 : 0);
            }
            r = topk_sum4(r);
            if (sub == 0 && mine != 0ull && r < k) {
                a.oval[q * k + r] = topk_key_val(mine);
                a.oidx[q * k + r] = (int64_t)topk_key_row(mine) + a.row_offset;
                if (r == k - 1 && dall >= topk_key_val(mine) - eps2) m.sh_i[1] = 1;
            }
        }
        if (p >= S && p < k) {                                // (fewer than k rows exist, or the lists lost some)
            a.oval[q * k + p] = -INFINITY;
            a.oidx[q * k + p] = (int64_t)-1;
        }
        __syncthreads();
        RANGE_TT_STAMP(7);
        // (nothing can have been dropped while fewer than k rows exist)
        unsafe = a.force_exact || m.sh_i[1] != 0 || (S < k && dall > -INFINITY);
        if (!unsafe) return;
    } else {
        // a crowd of near-equal similarities: ranked by (approximate) value first; prefilter form:
        // only those within 2 eps of the k-th best approximate value are recomputed, and ranked again
        topk_rank_into(m.surv, S, m.res);
        __syncthreads();
        const unsigned long long kth = m.res[k - 1];
        unsafe = a.force_exact || (kth != 0ull && dall >= topk_key_val(kth) - eps2) || (kth == 0ull && dall > -INFINITY);
        if (!unsafe && approx) {
            const float vmin = kth != 0ull ? topk_key_val(kth) - eps2 : -INFINITY;
            __syncthreads();                                   // (every thread has read res)
            if (p < MAX_TOPK) m.res[p] = 0ull;
            topk_exact_values(a, m.sh_q, lds + TOPKM_OFF_X, m.surv, m.surv2, S, vmin);
            __syncthreads();
            topk_rank_into(m.surv2, S, m.res);
            __syncthreads();
        }
    }
    if (unsafe) {                                              // (workgroup-uniform: every thread computed it from LDS)
        if (p == 0 && a.exact_count) atomicAdd(a.exact_count, 1);
        __syncthreads();                                       // (res is rewritten)
        topk_brute_force(a.keys, a.n_valid, m.sh_q, m.sh, m.res);
    }
    if (p < k) {
        const unsigned long long mm = m.res[p];
        a.oval[q * k + p] = mm ? topk_key_val(mm) : -INFINITY;
        a.oidx[q * k + p] = mm ? (int64_t)topk_key_row(mm) + a.row_offset : (int64_t)-1;
    }
}

// the merge as a launch of its own: one workgroup per query (batches larger than the stream grid)
template <int L>
__global__ __launch_bounds__(256, 2) void topk_merge_kernel(TopkStreamArgs a, int n_parts) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    topk_merge_prefetch<L>(smem, (int64_t)blockIdx.x, a);
    __syncthreads();
    topk_merge_query<L>(smem, (int64_t)blockIdx.x, a, n_parts);
}

}  // namespace range_hip

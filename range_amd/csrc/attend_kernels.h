// Kernel B of the RANGE engine: streaming soft-attention over the embedding bank.
//
// Reference semantics (range/range.py:213-217, 231-238): for every query
//     H = softmax_N(tau_sem * e . K^T) @ V,   G = softmax_N(tau_geo * x . X^T) @ V,
//     M = (1-beta) * G + beta * H
// with N = ALL bank rows (dense soft attention, no top-k truncation).  The reference materialises
// two (B,N) probability matrices and multiplies each with V; here:
//
//   pass 1  scan_stats_kernel     per query running (max, sum-exp) of both logit rows; the raw
//                                 semantic logits are kept in HBM (4 B per (query,row) pair)
//   pass 2  attend_stored_kernel  reads them back, forms ONE combined weight
//                                 w = beta*p_sem + (1-beta)*p_geo and accumulates w @ V once
//                                 (2572 FLOP per pair over both passes instead of 4614)
//           attend_kernel         the same, recomputing the logits (when they were not kept)
//
// Both passes are FP32-MFMA bound (v_mfma_f32_16x16x4_f32: exact f32 products, bitwise an fmaf
// chain), not HBM bound - see DESIGN.md.  Work decomposition (identical in both passes):
//
//   workgroup = 4 waves = 64 queries; wave w owns queries 16w..16w+15 and, in pass 2, the FULL
//   1024-wide output row of each (64 accumulator tiles of 16x16 = 256 VGPRs).  Bank rows arrive
//   in blocks of 16 through LDS by LDS-DMA (global_load_lds, no VGPR staging) and are shared by
//   the 4 waves.  The logit tile is computed TRANSPOSED, S^T = K_blk . Q^T (bank row on the MFMA
//   row index, query on the lane), so its accumulator registers are directly the A operand of the
//   w @ V product - no LDS round trip and no inter-wave exchange for the weights.
//
//   grid = (query tiles) x (bank splits); a split is a contiguous range of 16-row blocks.  Because
//   pass 2 uses GLOBAL softmax statistics its per-split partial outputs simply add, so splits
//   give full-chip occupancy for any batch size and the same kernel serves a row-sharded bank.
//   blockIdx is mapped so that the workgroups resident on one XCD stream the SAME split(s)
//   (decode_block): the bank rows are fetched once per XCD L2, not once per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "host_plan.h"

namespace range_hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KEY_DIM = 256;
constexpr int VAL_DIM = 1024;
constexpr int QTILE = 64;        // queries per workgroup
constexpr int BLK = 16;          // bank rows per block
constexpr int MAX_TOPK = 16;
constexpr float NEG_BIG = -1.0e30f;

#define RANGE_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// MFMA row index i (0..15) of the transposed logit tile -> bank row inside the 16-row block.
// With i = 4g + r (g = lane group that will hold it, r = accumulator register):
//   row = 8*(r>>1) + 2*g + (r&1)
// so registers r=0,1 of every lane group cover the first 8-row half of the block and r=2,3 the
// second: the w @ V product can consume V in 8-row (32 KB) LDS slots.
__device__ __forceinline__ int pi_row(int i) { return ((i & 2) << 2) | ((i >> 2) << 1) | (i & 1); }

struct ScanArgs {
    const float* keys;     // (n_pad,256)
    const float* xyz4;     // (n_pad,4)
    const float* values;   // (n_pad,1024)   (pass 2)
    const float* ehat;     // (B,256)
    const float* xq;       // (B,4)
    const float* stats;    // (B,4) global stats (pass 2)
    float* out;            // pass 1: (nsplit,B,4) ; pass 2: (nsplit,B,1024)
    float* cand_val;       // pass 1 top-k candidates (nsplit,B,4,K) or null
    int32_t* cand_idx;
    int64_t B;
    int64_t n_valid;       // real bank rows
    int32_t n_blocks;      // ceil(n_valid/16)
    int32_t n_qtiles;
    int32_t n_splits;
    float k_sem;           // tau_sem * log2(e)
    float k_geo;           // tau_geo * log2(e)
    float beta;
    unsigned long long* diag;   // diagnostic build only: per (workgroup, wave) cycle sums
    // kept logits: the raw semantic dot products of pass 1, one 1 KB tile per (query tile, bank
    // block, wave) in accumulator-register order (lane-linear float4).  Pass 1 writes them when
    // non-null; attend_stored_kernel reads them instead of recomputing K . Q^T.
    float* logits;
    int32_t qt_offset;          // pass 2 on a sub-range of the scanned queries: first query / 64
    // pass 1, optional: (nsplit,B,4) largest semantic similarity each lane group met - disjoint
    // row subsets, so the 16th largest of a query's entries bounds its 16th best similarity from
    // below (the threshold of topk_from_logits_kernel)
    float* rowmax;
    // pass 2, stream-K decomposition (round 5): > 0 = the launch has exactly this many workgroups, each
    // walking a contiguous range of (query tile, bank block) units per bank column (SlabMap below);
    // 0 = one workgroup per (bank split, query tile) item (decode_block: the bf16-plane kernel, the
    // diagnostic launches)
    int32_t sk_groups;
    int32_t sk_cols;       // bank columns of the stream-K walk (>= 1)
};

// Where pass 2 leaves its partial outputs, and how their consumers find the parts of a query.
//   split-major (sk_groups == 0): n_parts planes of (B, 1024): part p of query q at (p B + q) 1024.
//   stream-K    (sk_groups  > 0): the bank's blocks are cut into sk_cols COLUMNS (contiguous, near-
//     equal: column c = blocks [c n_blocks / C, (c+1) n_blocks / C)), visited one after the other by
//     ALL workgroups - a column's rows (the values above all: 4 KB per row) are then re-read by the
//     workgroups while they are in the Infinity Cache; one column for a bank or shard that fits it.
//     Inside a column the units u = qtile * column_blocks + block, qtile-major, are cut into
//     sk_groups contiguous, near-equal ranges [start(w), start(w+1)), start(w) = floor(w U / G):
//     workgroup w walks its range in order - at most the tail of one query tile, whole tiles, the head
//     of another - and writes one (64, 1024) slab per query tile it touches, slab index
//     c (G + n_qtiles) + w + qtile (unique: every next segment of the walk increases w or qtile).
//     The parts of query tile qt in column c are the slabs w + qt for w = owner(qt cb) ..
//     owner((qt + 1) cb - 1), in block order; owner(u) = floor(((u + 1) G - 1) / U).  With G = the CUs
//     every workgroup has the same work (+- one block per column): no last partial round,
//     C (G + n_qtiles) slabs instead of n_splits n_qtiles, and a workgroup's fixed costs (~7.5 us:
//     first tiles, 256 KB of stores, dispatch) paid 1-2 times per CU and column.
struct SlabMap {
    int32_t n_parts;      // split-major: planes
    int32_t sk_groups;    // stream-K: workgroups of the launch (0 = split-major)
    int32_t n_blocks;
    int32_t n_qtiles;
    int32_t sk_cols;
};
using range_host::sk_col_begin;     // (host_plan.h: the partition arithmetic, also run under sanitizers on the CPU)
using range_host::sk_owner;
using range_host::sk_start;
// parts of query q in column c (split-major: the one "column" holds all planes): float4 index of the
// first, the stride between parts and their number
__device__ __forceinline__ void slab_parts(const SlabMap& m, int64_t B, int64_t q, int c, int64_t& first4, int64_t& stride4, int& count) {
    if (m.sk_groups == 0) {
        first4 = q * (VAL_DIM / 4);
        stride4 = B * (VAL_DIM / 4);
        count = m.n_parts;
        return;
    }
    const int64_t cb = sk_col_begin(c + 1, m.n_blocks, m.sk_cols) - sk_col_begin(c, m.n_blocks, m.sk_cols);
    const int64_t qt = q / QTILE, U = (int64_t)m.n_qtiles * cb;
    const int64_t w0 = sk_owner(qt * cb, U, m.sk_groups), w1 = sk_owner((qt + 1) * cb - 1, U, m.sk_groups);
    first4 = (((int64_t)c * (m.sk_groups + m.n_qtiles) + w0 + qt) * QTILE + (q - qt * QTILE)) * (VAL_DIM / 4);
    stride4 = (int64_t)QTILE * (VAL_DIM / 4);
    count = (int)(w1 - w0) + 1;
}
__device__ __forceinline__ int slab_cols(const SlabMap& m) { return m.sk_groups == 0 ? 1 : m.sk_cols; }

// float offset of the kept-logit tile of (query tile, bank block, wave)
__device__ __forceinline__ int64_t logit_tile(int64_t qtile, int32_t n_blocks, int block, int wave) {
    return ((qtile * n_blocks + block) * 4 + wave) * 256;
}

// blockIdx -> (split, query tile).  Work items are numbered split-major (item = split * n_qtiles +
// tile).  Blocks b and b+8 share an XCD (measured: XCC_ID == blockIdx % 8), so the blocks of XCD x
// (b = 8j + x) take a CONTIGUOUS run of items: the workgroups resident on one XCD then stream the
// same one or two splits and the bank rows are fetched once per XCD L2, not once per CU.
// Bijective for any item count: XCD x owns cnt(x) = q + (x < r) items, total = 8q + r.
__device__ __forceinline__ void decode_block(const ScanArgs& a, int& split, int& qt) {
    const int total = a.n_splits * a.n_qtiles;
    const int b = blockIdx.x;
    const int x = b & 7, j = b >> 3;
    const int q = total >> 3, r = total & 7;
#ifdef RANGE_EXP_P2_SCATTER     // timing experiment: workgroups of an XCD on unrelated splits (no sharing of V in its L2)
    const int item = (int)(((int64_t)b * 997) % total);
#else
    const int item = x * q + (x < r ? x : r) + j;
#endif
    split = item / a.n_qtiles;
    qt = item - split * a.n_qtiles;
}

struct QFrag {
    f32x4 q[16];   // B operand of S^T = K . Q^T: lane (j = query, g) holds Q[j][16s + 4g + 0..3]
    float xq;      // geo head: xq[j][g]
};

__device__ __forceinline__ void load_qfrag(QFrag& f, const float* ehat, const float* xq, int64_t B,
                                           int64_t q, int g) {
    const int64_t qq = q < B ? q : B - 1;
    const f32x4* row = reinterpret_cast<const f32x4*>(ehat + qq * KEY_DIM);
#pragma unroll
    for (int s = 0; s < 16; ++s) f.q[s] = row[4 * s + g];
    f.xq = xq[qq * 4 + g];
}

// f32 MFMA with the accumulator pinned to arch VGPRs (inline asm).  Why not the builtin: with a
// 512-register budget hipcc (ROCm 7.2) selects every builtin MFMA in its AGPR form; pass 2 already
// fills all 256 AGPRs with the output accumulators, and any further AGPR-form accumulator makes
// the allocator shuttle ~1000 registers per block through v_accvgpr_read/write.  The logit tile
// therefore accumulates in VGPRs through these statements.  hipcc pads nothing around an asm
// MFMA: the operands here come from LDS reads and long-lived registers (never a just-executed
// VALU write; the first MFMA of a chain still carries `s_nop 1`), and a chain ends with
// QKAcc::fence() before any non-MFMA instruction may read the results.
//
// WAR hazard on the A/B operands (measured on gfx950, tools/check_mfma_war.py): the compiler
// treats an asm statement's inputs as dead once the statement has issued and may give their
// registers to the very next VALU instruction; an MFMA is still reading them then, and the
// product comes out wrong (deterministically).  One wait state after the MFMA was enough in every
// experiment; each asm MFMA below carries a trailing `s_nop 1` (two), and tests/test_host_cpu.py checks the
// generated code for MFMA sources written by the following instruction.
__device__ __forceinline__ void mfma_v_first(f32x4& d, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0\n\ts_nop 1" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x4& d, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+v"(d) : "v"(a), "v"(b));
}
// (8-pass MFMA result -> VALU read needs 11 wait states: QKAcc::fence gives 16.)

// The query fragments come from ordinary global loads that hipcc counts; "using" them here puts
// its vmcnt wait for them in front of the main loop.  Otherwise the wait lands at their first use
// INSIDE the loop as vmcnt(0) and drains the hand-counted LDS-DMA ring every iteration.
__device__ __forceinline__ void pin_qfrag(QFrag& f) {
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(f.q[s]));
    asm volatile("" : "+v"(f.xq));
}

// One 16-row block of transposed logits.  kt: LDS K tile [16][256] f32 whose 16-byte chunks were
// permuted at load time (chunk c of row R sits at position c ^ R, see issue_k_tile), which makes
// the ds_read_b128 below bank-conflict free.  The k index is consumed in a permuted order that is
// identical for both operands.  The 64 MFMAs of a tile form ONE dependent chain on a single
// accumulator: back-to-back MFMAs that accumulate into their own result issue at full rate on
// gfx950 (tools/micro/mfma_f32_chains.hip: 99 % with one chain), so the sum needs no VALU adds
// and every kernel that forms logits (both passes, the top-k scans) gets the same value.

// Accumulators of one transposed logit tile: the semantic chain and the geographic tile.
struct QKAcc {
    f32x4 a0, g;
    __device__ __forceinline__ float sem(int r) const { return a0[r]; }
    __device__ __forceinline__ void fence() {   // MFMA results -> VALU readers
        asm volatile("s_nop 15" : "+v"(a0), "+v"(g));
    }
};

// Per-lane LDS byte offsets of the K-tile reads (relative to the tile): with R = this lane's bank
// row and chunk index c = 4s + g, the swizzled position is c ^ R = 4(s ^ (R>>2)) + (g ^ (R&3));
// for s = 4a + b that is base[b] + 256*a bytes, so 4 VGPRs + immediates address all 16 reads.
struct KAddr {
    uint32_t b[4];
    uint32_t x;
    __device__ __forceinline__ void init(int lane) {
        const int g = lane >> 4;
        const int R = pi_row(lane & 15);
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
            b[bb] = (uint32_t)(R * (KEY_DIM * 4) + 64 * (bb ^ (R >> 2)) + 16 * (g ^ (R & 3)));
        x = (uint32_t)((R * 4 + g) * 4);
    }
};

struct KFirst { f32x4 k0, k1; float xa; };   // the reads of a tile's first two steps, issued early

template <bool GEO>
__device__ __forceinline__ KFirst qk_first_reads(const char* kt, const char* xt, const KAddr& ka_) {
    KFirst r;
    r.k0 = *reinterpret_cast<const f32x4*>(kt + ka_.b[0]);
    r.k1 = *reinterpret_cast<const f32x4*>(kt + ka_.b[1]);
    r.xa = GEO ? *reinterpret_cast<const float*>(xt + ka_.x) : 0.f;
    return r;
}

// hook(s) is inlined after the MFMAs of step s (used to start the next phase's LDS reads early).
template <bool GEO, class Hook>
__device__ __forceinline__ void qk_mfma(const char* kt, const KFirst& first, const KAddr& ka_,
                                        const QFrag& f, QKAcc& c, Hook&& hook) {
    // asm statements are scheduling boundaries for hipcc, so the LDS reads stay where the source
    // puts them: 16-byte K reads (4 k-steps each) two steps ahead of the MFMAs that hide their latency.
    f32x4 kn = first.k0, kn2 = first.k1;
    const float xa = first.xa;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 ka = kn;
        kn = kn2;
#ifdef RANGE_EXP_P1_NOLDS
        if (s < 14) kn2 = ka;
#else
        if (s < 14) kn2 = *reinterpret_cast<const f32x4*>(kt + ka_.b[(s + 2) & 3] + 256 * ((s + 2) >> 2));
#endif
        if (s == 0) mfma_v_first(c.a0, ka.x, f.q[s].x);
        else mfma_v(c.a0, ka.x, f.q[s].x);
        mfma_v(c.a0, ka.y, f.q[s].y);
        mfma_v(c.a0, ka.z, f.q[s].z);
        mfma_v(c.a0, ka.w, f.q[s].w);
        hook(s);
    }
    if (GEO) mfma_v_first(c.g, xa, f.xq);
    else c.g = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- LDS-DMA (global_load_lds) by inline asm -------------------------------------------------
// The builtin form makes hipcc (ROCm 7.2) drain vmcnt(0) before the next LDS read because it
// cannot tell which LDS bytes the DMA writes; that would serialise the whole ring.  In asm the
// compiler neither counts nor waits for these operations: every wait on them below is a
// hand-counted s_waitcnt vmcnt(N) followed by a workgroup barrier.  M0 carries the wave-uniform
// LDS destination and is written inside the statement that uses it.  It is not restored: nothing
// else in these kernels uses M0 (gfx9+ DS instructions do not need it, no other LDS-DMA, movrel,
// GWS or sendmsg), and every statement that needs it sets it.
// sbase must be wave-uniform (SGPR pair), voff is the per-lane byte offset.
__device__ __forceinline__ void dma_b128(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                 :: "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
// (non-temporal cache policy: data streamed once per launch)
__device__ __forceinline__ void dma_b128_nt(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 nt"
                 :: "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
// Group form: the instruction's immediate offset is applied to BOTH the global and the LDS
// address, so a run of pieces that is contiguous in both spaces (the 4 quarter rows of a V row,
// the 4 rows of a K tile) needs M0 and the SGPR base only once.  dma_group_begin sets M0;
// dma_b128_q(q) issues piece q (byte offset q*1024, q = 0..3) relative to it.  M0 must survive
// between the statements of a group: nothing else in these kernels writes M0 (checked on the
// generated code by tests/test_host_cpu.py::test_no_foreign_m0_writes).
__device__ __forceinline__ void dma_group_begin(uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma_b128_q(const void* sbase, uint32_t voff, int q) {
    switch (q) {
        case 0: asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase) : "memory"); break;
        case 1: asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(voff), "s"(sbase) : "memory"); break;
        case 2: asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(voff), "s"(sbase) : "memory"); break;
        default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff), "s"(sbase) : "memory"); break;
    }
}
// the same with the non-temporal cache policy (streamed-once data: the key scan of a single pass)
__device__ __forceinline__ void dma_b128_q_nt(const void* sbase, uint32_t voff, int q) {
    switch (q) {
        case 0: asm volatile("global_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(sbase) : "memory"); break;
        case 1: asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024 nt" :: "v"(voff), "s"(sbase) : "memory"); break;
        case 2: asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048 nt" :: "v"(voff), "s"(sbase) : "memory"); break;
        default: asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072 nt" :: "v"(voff), "s"(sbase) : "memory"); break;
    }
}
__device__ __forceinline__ void dma_b32(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, %2"
                 :: "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}

// One K tile (16 rows x 1 KB) + its X tile (16 x 4 f32).  Wave w moves rows 4w..4w+3, one
// dwordx4 DMA per row: lane ln fetches chunk (ln ^ R) of row R and lands at LDS position ln (the
// LDS side of LDS-DMA is always lane-linear; the swizzle lives on the source address).
// Every wave also issues the (identical) 256-byte X copy so that all waves keep the same count
// of outstanding vector-memory operations: 5 per tile.
// kt_lds / xt_lds are LDS byte addresses; swz = (lane ^ 4*wave) precomputed.
__device__ __forceinline__ void issue_k_tile(const float* keys, const float* xyz4, int64_t row0,
                                             uint32_t kt_lds, uint32_t xt_lds, int wave, int lane,
                                             int swz) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int R = 4 * wave + rr;
        dma_b128(keys + (row0 + R) * KEY_DIM, (uint32_t)((swz ^ rr) << 4), kt_lds + R * (KEY_DIM * 4));
    }
    dma_b32(xyz4 + row0 * 4, (uint32_t)(lane << 2), xt_lds);
}

// One 8-row half block of V (32 KB, row-major, linear): 32 pieces of 1 KB, 8 per wave.
__device__ __forceinline__ void issue_v_half(const float* values, int64_t row0, uint32_t vslot_lds,
                                             int wave, int lane) {
#pragma unroll
    for (int ii = 0; ii < 8; ++ii) {
        const int i = 8 * wave + ii;
        dma_b128(values + (row0 + (i >> 2)) * VAL_DIM + (i & 3) * 256, (uint32_t)(lane << 4),
                 vslot_lds + i * 1024);
    }
}

// wait for all but the n youngest vector-memory operations of this wave, then workgroup barrier.
// One asm statement with a memory clobber: no LDS access may be moved across it by the compiler.
// RANGE_EXP_NOBAR / RANGE_EXP_NODMA / RANGE_EXP_NOLDS: tuning experiments only (./build.sh -D...):
// they drop the barriers / the in-loop LDS-DMA issue / the V operand reads of pass 2 to price
// each of them; the results of such a build are garbage.
#ifdef RANGE_EXP_NOBAR
#define RANGE_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory")
#else
#define RANGE_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// Diagnostic build only (attend_kernel<GEO, true>, never on the product path): s_memtime stamps
// around the two parts of a wait so that their cycles can be summed per wave.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define RANGE_WAIT_BARRIER_DIAG(n, vm, bar)                                   \
    do {                                                                      \
        const unsigned long long t0_ = stamp();                               \
        asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)" ::: "memory");      \
        const unsigned long long t1_ = stamp();                               \
        asm volatile("s_barrier" ::: "memory");                               \
        const unsigned long long t2_ = stamp();                               \
        vm += t1_ - t0_; bar += t2_ - t1_;                                    \
    } while (0)
#define RANGE_WB(n, vm, bar)                                                  \
    do { if (DIAG) RANGE_WAIT_BARRIER_DIAG(n, vm, bar); else RANGE_WAIT_BARRIER(n); } while (0)


__device__ __forceinline__ void merge_ml(float& m, float& l, float m2, float l2) {
    const float mm = fmaxf(m, m2);
    l = l * __builtin_amdgcn_exp2f(m - mm) + l2 * __builtin_amdgcn_exp2f(m2 - mm);
    m = mm;
}

// ------------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------------
template <int K>
struct TopK {
    float v[K];
    int32_t i[K];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int k = 0; k < K; ++k) { v[k] = -INFINITY; i[k] = 0x7fffffff; }
    }
    // strict '>' keeps the earlier (lower) row among equal values: a lane meets rows in
    // increasing order.
    __device__ __forceinline__ void push(float x, int32_t idx) {
        if (x > v[K - 1]) {
            v[K - 1] = x; i[K - 1] = idx;
#pragma unroll
            for (int k = K - 1; k > 0; --k) {
                if (v[k] > v[k - 1]) {
                    const float tv = v[k]; v[k] = v[k - 1]; v[k - 1] = tv;
                    const int32_t ti = i[k]; i[k] = i[k - 1]; i[k - 1] = ti;
                }
            }
        }
    }
};

template <bool GEO, bool TOPK>
__global__ __launch_bounds__(256) void scan_stats_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: K ring 2 x [16][256] f32 | X ring 2 x [16][4] f32 (33 KB: four workgroups per CU)
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const uint32_t kring_lds = lds0, xring_lds = lds0 + 2 * BLK * KEY_DIM * 4;
    constexpr uint32_t KT_BYTES = BLK * KEY_DIM * 4;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int swz = lane ^ (4 * wave);
    int split, qt;
    decode_block(a, split, qt);
    const int b0 = (int)(((int64_t)split * a.n_blocks) / a.n_splits);
    const int b1 = (int)(((int64_t)(split + 1) * a.n_blocks) / a.n_splits);
    const int nb = b1 - b0;
    const int64_t q = (int64_t)qt * QTILE + wave * 16 + (lane & 15);

    QFrag f;
    load_qfrag(f, a.ehat, a.xq, a.B, q, g);
    pin_qfrag(f);
    KAddr kaddr;
    kaddr.init(lane);

    // Softmax statistics with a CONSTANT shift.  Both logit rows are dot products of unit vectors
    // (range.py:212 normalises e-hat, :85-89 the keys; utils.py:11-16 gives unit xyz), so
    // t = k * s <= k: the shift m = k (tau * log2 e: 17.3 / 21.6 / 57.7) replaces the running
    // maximum of an online softmax.  2^(t - m) then lies in [2^-2k, 1] - at least 2^-116, a normal
    // float32, for tau <= 43 (checked on the host) - so the statistic of an element is one fma,
    // one exp2 and one add, there is no rescaling, and the statistics of lanes, splits and bank
    // shards merge by plain sums.  Floating point keeps the relative precision of the sum whatever
    // the shift.
    float l1 = 0.f, l2 = 0.f;
    const float nm1 = -a.k_sem, nm2 = -a.k_geo;
    float smax = -INFINITY;      // largest similarity of this lane's rows (a.rowmax)
    TopK<TOPK ? MAX_TOPK : 1> tk;
    if (TOPK) tk.init();

    // ring of 2 K tiles: tile t+1 is requested right after barrier t (every wave is then done
    // with tile t-1, whose slot it re-uses) and waited for before barrier t+1.
    if (nb > 0) {
        issue_k_tile(a.keys, a.xyz4, (int64_t)b0 * BLK, kring_lds, xring_lds, wave, lane, swz);
    }
    int slot = 0;
    for (int t = 0; t < nb; ++t) {
        RANGE_WAIT_BARRIER(0);
#ifndef RANGE_EXP_P1_NODMA
        if (t + 1 < nb) {
            const int s2 = slot ^ 1;
            issue_k_tile(a.keys, a.xyz4, (int64_t)(b0 + t + 1) * BLK, kring_lds + s2 * KT_BYTES,
                         xring_lds + s2 * 256, wave, lane, swz);
        }
#endif
        QKAcc c;
        qk_mfma<GEO>(smem + slot * KT_BYTES,
                     qk_first_reads<GEO>(smem + slot * KT_BYTES, smem + 2 * KT_BYTES + slot * 256, kaddr),
                     kaddr, f, c, [](int) __attribute__((always_inline)) {});
        c.fence();
        const f32x4 ss = {c.sem(0), c.sem(1), c.sem(2), c.sem(3)};
        const f32x4 sg = c.g;
        if (a.logits)   // keep the tile for pass 2 (the barrier's vmcnt(0) also covers this store)
            // (non-temporal: 4 GB per 10^4 x 10^5 launch that nobody reads before pass 2 - measured
            // three A/B pairs, 10 000 queries: pass 1 3.937 -> 3.916 ms, the pass 2 behind it 14.690 ->
            // 14.616 ms; -DRANGE_EXP_P1_TSTORE restores the default policy)
#ifdef RANGE_EXP_P1_TSTORE
            *reinterpret_cast<f32x4*>(a.logits + logit_tile((int64_t)qt + a.qt_offset, a.n_blocks, b0 + t, wave) + 4 * lane) = ss;
#else
            __builtin_nontemporal_store(ss, reinterpret_cast<f32x4*>(a.logits + logit_tile((int64_t)qt + a.qt_offset, a.n_blocks, b0 + t, wave) + 4 * lane));
#endif
        // statistics of this tile.  Only the bank's last block can hold pad rows: every other
        // tile takes the unmasked form
        const int64_t row0 = (int64_t)(b0 + t) * BLK;
        const int n_here = (int)(a.n_valid - row0 < BLK ? a.n_valid - row0 : BLK);   // valid rows
        auto tile_stats = [&](auto masked_tag) __attribute__((always_inline)) {
            constexpr bool MASKED = decltype(masked_tag)::value;
            bool ok[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = pi_row(4 * g + r);
                ok[r] = !MASKED || pr < n_here;
                if (TOPK) { if (ok[r]) tk.push(ss[r], (int32_t)(row0 + pr)); }
            }
            if (a.rowmax)
                smax = fmaxf(smax, fmaxf(fmaxf(ok[0] ? ss[0] : -INFINITY, ok[1] ? ss[1] : -INFINITY),
                                         fmaxf(ok[2] ? ss[2] : -INFINITY, ok[3] ? ss[3] : -INFINITY)));
#ifndef RANGE_EXP_P1_NOVALU
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p1 = __builtin_amdgcn_exp2f(fmaf(ss[r], a.k_sem, nm1));
                l1 += ok[r] ? p1 : 0.f;
                if (GEO) {
                    const float p2 = __builtin_amdgcn_exp2f(fmaf(sg[r], a.k_geo, nm2));
                    l2 += ok[r] ? p2 : 0.f;
                }
            }
#else
            l1 += ss[0] + ss[3]; if (GEO) l2 += sg[1];
#endif
        };
        if (n_here == BLK) tile_stats(std::false_type{});
        else tile_stats(std::true_type{});
        slot ^= 1;
    }
    // lanes j, j+16, j+32, j+48 hold disjoint row subsets of the same query
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        l1 += __shfl_xor(l1, off);
        if (GEO) l2 += __shfl_xor(l2, off);
    }
    const float m1 = a.k_sem;
    float m2 = a.k_geo;
    if (!GEO) { m2 = NEG_BIG; l2 = 0.f; }   // "no rows": stays so under any merge
    if (q < a.B) {
        if (a.rowmax) a.rowmax[((int64_t)split * a.B + q) * 4 + g] = smax;   // before the lane merge
        if (g == 0) {
            f32x4 o = {m1, l1, m2, l2};
            *reinterpret_cast<f32x4*>(a.out + ((int64_t)split * a.B + q) * 4) = o;
        }
        if (TOPK) {
            const int64_t base = (((int64_t)split * a.B + q) * 4 + g) * MAX_TOPK;
#pragma unroll
            for (int k = 0; k < MAX_TOPK; ++k) {
                a.cand_val[base + k] = tk.v[TOPK ? k : 0];
                a.cand_idx[base + k] = tk.i[TOPK ? k : 0];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Top-k lists: 64-bit keys (ordered value bits << 32 | ~row): larger key = larger similarity,
// ties -> lower row index.  (The small-batch HBM-streaming scan that uses them: topk_stream.h.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long topk_key(float v, uint32_t row) {
    const uint32_t b = __float_as_uint(v);
    const uint32_t o = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float topk_key_val(unsigned long long k) {
    const uint32_t o = (uint32_t)(k >> 32);
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ uint32_t topk_key_row(unsigned long long k) { return 0xFFFFFFFFu - (uint32_t)k; }

struct KeyList {                       // sorted descending; 0 = empty slot
    unsigned long long k[MAX_TOPK];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int i = 0; i < MAX_TOPK; ++i) k[i] = 0ull;
    }
    __device__ __forceinline__ void push(unsigned long long x) {
        if (x > k[MAX_TOPK - 1]) {
            k[MAX_TOPK - 1] = x;
#pragma unroll
            for (int i = MAX_TOPK - 1; i > 0; --i) {
                const unsigned long long a = k[i - 1], b = k[i];
                k[i - 1] = a > b ? a : b;
                k[i] = a > b ? b : a;
            }
        }
    }
    __device__ __forceinline__ void pop() {
#pragma unroll
        for (int i = 0; i + 1 < MAX_TOPK; ++i) k[i] = k[i + 1];
        k[MAX_TOPK - 1] = 0ull;
    }
};

// The value lane ^ 16 / lane ^ 32 holds, through gfx950's row-swap instructions instead of the LDS
// crossbar (ds_bpermute: ~120 cycles a round trip, and a wave alone on its SIMD has nothing to put
// into that time): v_permlane16_swap swaps the odd 16-lane rows of its first operand with the even
// rows of the second, v_permlane32_swap the upper half of the first with the lower half of the
// second.  With both operands = x the partner's value ends up in the second operand for lanes of
// even rows / the lower half and in the first for the others: one move, one swap, one select.
typedef uint32_t range_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lane_xor16(uint32_t x) {
    const range_u32x2 r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    return (__lane_id() & 16) ? r[0] : r[1];
}
__device__ __forceinline__ uint32_t lane_xor32(uint32_t x) {
    const range_u32x2 r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return (__lane_id() & 32) ? r[0] : r[1];
}
__device__ __forceinline__ float lane_xor16(float x) { return __uint_as_float(lane_xor16(__float_as_uint(x))); }
__device__ __forceinline__ float lane_xor32(float x) { return __uint_as_float(lane_xor32(__float_as_uint(x))); }

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long x, int m) {
    const uint32_t lo = __shfl_xor((uint32_t)x, m), hi = __shfl_xor((uint32_t)(x >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}
template <int M>
__device__ __forceinline__ unsigned long long lane_xor_u64(unsigned long long x) {
    static_assert(M == 16 || M == 32, "row swaps exist for lane ^ 16 and lane ^ 32");
    const uint32_t lo = M == 16 ? lane_xor16((uint32_t)x) : lane_xor32((uint32_t)x);
    const uint32_t hi = M == 16 ? lane_xor16((uint32_t)(x >> 32)) : lane_xor32((uint32_t)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// merge the sorted lists held by lanes (j, g=0..3) of one query j: afterwards every such lane
// holds the same top-MAX_TOPK list.  Keys are unique (the row is part of the key).
__device__ __forceinline__ void merge_lane_groups(KeyList& L) {
    KeyList R;
#pragma unroll
    for (int i = 0; i < MAX_TOPK; ++i) {
        const unsigned long long h = L.k[0];
        unsigned long long m = h;
        unsigned long long o = lane_xor_u64<16>(m); m = o > m ? o : m;
        o = lane_xor_u64<32>(m); m = o > m ? o : m;
        R.k[i] = m;
        if (h == m && m != 0ull) L.pop();
    }
    L = R;
}

// top list of the 64 sorted lists held by the lanes of one wave (no barrier: shuffles only)
__device__ __forceinline__ void merge_wave(KeyList& L) {
    KeyList R;
#pragma unroll
    for (int i = 0; i < MAX_TOPK; ++i) {
        const unsigned long long h = L.k[0];
        unsigned long long m = h;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = shfl_xor_u64(m, off);
            m = o > m ? o : m;
        }
        R.k[i] = m;
        if (h == m && m != 0ull) L.pop();
    }
    L = R;
}

// theta[q] = the 16th largest of the n_parts*4 per-lane-group maxima pass 1 recorded for query q
// (-inf when there are fewer than 16): a lower bound of the query's 16th best similarity, and a
// tight one - with 52 groups about 20 of 100 000 values reach it.  One wave per query.
__global__ __launch_bounds__(256) void topk_threshold_kernel(const float* __restrict__ rowmax,
                                                             int n_parts, int64_t B,
                                                             float* __restrict__ theta) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= B) return;
    KeyList L;
    L.init();
    const int total = n_parts * 4;
    for (int e = lane; e < total; e += 64) {
        const float v = rowmax[((int64_t)(e >> 2) * B + q) * 4 + (e & 3)];
        if (v > -INFINITY) L.push(topk_key(v, (uint32_t)e));
    }
    merge_wave(L);
    if (lane == 0) theta[q] = L.k[MAX_TOPK - 1] ? topk_key_val(L.k[MAX_TOPK - 1]) : -INFINITY;
}

// Top-k from the KEPT logits (large batches): the semantic similarities of pass 1 are already in
// HBM, one 1 KB tile per (query tile, 16-row block, wave slot); this kernel streams them back
// (HBM-bound: 4 B per (query,row) pair) and keeps a running top-16 per lane.  A lane sees thousands
// of values here, so its list saturates and nearly every value fails the first comparison -
// unlike inside pass 1, where the list maintenance of the TOPK variant costs more than the MFMAs.
// One wave per (wave slot of 16 queries, chunk of blocks); lane (j,g) reads the float4 of rows
// pi_row(4g+r); the 4 lane groups of a query are merged and one sorted list of 16 (value, local
// row) goes to cval/cidx[(chunk*B + query)*16 ...], merged across chunks by merge_topk_wave_kernel.
__global__ __launch_bounds__(256) void topk_from_logits_kernel(const float* __restrict__ logits,
                                                               int32_t n_blocks, int64_t B,
                                                               int64_t n_valid, int32_t n_chunks,
                                                               const float* __restrict__ theta,
                                                               float* __restrict__ cval,
                                                               int32_t* __restrict__ cidx) {
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (query tile, wave)
    const int64_t n_slots = (B + 15) / 16;
    if (slot >= n_slots) return;
    const int chunk = blockIdx.y;
    const int b0 = (int)(((int64_t)chunk * n_blocks) / n_chunks);
    const int b1 = (int)(((int64_t)(chunk + 1) * n_blocks) / n_chunks);
    // tile (qtile, b, wave) sits at ((qtile * n_blocks + b) * 4 + wave) * 256 floats
    const float* base = logits + ((slot >> 2) * (int64_t)n_blocks * 4 + (slot & 3)) * 256 + 4 * lane;
    int prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = pi_row(4 * g + r);
    KeyList L;
    L.init();
    // lower bound of this lane's query's 16th best similarity (topk_threshold_kernel): only the
    // few values that reach it are candidates, so the (wave-divergent) insertion is rare
    const int64_t qq = slot * 16 + (lane & 15);
    const float th = theta[qq < B ? qq : B - 1];
    constexpr int UNROLL = 8;
    for (int b = b0; b < b1; b += UNROLL) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int bb = b + u < b1 ? b + u : b1 - 1;
            v[u] = *reinterpret_cast<const f32x4*>(base + (int64_t)bb * 1024);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (b + u < b1) {
                const int64_t row0 = (int64_t)(b + u) * BLK;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = row0 + prow[r];
                    if (v[u][r] >= th && row < n_valid) L.push(topk_key(v[u][r], (uint32_t)row));
                }
            }
        }
    }
    merge_lane_groups(L);
    const int64_t q = slot * 16 + (lane & 15);
    if (g == 0 && q < B) {
        float* ov = cval + ((int64_t)chunk * B + q) * MAX_TOPK;
        int32_t* oi = cidx + ((int64_t)chunk * B + q) * MAX_TOPK;
#pragma unroll
        for (int i = 0; i < MAX_TOPK; ++i) {
            const unsigned long long mm = L.k[i];
            ov[i] = mm ? topk_key_val(mm) : -INFINITY;
            oi[i] = mm ? (int32_t)topk_key_row(mm) : 0x7fffffff;
        }
    }
}

// Top-k of the pass-1 candidates of one query (scan_stats_kernel<.., true>): one WAVE per query.
// cval / cidx: (n_parts, B, per_part) sorted-by-lane-group candidate values and LOCAL rows
// (0x7fffffff = empty).  Lane l folds candidates l, l+64, ... into a register list, the wave then
// extracts the k best (ties -> lower row) by shuffles.
__global__ __launch_bounds__(256) void merge_topk_wave_kernel(const float* __restrict__ cval,
                                                              const int32_t* __restrict__ cidx,
                                                              int n_parts, int64_t B, int per_part,
                                                              int k, int64_t row_offset,
                                                              float* __restrict__ oval,
                                                              int64_t* __restrict__ oidx) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= B) return;
    KeyList L;
    L.init();
    const int total = n_parts * per_part;
    for (int e = lane; e < total; e += 64) {
        const int p = e / per_part, c = e - p * per_part;
        const int64_t at = ((int64_t)p * B + q) * per_part + c;
        const int32_t row = cidx[at];
        if (row != 0x7fffffff) L.push(topk_key(cval[at], (uint32_t)row));
    }
    merge_wave(L);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < MAX_TOPK; ++i) {
            if (i < k) {
                const unsigned long long mm = L.k[i];
                oval[q * k + i] = mm ? topk_key_val(mm) : -INFINITY;
                oidx[q * k + i] = mm ? (int64_t)topk_key_row(mm) + row_offset : (int64_t)-1;
            }
        }
    }
}

// (n_parts,B,4) -> (B,4): exact log-sum-exp merge, fixed order.
__global__ void merge_stats_kernel(const float* parts, int n_parts, int64_t B, float* out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B) return;
    float m1 = NEG_BIG, l1 = 0.f, m2 = NEG_BIG, l2 = 0.f;
    for (int p = 0; p < n_parts; ++p) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(parts + ((int64_t)p * B + q) * 4);
        merge_ml(m1, l1, s.x, s.y);
        merge_ml(m2, l2, s.z, s.w);
    }
    f32x4 o = {m1, l1, m2, l2};
    *reinterpret_cast<f32x4*>(out + q * 4) = o;
}

// The same merge with one WAVE per query, for the many-split launches of small batches (a
// thread walking 1000+ parts one dependent load at a time takes 0.4 ms): lane l folds parts
// l, l+64, ..., then a fixed butterfly of shuffles.
__global__ __launch_bounds__(256) void merge_stats_wave_kernel(const float* __restrict__ parts,
                                                               int n_parts, int64_t B,
                                                               float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= B) return;
    float m1 = NEG_BIG, l1 = 0.f, m2 = NEG_BIG, l2 = 0.f;
    for (int p = lane; p < n_parts; p += 64) {
        const f32x4 s = *reinterpret_cast<const f32x4*>(parts + ((int64_t)p * B + q) * 4);
        merge_ml(m1, l1, s.x, s.y);
        merge_ml(m2, l2, s.z, s.w);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        merge_ml(m1, l1, __shfl_xor(m1, off), __shfl_xor(l1, off));
        merge_ml(m2, l2, __shfl_xor(m2, off), __shfl_xor(l2, off));
    }
    if (lane == 0) {
        f32x4 o = {m1, l1, m2, l2};
        *reinterpret_cast<f32x4*>(out + q * 4) = o;
    }
}

// top-k of n_cand candidates per query (values desc, ties -> lower index), k <= 16.
// One thread per query; candidate lists are tiny (n_parts * 64 or n_parts * k entries).
__global__ void merge_topk_kernel(const float* cval, const int32_t* cidx32, const int64_t* cidx64,
                                  int n_parts, int64_t B, int per_part, int k, int64_t row_offset,
                                  float* oval, int64_t* oidx) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B) return;
    float bv[MAX_TOPK];
    int64_t bi[MAX_TOPK];
    for (int j = 0; j < MAX_TOPK; ++j) { bv[j] = -INFINITY; bi[j] = INT64_MAX; }
    for (int p = 0; p < n_parts; ++p) {
        const int64_t base = ((int64_t)p * B + q) * per_part;
        for (int c = 0; c < per_part; ++c) {
            const float v = cval[base + c];
            const int64_t i = cidx32 ? (cidx32[base + c] == 0x7fffffff
                                            ? INT64_MAX : (int64_t)cidx32[base + c] + row_offset)
                                     : cidx64[base + c];
            if (i == INT64_MAX) continue;
            // insert if better than the current worst
            if (v > bv[k - 1] || (v == bv[k - 1] && i < bi[k - 1])) {
                int j = k - 1;
                while (j > 0 && (v > bv[j - 1] || (v == bv[j - 1] && i < bi[j - 1]))) {
                    bv[j] = bv[j - 1]; bi[j] = bi[j - 1]; --j;
                }
                bv[j] = v; bi[j] = i;
            }
        }
    }
    for (int j = 0; j < k; ++j) { oval[q * k + j] = bv[j]; oidx[q * k + j] = bi[j] == INT64_MAX ? -1 : bi[j]; }
}

// ------------------------------------------------------------------------------------------------
// pass 2
// ------------------------------------------------------------------------------------------------
// Output accumulators live in the 256 AGPRs for the whole kernel ("+a"): written as asm for the
// same reason as mfma_v - the builtin lets hipcc migrate accumulator tiles between the AGPR and
// VGPR halves of the register file inside the loop.
__device__ __forceinline__ void mfma_a(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0\n\ts_nop 1" : "+a"(acc) : "v"(a), "v"(b));
}

// 8 bank rows x 1024 columns of w @ V for this wave's 16 queries, as 16 steps of 8 MFMAs (4
// accumulator tiles x 2 rows).  Two devices keep the MFMA pipe fed with one wave per SIMD:
//  * hook(h), h = 0..59, is inlined after every second MFMA: scalar/vector work placed there in
//    pieces of <= ~6 instructions (one LDS-DMA for a later tile, a slice of the next block's
//    softmax weights) issues in the shadow of the 32-cycle MFMAs (an MFMA occupies the issue port
//    for 8 of its 32 cycles);
//  * the LAST two steps of a phase are not executed but handed on as a PvCarry (their V operands are
//    already in registers): the next phase runs them right after its barrier, behind the first LDS
//    reads of the new phase, so no LDS latency is exposed at a phase boundary.
struct PvCarry {          // the last PV_CARRY = 2 steps of a half (16 MFMAs = 512 cycles of cover)
    f32x4 v0a, v1a, v0b, v1b;
    float w0, w1;
};

struct PvOps { f32x4 v0, v1; };   // V operands of one step: rows 2g and 2g+1, 4 columns each

// operands of steps 0 and 1 of a half (the read pipeline is two steps deep)
__device__ __forceinline__ void pv_first_reads(const float* vslot, int lane, PvOps& s0, PvOps& s1) {
    const float* base = vslot + (2 * (lane >> 4)) * VAL_DIM + 4 * (lane & 15);
    s0.v0 = *reinterpret_cast<const f32x4*>(base);
    s0.v1 = *reinterpret_cast<const f32x4*>(base + VAL_DIM);
    s1.v0 = *reinterpret_cast<const f32x4*>(base + 64);
    s1.v1 = *reinterpret_cast<const f32x4*>(base + VAL_DIM + 64);
}

__device__ __forceinline__ void pv_exec_carry(f32x4 (&acc)[64], const PvCarry& c) {
    mfma_a(acc[56], c.w0, c.v0a.x);
    mfma_a(acc[57], c.w0, c.v0a.y);
    mfma_a(acc[58], c.w0, c.v0a.z);
    mfma_a(acc[59], c.w0, c.v0a.w);
    mfma_a(acc[56], c.w1, c.v1a.x);
    mfma_a(acc[57], c.w1, c.v1a.y);
    mfma_a(acc[58], c.w1, c.v1a.z);
    mfma_a(acc[59], c.w1, c.v1a.w);
    mfma_a(acc[60], c.w0, c.v0b.x);
    mfma_a(acc[61], c.w0, c.v0b.y);
    mfma_a(acc[62], c.w0, c.v0b.z);
    mfma_a(acc[63], c.w0, c.v0b.w);
    mfma_a(acc[60], c.w1, c.v1b.x);
    mfma_a(acc[61], c.w1, c.v1b.y);
    mfma_a(acc[62], c.w1, c.v1b.z);
    mfma_a(acc[63], c.w1, c.v1b.w);
}

// steps T = 0..13 of one half; s0/s1 = operands of steps 0 and 1 (already requested by the
// caller); steps 14 and 15 are returned in `carry`.  hook(h), h = 0..111, runs after MFMA h.  The sched_barriers pin
// each piece into its own MFMA gap: without them hipcc sinks the pieces behind groups of four
// MFMAs, where only the last MFMA's shadow (24 issue cycles) is left to hide them.
#define RANGE_PV_MFMA(tile, w, v, h)                 \
    mfma_a(acc[tile], w, v);                         \
    __builtin_amdgcn_sched_barrier(0);               \
    hook(h);                                         \
    __builtin_amdgcn_sched_barrier(0)

template <class Hook>
__device__ __forceinline__ void pv_steps(const float* vslot, float w0, float w1, PvOps s0, PvOps s1,
                                         f32x4 (&acc)[64], int lane, PvCarry& carry, Hook&& hook) {
    const float* base = vslot + (2 * (lane >> 4)) * VAL_DIM + 4 * (lane & 15);
    f32x4 v0 = s0.v0, v1 = s0.v1, n0 = s1.v0, n1 = s1.v1;
#pragma unroll
    for (int T = 0; T < 14; ++T) {
        // lane (j,g) reads V[row 2g+rr][64T + 4j .. +3]: one ds_read_b128 feeds 4 accumulator
        // tiles; the reads of step T+2 sit in front of step T's 8 MFMAs (512 cycles of cover)
#ifdef RANGE_EXP_NOLDS
        const f32x4 m0 = v1, m1 = v0;
#else
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(base + 64 * (T + 2));
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(base + VAL_DIM + 64 * (T + 2));
#endif
        RANGE_PV_MFMA(4 * T + 0, w0, v0.x, 8 * T + 0);
        RANGE_PV_MFMA(4 * T + 1, w0, v0.y, 8 * T + 1);
        RANGE_PV_MFMA(4 * T + 2, w0, v0.z, 8 * T + 2);
        RANGE_PV_MFMA(4 * T + 3, w0, v0.w, 8 * T + 3);
        RANGE_PV_MFMA(4 * T + 0, w1, v1.x, 8 * T + 4);
        RANGE_PV_MFMA(4 * T + 1, w1, v1.y, 8 * T + 5);
        RANGE_PV_MFMA(4 * T + 2, w1, v1.z, 8 * T + 6);
        RANGE_PV_MFMA(4 * T + 3, w1, v1.w, 8 * T + 7);
        v0 = n0; v1 = n1; n0 = m0; n1 = m1;
    }
    carry.v0a = v0; carry.v1a = v1; carry.v0b = n0; carry.v1b = n1; carry.w0 = w0; carry.w1 = w1;
}
#undef RANGE_PV_MFMA

// MFMA results -> any non-MFMA reader: wait states first (hipcc pads nothing after an asm MFMA).
__device__ __forceinline__ void acc_fence(f32x4 (&acc)[64]) {
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int i = 0; i < 64; i += 16)
        asm volatile("" : "+a"(acc[i]), "+a"(acc[i + 1]), "+a"(acc[i + 2]), "+a"(acc[i + 3]),
                          "+a"(acc[i + 4]), "+a"(acc[i + 5]), "+a"(acc[i + 6]), "+a"(acc[i + 7]),
                          "+a"(acc[i + 8]), "+a"(acc[i + 9]), "+a"(acc[i + 10]), "+a"(acc[i + 11]),
                          "+a"(acc[i + 12]), "+a"(acc[i + 13]), "+a"(acc[i + 14]), "+a"(acc[i + 15]));
}

// The segments of this workgroup (pass 2): one (split, query tile) item, or the walk of a stream-K
// range.  next() yields the query tile, its block range [b0, b1) and the (64, 1024) tile of the slab
// the partial goes to.  Everything here is wave-uniform (blockIdx only).
struct SegWalk {
    int64_t u, u_end;
    int split, col, cb0, cb;       // split-major: the item's split; stream-K: column, its first block / block count
    __device__ __forceinline__ void init(const ScanArgs& a) {
        col = -1;
        u = u_end = 0;
        cb0 = 0;
        cb = a.n_blocks;
        split = 0;
        if (a.sk_groups == 0) {
            int qt;
            decode_block(a, split, qt);
            const int b0 = (int)(((int64_t)split * a.n_blocks) / a.n_splits);
            const int b1 = (int)(((int64_t)(split + 1) * a.n_blocks) / a.n_splits);
            u = (int64_t)qt * a.n_blocks + b0;
            u_end = u + (b1 - b0);
        }
    }
    __device__ __forceinline__ bool next(const ScanArgs& a, int& qt, int& b0, int& b1, float*& out_tile) {
        if (a.sk_groups == 0) {
            if (u >= u_end) return false;
            qt = (int)(u / a.n_blocks);
            b0 = (int)(u - (int64_t)qt * a.n_blocks);
            b1 = b0 + (int)(u_end - u);
            out_tile = a.out + ((int64_t)split * a.B + (int64_t)qt * QTILE) * VAL_DIM;
            u = u_end;
            return true;
        }
        while (u >= u_end) {                       // next column
            if (++col >= a.sk_cols) return false;
            cb0 = sk_col_begin(col, a.n_blocks, a.sk_cols);
            cb = sk_col_begin(col + 1, a.n_blocks, a.sk_cols) - cb0;
            const int64_t U = (int64_t)a.n_qtiles * cb;
            u = sk_start(blockIdx.x, U, a.sk_groups);
            u_end = sk_start((int64_t)blockIdx.x + 1, U, a.sk_groups);
        }
        qt = (int)(u / cb);
        const int bo = (int)(u - (int64_t)qt * cb);
        const int64_t left = u_end - u;
        const int n = left < (int64_t)(cb - bo) ? (int)left : cb - bo;
        b0 = cb0 + bo;
        b1 = b0 + n;
        out_tile = a.out + ((int64_t)col * (a.sk_groups + a.n_qtiles) + blockIdx.x + qt) * (QTILE * VAL_DIM);
        u += n;
        return true;
    }
};

// LDS map (bytes): V ring 3 x 32 KB | K ring 2 x 16 KB | X ring 2 x 256 B  = 131,584 B
constexpr int ATTEND_LDS_BYTES = (3 * 8 * VAL_DIM + 2 * BLK * KEY_DIM + 2 * 64) * 4;
// pass 1: workgroups per CU (4 = what the 33 KB of LDS allow; fewer by padding the allocation:
// measured 3 and 2 per CU slower, tools/README.md)
#ifndef RANGE_P1_WG_PER_CU
#define RANGE_P1_WG_PER_CU 4
#endif
constexpr int SCAN_LDS_BYTES = RANGE_P1_WG_PER_CU >= 4 ? (2 * BLK * KEY_DIM + 2 * 64) * 4
                                                       : (160 * 1024 / RANGE_P1_WG_PER_CU) / 256 * 256 - 512;

template <bool GEO, bool DIAG = false>
__global__ __launch_bounds__(256, 1) void attend_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vring = reinterpret_cast<float*>(smem);          // 3 slots x [8][1024]
    // then K ring 2 slots x [16][256] and X ring 2 slots x [16][4]

    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const uint32_t vring_lds = lds0;
    const uint32_t kring_lds = lds0 + 3 * 8 * VAL_DIM * 4;
    const uint32_t xring_lds = kring_lds + 2 * BLK * KEY_DIM * 4;
    constexpr uint32_t VS_BYTES = 8 * VAL_DIM * 4, KT_BYTES = BLK * KEY_DIM * 4;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    const int swz = lane ^ (4 * wave);
    SegWalk walk;
    walk.init(a);
    int qt, b0, b1;
    float* out_tile;
    for (bool first_seg = true; walk.next(a, qt, b0, b1, out_tile); first_seg = false) {
    // (a later segment re-uses the LDS rings: every wave must be done with the previous one)
    if (!first_seg) __syncthreads();
    const int nb = b1 - b0;
    const int64_t q = (int64_t)qt * QTILE + wave * 16 + (lane & 15);

    QFrag f;
    load_qfrag(f, a.ehat, a.xq, a.B, q, g);
    // per-query constants: w = ca * 2^(k_sem*s - m1) + cb * 2^(k_geo*g - m2)
    float ca, cb, m1, m2;
    {
        const f32x4 st = *reinterpret_cast<const f32x4*>(a.stats + (q < a.B ? q : a.B - 1) * 4);
        m1 = st.x; m2 = st.z;
        ca = a.beta / st.y;
        cb = GEO ? (1.0f - a.beta) / st.w : 0.f;
    }
    pin_qfrag(f);
    asm volatile("" : "+v"(ca), "+v"(cb), "+v"(m1), "+v"(m2));

    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    KAddr kaddr;
    kaddr.init(lane);
    const char* kring_b = smem + 3 * 8 * VAL_DIM * 4;
    const char* xring_b = kring_b + 2 * BLK * KEY_DIM * 4;
    // per-lane bank row of accumulator register r, relative to the block, and the number of
    // valid rows from this split's first row on (pad rows of the last block get weight 0)
    int prow[4];
    uint32_t kvoff[4];   // per-lane source offsets of the 4 K rows this wave moves (swizzled)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        prow[r] = pi_row(4 * g + r);
        kvoff[r] = (uint32_t)((swz ^ r) << 4);
    }
    const int n_left = (int)(a.n_valid - (int64_t)b0 * BLK);

    // Schedule (per wave; "half" = 8 bank rows, two per 16-row block t):
    //   half 2t   : wait+barrier | PV rows 0-7 of block t            (+ issue V half 2t+2)
    //   half 2t+1 : wait+barrier | QK of block t+1 | PV rows 8-15    (+ issue V half 2t+3,
    //               K/X tile t+2; the weights of block t+1 are formed between the PV MFMAs)
    // LDS-DMA groups in issue order: ... E(t-1)=8 | O(t-1)=8+5 | E(t)=8 | O(t)=8+5 ...; the
    // wait before barrier 2t leaves O(t-1) in flight, the one before barrier 2t+1 leaves E(t).
    // V half h lives in ring slot h%3 (re-filled two halves after its last read), K/X tile t in
    // slot t&1.  The steady state is branch-free: past the split's last block the prefetches
    // re-read that block (clamped) into slots nobody reads any more, so every group has its full
    // count and the waits are constants.
    f32x4 w_cur = {0.f, 0.f, 0.f, 0.f};
    if (nb > 0) {
        const int64_t r0 = (int64_t)b0 * BLK;
        const int64_t r1 = (int64_t)(nb > 1 ? b0 + 1 : b0) * BLK;
        issue_k_tile(a.keys, a.xyz4, r0, kring_lds, xring_lds, wave, lane, swz);
        issue_v_half(a.values, r0, vring_lds, wave, lane);
        issue_v_half(a.values, r0 + 8, vring_lds + VS_BYTES, wave, lane);
        issue_k_tile(a.keys, a.xyz4, r1, kring_lds + KT_BYTES, xring_lds + 256, wave, lane, swz);
        RANGE_WAIT_BARRIER(21);
        QKAcc c;
        qk_mfma<GEO>(kring_b, qk_first_reads<GEO>(kring_b, xring_b, kaddr), kaddr, f, c,
                     [](int) __attribute__((always_inline)) {});
        c.fence();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float wr = ca * __builtin_amdgcn_exp2f(fmaf(c.sem(r), a.k_sem, -m1));
            if (GEO) wr = fmaf(cb, __builtin_amdgcn_exp2f(fmaf(c.g[r], a.k_geo, -m2)), wr);
            w_cur[r] = prow[r] < n_left ? wr : 0.f;
        }
    }
    int vs = 0;   // V slot of half 2t
    PvCarry carry;
    carry.v0a = carry.v1a = carry.v0b = carry.v1b = f32x4{0.f, 0.f, 0.f, 0.f};
    carry.w0 = carry.w1 = 0.f;
    unsigned long long d_vm0 = 0, d_bar0 = 0, d_pv0 = 0, d_vm1 = 0, d_bar1 = 0, d_qk = 0, d_pv1 = 0;
    unsigned long long d_mark = 0;
    const unsigned long long d_start = DIAG ? stamp() : 0;
    const int b_last = b1 - 1;
    for (int t = 0; t < nb; ++t) {
        const int vs1 = vs == 2 ? 0 : vs + 1;
        const int vs2 = vs1 == 2 ? 0 : vs1 + 1;
        const int bn1 = min(b0 + t + 1, b_last), bn2 = min(b0 + t + 2, b_last);
        // wave-uniform source / destination bases of this wave's pieces
        const float* vsrc1 = a.values + ((int64_t)bn1 * BLK + 2 * wave) * VAL_DIM;     // rows 2w, 2w+1
        const float* ksrc2 = a.keys + ((int64_t)bn2 * BLK + 4 * wave) * KEY_DIM;       // rows 4w..4w+3
        const float* xsrc2 = a.xyz4 + (int64_t)bn2 * BLK * 4;
        const uint32_t vdst_e = vring_lds + vs2 * VS_BYTES + wave * 8192;   // half 2t+2
        const uint32_t vdst_o = vring_lds + vs * VS_BYTES + wave * 8192;    // half 2t+3
        const uint32_t kdst = kring_lds + (t & 1) * KT_BYTES + wave * 4096;
        const uint32_t xdst = xring_lds + (t & 1) * 256;
        const uint32_t vvoff = (uint32_t)(lane << 4);
        // ---- half 2t
        RANGE_WB(13, d_vm0, d_bar0);
        if (DIAG) d_mark = stamp();
        {
            PvOps s0, s1;
            pv_first_reads(vring + vs * 8 * VAL_DIM, lane, s0, s1);
            pv_exec_carry(acc, carry);                       // last step of the previous half
            pv_steps(vring + vs * 8 * VAL_DIM, w_cur[0], w_cur[1], s0, s1, acc, lane, carry,
                     [&](int h) __attribute__((always_inline)) {
                         if (h % 14 == 3) {                  // 8 pieces: V half 2t+2
                             const int ii = h / 14;
                             if ((ii & 3) == 0) dma_group_begin(vdst_e + (ii >> 2) * 4096);
                             dma_b128_q(vsrc1 + (ii >> 2) * VAL_DIM, vvoff, ii & 3);
                         }
                     });
        }
        // ---- half 2t+1
        if (DIAG) d_pv0 += stamp() - d_mark;
        RANGE_WB(8, d_vm1, d_bar1);
        if (DIAG) d_mark = stamp();
        QKAcc c;
        PvOps s0, s1;
        {
            const char* kt = kring_b + ((t + 1) & 1) * KT_BYTES;
            const KFirst kf = qk_first_reads<GEO>(kt, xring_b + ((t + 1) & 1) * 256, kaddr);
            pv_exec_carry(acc, carry);                       // last step of half 2t
            qk_mfma<GEO>(kt, kf, kaddr, f, c, [&](int s_) __attribute__((always_inline)) {
                if (s_ == 13) pv_first_reads(vring + vs1 * 8 * VAL_DIM, lane, s0, s1);
            });
        }
        if (DIAG) { const unsigned long long x_ = stamp(); d_qk += x_ - d_mark; d_mark = x_; }
        f32x4 w_next = {0.f, 0.f, 0.f, 0.f};
        float e1[4], e2[4];
        const int n_left1 = n_left - (t + 1) * BLK;
        pv_steps(vring + vs1 * 8 * VAL_DIM, w_cur[2], w_cur[3], s0, s1, acc, lane, carry,
                [&](int h) __attribute__((always_inline)) {
                    if ((h & 7) == 3) {
                        const int ii = h >> 3;               // 13 pieces: V half 2t+3, K/X tile t+2
                        if (ii < 8) {
                            if ((ii & 3) == 0) dma_group_begin(vdst_o + (ii >> 2) * 4096);
                            dma_b128_q(vsrc1 + (8 + (ii >> 2)) * VAL_DIM, vvoff, ii & 3);
                        } else if (ii < 12) {
                            if (ii == 8) dma_group_begin(kdst);
                            dma_b128_q(ksrc2, kvoff[ii - 8], ii - 8);
                        } else if (ii == 12) {
                            dma_b32(xsrc2, (uint32_t)(lane << 2), xdst);
                        }
                    } else if (h >= 21 && h < 101 && ((h - 21) & 3) == 0) {
                        // weights of block t+1 in 20 slices of <= 4 VALU instructions; the first
                        // runs >= 20 MFMAs after the last QK MFMA, whose results are long readable
                        const int k = (h - 21) >> 2, r = k / 5, part = k % 5;
                        if (part == 0) {
                        } else if (part == 1) {
                            e1[r] = fmaf(c.a0[r], a.k_sem, -m1);
                            if (GEO) e2[r] = fmaf(c.g[r], a.k_geo, -m2);
                        } else if (part == 2) {
                            e1[r] = __builtin_amdgcn_exp2f(e1[r]);
                        } else if (part == 3) {
                            if (GEO) e2[r] = __builtin_amdgcn_exp2f(e2[r]);
                        } else {
                            float wr = ca * e1[r];
                            if (GEO) wr = fmaf(cb, e2[r], wr);
                            w_next[r] = prow[r] < n_left1 ? wr : 0.f;
                        }
                    }
                });
        if (DIAG) d_pv1 += stamp() - d_mark;
        w_cur = w_next;
        vs = vs2;
    }
    if (nb > 0) pv_exec_carry(acc, carry);   // last step of the last half
    // the clamped prefetches of the last iterations are still in flight into this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DIAG && lane == 0 && a.diag) {
        unsigned long long* d = a.diag + ((size_t)blockIdx.x * 4 + wave) * 16;
        d[0] = d_vm0; d[1] = d_bar0; d[2] = d_pv0; d[3] = d_vm1; d[4] = d_bar1; d[5] = d_qk;
        d[6] = d_pv1; d[7] = stamp() - d_start; d[8] = (unsigned long long)nb; d[9] = d_start;
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[10] = xcc & 0xf;
    }

    acc_fence(acc);
    // accumulator tile 4T+c, register r, lane (j,g)  ->  out[query 4g+r of this wave][64T + 4j + c]
    const int j = lane & 15;
    const int64_t qw = (int64_t)qt * QTILE + wave * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t qo = qw + 4 * g + r;
        if (qo < a.B) {
            float* orow = out_tile + (wave * 16 + 4 * g + r) * VAL_DIM + 4 * j;
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                f32x4 o = {acc[4 * T + 0][r], acc[4 * T + 1][r], acc[4 * T + 2][r], acc[4 * T + 3][r]};
                *reinterpret_cast<f32x4*>(orow + 64 * T) = o;
            }
        }
    }
    }   // segments
}

// ------------------------------------------------------------------------------------------------
// pass 2 on KEPT logits.  Same schedule, same V ring, same weight arithmetic and summation order as
// attend_kernel (the outputs are bit-identical), but the semantic logits of block t+1 are not
// recomputed (64 of the 321 MFMAs per block and wave): pass 1 left them in HBM in accumulator
// order, and they arrive like a K tile did - one 1 KB LDS-DMA per wave and block into a 2-slot
// ring - at 4 B per (query, row) of extra HBM traffic each way, on a kernel that is MFMA-bound.
// The geographic tile (one MFMA per block) is still recomputed from the X ring.
// LDS map (bytes): V ring 3 x 32 KB | S ring 2 x 4 KB (1 KB per wave) | X ring 2 x 256 B
// LDS-DMA groups: E(t) = 8 V pieces, O(t) = 8 V + 1 S + 1 X = 10.
// ------------------------------------------------------------------------------------------------
constexpr int ATTEND_STORED_LDS_BYTES = (3 * 8 * VAL_DIM + 2 * 1024 + 2 * 64) * 4;

template <bool GEO>
__global__ __launch_bounds__(256, 1) void attend_stored_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vring = reinterpret_cast<float*>(smem);          // 3 slots x [8][1024]
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const uint32_t vring_lds = lds0;
    const uint32_t sring_lds = lds0 + 3 * 8 * VAL_DIM * 4;
    const uint32_t xring_lds = sring_lds + 2 * 4096;
    constexpr uint32_t VS_BYTES = 8 * VAL_DIM * 4;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
#ifdef RANGE_EXP_P2_STAMPS
    // timing build: s_memtime (shader clocks) / s_memrealtime (100 MHz) at entry, loop start, loop
    // end, stores issued, stores done + the CU this workgroup ran on (tools/pass2_stamps.py)
    unsigned long long st_c[5], st_r[5];
    st_c[0] = stamp(); st_r[0] = __builtin_amdgcn_s_memrealtime();
#endif
    SegWalk walk;
    walk.init(a);
    int qt, b0, b1;
    float* out_tile;
    for (bool first_seg = true; walk.next(a, qt, b0, b1, out_tile); first_seg = false) {
    // (a later segment re-uses the LDS rings: every wave must be done with the previous one)
    if (!first_seg) __syncthreads();
    const int nb = b1 - b0;
    const int64_t q = (int64_t)qt * QTILE + wave * 16 + (lane & 15);
    const int64_t qtile_kept = (int64_t)qt + a.qt_offset;

    // per-query constants: w = ca * 2^(k_sem*s - m1) + cb * 2^(k_geo*g - m2)
    float ca, cb, m1, m2, fxq;
    {
        const int64_t qq = q < a.B ? q : a.B - 1;
        const f32x4 st = *reinterpret_cast<const f32x4*>(a.stats + qq * 4);
        m1 = st.x; m2 = st.z;
        ca = a.beta / st.y;
        cb = GEO ? (1.0f - a.beta) / st.w : 0.f;
        fxq = a.xq[qq * 4 + g];
    }
    // (ordinary loads: put hipcc's wait for them in front of the loop, see pin_qfrag)
    asm volatile("" : "+v"(ca), "+v"(cb), "+v"(m1), "+v"(m2), "+v"(fxq));

    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* sring_b = smem + 3 * 8 * VAL_DIM * 4;
    const char* xring_b = sring_b + 2 * 4096;
    const uint32_t s_rd = (uint32_t)(wave * 1024 + lane * 16);                  // this lane's logits
    const uint32_t x_rd = (uint32_t)((pi_row(lane & 15) * 4 + g) * 4);          // as KAddr::x
    int prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = pi_row(4 * g + r);
    const int n_left = (int)(a.n_valid - (int64_t)b0 * BLK);
    const uint32_t vvoff = (uint32_t)(lane << 4);

    // one S tile (this wave's 1 KB) + the X tile: 2 vector-memory operations per wave
    auto issue_sx = [&](int block, int slot) __attribute__((always_inline)) {
        dma_b128(a.logits + logit_tile(qtile_kept, a.n_blocks, block, wave), vvoff,
                 sring_lds + slot * 4096 + wave * 1024);
        dma_b32(a.xyz4 + (int64_t)block * BLK * 4, (uint32_t)(lane << 2), xring_lds + slot * 256);
    };

    f32x4 w_cur = {0.f, 0.f, 0.f, 0.f};
    if (nb > 0) {
        const int64_t r0 = (int64_t)b0 * BLK;
        issue_sx(b0, 0);
        issue_v_half(a.values, r0, vring_lds, wave, lane);
        issue_v_half(a.values, r0 + 8, vring_lds + VS_BYTES, wave, lane);
        issue_sx(nb > 1 ? b0 + 1 : b0, 1);
        RANGE_WAIT_BARRIER(18);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sring_b + s_rd);
        f32x4 cg = {0.f, 0.f, 0.f, 0.f};
        if (GEO) {
            const float xa = *reinterpret_cast<const float*>(xring_b + x_rd);
            mfma_v_first(cg, xa, fxq);
            asm volatile("s_nop 15" : "+v"(cg));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float wr = ca * __builtin_amdgcn_exp2f(fmaf(sv[r], a.k_sem, -m1));
            if (GEO) wr = fmaf(cb, __builtin_amdgcn_exp2f(fmaf(cg[r], a.k_geo, -m2)), wr);
            w_cur[r] = prow[r] < n_left ? wr : 0.f;
        }
    }
    int vs = 0;   // V slot of half 2t
    PvCarry carry;
    carry.v0a = carry.v1a = carry.v0b = carry.v1b = f32x4{0.f, 0.f, 0.f, 0.f};
    carry.w0 = carry.w1 = 0.f;
    const int b_last = b1 - 1;
#ifdef RANGE_EXP_P2_STAMPS
    st_c[1] = stamp(); st_r[1] = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = 0; t < nb; ++t) {
        const int vs1 = vs == 2 ? 0 : vs + 1;
        const int vs2 = vs1 == 2 ? 0 : vs1 + 1;
        const int bn1 = min(b0 + t + 1, b_last), bn2 = min(b0 + t + 2, b_last);
        const float* vsrc1 = a.values + ((int64_t)bn1 * BLK + 2 * wave) * VAL_DIM;     // rows 2w, 2w+1
        const float* ssrc2 = a.logits + logit_tile(qtile_kept, a.n_blocks, bn2, wave);
        const float* xsrc2 = a.xyz4 + (int64_t)bn2 * BLK * 4;
        const uint32_t vdst_e = vring_lds + vs2 * VS_BYTES + wave * 8192;   // half 2t+2
        const uint32_t vdst_o = vring_lds + vs * VS_BYTES + wave * 8192;    // half 2t+3
        const uint32_t sdst = sring_lds + (t & 1) * 4096 + wave * 1024;
        const uint32_t xdst = xring_lds + (t & 1) * 256;
        // ---- half 2t : leaves O(t-1) = 10 operations in flight
        RANGE_WAIT_BARRIER(10);
        {
            PvOps s0, s1;
            pv_first_reads(vring + vs * 8 * VAL_DIM, lane, s0, s1);
            pv_exec_carry(acc, carry);                       // last step of the previous half
            pv_steps(vring + vs * 8 * VAL_DIM, w_cur[0], w_cur[1], s0, s1, acc, lane, carry,
                     [&](int h) __attribute__((always_inline)) {
#ifndef RANGE_EXP_NODMA
                         if (h % 14 == 3) {                  // 8 pieces: V half 2t+2
                             const int ii = h / 14;
                             if ((ii & 3) == 0) dma_group_begin(vdst_e + (ii >> 2) * 4096);
                             dma_b128_q(vsrc1 + (ii >> 2) * VAL_DIM, vvoff, ii & 3);
                         }
#endif
                     });
        }
        // ---- half 2t+1 : leaves E(t) = 8 operations in flight
        RANGE_WAIT_BARRIER(8);
        PvOps s0, s1;
        f32x4 cg = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sring_b + ((t + 1) & 1) * 4096 + s_rd);
        {
            const float xa = GEO ? *reinterpret_cast<const float*>(xring_b + ((t + 1) & 1) * 256 + x_rd) : 0.f;
            pv_first_reads(vring + vs1 * 8 * VAL_DIM, lane, s0, s1);
            pv_exec_carry(acc, carry);                       // last step of half 2t
            if (GEO) mfma_v_first(cg, xa, fxq);
        }
        f32x4 w_next = {0.f, 0.f, 0.f, 0.f};
        float e1[4], e2[4];
        const int n_left1 = n_left - (t + 1) * BLK;
        pv_steps(vring + vs1 * 8 * VAL_DIM, w_cur[2], w_cur[3], s0, s1, acc, lane, carry,
                [&](int h) __attribute__((always_inline)) {
#ifdef RANGE_EXP_NODMA
                    if (false) {
                        const int ii = 0;
#else
                    if ((h & 7) == 3) {
                        const int ii = h >> 3;               // 10 pieces: V half 2t+3, S/X tile t+2
#endif
                        if (ii < 8) {
                            if ((ii & 3) == 0) dma_group_begin(vdst_o + (ii >> 2) * 4096);
                            dma_b128_q(vsrc1 + (8 + (ii >> 2)) * VAL_DIM, vvoff, ii & 3);
                        } else if (ii == 8) {
                            dma_b128(ssrc2, vvoff, sdst);
                        } else if (ii == 9) {
                            dma_b32(xsrc2, (uint32_t)(lane << 2), xdst);
                        }
                    } else if (h >= 22 && h < 78 && (h & 1) == 0) {
                        // weights of block t+1, ONE VALU instruction per MFMA gap (an exp2 is a
                        // quarter-rate instruction: two of them in one gap delay the next MFMA);
                        // even h only, the LDS-DMA pieces sit on odd h.  The first runs > 20
                        // MFMAs after the geo MFMA, whose result is long readable.
                        const int k = (h - 22) >> 1, r = k / 7, op = k % 7;
                        if (op == 0) e1[r] = fmaf(sv[r], a.k_sem, -m1);
                        else if (op == 1) { if (GEO) e2[r] = fmaf(cg[r], a.k_geo, -m2); }
                        else if (op == 2) e1[r] = __builtin_amdgcn_exp2f(e1[r]);
                        else if (op == 3) { if (GEO) e2[r] = __builtin_amdgcn_exp2f(e2[r]); }
                        else if (op == 4) e1[r] = ca * e1[r];
                        else if (op == 5) { if (GEO) e1[r] = fmaf(cb, e2[r], e1[r]); }
                        else w_next[r] = e1[r];
                    }
                });
        // pad rows exist only in the bank's last block: their weights are zeroed here, outside
        // the MFMA gaps (a VALU instruction in a gap costs MFMA issue time)
        if (n_left1 < BLK) {
#pragma unroll
            for (int r = 0; r < 4; ++r) w_next[r] = prow[r] < n_left1 ? w_next[r] : 0.f;
        }
        w_cur = w_next;
        vs = vs2;
    }
    if (nb > 0) pv_exec_carry(acc, carry);   // last step of the last half
    // the clamped prefetches of the last iterations are still in flight into this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef RANGE_EXP_P2_STAMPS
    st_c[2] = stamp(); st_r[2] = __builtin_amdgcn_s_memrealtime();
#endif

    acc_fence(acc);
    // accumulator tile 4T+c, register r, lane (j,g)  ->  out[query 4g+r of this wave][64T + 4j + c]
    const int j = lane & 15;
    const int64_t qw = (int64_t)qt * QTILE + wave * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t qo = qw + 4 * g + r;
        if (qo < a.B) {
            float* orow = out_tile + (wave * 16 + 4 * g + r) * VAL_DIM + 4 * j;
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                f32x4 o = {acc[4 * T + 0][r], acc[4 * T + 1][r], acc[4 * T + 2][r], acc[4 * T + 3][r]};
                *reinterpret_cast<f32x4*>(orow + 64 * T) = o;
            }
        }
    }
    }   // segments
#ifdef RANGE_EXP_P2_STAMPS
    st_c[3] = stamp(); st_r[3] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st_c[4] = stamp(); st_r[4] = __builtin_amdgcn_s_memrealtime();
    if (a.diag && threadIdx.x == 0) {
        unsigned long long* d = a.diag + (size_t)blockIdx.x * 16;
        for (int i = 0; i < 5; ++i) { d[i] = st_c[i]; d[5 + i] = st_r[i]; }
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[10] = hw; d[11] = xcc & 0xf; d[12] = (unsigned long long)(a.sk_groups > 0 ? ((int64_t)a.n_qtiles * a.n_blocks) / a.sk_groups : a.n_blocks / a.n_splits);
    }
#endif
}


// the parts of pass 2 (SlabMap) -> (B, 1024) f32, fixed summation order (ascending bank blocks).
__global__ void reduce_parts_kernel(const float* parts, SlabMap m, int64_t B, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (VAL_DIM / 4)) return;
    const int64_t q = i / (VAL_DIM / 4);
    const int c4 = (int)(i - q * (VAL_DIM / 4));
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int col = 0; col < slab_cols(m); ++col) {
        int64_t first4, stride4;
        int n;
        slab_parts(m, B, q, col, first4, stride4, n);
        const f32x4* p4 = reinterpret_cast<const f32x4*>(parts) + first4 + c4;
        if (col == 0) s = p4[0];
#pragma unroll 4   // loads of 4 parts in flight; the additions keep their order
        for (int p = col == 0 ? 1 : 0; p < n; ++p) s += p4[(int64_t)p * stride4];
    }
    reinterpret_cast<f32x4*>(out)[i] = s;
}

// out = (1-beta)*G + beta*H, elementwise f32, with the reference's rounding (two products, one
// sum, no FMA contraction): range/range.py:238.  Used by the beta sweep, where H (beta=1) and G
// (beta=0) are computed once and blended for every beta.
__global__ void blend_kernel(const float* G, const float* H, float beta, int64_t n4, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 g = reinterpret_cast<const f32x4*>(G)[i];
    const f32x4 h = reinterpret_cast<const f32x4*>(H)[i];
    const float a = 1.0f - beta;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = __fadd_rn(__fmul_rn(a, g[k]), __fmul_rn(beta, h[k]));
    reinterpret_cast<f32x4*>(out)[i] = o;
}

// out (B,1280) f64 = [ sum_p partial_p (f32, widened) | ehat64 ]      (range/range.py:222, :240)
// for queries [q0, q0 + nq) of a batch of B (parts: (n_parts,B,1024), ehat64 / out: (B,..)).
__global__ void finalize_kernel(const float* parts, SlabMap m, const double* ehat64, int64_t B,
                                int64_t q0, int64_t nq, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over nq * 320 quads
    if (i >= nq * 320) return;
    const int64_t q = q0 + i / 320;
    const int c = (int)(i % 320);
    double* o = out + q * 1280 + 4 * c;
    if (c < 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int col = 0; col < slab_cols(m); ++col) {
            int64_t first4, stride4;
            int n;
            slab_parts(m, B, q, col, first4, stride4, n);
            const f32x4* p4 = reinterpret_cast<const f32x4*>(parts) + first4 + c;
            if (col == 0) s = p4[0];
#pragma unroll 8   // loads of 8 parts in flight; the additions keep their order
            for (int p = col == 0 ? 1 : 0; p < n; ++p) s += p4[(int64_t)p * stride4];
        }
        o[0] = (double)s.x; o[1] = (double)s.y; o[2] = (double)s.z; o[3] = (double)s.w;
    } else {
        const double* e = ehat64 + q * 256 + 4 * (c - 256);
        o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = e[3];
    }
}

}  // namespace range_hip

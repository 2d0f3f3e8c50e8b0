// Kernel B of the RANGE engine: streaming soft-attention over the embedding bank.
//
// Reference semantics (range/range.py:213-217, 231-238): for every query
//     H = softmax_N(tau_sem * e . K^T) @ V,   G = softmax_N(tau_geo * x . X^T) @ V,
//     M = (1-beta) * G + beta * H
// with N = ALL bank rows (dense soft attention, no top-k truncation).  The reference materialises
// the (B,N) matrices; here nothing of size B*N ever reaches HBM:
//
//   pass 1  scan_stats_kernel   per query running (max, sum-exp) of both logit rows
//   pass 2  attend_kernel       recomputes the logits, forms ONE combined weight
//                               w = beta*p_sem + (1-beta)*p_geo and accumulates w @ V once
//                               (3084 FLOP per (query,row) pair instead of 4614).
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
//   blockIdx is mapped so that the workgroups resident on one XCD stream the SAME split
//   (split % 8 == blockIdx % 8): the bank rows are fetched once per XCD L2, not once per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace range_hip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KEY_DIM = 256;
constexpr int VAL_DIM = 1024;
constexpr int QTILE = 64;        // queries per workgroup
constexpr int BLK = 16;          // bank rows per block
constexpr int MAX_TOPK = 16;
constexpr float NEG_BIG = -1.0e30f;

#define RANGE_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
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
    int32_t n_splits;      // multiple of 8
    float k_sem;           // tau_sem * log2(e)
    float k_geo;           // tau_geo * log2(e)
    float beta;
};

// blockIdx -> (split, query tile); blocks b and b+8 share an XCD, so each XCD gets splits
// {xcd, xcd+8, ...} and walks each split's query tiles consecutively.
__device__ __forceinline__ void decode_block(const ScanArgs& a, int& split, int& qt) {
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int idx = b >> 3;
    split = (idx / a.n_qtiles) * 8 + xcd;
    qt = idx % a.n_qtiles;
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
// MFMA: `s_nop 1` in front covers a VALU-written operand, and qk_block ends with mfma_fence()
// before any non-MFMA instruction may read the results.
__device__ __forceinline__ void mfma_v_first(f32x4& d, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_v(f32x4& d, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
// 8-pass MFMA result -> VALU read needs 11 wait states; 16 given.
__device__ __forceinline__ void mfma_fence(f32x4& x, f32x4& y, f32x4& z) {
    asm volatile("s_nop 15" : "+v"(x), "+v"(y), "+v"(z));
}

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
// identical for both operands.  Two accumulation chains hide the 40-cycle dependent-MFMA latency;
// the SAME summation order is used in both passes so pass 2 reproduces pass 1's logits bit for bit.
template <bool GEO>
__device__ __forceinline__ void qk_block(const float* kt, const float* xt, const QFrag& f, int lane,
                                         f32x4& s_sem, f32x4& s_geo) {
    const int g = lane >> 4;
    const int R = pi_row(lane & 15);
    const float* krow = kt + R * KEY_DIM;
    f32x4 a0, a1, ag = {0.f, 0.f, 0.f, 0.f};
    // asm statements are scheduling boundaries for hipcc, so the LDS reads stay where the source
    // puts them: one 16-byte K read (4 k-steps) ahead of the 4 MFMAs that hide its latency.
    f32x4 kn = *reinterpret_cast<const f32x4*>(krow + ((g ^ R) << 2));
    float xa = 0.f;
    if (GEO) xa = xt[R * 4 + g];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 ka = kn;
        if (s < 15) kn = *reinterpret_cast<const f32x4*>(krow + (((4 * (s + 1) + g) ^ R) << 2));
        if (s == 0) {
            mfma_v_first(a0, ka.x, f.q[s].x);
            mfma_v_first(a1, ka.y, f.q[s].y);
        } else {
            mfma_v(a0, ka.x, f.q[s].x);
            mfma_v(a1, ka.y, f.q[s].y);
        }
        mfma_v(a0, ka.z, f.q[s].z);
        mfma_v(a1, ka.w, f.q[s].w);
    }
    if (GEO) mfma_v_first(ag, xa, f.xq);
    mfma_fence(a0, a1, ag);
    s_sem = a0 + a1;
    s_geo = ag;
}

// ---- LDS-DMA (global_load_lds) by inline asm -------------------------------------------------
// The builtin form makes hipcc (ROCm 7.2) drain vmcnt(0) before the next LDS read because it
// cannot tell which LDS bytes the DMA writes; that would serialise the whole ring.  In asm the
// compiler neither counts nor waits for these operations: every wait on them below is a
// hand-counted s_waitcnt vmcnt(N) followed by a workgroup barrier.  M0 carries the wave-uniform
// LDS destination; it is written and restored inside the statement that uses it.
// sbase must be wave-uniform (SGPR pair), voff is the per-lane byte offset.
__device__ __forceinline__ void dma_b128(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dma_b32(const void* sbase, uint32_t voff, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_addr), "s"(sbase) : "memory");
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
#define RANGE_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory")


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
    float* kring = reinterpret_cast<float*>(smem);        // 3 x [16][256]
    float* xring = kring + 3 * BLK * KEY_DIM;             // 3 x [16][4]
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const uint32_t kring_lds = lds0, xring_lds = lds0 + 3 * BLK * KEY_DIM * 4;
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

    float m1 = NEG_BIG, l1 = 0.f, m2 = NEG_BIG, l2 = 0.f;
    TopK<TOPK ? MAX_TOPK : 1> tk;
    if (TOPK) tk.init();

    // ring of 3 K tiles, prefetch distance 2; 5 LDS-DMA operations per wave and tile.
    if (nb > 0) {
        issue_k_tile(a.keys, a.xyz4, (int64_t)b0 * BLK, kring_lds, xring_lds, wave, lane, swz);
        if (nb > 1)
            issue_k_tile(a.keys, a.xyz4, (int64_t)(b0 + 1) * BLK, kring_lds + KT_BYTES,
                         xring_lds + 256, wave, lane, swz);
    }
    int slot = 0;
    for (int t = 0; t < nb; ++t) {
        // tile t landed (mine: counted wait; everyone's: barrier).  The barrier also says every
        // wave is done with tile t-1, whose slot tile t+2 re-uses.
        if (t + 1 < nb) RANGE_WAIT_BARRIER(5); else RANGE_WAIT_BARRIER(0);
        if (t + 2 < nb) {
            const int s2 = slot >= 1 ? slot - 1 : 2;   // (slot + 2) % 3
            issue_k_tile(a.keys, a.xyz4, (int64_t)(b0 + t + 2) * BLK, kring_lds + s2 * KT_BYTES,
                         xring_lds + s2 * 256, wave, lane, swz);
        }
        f32x4 ss, sg;
        qk_block<GEO>(kring + slot * BLK * KEY_DIM, xring + slot * 64, f, lane, ss, sg);
        const int64_t row0 = (int64_t)(b0 + t) * BLK;
        float t1[4], t2[4];
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + pi_row(4 * g + r);
            ok[r] = row < a.n_valid;
            t1[r] = ok[r] ? ss[r] * a.k_sem : NEG_BIG;
            if (GEO) t2[r] = ok[r] ? sg[r] * a.k_geo : NEG_BIG;
            if (TOPK) { if (ok[r]) tk.push(ss[r], (int32_t)row); }
        }
        {
            const float mx = fmaxf(fmaxf(t1[0], t1[1]), fmaxf(t1[2], t1[3]));
            const float mn = fmaxf(m1, mx);
            float acc = l1 * __builtin_amdgcn_exp2f(m1 - mn);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += ok[r] ? __builtin_amdgcn_exp2f(t1[r] - mn) : 0.f;
            l1 = acc; m1 = mn;
        }
        if (GEO) {
            const float mx = fmaxf(fmaxf(t2[0], t2[1]), fmaxf(t2[2], t2[3]));
            const float mn = fmaxf(m2, mx);
            float acc = l2 * __builtin_amdgcn_exp2f(m2 - mn);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += ok[r] ? __builtin_amdgcn_exp2f(t2[r] - mn) : 0.f;
            l2 = acc; m2 = mn;
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
    // lanes j, j+16, j+32, j+48 hold disjoint row subsets of the same query
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        merge_ml(m1, l1, __shfl_xor(m1, off), __shfl_xor(l1, off));
        if (GEO) merge_ml(m2, l2, __shfl_xor(m2, off), __shfl_xor(l2, off));
    }
    if (!GEO) { m2 = 0.f; l2 = 1.f; }
    if (q < a.B) {
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
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int HB>
__device__ __forceinline__ void pv_half(const float* vslot, const f32x4& w, f32x4 (&acc)[64],
                                        int lane) {
    const int j = lane & 15, g = lane >> 4;
    const float* base = vslot + (2 * g) * VAL_DIM + 4 * j;
    float w0 = w[2 * HB], w1 = w[2 * HB + 1];
    asm volatile("s_nop 3" : "+v"(w0), "+v"(w1));   // VALU-written MFMA operands: settle once
    // lane (j,g) reads V[row 2g+rr][64T + 4j .. +3]: one ds_read_b128 feeds 4 accumulator tiles.
    // Reads for step T+1 sit in front of step T's 8 MFMAs (256 cycles of cover).
    f32x4 v0 = *reinterpret_cast<const f32x4*>(base);
    f32x4 v1 = *reinterpret_cast<const f32x4*>(base + VAL_DIM);
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        f32x4 n0 = v0, n1 = v1;
        if (T < 15) {
            n0 = *reinterpret_cast<const f32x4*>(base + 64 * (T + 1));
            n1 = *reinterpret_cast<const f32x4*>(base + VAL_DIM + 64 * (T + 1));
        }
        mfma_a(acc[4 * T + 0], w0, v0.x);
        mfma_a(acc[4 * T + 1], w0, v0.y);
        mfma_a(acc[4 * T + 2], w0, v0.z);
        mfma_a(acc[4 * T + 3], w0, v0.w);
        mfma_a(acc[4 * T + 0], w1, v1.x);
        mfma_a(acc[4 * T + 1], w1, v1.y);
        mfma_a(acc[4 * T + 2], w1, v1.z);
        mfma_a(acc[4 * T + 3], w1, v1.w);
        v0 = n0; v1 = n1;
    }
}

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

// LDS map (bytes): V ring 3 x 32 KB | K ring 2 x 16 KB | X ring 2 x 256 B  = 131,584 B
constexpr int ATTEND_LDS_BYTES = (3 * 8 * VAL_DIM + 2 * BLK * KEY_DIM + 2 * 64) * 4;
constexpr int SCAN_LDS_BYTES = (3 * BLK * KEY_DIM + 3 * 64) * 4;

template <bool GEO>
__global__ __launch_bounds__(256, 1) void attend_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* vring = reinterpret_cast<float*>(smem);          // 3 slots x [8][1024]
    float* kring = vring + 3 * 8 * VAL_DIM;                 // 2 slots x [16][256]
    float* xring = kring + 2 * BLK * KEY_DIM;               // 2 slots x [16][4]

    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const uint32_t vring_lds = lds0;
    const uint32_t kring_lds = lds0 + 3 * 8 * VAL_DIM * 4;
    const uint32_t xring_lds = kring_lds + 2 * BLK * KEY_DIM * 4;
    constexpr uint32_t VS_BYTES = 8 * VAL_DIM * 4, KT_BYTES = BLK * KEY_DIM * 4;

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

    // Vector-memory groups per wave: "even" half h=2t carries V(8) + K(4) + X(1) = 13 LDS-DMA
    // instructions, "odd" half carries V(8).  Group for half h is issued right after barrier h-2
    // into V slot h%3 (last read in half h-3) and K/X slot (h/2)&1 (last read in half h-4).
    if (nb > 0) {
        const int64_t r0 = (int64_t)b0 * BLK;
        issue_v_half(a.values, r0, vring_lds, wave, lane);
        issue_k_tile(a.keys, a.xyz4, r0, kring_lds, xring_lds, wave, lane, swz);
        issue_v_half(a.values, r0 + 8, vring_lds + VS_BYTES, wave, lane);
    }
    int vs = 0;   // V slot of half 2t
    for (int t = 0; t < nb; ++t) {
        const int64_t row0 = (int64_t)(b0 + t) * BLK;
        const bool more = t + 1 < nb;
        // ---- half 2t: logits + first 8 rows of w @ V
        RANGE_WAIT_BARRIER(8);
        const int vs1 = vs == 2 ? 0 : vs + 1;
        const int vs2 = vs1 == 2 ? 0 : vs1 + 1;
        if (more) {
            issue_v_half(a.values, row0 + BLK, vring_lds + vs2 * VS_BYTES, wave, lane);
            issue_k_tile(a.keys, a.xyz4, row0 + BLK, kring_lds + ((t + 1) & 1) * KT_BYTES,
                         xring_lds + ((t + 1) & 1) * 256, wave, lane, swz);
        }
        f32x4 ss, sg, w;
        qk_block<GEO>(kring + (t & 1) * BLK * KEY_DIM, xring + (t & 1) * 64, f, lane, ss, sg);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ok = row0 + pi_row(4 * g + r) < a.n_valid;
            float wr = ca * __builtin_amdgcn_exp2f(fmaf(ss[r], a.k_sem, -m1));
            if (GEO) wr = fmaf(cb, __builtin_amdgcn_exp2f(fmaf(sg[r], a.k_geo, -m2)), wr);
            w[r] = ok ? wr : 0.f;
        }
        pv_half<0>(vring + vs * 8 * VAL_DIM, w, acc, lane);
        // ---- half 2t+1: last 8 rows
        if (more) RANGE_WAIT_BARRIER(13); else RANGE_WAIT_BARRIER(0);
        if (more) issue_v_half(a.values, row0 + BLK + 8, vring_lds + vs * VS_BYTES, wave, lane);
        pv_half<1>(vring + vs1 * 8 * VAL_DIM, w, acc, lane);
        vs = vs2;
    }

    acc_fence(acc);
    // accumulator tile 4T+c, register r, lane (j,g)  ->  out[query 4g+r of this wave][64T + 4j + c]
    const int j = lane & 15;
    const int64_t qw = (int64_t)qt * QTILE + wave * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t qo = qw + 4 * g + r;
        if (qo < a.B) {
            float* orow = a.out + ((int64_t)split * a.B + qo) * VAL_DIM + 4 * j;
#pragma unroll
            for (int T = 0; T < 16; ++T) {
                f32x4 o = {acc[4 * T + 0][r], acc[4 * T + 1][r], acc[4 * T + 2][r], acc[4 * T + 3][r]};
                *reinterpret_cast<f32x4*>(orow + 64 * T) = o;
            }
        }
    }
}

// (n_parts, B, 1024) f32 -> (B, 1024) f32, fixed summation order.
__global__ void reduce_parts_kernel(const float* parts, int n_parts, int64_t total4, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    f32x4 s = reinterpret_cast<const f32x4*>(parts)[i];
    for (int p = 1; p < n_parts; ++p) s += reinterpret_cast<const f32x4*>(parts)[(int64_t)p * total4 + i];
    reinterpret_cast<f32x4*>(out)[i] = s;
}

// out (B,1280) f64 = [ sum_p partial_p (f32, widened) | ehat64 ]      (range/range.py:222, :240)
__global__ void finalize_kernel(const float* parts, int n_parts, const double* ehat64, int64_t B,
                                double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * 320 quads
    if (i >= B * 320) return;
    const int64_t q = i / 320;
    const int c = (int)(i % 320);
    double* o = out + q * 1280 + 4 * c;
    if (c < 256) {
        const int64_t off = q * 256 + c;
        const int64_t stride = B * 256;
        f32x4 s = reinterpret_cast<const f32x4*>(parts)[off];
        for (int p = 1; p < n_parts; ++p) s += reinterpret_cast<const f32x4*>(parts)[(int64_t)p * stride + off];
        o[0] = (double)s.x; o[1] = (double)s.y; o[2] = (double)s.z; o[3] = (double)s.w;
    } else {
        const double* e = ehat64 + q * 256 + 4 * (c - 256);
        o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = e[3];
    }
}

}  // namespace range_hip

// Opt-in reduced-precision pass 2: w @ V on the bf16 matrix cores with both operands split into
// three bf16 planes (x = x_h + x_m + x_l, 8 significant bits each = the 24 of a float32) and the
// six largest of the nine cross products accumulated in float32:
//     w.V ~= w_h V_h + (w_h V_m + w_m V_h) + (w_h V_l + w_m V_m + w_l V_h)
// The dropped terms (w_m V_l, w_l V_m, w_l V_l) are below 2^-24 |w V|; every kept product is
// exact in float32 (8 x 8 bits).  v_mfma_f32_16x16x32_bf16 does 16 x the FLOP per cycle of the
// exact v_mfma_f32_16x16x4_f32, so the six terms cost 6/16 of the exact kernel's MFMA time.
//
// NEVER the default: load_model(..., pv_mode="bf16x3") / range_set_pv_mode.  The default pass 2
// (attend_stored_kernel) multiplies exact float32 products like the reference (range/range.py:217,
// :236); this one reproduces them to ~2^-22 relative (measured against the float64 oracle in
// tests/test_gpu_bf16x3.py and profiles/).
//
// Layout.  The planes of V are prepared once per bank (vplanes_kernel) in MFMA fragment order:
//   group R = bank rows 32R .. 32R+31 (one MFMA K step), piece P = 256 output columns,
//   column tile ct (16 columns), plane p, lane (n, g), 8 bf16:
//   byte address ((((R*4 + P)*16 + ct)*3 + p)*64 + lane)*16
// so a B operand is one coalesced 1 KB load per wave.  Element i of lane (n, g) is row
// 32R + (i < 4 ? pi_row(4g+i) : 16 + pi_row(4g+i-4)) - the order in which a lane of pass 1 holds
// its logits (attend_kernels.h: pi_row) - and column 64*(t>>2) + 4n + (t&3) of tile t = 16P + ct.
// The weights come straight from the lane's own kept logits.
//
// Work decomposition: workgroup = 4 waves = 64 queries; wave w owns the 256 COLUMNS of piece w for
// all 64 queries: 4 query tiles x 16 column tiles = 64 accumulator tiles = 256 AGPRs.  A B fragment
// is used by exactly one wave, so it goes from global memory straight to that wave's registers (a
// ring of 8 column tiles = 24 KB per wave in flight) - no LDS staging of V - and only the A
// operands cross waves: each wave forms the weights of its own 16 queries and publishes the three
// planes (3 KB) through a double-buffered LDS area that all four waves read at the start of a
// group, behind the one barrier per 32 rows.
//
// History: the first version kept attend_stored_kernel's decomposition - wave =
// 16 queries x 1024 columns, V pieces through a 3-slot LDS ring by LDS-DMA - and ran in 8.8 ms:
// every wave read every piece from LDS (128 B/clk of operand reads next to the DMA writes); with
// a quarter of those reads (timing experiment) it took 7.0 ms.  This tiling: 7.9 ms.
#pragma once
#include "attend_kernels.h"

namespace range_hip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int PVB_PIECE_BYTES = 16 * 3 * 1024;                 // 16 column tiles x 3 planes x 1 KB: a wave's share of a group
constexpr int PVB_GROUP_BYTES = 4 * PVB_PIECE_BYTES;           // 32 rows x 1024 columns x 6 B

// round-to-nearest-even float32 -> bf16 of two values, packed (lo = a, hi = b)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// the three bf16 planes of a pair of floats: hi/mid/lo packed like cvt_pk_bf16
__device__ __forceinline__ void split3(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xFFFF0000u);
    m = cvt_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xFFFF0000u);
    l = cvt_pk_bf16(sa, sb);
}

// bank row (inside its 32-row group) of element i of lane group g
__device__ __forceinline__ int pvb_row(int g, int i) {
    return i < 4 ? pi_row(4 * g + i) : 16 + pi_row(4 * g + i - 4);
}

// values (n_alloc rows x 1024 f32, row-major) -> planes in fragment order; rows >= n_alloc are 0.
// One thread per (group, tile, lane): 8 strided reads, three 16-byte writes.  One-time cost.
__global__ __launch_bounds__(256) void vplanes_kernel(const float* __restrict__ values, int64_t n_alloc,
                                                      int64_t n_groups, u32x4* __restrict__ planes) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_groups * 64 * 64) return;
    const int lane = (int)(id & 63);
    const int t = (int)((id >> 6) & 63);
    const int64_t R = id >> 12;
    const int n = lane & 15, g = lane >> 4;
    const int col = 64 * (t >> 2) + 4 * n + (t & 3);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t row = R * 32 + pvb_row(g, i);
        v[i] = row < n_alloc ? values[row * VAL_DIM + col] : 0.f;
    }
    u32x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t hh, mm, ll;
        split3(v[2 * i], v[2 * i + 1], hh, mm, ll);
        h[i] = hh; m[i] = mm; l[i] = ll;
    }
    u32x4* dst = planes + (((R * 4 + (t >> 4)) * 16 + (t & 15)) * 3) * 64 + lane;
    dst[0] = h;
    dst[64] = m;
    dst[128] = l;
}

struct PvbA {            // the A operand of one group: 8 weights per lane in three planes
    u32x4 h, m, l;
};
struct PvbB {            // the B operands of one column tile
    u32x4 h, m, l;
};

// acc += A (16 queries x 32 rows) . B (32 rows x 16 columns), accumulator pinned to its AGPRs
// (asm for the reason given at mfma_a: the builtin lets hipcc move accumulator tiles between the
// two halves of the register file inside the loop - 300 v_accvgpr moves per group here).
// No wait state follows the MFMA (a 16-cycle MFMA leaves 8 issue cycles, s_nop 1 cost 10 % of the
// kernel): hipcc may hand a dead operand register to the very next vector instruction while the
// MFMA still reads it, so (i) the kernel puts no vector ALU instruction directly behind an MFMA -
// every gap starts with an LDS read, an LDS-DMA or a scalar instruction - and (ii)
// tools/check_mfma_war.py checks the generated code for exactly that (tests/test_host_cpu.py).
__device__ __forceinline__ void mfma_bf16_a(f32x4& acc, const u32x4& a, const u32x4& b) {
#ifdef RANGE_EXP_PVB_NOP
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 1" : "+a"(acc) : "v"(a), "v"(b));
#else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#endif
}
#ifndef RANGE_PVB_PF
#define RANGE_PVB_PF 8
#endif
constexpr int PVB2_PF = RANGE_PVB_PF;                        // ring slots of B per wave (a power of two <= 8)
constexpr int PVB2_D = PVB2_PF - 1;                          // column tiles of B in flight per wave
constexpr int PVB2_LDS_BYTES = 2 * 4 * 3 * 1024;             // A planes: 2 buffers x 4 query tiles x 3 planes

template <bool GEO>
__global__ __launch_bounds__(256, 1) void attend_bf16x3_kernel(ScanArgs a, const char* __restrict__ vplanes,
                                                                 int32_t n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    int split, qt;
    decode_block(a, split, qt);
    const int g0 = (int)(((int64_t)split * n_groups) / a.n_splits);
    const int g1 = (int)(((int64_t)(split + 1) * n_groups) / a.n_splits);
    const int nG = g1 - g0;
    const int64_t q = (int64_t)qt * QTILE + wave * 16 + (lane & 15);
    const int64_t qtile_kept = (int64_t)qt + a.qt_offset;

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
    f32x4 acc[64];                 // [query tile m][column tile ct] at 16 m + ct
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    int prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = pi_row(4 * g + r);
    const int b_last = a.n_blocks - 1;
    const int xrow = pi_row(lane & 15) * 4 + g;

    // this wave's B operands: piece `wave` of every group, 48 KB contiguous, tile ct plane p at
    // (ct*3+p) KB + 16 B per lane
    // (addresses: scalar base + the lane's constant byte offset - no vector ALU work per load, which
    // hipcc would place right behind an MFMA, into that MFMA's dying operand registers)
    const char* vb = vplanes + ((int64_t)g0 * 4 + wave) * PVB_PIECE_BYTES;
    const uint32_t voff16 = (uint32_t)(lane * 16), voffx = (uint32_t)(xrow * 4);
    // Every vector-memory load of the loop is an asm statement with a hand-counted wait in front of
    // its first use: hipcc's own counting does not follow a register ring across the loop's back
    // edge - it waited for ALL loads of the previous group at the top of every group (8.07 ms).
    // The loaded registers are touched by nothing but the asm MFMAs / the weights code behind
    // their wait.  Loads retire in issue order: vmcnt(N) = "all but the N youngest are done".
    // (the destination is written by the asm statements themselves: a returned temporary would be
    // COPIED into the ring by the compiler, i.e. read before the load has landed)
    // A vector-memory instruction must not read an SGPR the scalar ALU wrote less than five wait
    // states ago, and hipcc does not see the instruction inside an asm statement: addr_b() is
    // called in FRONT of an accumulation chain (the empty asm makes hipcc put the address into its
    // SGPRs there), load_b1() behind it.  (`early` = false: s_nop 4 instead, prologue only.)
    auto addr_b = [&](int gi, int ct, int plane) __attribute__((always_inline)) {
        const char* p = vb + (int64_t)min(gi, nG - 1) * PVB_GROUP_BYTES + ct * 3072 + plane * 1024;
        asm volatile("" : "+s"(p));
        return p;
    };
    auto load_b1 = [&](const char* p, u32x4& dst) __attribute__((always_inline)) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff16), "s"(p));
    };
    auto load_b = [&](int gi, int ct, PvbB& b) __attribute__((always_inline)) {
        const char* p0 = addr_b(gi, ct, 0);
        const char* p1 = addr_b(gi, ct, 1);
        const char* p2 = addr_b(gi, ct, 2);
        asm volatile("s_nop 4");
        load_b1(p0, b.h);
        load_b1(p1, b.m);
        load_b1(p2, b.l);
    };
    // kept logits (this wave's tile) and xyz of bank block b (two loads, also when there is no geo head:
    // the counts below do not depend on it)
    auto load_sx = [&](int b, f32x4& sv, float& xa) __attribute__((always_inline)) {
        const float* ps = a.logits + logit_tile(qtile_kept, a.n_blocks, min(b, b_last), wave);
        const float* px = a.xyz4 + (int64_t)min(b, b_last) * BLK * 4;
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(sv) : "v"(voff16), "s"(ps));
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(xa) : "v"(voffx), "s"(px));
    };
    // weights of one block from its logits -> two packed words per plane
    auto weights = [&](int b, const f32x4& sv, float xa, uint32_t (&h)[2], uint32_t (&m)[2], uint32_t (&l)[2]) __attribute__((always_inline)) {
        f32x4 cg = {0.f, 0.f, 0.f, 0.f};
        if (GEO) {
            mfma_v_first(cg, xa, fxq);
            asm volatile("s_nop 15" : "+v"(cg));
        }
        const int n_left = (int)(a.n_valid - (int64_t)b * BLK);
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float wr = ca * __builtin_amdgcn_exp2f(fmaf(sv[r], a.k_sem, -m1));
            if (GEO) wr = fmaf(cb, __builtin_amdgcn_exp2f(fmaf(cg[r], a.k_geo, -m2)), wr);
            w[r] = prow[r] < n_left ? wr : 0.f;
        }
        split3(w[0], w[1], h[0], m[0], l[0]);
        split3(w[2], w[3], h[1], m[1], l[1]);
    };
    // publish two words (one bank block) of this wave's planes of a group - the A operand of query
    // tile `wave` - into exchange buffer bf: words 2*half, 2*half+1 of each plane
    auto publish = [&](int bf, int half, const uint32_t (&h)[2], const uint32_t (&m)[2], const uint32_t (&l)[2]) __attribute__((always_inline)) {
        char* o = smem + ((bf * 4 + wave) * 3) * 1024 + lane * 16 + half * 8;
        *reinterpret_cast<uint2*>(o) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2*>(o + 1024) = make_uint2(m[0], m[1]);
        *reinterpret_cast<uint2*>(o + 2048) = make_uint2(l[0], l[1]);
    };

    if (nG > 0) {
        // group 0's weights
        const int b0 = 2 * g0;
        uint32_t h[2], m[2], l[2];
        f32x4 sv;
        float xa;
        load_sx(b0, sv, xa);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv), "+v"(xa));
        weights(b0, sv, xa, h, m, l);
        publish(0, 0, h, m, l);
        load_sx(b0 + 1, sv, xa);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sv), "+v"(xa));
        weights(b0 + 1, sv, xa, h, m, l);
        publish(0, 1, h, m, l);
    }
    // B ring of PVB2_PF slots: a tile is requested PVB2_D = PVB2_PF - 1 tiles ahead, into the slot
    // of the tile just consumed.  The first PVB2_D column tiles of group 0:
    PvbB bq[PVB2_PF];
#pragma unroll
    for (int i = 0; i < PVB2_D; ++i) load_b(0, i, bq[i]);
    bq[PVB2_D].h = bq[PVB2_D].m = bq[PVB2_D].l = u32x4{0u, 0u, 0u, 0u};

    for (int gi = 0; gi < nG; ++gi) {
        __syncthreads();                                     // every wave's planes of group gi are published
        PvbA am[4];
#pragma unroll
        for (int mq = 0; mq < 4; ++mq) {
            const char* o = smem + (((gi & 1) * 4 + mq) * 3) * 1024 + lane * 16;
            am[mq].h = *reinterpret_cast<const u32x4*>(o);
            am[mq].m = *reinterpret_cast<const u32x4*>(o + 1024);
            am[mq].l = *reinterpret_cast<const u32x4*>(o + 2048);
        }
        // the next group's logits and xyz (two blocks): four loads here, used from column tile 4 on
        const int bn = 2 * (g0 + gi + 1);
        f32x4 svA, svB;
        float xaA, xaB;
        load_sx(bn, svA, xaA);
        load_sx(bn + 1, svB, xaB);
        // The weights of the next group are formed in the gaps between the accumulation chains, two
        // statements per gap (wop below): block A during column tiles 4-9, block B during 10-15.
        f32x4 cg = {0.f, 0.f, 0.f, 0.f};
        float e1[4], e2[4], ww[4], ra = 0.f, rb = 0.f;
        uint32_t hh[2] = {0u, 0u}, mm[2] = {0u, 0u}, ll[2] = {0u, 0u};
#pragma unroll
        for (int r = 0; r < 4; ++r) e1[r] = e2[r] = ww[r] = 0.f;
        auto wop = [&](int half, int k) __attribute__((always_inline)) {
#ifndef RANGE_EXP_PVB_NOW
            const f32x4& sv = half ? svB : svA;
            const float xa = half ? xaB : xaA;
            const int n_left = (int)(a.n_valid - (int64_t)(bn + half) * BLK);
            if (k == 0) { if (GEO) mfma_v_first(cg, xa, fxq); }
            else if (k >= 1 && k < 5) e1[k - 1] = fmaf(sv[k - 1], a.k_sem, -m1);
            else if (k >= 5 && k < 9) e1[k - 5] = __builtin_amdgcn_exp2f(e1[k - 5]);
            else if (k >= 10 && k < 14) { if (GEO) e2[k - 10] = fmaf(cg[k - 10], a.k_geo, -m2); }
            else if (k >= 14 && k < 18) { if (GEO) e2[k - 14] = __builtin_amdgcn_exp2f(e2[k - 14]); }
            else if (k >= 18 && k < 22) ww[k - 18] = ca * e1[k - 18];
            else if (k >= 22 && k < 26) { if (GEO) ww[k - 22] = fmaf(cb, e2[k - 22], ww[k - 22]); }
            else if (k == 26) {
                if (n_left < BLK) {              // pad rows exist only in the bank's last block
#pragma unroll
                    for (int r = 0; r < 4; ++r) ww[r] = prow[r] < n_left ? ww[r] : 0.f;
                }
            }
            else if (k == 27 || k == 34) hh[k == 34] = cvt_pk_bf16(ww[2 * (k == 34)], ww[2 * (k == 34) + 1]);
            else if (k == 28 || k == 35) ra = ww[2 * (k == 35)] - __uint_as_float(hh[k == 35] << 16);
            else if (k == 29 || k == 36) rb = ww[2 * (k == 36) + 1] - __uint_as_float(hh[k == 36] & 0xFFFF0000u);
            else if (k == 30 || k == 37) mm[k == 37] = cvt_pk_bf16(ra, rb);
            else if (k == 31 || k == 38) ra = ra - __uint_as_float(mm[k == 38] << 16);
            else if (k == 32 || k == 39) rb = rb - __uint_as_float(mm[k == 39] & 0xFFFF0000u);
            else if (k == 33 || k == 40) ll[k == 40] = cvt_pk_bf16(ra, rb);
            else if (k == 41) publish((gi + 1) & 1, half, hh, mm, ll);
#endif
        };
#pragma unroll
        for (int ct = 0; ct < 16; ++ct) {
            // Tile ct's three loads (requested during tile ct - PVB2_D, one behind each of its first
            // three accumulation chains) are done when only the younger loads are outstanding: those
            // of the PVB2_D - 1 tiles since, plus the four S / X loads of this group's start if the
            // tile was requested in the previous group (ct < PVB2_D).
            PvbB& b = bq[ct & (PVB2_PF - 1)];
            PvbB& nb = bq[(ct + PVB2_D) & (PVB2_PF - 1)];      // the slot of tile ct - 1: free
            constexpr int RING = 3 * (PVB2_D - 1);
            if (ct < PVB2_D) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b.h), "+v"(b.m), "+v"(b.l) : "n"(RING + 4));
            else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b.h), "+v"(b.m), "+v"(b.l) : "n"(RING));
            // the S / X loads of this group's start: younger are the loads of column tiles 0-3
            if (ct == 4) asm volatile("s_waitcnt vmcnt(12)" : "+v"(svA), "+v"(xaA), "+v"(svB), "+v"(xaB));
            // one chain of six per accumulator (interleaving the four query tiles term by term was
            // slower, 8.5 ms against 8.0); one load of the tile PVB2_D ahead behind each of the first
            // three chains (three loads back to back behind the tile measured the same, and so did a
            // ring of 4 instead of 8 slots: the loads cost 1.2 ms of the kernel - timing experiment
            // without them - but neither their latency nor their clustering is what costs it)
#pragma unroll
            for (int mq = 0; mq < 4; ++mq) {
                f32x4& c = acc[16 * mq + ct];
#ifndef RANGE_EXP_PVB_NODMA
                const char* pn = nullptr;
                if (mq < 3) pn = ct + PVB2_D < 16 ? addr_b(gi, ct + PVB2_D, mq) : addr_b(gi + 1, ct + PVB2_D - 16, mq);
#endif
                mfma_bf16_a(c, am[mq].l, b.h);
                mfma_bf16_a(c, am[mq].m, b.m);
                mfma_bf16_a(c, am[mq].h, b.l);
                mfma_bf16_a(c, am[mq].m, b.h);
                mfma_bf16_a(c, am[mq].h, b.m);
                mfma_bf16_a(c, am[mq].h, b.h);
#ifndef RANGE_EXP_PVB_NODMA
                if (mq < 3) load_b1(pn, mq == 0 ? nb.h : mq == 1 ? nb.m : nb.l);
                else asm volatile("s_nop 0");                 // (no vector ALU directly behind an MFMA)
#else
                asm volatile("s_nop 0");
#endif
                if (ct >= 4) {                               // two statements of the next group's weights
                    const int half = ct >= 10, k = 8 * (ct - (half ? 10 : 4)) + 2 * mq;
                    __builtin_amdgcn_sched_barrier(0);
                    wop(half, k);
                    wop(half, k + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        asm volatile("s_nop 1");
    }
    // The clamped prefetches past the end are still in flight INTO the ring's registers: nothing
    // else may get those registers before they have landed (hipcc would hand them to the
    // epilogue's accumulator reads at once - the wait carries them as operands).
#pragma unroll
    for (int i = 0; i < PVB2_PF; ++i)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[i].h), "+v"(bq[i].m), "+v"(bq[i].l));
    // wait states between the last MFMAs and the readers of the accumulators (see above)
#pragma unroll
    for (int mq = 0; mq < 4; ++mq) asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[16 * mq + 15]));
    acc_fence(acc);
    // accumulator [m][ct], register r, lane (j,g) -> out[query 16 m + 4 g + r][64 (4 wave + ct/4) + 4 j + ct%4]
    const int j = lane & 15;
#pragma unroll
    for (int mq = 0; mq < 4; ++mq) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t qo = (int64_t)qt * QTILE + 16 * mq + 4 * g + r;
            if (qo < a.B) {
                float* orow = a.out + ((int64_t)split * a.B + qo) * VAL_DIM + 256 * wave + 4 * j;
#pragma unroll
                for (int T = 0; T < 4; ++T) {
                    f32x4 o = {acc[16 * mq + 4 * T + 0][r], acc[16 * mq + 4 * T + 1][r],
                               acc[16 * mq + 4 * T + 2][r], acc[16 * mq + 4 * T + 3][r]};
                    *reinterpret_cast<f32x4*>(orow + 64 * T) = o;
                }
            }
        }
    }
}

}  // namespace range_hip

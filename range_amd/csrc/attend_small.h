// The whole retrieval of a SMALL batch (up to 32 queries) in ONE pass over the bank.
//
// The two-pass kernels of attend_kernels.h are built for the FP32-MFMA-bound regime (64 queries per
// workgroup, hundreds of workgroups per bank pass).  A handful of queries is the opposite regime:
// the 513 MB of the bank (keys, locations, values: range/range.py:85-95) are what costs, and two
// passes over the keys plus a pass over the values with one busy wave per CU took 0.15 ms for 16
// queries.  Here every CU streams its share of the bank ONCE:
//
//   * no softmax statistics are needed in advance: the logits of unit vectors are bounded, the
//     exponent shift is the constant m = tau * log2(e) (attend_kernels.h, pass 1), so a workgroup
//     accumulates the UN-NORMALISED products  O_h = sum_n 2^(t_n - m) V_n  and  Z_h = sum_n 2^(t_n - m)
//     of both heads (h = semantic, geographic: range.py:213-217, :231-236) over its rows, and
//     small_finalize_kernel sums the workgroups' partials in a fixed order, divides and blends
//     (range.py:238).  Two products instead of pass 2's one combined weight - the extra MFMAs are
//     free while HBM-bound, and it is the reference's own arithmetic (two P @ V products).
//   * workgroup = 4 waves over the SAME NQ x 16 queries (NQ = 1, or 2 for 17..32 queries): wave w
//     owns output columns [256 w, 256 w + 256) of both heads (32 accumulator tiles = 128 registers
//     per query tile) and the k-slice [64 w, 64 w + 64) of the logits: 16 MFMAs per 16-row block,
//     query tile and wave instead of 64; the four partial logit tiles meet in LDS (one workgroup
//     barrier per block - the only one) and are summed in a fixed order.  With two query tiles
//     the kernel is at the MFMA / HBM ridge (290 MFMAs per block and wave = 3.9 us against 3.3 us
//     of bank): 32 queries take the time of 16 and a little.
//   * every wave streams ONLY what it uses, through wave-private LDS rings filled by LDS-DMA with
//     hand-counted vmcnt waits: its 256-byte slice of the key rows (2 slots), its 1 KB slice of the
//     value rows in 8-row halves (3 slots), the 16 locations.  Nothing else crosses waves, so the
//     rings need no barrier; ~20 KB per wave (80 KB per CU) are in flight all the time.
//   * MFMAs are v_mfma_f32_16x16x4_f32 (exact float32 products), compiler builtins where hipcc can
//     be left to place them.  What bounds the kernel (round 3, profiles/NOTES.md A.2): the 16-query
//     loop streams its 513 MB in 80 us = 6.4 TB/s, the rate a plain copy reaches on this chip - it
//     is HBM-bound, and timing builds without the exchange, without the logit MFMAs, without the
//     LDS reads of V (-DRANGE_EXP_AS_NOEXCH / _NOLOGITS / _NOVREAD, results invalid) take the same
//     time to the microsecond.  The waves do not wait in their vmcnt waits (1 us of 80) but at the
//     ISSUE of the next LDS-DMA request, which blocks while the memory pipeline is full - stamps
//     around the waits alone read as "compute-bound".  The rest of the kernel's 93-99 us: ~6 us
//     before the first data, ~5 us to write the 33 MB of partials, the spread of the workgroups'
//     ends.  With two query tiles (290 MFMAs per block and wave: 100 us of matrix-core time against
//     80 us of bank) the two do not overlap perfectly - an MFMA behind a blocked request waits
//     with it: 131 us.
//
// Logit tile transposed (bank row on the MFMA row index, query on the lane) with the row
// permutation pi_row, as everywhere: accumulator registers 0,1 of the logit tile are rows of the
// block's first half, 2,3 of the second, and are directly the A operand of the P @ V MFMAs.
#pragma once
#include "attend_kernels.h"

namespace range_hip {

struct SmallArgs {
    const float* keys;     // (n_pad,256)
    const float* xyz4;     // (n_pad,4)
    const float* values;   // (n_pad,1024)
    const float* ehat;     // (B,256)
    const float* xq;       // (B,4)
    float* osum;           // (n_wg, 2 heads, 16 NQ queries, 1024): un-normalised partial products
    float* zsum;           // (n_wg, 16 NQ queries, 2 heads): partial sums of the weights
    int64_t B;             // 1..16 NQ
    int64_t n_valid;
    int32_t n_blocks;
    float k_sem, k_geo;    // tau * log2(e); k_geo = 0: no geographic head (plain RANGE)
};

constexpr int AS_VSLOTS = 3;
constexpr int AS_V_BYTES = 8 * 1024;                       // 8 rows x this wave's 256 columns
constexpr int AS_K_BYTES = 16 * 256;                       // 16 rows x this wave's 64 dims
constexpr int AS_X_BYTES = 256;                            // 16 rows x (x, y, z, 0)
constexpr int AS_OFF_K = AS_VSLOTS * AS_V_BYTES;
constexpr int AS_OFF_X = AS_OFF_K + 2 * AS_K_BYTES;
constexpr int AS_WAVE_LDS = AS_OFF_X + 2 * AS_X_BYTES;     // 33 280 B per wave
constexpr int AS_OFF_XBUF = 4 * AS_WAVE_LDS;               // partial logit tiles: 2 buffers x NQ x 4 waves x 1 KB
constexpr int as_lds_bytes(int nq) { return AS_OFF_XBUF + 2 * nq * 4 * 1024; }

// the bank is streamed once per launch: non-temporal requests (measured, 16 queries against
// range_db_large: 99 -> 94 us per call; RANGE_EXP_AS_TEMPORAL restores the default policy)
#ifdef RANGE_EXP_AS_TEMPORAL
#define AS_DMA dma_b128
#else
#define AS_DMA dma_b128_nt
#endif
template <bool GEO, int NQ>
__global__ __launch_bounds__(256, 1) void attend_small_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    char* wl = smem + wave * AS_WAVE_LDS;
    const uint32_t wl_lds = (uint32_t)(uintptr_t)RANGE_LPTR(smem) + wave * AS_WAVE_LDS;
    f32x4* xbuf = reinterpret_cast<f32x4*>(smem + AS_OFF_XBUF);

    const int n_wg = (int)gridDim.x;
    const int b0 = (int)(((int64_t)a.n_blocks * blockIdx.x) / n_wg);
    const int b1 = (int)(((int64_t)a.n_blocks * (blockIdx.x + 1)) / n_wg);
    const int nb = b1 - b0;                                 // >= 1 (the host launches n_wg <= n_blocks)

    // ---- query operands (ordinary loads; pinned BEFORE the first LDS-DMA goes out: hipcc waits
    //      for them with vmcnt(0), which behind the ring's requests would wait for the ring)
    f32x4 qf[NQ][4];
    float xqv[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        const int64_t q = 16 * nq + j < a.B ? 16 * nq + j : a.B - 1;
        const f32x4* rowp = reinterpret_cast<const f32x4*>(a.ehat + q * KEY_DIM + 64 * wave);
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[nq][s] = rowp[4 * s + g];
        xqv[nq] = GEO ? a.xq[q * 4 + g] : 0.f;
    }
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[nq][s]));
        asm volatile("" : "+v"(xqv[nq]));
    }

    // ---- LDS-DMA of this wave's slices (sequence positions past the end re-fetch the last block:
    //      never consumed, they keep every wait below a constant)
    // key slice: LDS slot (16 B) p = R * 16 + (c ^ R) holds chunk c of row R - the swizzle makes the
    // A-fragment reads below bank-conflict free; instruction i fills slots 64 i .. 64 i + 63
    uint32_t kvoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = 4 * i + g;
        kvoff[i] = (uint32_t)(R * 1024 + 256 * wave + ((j ^ R) << 4));
    }
    auto issue_kx = [&](int t) __attribute__((always_inline)) {
        const int tt = t < nb ? t : nb - 1;
        const int slot = t & 1;
        const float* ksrc = a.keys + (int64_t)(b0 + tt) * BLK * KEY_DIM;
#pragma unroll
        for (int i = 0; i < 4; ++i) AS_DMA(ksrc, kvoff[i], wl_lds + AS_OFF_K + slot * AS_K_BYTES + i * 1024);
        if (GEO) dma_b32(a.xyz4 + (int64_t)(b0 + tt) * BLK * 4, (uint32_t)(lane << 2), wl_lds + AS_OFF_X + slot * AS_X_BYTES);
    };
    const uint32_t vvoff = (uint32_t)(wave * 1024 + (lane << 4));
    auto issue_v = [&](int v) __attribute__((always_inline)) {       // v = 2 t + half
        const int t = v >> 1, h = v & 1;
        const int tt = t < nb ? t : nb - 1;
        const uint32_t dst = wl_lds + (uint32_t)(v % AS_VSLOTS) * AS_V_BYTES;
        const float* vsrc = a.values + ((int64_t)(b0 + tt) * BLK + 8 * h) * VAL_DIM;
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) AS_DMA(vsrc + r8 * VAL_DIM, vvoff, dst + r8 * 1024);
    };
    // one row of the same request (RANGE_AS_SPREAD: the eight requests of a half go out between its
    // MFMAs, one per 16, instead of in a burst in front of them - a burst into a full memory
    // pipeline blocks the wave, and the matrix cores with it, until there is room for all eight)
    auto issue_v_row = [&](int v, int r8) __attribute__((always_inline)) {
        const int t = v >> 1, h = v & 1;
        const int tt = t < nb ? t : nb - 1;
        const uint32_t dst = wl_lds + (uint32_t)(v % AS_VSLOTS) * AS_V_BYTES;
        const float* vsrc = a.values + ((int64_t)(b0 + tt) * BLK + 8 * h) * VAL_DIM;
        AS_DMA(vsrc + r8 * VAL_DIM, vvoff, dst + r8 * 1024);
    };
#ifdef RANGE_EXP_AS_BURST
    constexpr bool SPREAD = false;
#else
    constexpr bool SPREAD = NQ * (GEO ? 2 : 1) > 2;     // (the asm-MFMA kernel: program order is issue order)
#endif
    constexpr int OPS_KX = GEO ? 5 : 4, OPS_V = 8;
    // the order of the steady state: [K X (t+2)] after the logits of t, [V (t+1, 1)] after the first
    // half of t, [V (t+2, 0)] after its second half
    issue_kx(0);
    issue_v(0);
    issue_kx(1);
    issue_v(1);
    issue_v(2);

    // LDS read offsets: A fragment of the logits, lane (m = j, kg = g): row R = pi_row(j), chunk 4 s + g
    const int R = pi_row(j);
    uint32_t koff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koff[s] = (uint32_t)((R * 16 + ((4 * s + g) ^ R)) << 4);
    const uint32_t xoff = (uint32_t)((R * 4 + g) << 2);
    // B fragment of P @ V, lane (n = j, kg = g): V row 2 g + (r & 1) of the half, this lane's 4 columns
    const uint32_t voff_r0 = (uint32_t)((2 * g) * 1024 + (j << 4));
    uint32_t prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = (uint32_t)pi_row(4 * g + r);

    f32x4 acc[NQ][2][16];
    float z1[NQ], z2[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
        z1[nq] = 0.f; z2[nq] = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nq][h][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float nm1 = -a.k_sem, nm2 = -a.k_geo;             // the constant shift m = tau * log2(e)

    for (int t = 0; t < nb; ++t) {
        // ---- logits of block t: this wave's k-slice (16 MFMAs), the geographic tile (1)
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"(OPS_KX + 2 * OPS_V) : "memory");   // K X (t), V (t, 0) landed
        const char* kt = wl + AS_OFF_K + (t & 1) * AS_K_BYTES;
        f32x4 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const f32x4*>(kt + koff[s]);
        float xa = 0.f;
        if (GEO) xa = *reinterpret_cast<const float*>(wl + AS_OFF_X + (t & 1) * AS_X_BYTES + xoff);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(kf[s]));
        asm volatile("" : "+v"(xa));
        issue_kx(t + 2);                                   // (the slots just read are free: wave-private)
        f32x4 c[NQ], cg[NQ];
        if (NQ * (GEO ? 2 : 1) > 2) {
            // every accumulator register is taken by the products (256, pinned below): the logit
            // tiles are formed in VECTOR registers by the asm forms - with the builtin hipcc parked
            // four accumulator tiles in vector registers around this chain, reading the results of
            // MFMAs it does not know to be MFMAs without their wait states (checked on the
            // generated code: tests/test_host_cpu.py)
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s == 0) mfma_v_first(c[nq], kf[s].x, qf[nq][s].x);
                    else mfma_v(c[nq], kf[s].x, qf[nq][s].x);
                    mfma_v(c[nq], kf[s].y, qf[nq][s].y);
                    mfma_v(c[nq], kf[s].z, qf[nq][s].z);
                    mfma_v(c[nq], kf[s].w, qf[nq][s].w);
                }
                if (GEO) mfma_v_first(cg[nq], xa, xqv[nq]);
                else cg[nq] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // (the tiles are read by ordinary instructions next: an 8-pass MFMA result needs 11 wait
            // states before that, which hipcc cannot know about - 16 here, like QKAcc::fence)
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) asm volatile("s_nop 15" : "+v"(c[nq]), "+v"(cg[nq]));
        } else {
#pragma unroll
            for (int nq = 0; nq < NQ; ++nq) {
                c[nq] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef RANGE_EXP_AS_NOLOGITS
                c[nq] = kf[0] + kf[1] + kf[2] + kf[3];
#else
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    c[nq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].x, qf[nq][s].x, c[nq], 0, 0, 0);
                    c[nq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].y, qf[nq][s].y, c[nq], 0, 0, 0);
                    c[nq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].z, qf[nq][s].z, c[nq], 0, 0, 0);
                    c[nq] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].w, qf[nq][s].w, c[nq], 0, 0, 0);
                }
#endif
                cg[nq] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (GEO) cg[nq] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xqv[nq], cg[nq], 0, 0, 0);
            }
        }
        // ---- the four k-slices meet: partial tiles through LDS, summed in a fixed order
        f32x4* xb = xbuf + (t & 1) * (NQ * 256);
#ifndef RANGE_EXP_AS_NOEXCH      // (timing experiments: results invalid)
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) xb[(nq * 4 + wave) * 64 + lane] = c[nq];
#ifndef RANGE_EXP_AS_NOBAR
        __syncthreads();
#endif
#endif
        // ---- un-normalised weights of both heads (pad rows of the bank's last block: 0)
        const uint32_t row0 = (uint32_t)(b0 + t) * BLK;
        float p1[NQ][4], p2[NQ][4];
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
#ifdef RANGE_EXP_AS_NOEXCH
            f32x4 sv = c[nq];
#else
            f32x4 sv = xb[(nq * 4 + 0) * 64 + lane];
            sv += xb[(nq * 4 + 1) * 64 + lane];
            sv += xb[(nq * 4 + 2) * 64 + lane];
            sv += xb[(nq * 4 + 3) * 64 + lane];
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = row0 + prow[r] < (uint32_t)a.n_valid;
                const float e1 = __builtin_amdgcn_exp2f(fmaf(sv[r], a.k_sem, nm1));
                p1[nq][r] = ok ? e1 : 0.f;
                z1[nq] += p1[nq][r];
                if (GEO) {
                    const float e2 = __builtin_amdgcn_exp2f(fmaf(cg[nq][r], a.k_geo, nm2));
                    p2[nq][r] = ok ? e2 : 0.f;
                    z2[nq] += p2[nq][r];
                } else {
                    p2[nq][r] = 0.f;
                }
            }
        }
        // ---- P @ V, half by half: 2 rows per lane group and half, 4 column groups of 64, a
        //      16-byte LDS read = the B operands of 4 MFMAs (this lane's 4 consecutive columns)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(OPS_KX + 2 * OPS_V) : "memory");   // V (t, 1) landed
            const char* vt = wl + ((2 * t + h) % AS_VSLOTS) * AS_V_BYTES + voff_r0;
            f32x4 vb[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
#ifdef RANGE_EXP_AS_NOVREAD
                for (int gg = 0; gg < 4; ++gg) vb[rr][gg] = f32x4{(float)t, (float)h, (float)rr, (float)gg};
#else
                for (int gg = 0; gg < 4; ++gg) vb[rr][gg] = *reinterpret_cast<const f32x4*>(vt + rr * 1024 + gg * 256);
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) asm volatile("" : "+v"(vb[rr][gg]));
            if (!SPREAD) issue_v(2 * t + h + 3);           // (into the slot just read)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 2 * h + rr;
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    if (SPREAD) issue_v_row(2 * t + h + 3, 4 * rr + gg);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int nq = 0; nq < NQ; ++nq) {
                            if (NQ * (GEO ? 2 : 1) > 2) {
                                // 256 accumulator registers: pinned to the AGPR file (mfma_a; left to
                                // the builtin hipcc moved 600 registers per block between the files)
                                mfma_a(acc[nq][0][gg * 4 + u], p1[nq][r], vb[rr][gg][u]);
                                if (GEO) mfma_a(acc[nq][1][gg * 4 + u], p2[nq][r], vb[rr][gg][u]);
                            } else {
                                acc[nq][0][gg * 4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1[nq][r], vb[rr][gg][u], acc[nq][0][gg * 4 + u], 0, 0, 0);
                                if (GEO)
                                    acc[nq][1][gg * 4 + u] = __builtin_amdgcn_mfma_f32_16x16x4f32(p2[nq][r], vb[rr][gg][u], acc[nq][1][gg * 4 + u], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the clamped prefetches past the end

    // ---- this workgroup's partials: accumulator tile (gg, u), register r of lane (n = j, mg = g)
    //      = query 4 g + r, column 256 w + 64 gg + 4 j + u: the 4 tiles u are 4 consecutive columns
    constexpr int QC = 16 * NQ;
    float* ob = a.osum + (int64_t)blockIdx.x * (2 * QC * VAL_DIM);
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) {
#pragma unroll
        for (int h = 0; h < (GEO ? 2 : 1); ++h) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x4 o = {acc[nq][h][gg * 4 + 0][r], acc[nq][h][gg * 4 + 1][r], acc[nq][h][gg * 4 + 2][r],
                                     acc[nq][h][gg * 4 + 3][r]};
                    *reinterpret_cast<f32x4*>(ob + ((int64_t)h * QC + 16 * nq + 4 * g + r) * VAL_DIM + 256 * wave + 64 * gg + 4 * j) = o;
                }
            }
        }
        float za = z1[nq], zb = z2[nq];
        za += __shfl_xor(za, 16); za += __shfl_xor(za, 32);
        zb += __shfl_xor(zb, 16); zb += __shfl_xor(zb, 32);
        if (wave == 0 && g == 0) {
            a.zsum[((int64_t)blockIdx.x * QC + 16 * nq + j) * 2 + 0] = za;
            a.zsum[((int64_t)blockIdx.x * QC + 16 * nq + j) * 2 + 1] = zb;
        }
    }
}

// Sums the workgroups' partials (fixed order: 32 groups of consecutive workgroups, each summed in
// order, then the 32 group sums in order), normalises, blends like range/range.py:238
// ((1 - beta) * G + beta * H in float32) and packs with e-hat: out (B,1280) float64.
// grid (B, 8), 1024 threads: block y handles columns [128 y, 128 y + 128) as 32 float4; thread =
// (part 0..31, column): 8 workgroups' partials per thread, their loads in flight together (with 8
// parts of 32 the kernel took 13 us of dependent-latency loads for 33 MB).
constexpr int SF_PARTS = 32;
__global__ __launch_bounds__(1024) void small_finalize_kernel(const float* __restrict__ osum, const float* __restrict__ zsum,
                                                              int n_wg, int qcap, int geo, float beta,
                                                              const double* __restrict__ ehat64, double* __restrict__ out) {
    __shared__ f32x4 sh_o[2][SF_PARTS][32];
    __shared__ float sh_z[2][1024];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int part = tid >> 5, c4 = blockIdx.y * 32 + (tid & 31);
    const int per = (n_wg + SF_PARTS - 1) / SF_PARTS;
    const int w0 = part * per, w1 = min(n_wg, w0 + per);
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const float* ob = osum + (int64_t)w0 * (2 * qcap * VAL_DIM) + (int64_t)q * VAL_DIM + 4 * c4;
    const int64_t wstride = (int64_t)2 * qcap * VAL_DIM;
    int w = w0;
    for (; w + 8 <= w1; w += 8, ob += 8 * wstride) {
        f32x4 t1[8], t2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            t1[i] = *reinterpret_cast<const f32x4*>(ob + i * wstride);
            t2[i] = geo ? *reinterpret_cast<const f32x4*>(ob + i * wstride + (int64_t)qcap * VAL_DIM) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += t1[i]; s2 += t2[i]; }
    }
    for (; w < w1; ++w, ob += wstride) {
        s1 += *reinterpret_cast<const f32x4*>(ob);
        if (geo) s2 += *reinterpret_cast<const f32x4*>(ob + (int64_t)qcap * VAL_DIM);
    }
    sh_o[0][part][tid & 31] = s1;
    sh_o[1][part][tid & 31] = s2;
    // the weight sums: 1024 values per head (zero beyond n_wg), a fixed tree
    {
        float za = 0.f, zb = 0.f;
        for (int ww = tid; ww < n_wg; ww += 1024) { za += zsum[((int64_t)ww * qcap + q) * 2]; zb += zsum[((int64_t)ww * qcap + q) * 2 + 1]; }
        sh_z[0][tid] = za;
        sh_z[1][tid] = zb;
    }
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if (tid < d) { sh_z[0][tid] += sh_z[0][tid + d]; sh_z[1][tid] += sh_z[1][tid + d]; }
        __syncthreads();
    }
    if (part == 0) {
        f32x4 h = sh_o[0][0][tid], gsum = sh_o[1][0][tid];
#pragma unroll
        for (int p = 1; p < SF_PARTS; ++p) { h += sh_o[0][p][tid]; gsum += sh_o[1][p][tid]; }
        const float zh = sh_z[0][0], zg = sh_z[1][0];
        double* o = out + (int64_t)q * 1280 + 4 * c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float H = h[e] / zh;
            float m = H;
            if (geo) {
                const float G = gsum[e] / zg;
                m = __fadd_rn(__fmul_rn(1.0f - beta, G), __fmul_rn(beta, H));     // range.py:238, float32
            }
            o[e] = (double)m;
        }
    }
    if (blockIdx.y == 0 && tid < 256) out[(int64_t)q * 1280 + 1024 + tid] = ehat64[(int64_t)q * 256 + tid];
}

}  // namespace range_hip

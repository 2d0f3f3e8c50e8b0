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
// so a piece (48 KB) is one contiguous LDS-DMA copy and a B operand one conflict-free
// ds_read_b128.  Element i of lane (n, g) is row 32R + (i < 4 ? pi_row(4g+i) : 16 + pi_row(4g+i-4))
// - the order in which a lane of pass 1 holds its logits (attend_kernels.h: pi_row) - and column
// 64*(t>>2) + 4n + (t&3) of tile t = 16P + ct (the accumulator order of attend_stored_kernel, so
// the two kernels share their epilogue layout).  The weights come straight from the lane's own
// kept logits: no lane exchange, no LDS round trip for the A operand.
//
// Work decomposition as in attend_stored_kernel: workgroup = 4 waves = 64 queries, wave w owns
// queries 16w..16w+15 and all 1024 output columns (64 accumulator tiles = 256 AGPRs); the four
// waves share every piece of V through a 3-slot LDS ring (144 KB: 96 KB in flight per CU while
// the third slot is read).  One step = one piece = 16 column tiles x (3 operand reads + 6 MFMAs).
#pragma once
#include "attend_kernels.h"

namespace range_hip {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// column tiles per LDS piece: 16 (3 ring slots of 48 KB, 96 KB in flight) or 8 (6 slots of 24 KB,
// 120 KB in flight, twice the barriers: measured 9.04 ms against 8.70 - the stream is not waiting
// for latency, more bytes in flight do not help)
#ifndef RANGE_PVB_TILES
#define RANGE_PVB_TILES 16
#endif
constexpr int PVB_TP = RANGE_PVB_TILES;
static_assert(PVB_TP == 16 || PVB_TP == 8, "piece of 16 or 8 column tiles");
constexpr int PVB_NP = 64 / PVB_TP;                            // pieces (= steps) per 32-row group
constexpr int PVB_PIECE_BYTES = PVB_TP * 3 * 1024;             // column tiles x 3 planes x 1 KB
constexpr int PVB_SLOTS = 144 * 1024 / PVB_PIECE_BYTES;        // V ring slots
constexpr int PVB_AHEAD = PVB_SLOTS - 1;                       // a piece is requested this many steps ahead
constexpr int PVB_OPS = PVB_TP * 3 / 4;                        // LDS-DMA operations per wave and piece
constexpr int PVB_WAVE_BYTES = PVB_PIECE_BYTES / 4;            // a wave's share of a piece
constexpr int PVB_GROUP_BYTES = 64 * 3 * 1024;                 // 32 rows x 1024 columns x 6 B
// LDS map (bytes): V ring 144 KB | S ring 3 x 4 KB | X ring 3 x 256 B = 160,512 B
constexpr int PVB_LDS_BYTES = PVB_SLOTS * PVB_PIECE_BYTES + 3 * 4096 + 3 * 256;

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
// the six kept cross products of one column tile, small terms first; hook(i) runs in the gap
// behind MFMA i (pinned there: hipcc would otherwise sink the pieces behind the whole group)
template <class Hook>
__device__ __forceinline__ void pvb_tile(f32x4& acc, const PvbA& a, const PvbB& b, Hook&& hook) {
#define RANGE_PVB_MFMA(x, y, i)                      \
    mfma_bf16_a(acc, x, y);                          \
    __builtin_amdgcn_sched_barrier(0);               \
    hook(i);                                         \
    __builtin_amdgcn_sched_barrier(0)
    RANGE_PVB_MFMA(a.l, b.h, 0);
    RANGE_PVB_MFMA(a.m, b.m, 1);
    RANGE_PVB_MFMA(a.h, b.l, 2);
    RANGE_PVB_MFMA(a.m, b.h, 3);
    RANGE_PVB_MFMA(a.h, b.m, 4);
    RANGE_PVB_MFMA(a.h, b.h, 5);
#undef RANGE_PVB_MFMA
}
__device__ __forceinline__ PvbB pvb_read(const char* vslot, int ct) {
    PvbB b;
    b.h = *reinterpret_cast<const u32x4*>(vslot + (ct * 3 + 0) * 1024);
    b.m = *reinterpret_cast<const u32x4*>(vslot + (ct * 3 + 1) * 1024);
    b.l = *reinterpret_cast<const u32x4*>(vslot + (ct * 3 + 2) * 1024);
    return b;
}

template <bool GEO>
__global__ __launch_bounds__(256, 1) void attend_bf16x3_kernel(ScanArgs a, const char* __restrict__ vplanes,
                                                               int32_t n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)RANGE_LPTR(smem);
    const uint32_t vring_lds = lds0;
    const uint32_t sring_lds = lds0 + PVB_SLOTS * PVB_PIECE_BYTES;
    const uint32_t xring_lds = sring_lds + 3 * 4096;
    const char* sring_b = smem + PVB_SLOTS * PVB_PIECE_BYTES;
    const char* xring_b = sring_b + 3 * 4096;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4;
    int split, qt;
    decode_block(a, split, qt);
    const int g0 = (int)(((int64_t)split * n_groups) / a.n_splits);
    const int g1 = (int)(((int64_t)(split + 1) * n_groups) / a.n_splits);
    const int nG = g1 - g0;
    const int n_steps = PVB_NP * nG;
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
    asm volatile("" : "+v"(ca), "+v"(cb), "+v"(m1), "+v"(m2), "+v"(fxq));

    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const uint32_t s_rd = (uint32_t)(wave * 1024 + lane * 16);                  // this lane's logits
    const uint32_t x_rd = (uint32_t)((pi_row(lane & 15) * 4 + g) * 4);          // as KAddr::x
    int prow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = pi_row(4 * g + r);
    const uint32_t vvoff = (uint32_t)(lane << 4);
    const int b_first = 2 * g0, b_last = a.n_blocks - 1;

    // S tile (this wave's 1 KB of kept logits) + X tile of bank block b_first + bb into ring slot
    // bb % 3: 2 vector-memory operations per wave
    auto issue_sx = [&](int bb) __attribute__((always_inline)) {
        const int b = min(b_first + bb, b_last);
        const int slot = bb % 3;
        dma_b128(a.logits + logit_tile(qtile_kept, a.n_blocks, b, wave), vvoff,
                 sring_lds + slot * 4096 + wave * 1024);
        dma_b32(a.xyz4 + (int64_t)b * BLK * 4, (uint32_t)(lane << 2), xring_lds + slot * 256);
    };
    // this wave's quarter of the piece of step st into ring slot vslot: op `i` of PVB_OPS
    auto issue_v = [&](int st, int vslot, int i) __attribute__((always_inline)) {
        const int sc = min(st, n_steps - 1);
        const char* src = vplanes + ((int64_t)g0 * PVB_NP + sc) * PVB_PIECE_BYTES + wave * PVB_WAVE_BYTES + (i >> 2) * 4096;
        if ((i & 3) == 0) dma_group_begin(vring_lds + vslot * PVB_PIECE_BYTES + wave * PVB_WAVE_BYTES + (i >> 2) * 4096);
        dma_b128_q(src, vvoff, i & 3);
    };
    // weights of bank block b_first + bb (4 per lane) -> packed bf16 planes, two words each
    auto weights = [&](int bb, uint32_t (&h)[2], uint32_t (&m)[2], uint32_t (&l)[2]) __attribute__((always_inline)) {
        const int slot = bb % 3;
        const f32x4 sv = *reinterpret_cast<const f32x4*>(sring_b + slot * 4096 + s_rd);
        f32x4 cg = {0.f, 0.f, 0.f, 0.f};
        if (GEO) {
            const float xa = *reinterpret_cast<const float*>(xring_b + slot * 256 + x_rd);
            mfma_v_first(cg, xa, fxq);
            asm volatile("s_nop 15" : "+v"(cg));
        }
        const int n_left = (int)(a.n_valid - (int64_t)(b_first + bb) * BLK);
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

    PvbA cur, nxt;
    cur.h = cur.m = cur.l = nxt.h = nxt.m = nxt.l = u32x4{0u, 0u, 0u, 0u};
    if (nG > 0) {
        issue_sx(0);
        issue_sx(1);
        issue_sx(2);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // blocks 0 and 1 have landed
        uint32_t h[2], m[2], l[2];
        weights(0, h, m, l);
        cur.h[0] = h[0]; cur.h[1] = h[1]; cur.m[0] = m[0]; cur.m[1] = m[1]; cur.l[0] = l[0]; cur.l[1] = l[1];
        weights(1, h, m, l);
        cur.h[2] = h[0]; cur.h[3] = h[1]; cur.m[2] = m[0]; cur.m[3] = m[1]; cur.l[2] = l[0]; cur.l[3] = l[1];
        // slot 0 is refilled next: this wave's S tile has been read (lgkmcnt), and the X tile - one
        // copy shared by the four waves, each of which writes it - by every wave (barrier)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        nxt = cur;
        // the first PVB_AHEAD pieces, with S / X of block 3 where the steady state has it relative to
        // the pieces (the waits of the first steps count the same operations as all later ones)
        if (PVB_NP == 4) issue_sx(3);
#pragma unroll
        for (int pc = 0; pc < PVB_AHEAD; ++pc) {
#pragma unroll
            for (int i = 0; i < PVB_OPS; ++i) issue_v(pc, pc, i);
            if (PVB_NP == 8 && pc == 0) issue_sx(3);
        }
    }

    int vs = 0;                                   // ring slot of the current step
    // The last column tile of a step is not executed in its step but carried (operands in
    // registers) behind the next step's barrier, where it covers the latency of that step's first
    // LDS reads.  (Before the first step the carried operands are zero.)
    PvbB carry;
    carry.h = carry.m = carry.l = u32x4{0u, 0u, 0u, 0u};
    auto nohook = [](int) __attribute__((always_inline)) {};
    for (int gi = 0; gi < nG; ++gi) {
#pragma unroll
        for (int P = 0; P < PVB_NP; ++P) {
            const int st = PVB_NP * gi + P;
            const int vs2 = vs == 0 ? PVB_SLOTS - 1 : vs - 1;       // slot of step st + PVB_AHEAD == that of st - 1
            // The piece of this step has landed when only the operations issued during the last
            // PVB_AHEAD - 1 steps are outstanding.  16-tile pieces: those of the previous step, 12 (V)
            // after steps 1 and 3, 14 (V + S + X) after steps 0 and 2; 8-tile pieces: four steps of
            // 6, one of them with S + X.
            if (PVB_NP == 8) RANGE_WAIT_BARRIER(26);
            else if (P & 1) RANGE_WAIT_BARRIER(14);
            else RANGE_WAIT_BARRIER(12);
            const char* vslot = smem + vs * PVB_PIECE_BYTES + lane * 16;
            PvbB b0 = pvb_read(vslot, 0), b1 = pvb_read(vslot, 1);
            pvb_tile(acc[PVB_TP * ((P + PVB_NP - 1) % PVB_NP) + PVB_TP - 1], cur, carry, nohook);
            if (P == 0) {                                     // (in the first group nxt == cur)
                asm volatile("s_nop 1");                      // the carried MFMAs still read cur
                cur = nxt;
            }
            // LDS-DMA of this step: the piece PVB_AHEAD steps ahead (this wave's quarter), then S / X
#ifdef RANGE_EXP_PVB_SAMESRC   // (timing experiment: every piece from the same, cache-resident source)
            const char* vsrc = vplanes + (int64_t)(st & 7) * PVB_PIECE_BYTES + wave * PVB_WAVE_BYTES;
#else
            const char* vsrc = vplanes + ((int64_t)g0 * PVB_NP + min(st + PVB_AHEAD, n_steps - 1)) * PVB_PIECE_BYTES + wave * PVB_WAVE_BYTES;
#endif
            const uint32_t vdst = vring_lds + vs2 * PVB_PIECE_BYTES + wave * PVB_WAVE_BYTES;
#pragma unroll
            for (int ct = 0; ct < PVB_TP - 1; ++ct) {
                PvbB b2;
                pvb_tile(acc[PVB_TP * P + ct], cur, b0, [&](int i) __attribute__((always_inline)) {
                    const int cr = ct + 2 < PVB_TP ? ct + 2 : PVB_TP - 1;
                    if (i == 0) b2.h = *reinterpret_cast<const u32x4*>(vslot + (cr * 3 + 0) * 1024);
                    else if (i == 1) b2.m = *reinterpret_cast<const u32x4*>(vslot + (cr * 3 + 1) * 1024);
                    else if (i == 2) b2.l = *reinterpret_cast<const u32x4*>(vslot + (cr * 3 + 2) * 1024);
                    else if (i == 3) {
#ifndef RANGE_EXP_PVB_NODMA
                        if (ct < PVB_OPS) {
                            if ((ct & 3) == 0) dma_group_begin(vdst + (ct >> 2) * 4096);
                            dma_b128_q(vsrc + (ct >> 2) * 4096, vvoff, ct & 3);
                        } else if (ct == PVB_OPS && P % (PVB_NP / 2) == 0) {
                            issue_sx(2 * gi + 4 + P / (PVB_NP / 2));
                        }
#endif
                    } else if (i == 5 && ct == PVB_TP - 2) {
                        asm volatile("s_nop 1");             // the step ends: whatever follows may be vector ALU
                    }
                });
                b0 = b1; b1 = b2;
            }
            carry = b0;                                      // the step's last column tile
            // the next group's weights: its first block after the first half of the steps, its second
            // after the last (cut into pieces and placed in the MFMA gaps they took the same time:
            // the vector ALU work costs its issue slots either way)
#ifdef RANGE_EXP_PVB_NOW
            if (false) {
#else
            if (P % (PVB_NP / 2) == PVB_NP / 2 - 1) {
#endif
                uint32_t h[2], m[2], l[2];
                weights(2 * gi + 2 + P / (PVB_NP / 2), h, m, l);
                const int o = 2 * (P / (PVB_NP / 2));
                nxt.h[o] = h[0]; nxt.h[o + 1] = h[1];
                nxt.m[o] = m[0]; nxt.m[o + 1] = m[1];
                nxt.l[o] = l[0]; nxt.l[o + 1] = l[1];
            }
            vs = vs == PVB_SLOTS - 1 ? 0 : vs + 1;
        }
    }
    if (nG > 0) {
        pvb_tile(acc[63], cur, carry, nohook);
        // hipcc knows nothing about the MFMA inside an asm statement: whatever it schedules next may
        // read this accumulator (it moves accumulators around in front of acc_fence's operands).
        // The wait states an MFMA result needs before a non-MFMA reader therefore sit in a
        // statement that carries the accumulator as an operand, directly behind the last MFMA.
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[63]));
    }
    // the clamped prefetches of the last steps are still in flight into this workgroup's LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

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

}  // namespace range_hip

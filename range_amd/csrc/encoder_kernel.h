// Kernel A of the RANGE engine: fused (lon,lat) -> spherical-harmonic features -> SirenNet ->
// L2 normalisation, all in float64 like the reference (satclip/model_old.py:326-330,
// range/range.py:83-84, 210-212), plus the query's unit xyz (range/range.py:225-229).
//
// Reference ops replaced: spherical_harmonics.py:27-42 (1600 TorchScript launches + stack for
// L=40), location_encoder.py:98-112 (3 F.linear + 2 sin), range.py:212 (norm + div).
//
// Design
//   * The SH basis is evaluated by the stable three-term recurrence on fully normalised
//     associated Legendre functions (one chain per order m), with the reference's conventions
//     folded into the chain seed: "analytic" = sqrt(2) N P cos/sin(m phi) without the
//     Condon-Shortley sign and pi * orthonormal for m = 0; "closed-form" = orthonormal m = 0 and
//     the Condon-Shortley sign kept.  The reference's expanded
//     polynomials are NOT reproduced: they are ill-conditioned in float64 above |lat| ~ 45 deg.
//   * The features are never written to HBM.  The first layer's K dimension (L*L) is re-ordered
//     on the host into "slots": slot 0 = order 0, slot s = orders {s, L-s}; every slot is one
//     thread's work of exactly L recurrence steps for one query, i.e. perfectly balanced.  Four
//     slots (<= 8L features) are generated per round into LDS and consumed by the MFMAs.
//   * GEMMs run on v_mfma_f64_16x16x4_f64.  Workgroup = 4 waves = 32 queries; wave w owns hidden
//     columns [w*H/4, (w+1)*H/4) for both 16-query tiles, so weights are never shared between
//     waves and stream straight from L2 in pre-packed fragment order (512 B per wave
//     instruction), while activations go through LDS in fragment order with an XOR swizzle that
//     keeps both the accumulator write-back and the operand reads bank-conflict free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "async_err.h"

namespace range_hip {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int ENC_QTILE = 32;
constexpr int ENC_MAX_LAYERS = 8;      // hidden layers + last
constexpr int ENC_SLOTS_PER_ROUND = 4;
constexpr int ENC_EMBED = 256;

// One entry per (l, m >= 0) of the reference's generated "analytic" functions (range_amd/sh_table.py):
//   Y = front * (a0 + a2 x^2)^(p2/2) * x^kx * sum_j coef[off + j] * x^pow[off + j],  x = cos(theta)
struct SHDesc {
    double front, a0, a2;
    int32_t off;
    int16_t cnt;
    int8_t p2, kx;
};
static_assert(sizeof(SHDesc) == 32, "SHDesc layout");

struct EncArgs {
    const double* lonlat;   // (B,2)
    double* ehat64;         // (B,256) normalised
    double* eraw64;         // (B,256) un-normalised SirenNet output, or null
    float* ehat32;          // (B,256)
    float* xq;              // (B,4)
    int64_t B;
    int32_t L;
    int32_t n_slots;
    int32_t n_rounds;
    int32_t n_layers;       // hidden layers (>=1); last layer is index n_layers
    int32_t H;
    int32_t kp0_total;      // k-step PAIRS (8 features) of the padded, permuted first-layer K
    int32_t lds_main_doubles;
    int32_t n_wg32;         // workgroups [0, n_wg32) take 32 queries each, the rest 16
    const int32_t* slot_base;   // [n_slots+1] padded feature offsets (multiples of 8)
    const double* coefA;        // [L*L] at l*L+m : a(l,m)            (0 for l<=m)
    const double* coefB;        // [L*L] at l*L+m : a(l,m)*b(l,m)
    const double* seedc;        // [L]  chain seed constant incl. convention factors
    // reference-faithful analytic SH (null: stable recurrence): the generated polynomials'
    // 15-digit coefficients, walked in the reference's operation order
    const SHDesc* sh_desc;      // [L*L] at l*L+m
    const double* sh_coef;
    const int32_t* sh_pow;
    // small batches: the first layer split over n_parts x n_kparts workgroups per 16-query tile
    // (each takes part_cols hidden columns and a range of the K slots, and writes its raw partial
    // sums to h1); the second kernel sums the K parts, adds the bias and activates
    double* h1;                 // (n_kparts, ceil(B/16)*16, H) f64
    int32_t n_parts;
    int32_t part_cols;
    int32_t n_kparts;
    // ... and the second layer over n_parts2 column parts per tile (MODE 3: activated slice to
    // h2); the last kernel then starts from h2 (rest_from = 1) instead of the partial sums of h1
    double* h2;                 // (ceil(B/16)*16, H) f64
    int32_t n_parts2;
    int32_t part2_cols;
    int32_t rest_from;
    // ... and, for a few tiles, the LAST layer over 4 parts of 64 outputs per tile (MODE 4: raw
    // outputs + bias to e3); encoder_norm_kernel then normalises and writes the results
    double* e3;                 // (ceil(B/16)*16, 256) f64
    double* h1a;                // encoder_tile_kernel: the ACTIVATED first layer of the tile (16, H) f64, or null
    // encoder_tile_kernel (one 16-query tile, one launch): 4 arrival counters, 64 words apart; zero
    // when the context is created, left zero by every launch
    uint32_t* sync;
    uint32_t* err;              // host-mapped word (or null): set when a bounded in-kernel wait gave up
    int32_t debug_giveup;       // test hook: every in-launch wait behaves as if it had expired
    const double* wp[ENC_MAX_LAYERS];     // packed weights, pair-fragment order (see gemm_kpairs)
    const double* bias[ENC_MAX_LAYERS];
};

// h1 / h2 / e3 travel between workgroups: inside ONE launch (encoder_tile_kernel) that needs
// agent-scope accesses - written through (sc1), read past the L1 (guide: Guideline 16, the
// "every load sc1" form); between launches they are ordinary global memory either way.
__device__ __forceinline__ void st_xwg(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_xwg(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(
        reinterpret_cast<unsigned long long*>(const_cast<double*>(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// LDS address (in doubles) of activation element (query q in 0..16 QT - 1, k) in fragment order:
// fragment (kstep, qtile) = 64 doubles = the A operand of one MFMA: lane (kq = k&3, ql = q&15);
// QT = 16-query tiles of the workgroup (a 16-query workgroup packs its one tile densely: that is
// what lets hidden widths up to 1024 - 16 x 1024 float64 = 128 KB - fit the LDS).
// The XOR term spreads the 16 lanes of an accumulator write-back (16 consecutive k, same q)
// over 16 bank pairs; a fragment read (fixed k, ql = 0..15) stays a contiguous permutation.
template <int QT>
__device__ __forceinline__ int act_addr(int q, int k) {
    const int kstep = k >> 2, kq = k & 3;
    return ((kstep * QT + (q >> 4)) << 6) + (kq << 4) + ((q & 15) ^ ((kq << 2) | (kstep & 3)));
}
template <int QT>
__device__ __forceinline__ int frag_addr(int kstep, int qt, int lane) {
    const int kq = lane >> 4;
    return ((kstep * QT + qt) << 6) + (kq << 4) + ((lane & 15) ^ ((kq << 2) | (kstep & 3)));
}

// acc[qt][i] += A(lds, 2*kpairs k-steps) x Wp rows owned by this wave (NTW n-tiles).
// Weights are packed per PAIR of k-steps: element [(ntile*kp_total + kpair)*64 + lane] is a
// double2 {W[n][8*kpair + (lane>>4)], W[n][8*kpair + 4 + (lane>>4)]}, n = ntile*16 + (lane&15):
// one 16-byte load per lane (1 KB per wave instruction) feeds two MFMA k-steps.
// wp points at this wave's first n-tile, pair 0, lane element; n-tile stride = kp_stride*64.
// The fragments come straight from L2 / Infinity Cache (each wave owns its n-tiles, nothing to
// share through LDS): one pair is 32 f64 MFMAs = 2048 cycles, less than that latency under load,
// so the loads run PF pairs ahead in a statically indexed register ring: 3 where a wave owns many
// n-tiles (the large-batch kernel: 16 bytes x n-tiles x PF registers), 12 in the small-batch
// kernels, whose waves own one or two n-tiles and walk ALL k-pairs of a layer alone - there the
// chain is bound by the latency of this stream (64 pairs at 3 in flight: ~35 us per layer; round 3).
constexpr int ENC_PF = 3;
constexpr int ENC_PF_SMALL = 3;     // (12 was measured: 111 -> 124 us for 16 queries - not the stream's latency then)
typedef double f64x2 __attribute__((ext_vector_type(2)));

// kp_rot rotates the order in which the pairs are visited (pair index = (i + kp_rot) mod kpairs):
// every workgroup streams the SAME weights, and without a per-workgroup rotation they all ask the
// same L2 lines at the same moment.
template <int NTW, int QT, int PF = ENC_PF>
__device__ __forceinline__ void gemm_kpairs(const double* lds, const f64x2* wp, int kpairs,
                                            int64_t kp_stride, int kp_rot, int lane,
                                            f64x4 (&acc)[QT][NTW], int ks_base = 0) {
    auto rot = [&](int i) __attribute__((always_inline)) {
        i += kp_rot;
        return i >= kpairs ? i - kpairs : i;
    };
    // (ks_base: first k-step of the LDS operand when wp points into the middle of the K range)
    f64x2 bq[PF][NTW];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kp = rot(d < kpairs ? d : kpairs - 1);
#pragma unroll
        for (int i = 0; i < NTW; ++i) bq[d][i] = wp[((int64_t)i * kp_stride + kp) * 64];
    }
    int kp0 = 0;
    for (; kp0 + PF <= kpairs; kp0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kp = rot(kp0 + d);
            const int ks = ks_base + 2 * kp;
            const double a00 = lds[frag_addr<QT>(ks, 0, lane)];
            const double a01 = QT > 1 ? lds[frag_addr<QT>(ks, 1, lane)] : 0.0;
            const double a10 = lds[frag_addr<QT>(ks + 1, 0, lane)];
            const double a11 = QT > 1 ? lds[frag_addr<QT>(ks + 1, 1, lane)] : 0.0;
            f64x2 b[NTW];
#pragma unroll
            for (int i = 0; i < NTW; ++i) b[i] = bq[d][i];
            // refill this ring slot with pair kp + PF (clamped: the tail re-reads the last)
            const int kn = rot(kp0 + d + PF < kpairs ? kp0 + d + PF : kpairs - 1);
#pragma unroll
            for (int i = 0; i < NTW; ++i) bq[d][i] = wp[((int64_t)i * kp_stride + kn) * 64];
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                acc[0][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, b[i].x, acc[0][i], 0, 0, 0);
                if (QT > 1) acc[QT - 1][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, b[i].x, acc[QT - 1][i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                acc[0][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a10, b[i].y, acc[0][i], 0, 0, 0);
                if (QT > 1) acc[QT - 1][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, b[i].y, acc[QT - 1][i], 0, 0, 0);
            }
        }
    }
    // remainder (< PF pairs): their fragments are already in the ring, in order
#pragma unroll
    for (int d = 0; d < PF - 1; ++d) {
        if (kp0 + d < kpairs) {
            const int kp = rot(kp0 + d);
            const int ks = ks_base + 2 * kp;
            const double a00 = lds[frag_addr<QT>(ks, 0, lane)];
            const double a01 = QT > 1 ? lds[frag_addr<QT>(ks, 1, lane)] : 0.0;
            const double a10 = lds[frag_addr<QT>(ks + 1, 0, lane)];
            const double a11 = QT > 1 ? lds[frag_addr<QT>(ks + 1, 1, lane)] : 0.0;
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                acc[0][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, bq[d][i].x, acc[0][i], 0, 0, 0);
                if (QT > 1) acc[QT - 1][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, bq[d][i].x, acc[QT - 1][i], 0, 0, 0);
                acc[0][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a10, bq[d][i].y, acc[0][i], 0, 0, 0);
                if (QT > 1) acc[QT - 1][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, bq[d][i].y, acc[QT - 1][i], 0, 0, 0);
            }
        }
    }
}

// f64 MFMA C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg.
template <int NTW, int QT>
__device__ __forceinline__ void store_act(double* lds, const double* bias, double w0, int wave,
                                          int lane, f64x4 (&acc)[QT][NTW]) {
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int n = (wave * NTW + i) * 16 + (lane & 15);
        const double bn = bias[n];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qt * 16 + (lane >> 4) + 4 * r;
                lds[act_addr<QT>(q, n)] = sin(w0 * (acc[qt][i][r] + bn));
            }
    }
}

#ifdef RANGE_EXP_ENC_STAMPS   // tuning only: phase stamps of workgroup 0 / thread 0 behind e3's rows (100 MHz counter)
#define ENC_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && a.e3) reinterpret_cast<unsigned long long*>(a.e3 + ((a.B + 15) / 16) * 16 * ENC_EMBED)[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ENC_STAMP(i) do { } while (0)
#endif

// NT = H / 64.  NW waves per workgroup (4 or 16) share the H/16 n-tiles of 16 hidden columns:
// NTW = 4*NT/NW per wave (and 16/NW of the 16 output n-tiles).  Sixteen waves (four per SIMD)
// are what the float64 MFMA pipe wants (tools/micro/mfma_f64_peak.hip: 36 TFLOP/s with 1-2 waves
// per SIMD, 47-49 with 3 or more); they need H to be a multiple of 256.
// QT = 16-query tiles of the workgroup: 2 (32 queries) or 1.  Workgroups are equal-cost, so a
// batch that needs 1.2 rounds of 32-query workgroups pays for 2; the host then gives the LAST
// round half-size workgroups (EncArgs::n_wg32), which finish in about half the time.
// MODE 0: the whole encoder in one workgroup.  Small batches leave most CUs idle and every
// workgroup streams all 9.5 MB of weights on its own (0.28 ms, whatever the batch): they run
// MODE 1 - the first layer for ONE column part (NT = the part's width / 64, `part` = which) and
// ONE range of K slots (`kpart`: a range of slots means that share of the features to generate
// and of the weights to stream) of a 16-query tile, raw partial sums to a.h1 - on n_parts x
// n_kparts times as many workgroups, then
// MODE 2 - everything after the first layer per tile: h1 = sin(30 (sum of the K parts + b)).
// Where there are CUs to spare for it too, the second layer (a third of one workgroup's remaining
// chain) runs as MODE 3 - ONE column part of the second layer, activated slice to a.h2 - and
// MODE 2 starts from a.h2 (a.rest_from = 1).
// NWT >= NW waves in the workgroup: waves NW.. only generate features (MODE 1, where a part of few
// columns has few n-tiles to compute but the whole feature set to generate).
template <int NT, int NW, int QT, int MODE = 0, int NWT = NW>
__device__ __forceinline__ void encoder_body(const EncArgs& a, int64_t q0, char* smem, int part = 0,
                                             int kpart = 0, bool skip_fill = false) {
    // (skip_fill: encoder_tile_kernel, a workgroup's 2nd.. part of a phase - its input is in LDS already)
    constexpr int NTW = 4 * NT / NW;   // hidden n-tiles per wave
    constexpr int EW = 16 / NW;        // output n-tiles per wave
    constexpr int PF = (MODE != 0 && NTW <= 2) ? ENC_PF_SMALL : ENC_PF;   // depth of the weight stream's register ring
    static_assert(MODE == 4 || (NTW * NW == 4 * NT && (MODE == 1 || MODE == 3 || EW * NW == 16)), "n-tiles must divide among the waves");
    double* lds = reinterpret_cast<double*>(smem);
    double* red = lds + a.lds_main_doubles;      // [NW waves][32 queries]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double DEG = 3.14159265358979323846 / 180.0;

    // ---- SH generator state: thread = (query gq, slot-in-round gslot); the other threads idle
    //      (reference-faithful mode: every thread generates - thread = (query, slot, sub), the
    //      polynomials of an order are dealt to the NSUB sub-threads by degree)
    constexpr int GQ = 16 * QT;                                     // queries of this workgroup
    constexpr int NSUB = NWT * 64 / (GQ * ENC_SLOTS_PER_ROUND);
    static_assert(MODE == 1 || NWT == NW, "generator-only waves exist in MODE 1 only");
    const int gq = tid & (GQ - 1);
    const int gslot = a.sh_desc ? (tid / GQ) & (ENC_SLOTS_PER_ROUND - 1) : tid / GQ;
    const int gsub = tid / (GQ * ENC_SLOTS_PER_ROUND);
    double cx = 0, sx = 0, phi = 0;
    if (MODE < 2 && gslot < ENC_SLOTS_PER_ROUND) {
        const int64_t q = (q0 + gq < a.B) ? q0 + gq : a.B - 1;
        phi = (a.lonlat[2 * q] + 180.0) * DEG;                    // spherical_harmonics.py:31
        const double theta = (a.lonlat[2 * q + 1] + 90.0) * DEG;  // :32
        cx = cos(theta);
        // |sin|: the reference forms sin(theta) as sqrt((1 - x)(1 + x)) (closed_form.py:11) / (1 - x^2)^(m/2)
        // (generated file), never negative - the same thing for theta in [0, pi], and what continues
        // the basis to a latitude beyond +-90 degrees the way the reference continues it
        sx = fabs(sin(theta));
    }

    f64x4 acc[QT][NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) acc[qt][i] = f64x4{0, 0, 0, 0};

    const int L = a.L;
    // Reference-faithful mode: the powers cos^k(theta), k < L, of this workgroup's queries, each the
    // CORRECTLY ROUNDED float64 of the true power (double-double running product) - what the
    // reference's torch.pow delivers up to its last bit.  The sums below cancel by up to 1e14, so
    // they see every bit of their terms; the powers are shared through LDS ([32 queries][L]).
    double* pwt = red + 16 * ENC_QTILE;
    if (MODE < 2 && a.sh_desc && tid < GQ) {
        double hi = 1.0, lo = 0.0;
        pwt[gq * L] = 1.0;
        for (int k = 1; k < L; ++k) {
            const double p = hi * cx;
            double e = fma(hi, cx, -p);
            e = fma(lo, cx, e);
            const double h2 = p + e;
            lo = e - (h2 - p);
            hi = h2;
            pwt[gq * L + k] = hi;
        }
    }
    // (made visible to the other generator threads by the barrier that opens the first round)
    // the slots of this workgroup: all of them, or the kpart-th of n_kparts near-equal ranges
    const int ks0 = MODE == 1 ? (a.n_slots * kpart) / a.n_kparts : 0;
    const int ks1 = MODE == 1 ? (a.n_slots * (kpart + 1)) / a.n_kparts : a.n_slots;
    const int my_rounds = (ks1 - ks0 + ENC_SLOTS_PER_ROUND - 1) / ENC_SLOTS_PER_ROUND;
    for (int rnd_i = 0; MODE < 2 && rnd_i < my_rounds; ++rnd_i) {
        // rounds are visited in a per-workgroup rotated order (same reason as kp_rot)
        const int rnd = (int)((rnd_i + blockIdx.x) % (unsigned)my_rounds);
        const int s_first = ks0 + rnd * ENC_SLOTS_PER_ROUND;
        const int s_last = min(s_first + ENC_SLOTS_PER_ROUND, ks1);
        const int kp0 = a.slot_base[s_first];
        const int kp1 = a.slot_base[s_last];
        __syncthreads();   // previous round's fragment reads are done
        const int slot = s_first + gslot;
#ifdef RANGE_EXP_ENC_NOGEN      // (timing experiment: no feature generation at all)
        if (false) {
#else
        if (a.sh_desc) {
#endif
            // the generated functions Yl{l}_m{m} of this slot's orders, in the reference's order of
            // operations: every product and every sum rounded on its own (no fused multiply-add),
            // the terms of a sum left to right; degrees l = m + gsub, m + gsub + NSUB, ...
            if (slot < s_last) {
                const int pos0 = a.slot_base[slot] - kp0;
                const int end = a.slot_base[slot + 1] - kp0;
                const int m_a = slot;
                const int m_b = (slot > 0 && L - slot > slot) ? L - slot : -1;
                const double* pw = pwt + gq * L;
                const double x2 = __dmul_rn(cx, cx);
                int pos_m = pos0;
                for (int c = 0; c < 2; ++c) {
                    const int m = c == 0 ? m_a : m_b;
                    if (m < 0) break;
                    double cm = 1.0, sm = 0.0;
                    if (m > 0) { cm = cos(m * phi); sm = sin(m * phi); }
                    double f_a0 = 0.0, f_a2 = 0.0, f_sp = 1.0;     // the (a0 + a2 x^2)^(p2/2) factor:
                    int f_p2 = 0;                                  // the same for nearly all l of an order
                    for (int l = m + gsub; l < L; l += NSUB) {
                        const SHDesc d = a.sh_desc[l * L + m];
                        double v = d.front;
                        if (d.p2) {
                            if (d.p2 != f_p2 || d.a0 != f_a0 || d.a2 != f_a2) {
                                f_p2 = d.p2; f_a0 = d.a0; f_a2 = d.a2;
                                const double s2 = __dadd_rn(d.a0, __dmul_rn(d.a2, x2));
                                double r = (d.p2 & 1) ? sqrt(s2) : 1.0;
                                for (int i = 0; i < (d.p2 >> 1); ++i) r *= s2;
                                f_sp = r;
                            }
                            v *= f_sp;
                        }
                        if (d.cnt) {
                            const double* cf = a.sh_coef + d.off;
                            const int32_t* pk = a.sh_pow + d.off;
                            double sum = __dmul_rn(cf[0], pw[pk[0]]);
                            int t = 1;
                            for (; t + 4 <= d.cnt; t += 4) {      // loads of 4 terms in flight
                                const double c0 = cf[t], c1 = cf[t + 1], c2 = cf[t + 2], c3 = cf[t + 3];
                                const double p0 = pw[pk[t]], p1 = pw[pk[t + 1]], p2 = pw[pk[t + 2]], p3 = pw[pk[t + 3]];
                                sum = __dadd_rn(sum, __dmul_rn(c0, p0));
                                sum = __dadd_rn(sum, __dmul_rn(c1, p1));
                                sum = __dadd_rn(sum, __dmul_rn(c2, p2));
                                sum = __dadd_rn(sum, __dmul_rn(c3, p3));
                            }
                            for (; t < d.cnt; ++t) sum = __dadd_rn(sum, __dmul_rn(cf[t], pw[pk[t]]));
                            v *= sum;
                        }
                        if (d.kx) v *= pw[d.kx];
                        if (m == 0) {
                            lds[act_addr<QT>(gq, pos_m + (l - m))] = v;
                        } else {
                            lds[act_addr<QT>(gq, pos_m + 2 * (l - m))] = v * cm;
                            lds[act_addr<QT>(gq, pos_m + 2 * (l - m) + 1)] = v * sm;
                        }
                    }
                    pos_m += (m == 0 ? 1 : 2) * (L - m);
                }
                if (gsub == 0)
                    for (int pos = pos_m; pos < end; ++pos) lds[act_addr<QT>(gq, pos)] = 0.0;
            }
#ifdef RANGE_EXP_ENC_NOGEN
        } else if (false) {
#else
        } else if (gslot < ENC_SLOTS_PER_ROUND && slot < s_last) {
#endif
            int pos = a.slot_base[slot] - kp0;
            const int end = a.slot_base[slot + 1] - kp0;
            const int m_a = slot;
            const int m_b = (slot > 0 && L - slot > slot) ? L - slot : -1;
            for (int c = 0; c < 2; ++c) {
                const int m = c == 0 ? m_a : m_b;
                if (m < 0) break;
                double p = 1.0;
                for (int i = 0; i < m; ++i) p *= sx;
                const double seed = a.seedc[m] * p;
                double cm = 1.0, sm = 0.0;
                if (m > 0) { cm = cos(m * phi); sm = sin(m * phi); }
                double q1 = 0.0, q2 = 0.0;
                for (int l = m; l < L; ++l) {
                    const double v = (l == m) ? seed
                                              : a.coefA[l * L + m] * cx * q1 - a.coefB[l * L + m] * q2;
                    q2 = q1; q1 = v;
                    if (m == 0) {
                        lds[act_addr<QT>(gq, pos++)] = v;
                    } else {
                        lds[act_addr<QT>(gq, pos++)] = v * cm;
                        lds[act_addr<QT>(gq, pos++)] = v * sm;
                    }
                }
            }
            for (; pos < end; ++pos) lds[act_addr<QT>(gq, pos)] = 0.0;
        }
        __syncthreads();
        const f64x2* wp = reinterpret_cast<const f64x2*>(a.wp[0]) +
                          ((int64_t)((MODE == 1 ? part * (a.part_cols >> 4) : 0) + wave * NTW) * a.kp0_total + (kp0 >> 3)) * 64 + lane;
        if (NWT == NW || wave < NW)
            gemm_kpairs<NTW, QT, PF>(lds, wp, (kp1 - kp0) >> 3, a.kp0_total, (int)((blockIdx.x * 7u) % (unsigned)((kp1 - kp0) >> 3)), lane, acc);
    }

    if (MODE == 1) {
        // this workgroup's raw partial sums of the first layer to HBM (plane kpart of a.h1)
        if (NWT != NW && wave >= NW) return;
        const int64_t rows = ((a.B + 15) / 16) * 16;
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int n = part * a.part_cols + (wave * NTW + i) * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t q = q0 + (lane >> 4) + 4 * r;      // (rows past B are scratch rows of h1)
                st_xwg(a.h1 + ((int64_t)kpart * rows + q) * a.H + n, acc[0][i][r]);
            }
        }
        return;
    }
    const int l0 = MODE == 2 ? a.rest_from : 0;   // first layer whose activation is already in LDS
    if ((MODE == 3 || MODE == 4) && skip_fill) {
        // (the previous part of this phase left the tile's input in LDS)
    } else if ((MODE == 2 && l0 == 1) || MODE == 4) {
        // the activated second layer of this tile, written by the MODE 3 workgroups
        for (int idx = tid; idx < QT * 16 * a.H; idx += blockDim.x) {
            const int q = idx / a.H, k = idx - q * a.H;
            lds[act_addr<QT>(q, k)] = ld_xwg(a.h2 + (q0 + q) * a.H + k);
        }
    } else if (MODE == 3 && a.h1a) {
        // (encoder_tile_kernel: the first layer was summed and activated by an earlier phase)
        for (int idx = tid; idx < QT * 16 * a.H; idx += blockDim.x) {
            const int q = idx / a.H, k = idx - q * a.H;
            lds[act_addr<QT>(q, k)] = ld_xwg(a.h1a + (q0 + q) * a.H + k);
        }
    } else if (MODE == 2 || MODE == 3) {
        // first layer of this tile: the K parts summed in a fixed order, then
        // h1 = sin(30 * (sum + b)) (location_encoder.py:119, 147-150)
        const int64_t rows = ((a.B + 15) / 16) * 16;
        for (int idx = tid; idx < QT * 16 * a.H; idx += blockDim.x) {
            const int q = idx / a.H, k = idx - q * a.H;
            double v = ld_xwg(a.h1 + (q0 + q) * a.H + k);
            for (int kp = 1; kp < a.n_kparts; ++kp) v += ld_xwg(a.h1 + ((int64_t)kp * rows + q0 + q) * a.H + k);
            lds[act_addr<QT>(q, k)] = sin(30.0 * (v + a.bias[0][k]));
        }
    }
    if (MODE == 4) {
        // 64 outputs of the last layer (Identity activation, location_encoder.py:95-96, 112): one n-tile
        // per wave (4 waves), raw value + bias to e3; the norm follows in encoder_norm_kernel
        __syncthreads();
        const int kpH4 = a.H >> 3;
        f64x4 ae4[1][1] = {{f64x4{0, 0, 0, 0}}};
        // encoder_tile_kernel runs this with 16 waves: 4 n-tiles x 4 quarters of K, the quarters' partial
        // tiles summed through the spare half of the activation area (a wave alone walks 64 k-pairs in 8 us)
        const int KS = (16 * a.H + 3 * 4 * 256 <= a.lds_main_doubles && (int)(blockDim.x >> 6) >= 16 && (kpH4 & 3) == 0) ? 4 : 1;
        const int wn = wave & 3, kq = wave >> 2;
        if (kq >= KS) return;
        const int kcnt = kpH4 / KS;
        const f64x2* wp = reinterpret_cast<const f64x2*>(a.wp[a.n_layers]) +
                          ((int64_t)(part * 4 + wn) * kpH4 + (int64_t)kq * kcnt) * 64 + lane;
        gemm_kpairs<1, 1, ENC_PF_SMALL>(lds, wp, kcnt, kpH4, (int)((blockIdx.x * 7u) % (unsigned)kcnt), lane, ae4, 2 * kq * kcnt);
        if (KS > 1) {
            f64x4* px = reinterpret_cast<f64x4*>(lds + 16 * a.H);
            if (kq > 0) px[((kq - 1) * 4 + wn) * 64 + lane] = ae4[0][0];
            __syncthreads();
            if (kq > 0) return;
            for (int kk = 0; kk < KS - 1; ++kk) ae4[0][0] += px[(kk * 4 + wn) * 64 + lane];
        }
        const int n = (part * 4 + wn) * 16 + (lane & 15);
        const double bn = a.bias[a.n_layers][n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t q = q0 + (lane >> 4) + 4 * r;          // (rows past B are scratch rows of e3)
            st_xwg(a.e3 + q * ENC_EMBED + n, ae4[0][0][r] + bn);
        }
        return;
    }
    if (MODE == 3) {
        // the part's columns of the second layer: h2 = sin(acc + b) (location_encoder.py:147-150, w0 = 1)
        __syncthreads();
        ENC_STAMP(10);
        const int kpH3 = a.H >> 3;
        // (encoder_tile_kernel runs this with 16 waves for NW = 4: 4 quarters of K per n-tile, as in MODE 4)
        const int KS = (NTW == 1 && 16 * a.H + 3 * NW * 256 <= a.lds_main_doubles && (int)(blockDim.x >> 6) >= 4 * NW && (kpH3 & 3) == 0) ? 4 : 1;
        const int wn = wave % NW, kq = wave / NW;
        if (kq >= KS) return;
        const int kcnt = kpH3 / KS;
        const f64x2* wp = reinterpret_cast<const f64x2*>(a.wp[1]) +
                          ((int64_t)(part * (a.part2_cols >> 4) + wn * NTW) * kpH3 + (int64_t)kq * kcnt) * 64 + lane;
        gemm_kpairs<NTW, QT, PF>(lds, wp, kcnt, kpH3, (int)((blockIdx.x * 7u) % (unsigned)kcnt), lane, acc, 2 * kq * kcnt);
        if (KS > 1) {
            f64x4* px = reinterpret_cast<f64x4*>(lds + 16 * a.H);
            if (kq > 0) px[((kq - 1) * NW + wn) * 64 + lane] = acc[0][0];
            __syncthreads();
            if (kq > 0) return;
            for (int kk = 0; kk < KS - 1; ++kk) acc[0][0] += px[(kk * NW + wn) * 64 + lane];
        }
        ENC_STAMP(11);
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
            const int n = part * a.part2_cols + (wn * NTW + i) * 16 + (lane & 15);
            const double bn = a.bias[1][n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t q = q0 + (lane >> 4) + 4 * r;      // (rows past B are scratch rows of h2)
                st_xwg(a.h2 + q * a.H + n, sin(acc[0][i][r] + bn));
            }
        }
        return;
    }

    // ---- hidden layers: h = sin(w0 * (acc + b)), w0 = 30 on the first layer only
    //      (location_encoder.py:83, 119, 147-150)
    const int kpH = a.H >> 3;
    for (int layer = l0; layer < a.n_layers; ++layer) {
        __syncthreads();   // all waves finished reading the previous operand
        if (!(MODE == 2 && layer == l0))
            store_act<NTW, QT>(lds, a.bias[layer], layer == 0 ? 30.0 : 1.0, wave, lane, acc);
        __syncthreads();
        if (layer + 1 < a.n_layers) {
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc[qt][i] = f64x4{0, 0, 0, 0};
            const f64x2* wp = reinterpret_cast<const f64x2*>(a.wp[layer + 1]) +
                              ((int64_t)(wave * NTW) * kpH) * 64 + lane;
            gemm_kpairs<NTW, QT, PF>(lds, wp, kpH, kpH, (int)((blockIdx.x * 7u) % (unsigned)kpH), lane, acc);
        }
    }

    // ---- last layer (Identity activation, location_encoder.py:95-96, 112): 256 outputs,
    //      EW n-tiles per wave
    f64x4 ae[QT][EW];
#pragma unroll
    for (int i = 0; i < EW; ++i)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) ae[qt][i] = f64x4{0, 0, 0, 0};
    {
        const f64x2* wp = reinterpret_cast<const f64x2*>(a.wp[a.n_layers]) +
                          ((int64_t)(wave * EW) * kpH) * 64 + lane;
        gemm_kpairs<EW, QT, PF>(lds, wp, kpH, kpH, (int)((blockIdx.x * 7u) % (unsigned)kpH), lane, ae);
    }
    const double* bl = a.bias[a.n_layers];
    double ss[QT][4];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ss[qt][r] = 0.0;
#pragma unroll
    for (int i = 0; i < EW; ++i) {
        const double bn = bl[(wave * EW + i) * 16 + (lane & 15)];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ae[qt][i][r] += bn;
                ss[qt][r] += ae[qt][i][r] * ae[qt][i][r];
            }
    }
    // ---- L2 norm over the 256 outputs of each query (range.py:212)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double v = ss[qt][r];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            if ((lane & 15) == 0) red[wave * 32 + qt * 16 + (lane >> 4) + 4 * r] = v;
        }
    __syncthreads();
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ql = qt * 16 + (lane >> 4) + 4 * r;
            double sq = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) sq += red[w * 32 + ql];
            const double nrm = sqrt(sq);
            const int64_t q = q0 + ql;
            if (q < a.B) {
#pragma unroll
                for (int i = 0; i < EW; ++i) {
                    const int n = (wave * EW + i) * 16 + (lane & 15);
                    const double e = ae[qt][i][r] / nrm;
                    if (a.eraw64) a.eraw64[q * ENC_EMBED + n] = ae[qt][i][r];
                    a.ehat64[q * ENC_EMBED + n] = e;
                    a.ehat32[q * ENC_EMBED + n] = (float)e;
                }
            }
        }
    // ---- query unit vector: float64 trig, then .float()  (range.py:225-231, utils.py:11-16)
    if (tid < 16 * QT && q0 + tid < a.B) {
        const int64_t q = q0 + tid;
        const double lon = a.lonlat[2 * q] * 3.14159265358979323846 / 180.0;
        const double lat = a.lonlat[2 * q + 1] * 3.14159265358979323846 / 180.0;
        const double cl = cos(lat);
        float4 o = make_float4((float)(cl * cos(lon)), (float)(cl * sin(lon)), (float)sin(lat), 0.f);
        *reinterpret_cast<float4*>(a.xq + q * 4) = o;
    }
}

// small-batch pair: first layer per (16-query tile, column part), then the rest per tile
// (NWP waves: 8 where the part has at least 8 n-tiles - more threads for the feature generation,
// which every part of a tile repeats)
constexpr int ENC_PART_WAVES = 16;     // 4 per SIMD: the generator waits on table loads most of its time
template <int NTP, int NWP>
__global__ __launch_bounds__(ENC_PART_WAVES * 64, 1) void encoder_l1_part_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // consecutive workgroups share a tile (its queries) and then a column part (its weights)
    const int per_tile = a.n_parts * a.n_kparts;
    const int tile = blockIdx.x / per_tile, rest = blockIdx.x - tile * per_tile;
    const int part = rest / a.n_kparts, kpart = rest - part * a.n_kparts;
    encoder_body<NTP, NWP, 1, 1, ENC_PART_WAVES>(a, (int64_t)tile * 16, smem, part, kpart);
}

// second layer per (16-query tile, column part): NTP = the part's width / 64, one n-tile per wave
template <int NTP>
__global__ __launch_bounds__(NTP * 256, 1) void encoder_l2_part_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x / a.n_parts2, part = blockIdx.x - tile * a.n_parts2;
    encoder_body<NTP, 4 * NTP, 1, 3>(a, (int64_t)tile * 16, smem, part);
}

// last layer per (16-query tile, part of 64 outputs): 4 waves, one n-tile each (NT = H / 64 only
// sizes the body's unused hidden-layer registers: 4 keeps them minimal)
__global__ __launch_bounds__(256, 1) void encoder_l3_part_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x >> 2, part = blockIdx.x & 3;
    encoder_body<4, 4, 1, 4>(a, (int64_t)tile * 16, smem, part);
}

// e = e3 / |e3| (range.py:212), its float32 copy, the raw output if asked for, and the query's unit
// vector (float64 trig, then .float(): range.py:225-231, utils.py:11-16).  One wave per query.
__device__ __forceinline__ void encoder_norm_query(const EncArgs& a, int64_t q, int lane) {
    double v[4], ss = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = ld_xwg(a.e3 + q * ENC_EMBED + 64 * i + lane); ss += v[i] * v[i]; }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) ss += __shfl_xor(ss, off);
    const double nrm = sqrt(ss);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = 64 * i + lane;
        const double e = v[i] / nrm;
        if (a.eraw64) a.eraw64[q * ENC_EMBED + n] = v[i];
        a.ehat64[q * ENC_EMBED + n] = e;
        a.ehat32[q * ENC_EMBED + n] = (float)e;
    }
    if (lane == 0) {
        const double lon = a.lonlat[2 * q] * 3.14159265358979323846 / 180.0;
        const double lat = a.lonlat[2 * q + 1] * 3.14159265358979323846 / 180.0;
        const double cl = cos(lat);
        *reinterpret_cast<float4*>(a.xq + q * 4) = make_float4((float)(cl * cos(lon)), (float)(cl * sin(lon)), (float)sin(lat), 0.f);
    }
}
__global__ __launch_bounds__(256) void encoder_norm_kernel(EncArgs a) {
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q < a.B) encoder_norm_query(a, q, threadIdx.x & 63);
}

// ONE launch for a FEW 16-query tiles (the latency regime: up to 512 queries), each tile on its own
// workgroups.  As separate launches the
// small-batch encoder costs ~110 us: the first layer's part kernel 33 us, then ONE workgroup's chain
// over the rest (90 us) - and every further launch that would spread that chain costs 10-20 us of its
// own (dispatch + 4-6 us before a kernel's first memory access returns).  Here the workgroups of the
// first layer stay and meet at counters (phase stamps of a -DRANGE_EXP_ENC_STAMPS build, round 3: the
// first layer 12.5 us; summing + activating its 16 x H outputs 58 us when 256 threads of a workgroup
// did it alone - one memory round trip and one float64 sin per element and thread):
//   phase 1  all workgroups (n_parts x n_kparts): first layer, one column part x one K range (MODE 1)
//   phase 2  the first 16 H / 1024 workgroups: sum of the K parts + bias, sin(30 .): ONE element per thread
//   phase 3  workgroups [0, n_parts2): second layer, 64 columns each (MODE 3 on the activated input)
//   phase 4  workgroups [0, 4): last layer, 64 outputs each (MODE 4)
//   phase 5  workgroup 0: norm, float32 copy, query unit vectors - a wave per query
// Hand-off between phases (guide: Guideline 16 / visibility table, first row): the buffers are
// written through (st_xwg), every storing wave waits vmcnt(0), workgroup barrier, ONE lane of the
// workgroup increments the phase's counter; a consuming workgroup polls it (sc1 load) and reads
// the buffers with sc1 loads (ld_xwg) behind a barrier that lane joins.  All workgroups are
// resident (grid <= CUs, one per CU) and arrive before they wait.  If CUs are held by someone
// else's kernels (another process on the same GPU) the launch is only slower, never stuck:
// workgroups are dispatched in blockIdx order, the producers of phase 1 never wait and leave when
// done, and from phase 2 on producers and consumers are the same K workgroups, resident by then;
// a wait that does not end within ENC_SPIN_LIMIT polls (seconds) gives up instead of hanging - and
// the tile's output rows are then NaN, the context's host-mapped error word is set, and the host
// takes the separate-launch path from the next call on (range_hip.hip: check_async_error).
// The counters wrap to zero by themselves: arrivals + one increment per consumer = the wrap limit
// of the atomic inc.
constexpr uint32_t ENC_SPIN_LIMIT = 1u << 21;
// returns 1 for a workgroup that goes on to the next phase, 0 for one that has no part in it, -1
// for one whose wait gave up.  A workgroup that gives up does NOT count itself out of the counter
// it polled: the tile's later phases can then never complete, every workgroup still waiting for
// them gives up too, and nothing is written over the NaN rows that workgroup 0 of the tile - a
// consumer of every phase - leaves behind (encoder_tile_kernel).
__device__ __forceinline__ int enc_phase_sync(uint32_t* ctr, int n_prod, int n_cons, bool consumer, int* flag,
                                              uint32_t* err, int debug_giveup) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (every wave: its stores have completed)
    __syncthreads();
    if (threadIdx.x == 0) atomicInc(ctr, (uint32_t)(n_prod + n_cons - 1));
    if (!consumer) return 0;
    if (threadIdx.x == 0) {
        int ok = 1;
        for (uint32_t spins = 0;
             __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)n_prod; ++spins) {
            if (spins > ENC_SPIN_LIMIT) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (debug_giveup) ok = 0;
        if (ok) {
            atomicInc(ctr, (uint32_t)(n_prod + n_cons - 1));     // (the last consumer's increment wraps the counter to 0)
        } else if (err) {
            // say so where the host sees it without a synchronisation (range_hip.hip: check_async_error)
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0 ? 1 : -1;
}

// what a tile whose in-launch wait gave up hands out: NaN in every output row (one thread per element)
__device__ __forceinline__ void encoder_poison_tile(const EncArgs& a, int64_t q0) {
    const double nan64 = __builtin_nan("");
    const float nan32 = __builtin_nanf("");
    for (int idx = threadIdx.x; idx < 16 * ENC_EMBED; idx += blockDim.x) {
        const int64_t q = q0 + idx / ENC_EMBED;
        if (q >= a.B) break;
        const int n = idx % ENC_EMBED;
        a.ehat64[q * ENC_EMBED + n] = nan64;
        a.ehat32[q * ENC_EMBED + n] = nan32;
        if (a.eraw64) a.eraw64[q * ENC_EMBED + n] = nan64;
        if (n < 4) a.xq[q * 4 + n] = nan32;
    }
}

template <int NTP, int NWP>
__global__ __launch_bounds__(ENC_PART_WAVES * 64, 1) void encoder_tile_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (a word of the norm's reduction area, unused by the part bodies: a static __shared__ variable
    // would shift the dynamic LDS base off its 16-byte alignment)
    int* flag = reinterpret_cast<int*>(reinterpret_cast<double*>(smem) + a.lds_main_doubles);
    // several tiles run side by side, each with its own workgroups, counters and buffer rows
    const int n_wg = a.n_parts * a.n_kparts;                 // workgroups per tile
    const int tile = (int)blockIdx.x / n_wg, b = (int)blockIdx.x - tile * n_wg;
    const int part = b / a.n_kparts, kpart = b - part * a.n_kparts;
    const int64_t q0 = (int64_t)tile * 16;
    const int64_t rows = ((a.B + 15) / 16) * 16;
    uint32_t* sync = a.sync + tile * 256;
    ENC_STAMP(0);
    {
        EncArgs a1 = a;
        a1.h1a = nullptr;
        encoder_body<NTP, NWP, 1, 1, ENC_PART_WAVES>(a1, q0, smem, part, kpart);
    }
    ENC_STAMP(1);
    // ---- h1a = sin(30 (sum of the K parts + b)) (location_encoder.py:119, 147-150): one element per thread
    // The first K = max(workgroups of any later phase) workgroups stay to the end and meet at every
    // counter, whether or not a phase has work for them (narrow encoders: 2 workgroups activate, 2 run
    // the second layer, but 4 the last one); the others leave after the first layer.
    // (up to 512 queries a tile has at least that many workgroups; beyond - 513 .. 2 048 queries, round 5:
    // 2-7 workgroups per tile - ALL of a tile's workgroups stay and take the parts of a phase in turns)
    const int n_act = (16 * a.H + (int)blockDim.x - 1) / (int)blockDim.x;
    const int K = min(n_wg, max(max(n_act, a.n_parts2), 4));
    // a phase hand-off; a workgroup that leaves because its wait gave up: workgroup 0 of the tile (a
    // consumer of every phase, and the only writer of the results) leaves NaN rows behind
#define ENC_HANDOFF(ctr, n_prod, n_cons, consumer)                                              \
    do {                                                                                        \
        const int go = enc_phase_sync(ctr, n_prod, n_cons, consumer, flag, a.err, a.debug_giveup); \
        if (go <= 0) {                                                                          \
            if (go < 0 && b == 0) encoder_poison_tile(a, q0);                                   \
            return;                                                                             \
        }                                                                                       \
    } while (0)
    ENC_HANDOFF(sync, n_wg, K, b < K);
    ENC_STAMP(2);
    for (int blk = b; blk < n_act; blk += K) {
        const int e = blk * (int)blockDim.x + (int)threadIdx.x;
        if (e < 16 * a.H) {
            const int q = e / a.H, k = e - q * a.H;
            const int64_t at = (q0 + q) * a.H + k;
            double v = ld_xwg(a.h1 + at);
            for (int kp = 1; kp < a.n_kparts; ++kp) v += ld_xwg(a.h1 + (int64_t)kp * rows * a.H + at);
            st_xwg(a.h1a + at, sin(30.0 * (v + a.bias[0][k])));
        }
    }
    ENC_STAMP(3);
    // ---- second layer on the first n_parts2 workgroups
    ENC_HANDOFF(sync + 64, K, K, true);
    ENC_STAMP(4);
    // (parts of 64 columns with K split four ways where a tile has >= H / 64 workgroups; wider parts -
    // 128 or 256 columns, one n-tile per wave over the whole K - where it has fewer: the host chooses)
    for (int part = b; part < a.n_parts2; part += K) {
        if (a.part2_cols == 64) encoder_body<1, 4, 1, 3>(a, q0, smem, part, 0, part != b);
        else if (a.part2_cols == 128) encoder_body<2, 8, 1, 3>(a, q0, smem, part, 0, part != b);
        else encoder_body<4, 16, 1, 3>(a, q0, smem, part, 0, part != b);
    }
    ENC_STAMP(5);
    // ---- last layer on workgroups 0..3
    ENC_HANDOFF(sync + 128, K, K, true);
    ENC_STAMP(6);
    for (int part = b; part < 4; part += K) encoder_body<4, 4, 1, 4>(a, q0, smem, part, 0, part != b);
    ENC_STAMP(7);
    // ---- norm on workgroup 0: a wave per query
    ENC_HANDOFF(sync + 192, K, 1, b == 0);
#undef ENC_HANDOFF
    {
        const int64_t q = q0 + (threadIdx.x >> 6);
        if (q < a.B) encoder_norm_query(a, q, threadIdx.x & 63);
    }
    ENC_STAMP(8);
}

template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void encoder_rest_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    encoder_body<NT, NW, 1, 2>(a, (int64_t)blockIdx.x * 16, smem);
}

template <int NT, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void encoder_kernel(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if constexpr (NT <= 8) {     // (wider encoders run 16-query workgroups only: 32 x H float64 would not fit the LDS)
        if (b < a.n_wg32) {
            encoder_body<NT, NW, 2>(a, (int64_t)b * 32, smem);
            return;
        }
    }
    encoder_body<NT, NW, 1>(a, (int64_t)a.n_wg32 * 32 + (int64_t)(b - a.n_wg32) * 16, smem);
}

}  // namespace range_hip

// Pure host arithmetic of the engine - launch geometry, the encoder's slot plan, recurrence
// tables, weight packing - free of any HIP dependency, so that the same code the library runs is
// also compiled with g++ under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on
// the CPU (tests/native/host_sanitize.cpp, tests/test_host_cpu.py).  GPU sanitizers are not
// available on the target pool; this is the part of the C-ABI shim that can be sanitised.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace range_host {

// Number of bank splits.  Workgroups of both scan kernels are equal-cost, so the chip runs them
// in near lock-step "rounds" of n_cu * wg_per_cu workgroups: pick the split count whose last
// round is best filled (e.g. 157 query tiles x 13 splits = 2041 workgroups = 7.97 rounds of 256),
// preferring fewer splits (less partial-result traffic) on near-ties.  wg_per_cu: 1 for pass 2
// (512 registers, 129 KB LDS), 4 for pass 1 (33 KB LDS, <= 128 VGPRs).
inline int choose_splits(int n_qtiles, int n_blocks, int n_cu, int wg_per_cu, int max_splits,
                         double split_cost = 0.002) {
    const double slots = (double)n_cu * wg_per_cu;
    const int cap = std::max(1, std::min(max_splits, n_blocks / 4));   // >= 4 blocks per split
    // small batches: first of all give every slot a workgroup
    const int ns_min = std::min(cap, (int)std::ceil(slots / n_qtiles));
    int best = ns_min;
    double best_score = -1e9;
    for (int ns = ns_min; ns <= cap; ++ns) {
        const double total = (double)n_qtiles * ns;
        const double rounds = std::ceil(total / slots);
        double score = total / (rounds * slots);            // fill of the rounds
        if (rounds < 4) score *= 0.85 + 0.0375 * rounds;    // few rounds: ragged finish hurts more
        score -= split_cost * (ns - ns_min);                // partial-result traffic
        if (score > best_score) { best_score = score; best = ns; }
        if (ns - ns_min > 64) break;
    }
    return best;
}

// Stream-K partition of pass 2 (attend_kernels.h: SlabMap / SegWalk; round 5): U units in order are cut
// into G contiguous, near-equal ranges [start(w), start(w + 1)); owner(u) is the range a unit falls in.
// The kernel walks by `start`, the reduction finds a query tile's parts by `owner`: both are these
// functions (compiled for the device too: RANGE_HD), and tests/native/host_sanitize.cpp checks that they
// agree for every unit.
#ifndef RANGE_HD
#if defined(__HIPCC__)
#define RANGE_HD __host__ __device__ __forceinline__
#else
#define RANGE_HD inline
#endif
#endif
RANGE_HD int64_t sk_start(int64_t w, int64_t U, int64_t G) { return (w * U) / G; }
RANGE_HD int64_t sk_owner(int64_t u, int64_t U, int64_t G) { return ((u + 1) * G - 1) / U; }
// first block of bank column c of n_cols (columns: contiguous, near-equal block ranges)
RANGE_HD int sk_col_begin(int c, int n_blocks, int n_cols) { return (int)(((int64_t)c * n_blocks) / n_cols); }

// Small-batch encoder: workgroups per 16-query tile = column parts S (a power of two, parts of
// 64 .. 512 columns: the widths a kernel exists for) x K parts KP (ranges of at least 3 of the
// first layer's slots).  Every workgroup gets its own CU (tiles * S * KP <= n_cu); K parts come
// first - a column part re-generates all the features of its K range, a K part generates only
// its share.  1 x 1: no split.
inline void choose_encoder_split(int n_cu, int n_slots, int H, long long tiles, int& S, int& KP) {
    S = 1;
    KP = 1;
    const int kp_max = std::max(1, std::min(7, n_slots / 3));
    for (int kp = 1; kp <= kp_max; ++kp)
        for (int s2 = 1; s2 <= 8; s2 *= 2) {
            const int part = H / s2;
            if (H % s2 || !(part == 64 || part == 128 || part == 256 || part == 512)) continue;
            if (tiles * s2 * kp > n_cu) continue;
            if (s2 * kp > S * KP || (s2 * kp == S * KP && kp > KP)) { S = s2; KP = kp; }
        }
}

// The encoder's first-layer K order: the L*L spherical-harmonic features permuted into "slots"
// (slot 0 = order 0; slot s >= 1 = orders {s, L-s}, or {s} when s == L-s), every slot padded to
// whole k-step pairs (8 features).  perm: padded position -> feature index l*l+l+m, or -1.
struct EncoderPlan {
    std::vector<int> perm;
    std::vector<int32_t> slot_base;   // [n_slots + 1]
    int n_slots = 0, n_rounds = 0, max_round = 0;
};

inline bool build_encoder_plan(int L, int slots_per_round, EncoderPlan& p) {
    std::vector<int> slot_m_a, slot_m_b;
    slot_m_a.push_back(0);
    slot_m_b.push_back(-1);
    for (int s = 1; s <= L - s && s < L; ++s) {
        slot_m_a.push_back(s);
        slot_m_b.push_back(L - s > s ? L - s : -1);
    }
    p.n_slots = (int)slot_m_a.size();
    p.slot_base.assign(p.n_slots + 1, 0);
    p.perm.clear();
    for (int s = 0; s < p.n_slots; ++s) {
        p.slot_base[s] = (int32_t)p.perm.size();
        for (int c2 = 0; c2 < 2; ++c2) {
            const int m = c2 == 0 ? slot_m_a[s] : slot_m_b[s];
            if (m < 0) break;
            for (int l = m; l < L; ++l) {
                if (m == 0) p.perm.push_back(l * l + l);
                else { p.perm.push_back(l * l + l + m); p.perm.push_back(l * l + l - m); }
            }
        }
        while (p.perm.size() % 8) p.perm.push_back(-1);   // whole k-step pairs (8 features)
    }
    p.slot_base[p.n_slots] = (int32_t)p.perm.size();
    // every feature exactly once
    std::vector<char> seen((size_t)L * L, 0);
    int cnt = 0;
    for (int f : p.perm)
        if (f >= 0) {
            if (f >= L * L || seen[f]) return false;
            seen[f] = 1;
            ++cnt;
        }
    if (cnt != L * L) return false;
    p.n_rounds = (p.n_slots + slots_per_round - 1) / slots_per_round;
    p.max_round = 0;
    for (int r = 0; r < p.n_rounds; ++r) {
        const int s1 = std::min((r + 1) * slots_per_round, p.n_slots);
        p.max_round = std::max(p.max_round, p.slot_base[s1] - p.slot_base[r * slots_per_round]);
    }
    return true;
}

// Three-term recurrence on fully normalised associated Legendre functions: q(l,m) =
// a(l,m) * (x q(l-1,m) - b(l,m) q(l-2,m)); coefA = a, coefB = a*b at [l*L+m]; seedc[m] = the chain
// seed's constant with the reference's convention folded in (analytic: pi-scaled m = 0, no
// Condon-Shortley sign; closed-form: orthonormal m = 0, sign kept).
inline void recurrence_tables(int L, bool analytic, std::vector<double>& coefA, std::vector<double>& coefB,
                              std::vector<double>& seedc) {
    coefA.assign((size_t)L * L, 0.0);
    coefB.assign((size_t)L * L, 0.0);
    seedc.assign(L, 0.0);
    const double PI = 3.14159265358979323846;
    double cm = std::sqrt(1.0 / (4.0 * PI));
    for (int m = 0; m < L; ++m) {
        if (m > 0) cm *= std::sqrt((2.0 * m + 1.0) / (2.0 * m));
        double scale;
        if (m == 0) scale = analytic ? PI : 1.0;
        else scale = std::sqrt(2.0) * ((!analytic && (m & 1)) ? -1.0 : 1.0);
        seedc[m] = cm * scale;
        for (int l = m + 1; l < L; ++l) {
            if (l == m + 1) {
                coefA[(size_t)l * L + m] = std::sqrt(2.0 * m + 3.0);
                coefB[(size_t)l * L + m] = 0.0;
            } else {
                const double a = std::sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
                const double b = std::sqrt((((double)l - 1.0) * (l - 1.0) - (double)m * m) /
                                           (4.0 * (l - 1.0) * (l - 1.0) - 1.0));
                coefA[(size_t)l * L + m] = a;
                coefB[(size_t)l * L + m] = a * b;
            }
        }
    }
}

// The hidden width the encoder kernels run a checkpoint's `capacity` at: kernels exist for multiples
// of 64 up to 512, and for 768 and 1024; any other width up to 1024 runs as the next of those with
// zero-padded weights (a padded unit is sin(w0 (0 . x + 0)) = 0 feeding zero weights: every product
// it adds is an exact +0.0, the unpadded network's result bit for bit).  0: unsupported.
inline int kernel_hidden_width(int hidden) {
    if (hidden < 1 || hidden > 1024) return 0;
    return hidden <= 512 ? (hidden + 63) / 64 * 64 : (hidden <= 768 ? 768 : 1024);
}

// (n_out x k_in) row-major -> (n_pad x k_pad), the new rows and columns zero
inline std::vector<double> pad_weights(const double* W, int n_out, int k_in, int n_pad, int k_pad) {
    std::vector<double> P((size_t)n_pad * k_pad, 0.0);
    for (int r = 0; r < n_out; ++r)
        std::copy(W + (size_t)r * k_in, W + (size_t)(r + 1) * k_in, P.begin() + (size_t)r * k_pad);
    return P;
}

// Query ranges of the numpy contract's pass 2 (range_forward_host): the batch in parts, the tail parts
// of the given nominal sizes (the last ones shortest: their copies are what the caller waits for), cuts
// rounded UP to a query tile, never past the batch and never backwards - whatever `tail` holds
// (RANGE_HOST_PARTS).  Returns the cut positions, first 0 and last B.
inline std::vector<int64_t> host_part_cuts(int64_t B, const std::vector<int64_t>& tail, int64_t tile) {
    std::vector<int64_t> cuts{0};
    int64_t rest = 0;
    for (auto v : tail) rest += v;
    if (rest < B) {
        int64_t at = B - rest;
        for (auto v : tail) {
            const int64_t cut = std::min<int64_t>(B, (at + tile - 1) / tile * tile);
            if (cut > cuts.back()) cuts.push_back(cut);
            at += v;
        }
        if (cuts.back() == B) cuts.pop_back();
    }
    cuts.push_back(B);
    return cuts;
}

// Weights (n_out, k_in) row-major -> MFMA B-fragment order, two k-steps per 16-byte lane element:
//   [((ntile*kpairs + kpair)*64 + lane)*2 + e] = W[ntile*16 + (lane&15)][kperm[kpair*8 + 4e + (lane>>4)]]
// (kperm: padded position -> column or -1 for a zero pad; null: identity).
inline std::vector<double> pack_weights(const double* W, int n_out, int k_in, const std::vector<int>* kperm,
                                        int Kpad) {
    const int kp = Kpad / 8, nt = n_out / 16;
    std::vector<double> out((size_t)nt * kp * 128);
    for (int t = 0; t < nt; ++t)
        for (int s = 0; s < kp; ++s)
            for (int ln = 0; ln < 64; ++ln)
                for (int e = 0; e < 2; ++e) {
                    const int n = t * 16 + (ln & 15);
                    const int kk = s * 8 + 4 * e + (ln >> 4);
                    const int k = kperm ? (*kperm)[kk] : kk;
                    out[(((size_t)t * kp + s) * 64 + ln) * 2 + e] = k >= 0 ? W[(size_t)n * k_in + k] : 0.0;
                }
    return out;
}

}  // namespace range_host

// Host-side code of librange_hip.so under sanitizers (CPU only; GPU sanitizers are not available
// on the target pool).  Built by tests/test_host_cpu.py with
//   g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize-recover=all   (run 1)
//   g++ -std=c++17 -g -O1 -fsanitize=thread                                        (run 2)
// from the very headers the library compiles: host_plan.h (launch geometry, encoder slot plan,
// recurrence tables, weight packing) and host_copy.h (the thread pool that fills the caller's
// host array).  Exits non-zero on a failed invariant; a sanitizer report aborts the process.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include "../../range_amd/csrc/host_copy.h"
#include "../../range_amd/csrc/host_plan.h"

using namespace range_host;

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

static void test_plan_and_packing() {
    std::mt19937_64 rng(7);
    for (int L : {1, 2, 3, 7, 10, 16, 33, 40, 64}) {
        EncoderPlan p;
        CHECK(build_encoder_plan(L, 4, p));
        CHECK((int)p.perm.size() % 8 == 0 && p.slot_base.front() == 0 && p.slot_base.back() == (int)p.perm.size());
        std::vector<int> seen((size_t)L * L, 0);
        for (int f : p.perm) if (f >= 0) { CHECK(f < L * L); ++seen[f]; }
        for (int v : seen) CHECK(v == 1);
        for (int s = 0; s < p.n_slots; ++s) CHECK(p.slot_base[s] % 8 == 0 && p.slot_base[s] < p.slot_base[s + 1]);
        CHECK(p.max_round <= 8 * L + 8 * 4 && p.n_rounds * 4 >= p.n_slots);
        std::vector<double> A, B, S;
        recurrence_tables(L, true, A, B, S);
        CHECK((int)A.size() == L * L && (int)S.size() == L);
        for (double v : S) CHECK(std::isfinite(v) && v > 0.0);
        recurrence_tables(L, false, A, B, S);
        for (int m = 0; m < L; ++m) CHECK((S[m] < 0.0) == (m > 0 && (m & 1)));
        // packing: every weight lands exactly once at the fragment position the kernel reads
        for (int H : {64, 192, 512}) {
            const int K = L * L, Kp = (int)p.perm.size();
            std::vector<double> W((size_t)H * K);
            for (auto& v : W) v = (double)(rng() % 1000003) + 1.0;
            const std::vector<double> P = pack_weights(W.data(), H, K, &p.perm, Kp);
            CHECK(P.size() == (size_t)H * Kp);
            double sw = 0, sp = 0;
            for (double v : W) sw += v;
            for (double v : P) sp += v;
            CHECK(sw == sp);
            const int kp = Kp / 8;
            for (int trial = 0; trial < 200; ++trial) {
                const int n = (int)(rng() % H), kk = (int)(rng() % Kp);
                const int t = n / 16, s = kk / 8, e = (kk % 8) / 4, ln = (n & 15) + 16 * (kk % 4);
                const double got = P[(((size_t)t * kp + s) * 64 + ln) * 2 + e];
                CHECK(got == (p.perm[kk] >= 0 ? W[(size_t)n * K + p.perm[kk]] : 0.0));
            }
        }
    }
}

// the stream-K partition of pass 2: every unit in exactly one range, owner() the inverse of start(),
// and the slab indices w + qtile of a walk unique (what SegWalk / slab_parts rely on)
static void test_streamk_partition() {
    const int64_t Gs[] = {1, 2, 7, 255, 256, 1024};
    const int shapes[][2] = {{1, 4}, {1, 782}, {5, 63}, {79, 782}, {157, 782}, {157, 1024}, {3, 6250}, {1250, 97}};
    for (int64_t G : Gs)
        for (auto& sh : shapes) {
            const int n_qtiles = sh[0], cb = sh[1];
            const int64_t U = (int64_t)n_qtiles * cb;
            CHECK(sk_start(0, U, G) == 0 && sk_start(G, U, G) == U);
            for (int64_t w = 0; w < G; ++w) CHECK(sk_start(w, U, G) <= sk_start(w + 1, U, G));
            // owner(u) = the w with start(w) <= u < start(w + 1)
            const int64_t step = U > 5000 ? 37 : 1;
            for (int64_t u = 0; u < U; u += step) {
                const int64_t w = sk_owner(u, U, G);
                CHECK(w >= 0 && w < G && sk_start(w, U, G) <= u && u < sk_start(w + 1, U, G));
            }
            for (int64_t w = 0; w < G; ++w) {      // (range boundaries exactly)
                const int64_t a = sk_start(w, U, G), b = sk_start(w + 1, U, G);
                if (a < b) CHECK(sk_owner(a, U, G) == w && sk_owner(b - 1, U, G) == w);
            }
            // the walk: segments (w, qtile) in order; slab index w + qtile strictly increases
            int64_t last_slab = -1, covered = 0;
            for (int64_t w = 0; w < G; ++w) {
                int64_t u = sk_start(w, U, G);
                const int64_t u_end = sk_start(w + 1, U, G);
                while (u < u_end) {
                    const int64_t qt = u / cb, bo = u - qt * cb;
                    const int64_t n = std::min<int64_t>(u_end - u, cb - bo);
                    CHECK(n > 0 && w + qt > last_slab && w + qt < G + n_qtiles);
                    last_slab = w + qt;
                    // the reduction's view of this query tile's parts contains this workgroup
                    CHECK(sk_owner(qt * cb, U, G) <= w && w <= sk_owner((qt + 1) * cb - 1, U, G));
                    covered += n;
                    u += n;
                }
            }
            CHECK(covered == U);
        }
    for (int n_blocks : {4, 63, 782, 6250})
        for (int n_cols : {1, 2, 3, 7}) {
            CHECK(sk_col_begin(0, n_blocks, n_cols) == 0 && sk_col_begin(n_cols, n_blocks, n_cols) == n_blocks);
            for (int c = 0; c < n_cols; ++c) CHECK(sk_col_begin(c, n_blocks, n_cols) <= sk_col_begin(c + 1, n_blocks, n_cols));
        }
}

static void test_choose_splits() {
    for (int qt : {1, 2, 16, 79, 157, 1563})
        for (int nb : {1, 3, 4, 64, 782, 3125, 6250})
            for (int per_cu : {1, 4})
                for (int cap : {1, 16, 128, 2048}) {
                    const int ns = choose_splits(qt, nb, 256, per_cu, cap, 0.002);
                    CHECK(ns >= 1 && ns <= std::max(1, std::min(cap, nb / 4 > 0 ? nb / 4 : 1)));
                }
    CHECK(choose_splits(157, 6250, 256, 1, 32, 0.0014) == 13);   // the bench geometry
}

static void test_encoder_split() {
    for (int H : {64, 128, 192, 256, 320, 384, 512})
        for (int n_slots : {1, 2, 4, 6, 9, 21})
            for (long long tiles : {1LL, 2LL, 16LL, 40LL, 79LL, 113LL, 128LL, 129LL, 157LL, 1000LL}) {
                int S = -1, KP = -1;
                choose_encoder_split(256, n_slots, H, tiles, S, KP);
                CHECK(S >= 1 && KP >= 1 && (S * KP == 1 || tiles * S * KP <= 256));
                CHECK(KP == 1 || n_slots / KP >= 3 || KP <= n_slots / 3);
                if (S * KP > 1) CHECK(H % S == 0 && (H / S == 64 || H / S == 128 || H / S == 256 || H / S == 512));
            }
    int S, KP;
    choose_encoder_split(256, 21, 512, 79, S, KP);   // a 1 250-query rank of an 8-GPU strong-scaling step
    CHECK(S == 1 && KP == 3);
    choose_encoder_split(256, 21, 512, 1, S, KP);    // one tile
    CHECK(S * KP == 56 && KP == 7);
    choose_encoder_split(256, 21, 512, 157, S, KP);  // more tiles than half the CUs: no split
    CHECK(S == 1 && KP == 1);
    choose_encoder_split(256, 21, 384, 10, S, KP);   // a width without part kernels
    CHECK(S == 1 && KP == 1);
}

static void test_copy_pool() {
    std::mt19937_64 rng(11);
    for (int threads : {1, 3, 8}) {
        HostCopyPool pool(threads);
        for (size_t bytes : {(size_t)0, (size_t)1, (size_t)4095, (size_t)1 << 20, ((size_t)1 << 20) + 7,
                             (size_t)10485760 + 13, (size_t)33 << 20}) {
            std::vector<unsigned char> src(bytes + 64), dst(bytes + 64, 0xEE);
            for (auto& v : src) v = (unsigned char)rng();
            pool.copy(dst.data() + 32, src.data() + 17, bytes);          // unaligned on purpose
            CHECK(std::memcmp(dst.data() + 32, src.data() + 17, bytes) == 0);
            for (int i = 0; i < 32; ++i) CHECK(dst[i] == 0xEE && dst[32 + bytes + i] == 0xEE);   // nothing beyond
        }
        // jobs back to back reuse the workers
        std::vector<int> hits(threads, 0);
        for (int rep = 0; rep < 50; ++rep) pool.run([&](int t, int n) { CHECK(n == threads); ++hits[t]; });
        for (int v : hits) CHECK(v == 50);
    }
    // several callers at once on ONE pool (Python threads around ctypes calls): every copy complete
    {
        HostCopyPool pool(4);
        const size_t bytes = ((size_t)3 << 20) + 5;
        std::vector<std::thread> callers;
        std::vector<int> ok(6, 0);
        for (int c = 0; c < 6; ++c)
            callers.emplace_back([&, c] {
                std::vector<unsigned char> src(bytes, (unsigned char)(c + 1)), dst(bytes, 0);
                for (int rep = 0; rep < 8; ++rep) {
                    std::fill(dst.begin(), dst.end(), 0);
                    pool.copy(dst.data(), src.data(), bytes);
                    if (std::memcmp(dst.data(), src.data(), bytes) != 0) return;
                }
                ok[c] = 1;
            });
        for (auto& t : callers) t.join();
        for (int v : ok) CHECK(v == 1);
    }
}

// hidden widths without a kernel run zero-padded as the next width that has one; the cuts of the
// numpy contract stay inside the batch whatever RANGE_HOST_PARTS holds
static void test_padding_and_host_parts() {
    for (int h = -3; h <= 1100; ++h) {
        const int k = kernel_hidden_width(h);
        if (h < 1 || h > 1024) { CHECK(k == 0); continue; }
        CHECK(k >= h && (k == 768 || k == 1024 || (k <= 512 && k % 64 == 0)));
        CHECK(k - h < 256 && (h % 64 != 0 || h > 512 || k == h) && kernel_hidden_width(k) == k);
    }
    std::mt19937_64 rng(11);
    for (int trial = 0; trial < 50; ++trial) {
        const int n = 1 + (int)(rng() % 70), kin = 1 + (int)(rng() % 90), np = n + (int)(rng() % 40), kp = kin + (int)(rng() % 40);
        std::vector<double> W((size_t)n * kin);
        for (auto& v : W) v = (double)(rng() % 9973) + 1.0;
        const std::vector<double> P = pad_weights(W.data(), n, kin, np, kp);
        CHECK(P.size() == (size_t)np * kp);
        for (int r = 0; r < np; ++r)
            for (int c = 0; c < kp; ++c)
                CHECK(P[(size_t)r * kp + c] == (r < n && c < kin ? W[(size_t)r * kin + c] : 0.0));
    }
    const std::vector<std::vector<int64_t>> tails = {{4096, 512}, {4096, 1}, {1}, {64}, {5000, 5000, 5000}, {100000}, {}, {63, 63, 63},
                                                     {2048, 256}, {1, 1, 1, 1}};
    for (int64_t B : {(int64_t)4096, (int64_t)4097, (int64_t)10000, (int64_t)16384, (int64_t)5000})
        for (const auto& t : tails) {
            const std::vector<int64_t> cuts = host_part_cuts(B, t, 64);
            CHECK(cuts.size() >= 2 && cuts.front() == 0 && cuts.back() == B);
            for (size_t i = 1; i < cuts.size(); ++i) CHECK(cuts[i] > cuts[i - 1]);
            for (size_t i = 1; i + 1 < cuts.size(); ++i) CHECK(cuts[i] % 64 == 0 && cuts[i] < B);
        }
    const std::vector<int64_t> d = host_part_cuts(10000, {4096, 512}, 64);          // the default of range_forward_host
    CHECK(d.size() == 4 && d[1] == 5440 && d[2] == 9536);
}

int main() {
    test_padding_and_host_parts();
    test_plan_and_packing();
    test_choose_splits();
    test_streamk_partition();
    test_encoder_split();
    test_copy_pool();
    std::puts("host_sanitize ok");
    return 0;
}

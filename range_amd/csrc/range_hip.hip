// librange_hip.so - C ABI (include/range_hip.h) over the hand-written gfx950 kernels.
// Host side of the engine: context, one-time weight/bank packing, launch geometry, workspace.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/range_hip.h"
#include "host_common.h"
#include "host_copy.h"
#include "host_plan.h"
#include "attend_kernels.h"
#include "topk_stream.h"
#include "topk_gemm.h"
#include "attend_bf16x3.h"
#include "attend_small.h"
#include "encoder_kernel.h"

using namespace range_hip;
using namespace range_host;

// waves per encoder workgroup when the hidden width is a multiple of 256 (4 or 16)
#ifndef RANGE_ENC_WAVES
#define RANGE_ENC_WAVES 16
#endif
// top-k batches up to this size keep per-lane lists inside pass 1 (scan_stats_kernel<.., true>),
// larger ones select from the kept logits.  Measured (tools/topk_total_time.py): the selection
// wins at every batch size, so the in-scan lists only serve contexts that cannot keep logits.
#ifndef RANGE_TOPK_INSCAN_MAX
#define RANGE_TOPK_INSCAN_MAX 0
#endif
// range_topk_stream: query groups (of 16) sharing one pass over the keys (0 = by batch size:
// 1 group up to 16 queries, 2 up to 32, 4 beyond), and the depth of the per-lane lists
#ifndef RANGE_TOPKS_GROUPS
#define RANGE_TOPKS_GROUPS 0
#endif
#ifndef RANGE_TOPKS_LIST
#define RANGE_TOPKS_LIST 4
#endif

struct range_ctx {
    int device = 0;
    int n_cu = 256;
    // encoder
    bool has_encoder = false;
    range_encoder_desc desc{};
    EncArgs enc{};
    size_t enc_lds_bytes = 0;
    DevBuf<int32_t> d_slot_base;
    DevBuf<double> d_coefA, d_coefB, d_seedc;
    DevBuf<double> d_wp[ENC_MAX_LAYERS], d_bias[ENC_MAX_LAYERS];
    DevBuf<SHDesc> d_sh_desc;
    DevBuf<double> d_sh_coef;
    DevBuf<int32_t> d_sh_pow;
    size_t enc_lds_base = 0;     // LDS of the encoder without the power table of the faithful SH mode
    // bank
    bool has_bank = false;
    int64_t n_rows = 0, n_pad = 0, row_offset = 0;
    DevBuf<float> d_keys, d_values, d_xyz4;
    // opt-in pass 2 on bf16 planes of the values (attend_bf16x3.h): RANGE_PV_EXACT unless asked for
    int pv_mode = RANGE_PV_EXACT;
    DevBuf<uint32_t> d_vplanes;       // (ceil(n_rows/32), 4 pieces, 16 tiles, 3 planes, 64 lanes, 8 bf16)
    int64_t vplanes_groups = 0;
    // workspace
    DevBuf<float> ws_stats_parts, ws_slabs, ws_stats, ws_ehat32, ws_xq, ws_partial, ws_cand_val;
    // logits kept by the last range_scan_stats(keep_logits = 1): kept_B queries x kept_blocks
    // bank blocks, 1 KB tiles (attend_kernels.h: logit_tile); kept_B == 0: nothing kept
    DevBuf<float> ws_logits, ws_rowmax, ws_theta;
    int64_t kept_B = 0, kept_total = 0;
    int32_t kept_blocks = 0;
    bool allow_keep = true;   // RANGE_KEEP_LOGITS=0 in the environment: never keep (pass 2 recomputes)
    bool warned_no_keep = false;
    bool enc_split = true;    // RANGE_ENC_SPLIT=0: small batches use the one-kernel encoder too
    bool enc_split2 = true;   // RANGE_ENC_SPLIT2=0: ... without the second layer's own split
    bool enc_split3 = false;  // RANGE_ENC_SPLIT3=1: the last layer split too for up to 8 tiles (measured slower: see launch_encoder_split)
    bool enc_tail_split = true;   // RANGE_ENC_TAIL=0: the last partial round of a large batch as 16-query workgroups
    DevBuf<int32_t> ws_cand_idx;
    DevBuf<unsigned long long> ws_cand_keys;
    DevBuf<float> ws_cand_dmax;
    DevBuf<int32_t> ws_exact_count;   // queries range_topk_stream recomputed by brute force
    int topks_groups = RANGE_TOPKS_GROUPS;   // RANGE_TOPKS_GROUPS in the environment overrides
    bool topks_force_exact = false;          // RANGE_TOPKS_FORCE_EXACT=1: tests of the fallback
    bool topks_bf16 = true;                  // RANGE_TOPKS_KEYS=f32: stream the float32 keys (no prefilter)
    bool topks_fused = true;                 // RANGE_TOPKS_FUSED=0: the merge as a second launch at every batch size (A/B)
    DevBuf<uint32_t> ws_topk_sync;           // TOPKS_SYNC_WORDS: 8 arrival counters (they only count up), key-norm scratch
    uint32_t topk_sync_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // what the counters read when the next fused launch starts
    bool debug_giveup_next = false;          // range_debug_raise_async_error: the next persistent launch gives up
    bool has_values = false;                 // false: keys-only bank (range_set_keys): top-k side channel only
    int tg_sample = TG_SAMPLE;               // RANGE_TG_SAMPLE=n: pass A of the batch top-k looks at every n-th tile (tuning)
    bool topk_gemm = true;                   // RANGE_TOPK_GEMM=0: batches beyond 256 queries through the streaming scan too (A/B)
    DevBuf<float> ws_tg_gmax, ws_tg_theta;   // topk_gemm.h: group maxima (n_splits * 2, B, 4), thresholds (B)
    DevBuf<uint32_t> ws_tg_qfrag;            // ... the call's queries as fp16 fragments (512 B per query)
    DevBuf<float> ws_tg_qscale;              // ... and their scales
    DevBuf<uint32_t> ws_tg_cnt, ws_tg_ovf;
    DevBuf<uint2> ws_tg_cand;                  // candidate lists: lengths (B, lists), rows (B, lists, TG_CAP_L); overflow flags (B)
    int p2_splits_forced = 0;                // RANGE_P2_SPLITS=n: pass 2 with n bank splits (tuning)
    bool small_forward = true;               // RANGE_SMALL_FORWARD=0: batches of <= 32 queries take the two-pass kernels too (A/B)
    DevBuf<float> ws_small_o, ws_small_z;    // attend_small_kernel: per-workgroup partial products / weight sums
    DevBuf<uint32_t> ws_read_sink;           // range_stream_read_timed: one word per workgroup
    DevBuf<uint32_t> d_keys_bf16;            // bf16 copy of the keys in MFMA fragment order (8 KB per 16 rows)
    DevBuf<uint32_t> d_keys_f16;             // fp16 copy x tg_key_scale, same order: the batch top-k's (built on its first call)
    float tg_key_scale = 0.f;                // 0: not built for the current keys
    float key_norm_max = 1.f;                // largest |key row| (error bound of the prefilter)
    float xyz_norm_max = 1.f;                // largest |location row| (the geo head's logits must be <= 1 too)
    DevBuf<double> ws_ehat64, ws_h1, ws_h1a, ws_h2, ws_e3;
    int64_t ws_queries = 0;         // queries whose e-hat the workspace holds (range_forward* / range_encode_raw): range_topk_last
    DevBuf<uint32_t> ws_enc_sync;   // encoder_tile_kernel: 4 phase counters, 64 words apart
    // words of host memory the kernels can write (hipHostMallocMapped; async_err.h): set by a persistent
    // kernel whose bounded wait for other workgroups gave up; read - without synchronising - by the
    // next call and behind every synchronising exit (check_async_error)
    uint32_t* h_async_err = nullptr;
    uint32_t* d_async_err = nullptr;
    bool enc_fused = true;          // RANGE_ENC_FUSED=0: up to 16 queries take the separate small-batch kernels
    bool enc_fused_mid = true;      // RANGE_ENC_FUSED_MID=0: 513 .. 2 048 queries as three launches (A/B)
    int enc_fused_mid_min_wg = 2;   // RANGE_ENC_FUSED_MID_MIN_WG=n: ... one launch while a tile gets >= n workgroups (tuning)
    int last_qtiles = 0, last_splits = 0;
    bool p2_streamk = true;         // RANGE_P2_STREAMK=0: pass 2 as one workgroup per (bank split, query tile) (A/B)
    int p2_col_rows = 16384;        // RANGE_P2_COL_ROWS=n: largest bank column of the stream-K walk (rows; tuning)
    int p2_streamk_rows = 50000;    // RANGE_P2_STREAMK_ROWS=n: banks / shards up to n rows take the stream-K walk
    // host contract (range_forward_host): device result, pinned staging, copy stream, copy threads
    DevBuf<double> ws_out64;
    void* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    hipStream_t copy_stream = nullptr;
    std::unique_ptr<HostCopyPool> pool;
    bool host_timing = false;   // RANGE_HOST_TIMING=1: phase times of range_forward_host on stderr
    // profiling: event pairs per kernel kind
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof[RANGE_PROF_KINDS];
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t get_event() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    ~range_ctx() {
        if (h_stage) (void)hipHostFree(h_stage);
        if (h_async_err) (void)hipHostFree(h_async_err);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto& v : prof) for (auto& p : v) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        for (auto e : ev_pool) (void)hipEventDestroy(e);
    }
};

namespace {
// records an event pair around one kernel launch when profiling is on
struct ProfScope {
    range_ctx* c; int which; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(range_ctx* c_, int which_, hipStream_t s_) : c(c_), which(which_), s(s_) {
        if (c->profile) { a = c->get_event(); b = c->get_event(); if (a) (void)hipEventRecord(a, s); }
    }
    ~ProfScope() {
        if (a && b) { (void)hipEventRecord(b, s); c->prof[which].emplace_back(a, b); }
    }
};
}  // namespace

namespace {

template <typename K>
int set_dyn_lds(K kernel, size_t bytes) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return RANGE_OK;
}

// Small batches (fewer 16-query tiles than half the CUs): the first layer - 70 % of the weights
// a workgroup streams - is split over S column parts per tile on S times as many workgroups
// (encoder_l1_part_kernel), the rest follows per tile (encoder_rest_kernel).  One workgroup's
// serial chain over all weights takes 0.28 ms whatever the batch; this pair takes about half.
// the first layer's part kernels exist for parts of 64, 128, 256 and 512 columns
inline bool split_width_ok(int H, int S) {
    const int part = H / S;
    return H % S == 0 && (part == 64 || part == 128 || part == 256 || part == 512);
}

int launch_encoder_split(range_ctx* c, EncArgs a, int S, int KP, hipStream_t s) {
    const int tiles = (int)((a.B + 15) / 16);
    if (c->ws_h1.ensure((size_t)KP * tiles * 16 * a.H) != hipSuccess) return fail(RANGE_ERR_NOMEM, "out of device memory");
    a.h1 = c->ws_h1.p;
    a.n_parts = S;
    a.part_cols = a.H / S;
    a.n_kparts = KP;
    a.n_wg32 = 0;
    const size_t lds = c->enc_lds_bytes;
    const int ntp = a.part_cols / 64;
    int rc = RANGE_OK;
    ProfScope ps(c, RANGE_PROF_ENCODER, s);
    // up to 32 tiles (512 queries): all phases in ONE launch (encoder_tile_kernel), every tile on its own
    // workgroups, where a tile's first-layer workgroups are enough to carry its later phases (H / 64 of
    // them the second layer, 4 the last)
    // ... and, round 5, up to 128 tiles (2 048 queries: a rank's share of an 8-GPU batch, the last partial
    // round of a large batch): the 2-7 workgroups a tile then gets take the parts of the later phases in
    // turns (second-layer parts of 128 / 256 columns where a tile has < 8 / < 4 workgroups).  One launch
    // against three (tools/encoder_mid.py, steady state): 513 queries 83 us / 114, 800: 93 / 118, 1 024:
    // 102 / 122, 1 250: 128 / 132, 1 536 - 2 048: 155-156 / 157.  (The first version of this looked SLOWER
    // beyond 816 queries and cost the <= 512-query path 15 us: per-part copies of the argument struct inside
    // the phase loops had put 456 B of it into scratch memory - found through the latency log, now refused by
    // tests/test_host_cpu.py.)  Phase stamps of a tile at 1 250 queries (us): first layer 49, wait 11,
    // activation 4, second layer 21 (6 of them filling LDS), wait 10, last layer 16 (two parts of 64 outputs on
    // the tile's first workgroup), norm 3.  RANGE_ENC_FUSED_MID=0: three launches (A/B).
    const bool few_tiles = tiles <= 32 && S * KP >= std::max(std::max(a.H / 64, 4), (16 * a.H + 1023) / 1024);
    const bool mid_tiles = tiles > 32 && tiles <= 128 && S * KP >= c->enc_fused_mid_min_wg && c->enc_fused_mid;
    if ((few_tiles || mid_tiles) && a.n_layers == 2 && c->enc_fused && a.H % 64 == 0 && a.H <= 512 && tiles * S * KP <= c->n_cu) {
        if (c->ws_h2.ensure((size_t)tiles * 16 * a.H) != hipSuccess || c->ws_h1a.ensure((size_t)tiles * 16 * a.H) != hipSuccess ||
            c->ws_e3.ensure((size_t)tiles * 16 * ENC_EMBED + 64) != hipSuccess || c->ws_enc_sync.ensure(128 * 256) != hipSuccess)
            return fail(RANGE_ERR_NOMEM, "out of device memory");
        // (the counters wrap to zero by themselves, but a launch whose bounded spin gave up would leave
        // them poisoned for good: zeroed in front of every launch - 2 us of a ~55 us kernel - as the
        // guide asks of every polled word)
        HIP_TRY(hipMemsetAsync(c->ws_enc_sync.p, 0, (size_t)tiles * 256 * 4, s));
        a.h2 = c->ws_h2.p;
        a.h1a = c->ws_h1a.p;
        a.e3 = c->ws_e3.p;
        a.sync = c->ws_enc_sync.p;
        a.err = c->d_async_err ? c->d_async_err + RANGE_ASYNC_WORD_ENCODER : nullptr;
        a.debug_giveup = c->debug_giveup_next ? 1 : 0;
        c->debug_giveup_next = false;
        // second-layer parts: 64 columns where the tile has a workgroup for each, else 128 / 256
        a.part2_cols = 64;
        while (a.part2_cols < 256 && S * KP < a.H / a.part2_cols && a.H % (2 * a.part2_cols) == 0) a.part2_cols *= 2;
        a.n_parts2 = a.H / a.part2_cols;
        a.rest_from = 1;
#define RANGE_ENC_TILE(NTP, NWP)                                                               \
    case NTP:                                                                                  \
        rc = set_dyn_lds(encoder_tile_kernel<NTP, NWP>, lds);                                  \
        if (rc) return rc;                                                                     \
        hipLaunchKernelGGL((encoder_tile_kernel<NTP, NWP>), dim3(tiles * S * KP), dim3(ENC_PART_WAVES * 64), lds, s, a); \
        break;
        switch (ntp) {
            RANGE_ENC_TILE(1, 4)
            RANGE_ENC_TILE(2, 8)
            RANGE_ENC_TILE(4, 8)
            RANGE_ENC_TILE(8, 16)
            default: return fail(RANGE_ERR_INVALID, "internal: encoder part width %d", a.part_cols);
        }
#undef RANGE_ENC_TILE
        HIP_TRY(hipGetLastError());
#ifdef RANGE_EXP_ENC_STAMPS
        if (std::getenv("RANGE_ENC_STAMPS")) {
            unsigned long long h[12];
            HIP_TRY(hipStreamSynchronize(s));
            HIP_TRY(hipMemcpy(h, c->ws_e3.p + (size_t)tiles * 16 * ENC_EMBED, sizeof h, hipMemcpyDeviceToHost));   // (the stamps sit behind the tiles' rows)
            std::fprintf(stderr, "encoder_tile stamps (us after start):");
            for (int i = 1; i < 12; ++i) std::fprintf(stderr, " %d: %.1f", i, (double)(h[i] - h[0]) * 0.01);
            std::fprintf(stderr, "  [1 first layer, 2 sync, 3 activation, 4 sync, 5 second layer (10 its input in LDS, 11 its products), "
                         "6 sync, 7 last layer, 8 sync + norm]\n");
        }
#endif
        return RANGE_OK;
    }
#define RANGE_ENC_PART(NTP, NWP)                                                               \
    case NTP:                                                                                  \
        rc = set_dyn_lds(encoder_l1_part_kernel<NTP, NWP>, lds);                               \
        if (rc) return rc;                                                                     \
        hipLaunchKernelGGL((encoder_l1_part_kernel<NTP, NWP>), dim3(tiles * S * KP), dim3(ENC_PART_WAVES * 64), lds, s, a); \
        break;
    switch (ntp) {
        RANGE_ENC_PART(1, 4)
        RANGE_ENC_PART(2, 8)
        RANGE_ENC_PART(4, 8)
        RANGE_ENC_PART(8, 16)
        default: return fail(RANGE_ERR_INVALID, "internal: encoder part width %d", a.part_cols);
    }
#undef RANGE_ENC_PART
    HIP_TRY(hipGetLastError());
    // the second layer over S2 column parts per tile where there are CUs for it (and a second
    // hidden layer exists); the last kernel then starts from its output.  Every part re-reads the
    // tile's partial sums and re-activates them, and a third launch costs its ~10 us: measured
    // (tools/encoder_latency.py, RANGE_ENC_SPLIT2=0 for A/B) 16 queries 106 -> 121 us, 256 queries
    // equal, 625 queries 127 -> 118, 1 250 queries 145 -> 135, 2 048 queries 170 -> 162 us: from 32
    // tiles on.
    a.rest_from = 0;
    // A FEW tiles (up to 8: the latency regime - a handful of queries): one workgroup's chain over the
    // second and the last layer is 12.6 MFLOP of float64 MFMA on ONE CU, 70-90 us whatever the batch.
    // RANGE_ENC_SPLIT3=1 splits both layers too (second: column parts; last: 4 parts of 64 outputs) with
    // a one-wave-per-query kernel to normalise - four short launches instead of two.  MEASURED SLOWER:
    // 133 us against 111 us for 16 queries (round 3; round 2 saw the same with three launches): every
    // dependent launch costs ~10-20 us (dispatch, then 4-6 us before a kernel's first memory access
    // returns), more than the split saves.  Off by default; what would help is ONE persistent launch.
    const bool few = tiles <= 8 && a.n_layers == 2 && c->enc_split3 && c->enc_split2;
    int S2 = 1;
    for (int s2 = 2; s2 <= 8 && (tiles >= 32 || few) && tiles * s2 <= c->n_cu && a.n_layers >= 2; s2 *= 2) {
        const int part = a.H / s2;
        if (a.H % s2 == 0 && (part == 64 || part == 128 || part == 256)) S2 = s2;
    }
    if (S2 > 1 && c->enc_split2) {
        if (c->ws_h2.ensure((size_t)tiles * 16 * a.H) != hipSuccess) return fail(RANGE_ERR_NOMEM, "out of device memory");
        a.h2 = c->ws_h2.p;
        a.n_parts2 = S2;
        a.part2_cols = a.H / S2;
        a.rest_from = 1;
#define RANGE_ENC_PART2(NTP)                                                                   \
    case NTP:                                                                                  \
        rc = set_dyn_lds(encoder_l2_part_kernel<NTP>, lds);                                    \
        if (rc) return rc;                                                                     \
        hipLaunchKernelGGL((encoder_l2_part_kernel<NTP>), dim3(tiles * S2), dim3(NTP * 256), lds, s, a); \
        break;
        switch (a.part2_cols / 64) {
            RANGE_ENC_PART2(1)
            RANGE_ENC_PART2(2)
            RANGE_ENC_PART2(4)
            default: return fail(RANGE_ERR_INVALID, "internal: encoder part width %d", a.part2_cols);
        }
#undef RANGE_ENC_PART2
        HIP_TRY(hipGetLastError());
        if (few) {
            if (c->ws_e3.ensure((size_t)tiles * 16 * ENC_EMBED) != hipSuccess) return fail(RANGE_ERR_NOMEM, "out of device memory");
            a.e3 = c->ws_e3.p;
            rc = set_dyn_lds(encoder_l3_part_kernel, lds);
            if (rc) return rc;
            hipLaunchKernelGGL(encoder_l3_part_kernel, dim3(tiles * 4), dim3(256), lds, s, a);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(encoder_norm_kernel, dim3((unsigned)((a.B + 3) / 4)), dim3(256), 0, s, a);
            HIP_TRY(hipGetLastError());
            return RANGE_OK;
        }
    }
#define RANGE_ENC_REST(NT, NW)                                                                 \
    case NT:                                                                                   \
        rc = set_dyn_lds(encoder_rest_kernel<NT, NW>, lds);                                    \
        if (rc) return rc;                                                                     \
        hipLaunchKernelGGL((encoder_rest_kernel<NT, NW>), dim3(tiles), dim3(NW * 64), lds, s, a); \
        break;
    switch (a.H / 64) {
        RANGE_ENC_REST(1, 4)
        RANGE_ENC_REST(2, 4)
        RANGE_ENC_REST(4, RANGE_ENC_WAVES)
        RANGE_ENC_REST(6, 4)
        RANGE_ENC_REST(8, RANGE_ENC_WAVES)
        RANGE_ENC_REST(12, 16)
        RANGE_ENC_REST(16, 16)
        default: return fail(RANGE_ERR_INVALID, "internal: hidden width %d", a.H);
    }
#undef RANGE_ENC_REST
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int launch_encoder(range_ctx* c, const EncArgs& a_in, hipStream_t s) {
    EncArgs a = a_in;
    {
        // small batches: the first layer split over column parts and K ranges (host_plan.h)
        const int64_t tiles = (a.B + 15) / 16;
        int S, KP;
        choose_encoder_split(c->n_cu, a.n_slots, a.H, tiles, S, KP);
        if (S * KP > 1 && c->enc_split && split_width_ok(a.H, S)) return launch_encoder_split(c, a, S, KP, s);
    }
    const bool only16 = a.H > 512;       // (32 queries x H float64 of activations do not fit the LDS)
    // Workgroups take 32 queries and cost the same, one per CU at a time.  When the last round of
    // them would be less than half full, it is run with 16-query workgroups instead (about half
    // the time each): 10 000 queries = 256 x 32 + 113 x 16 instead of 313 x 32.
    const int64_t wg32 = (a.B + ENC_QTILE - 1) / ENC_QTILE;
    const int64_t full_rounds = wg32 / c->n_cu;
    const int64_t rem = a.B - full_rounds * c->n_cu * ENC_QTILE;      // queries after the full rounds
    // A tail of up to 2 048 queries after full rounds runs as the small-batch kernels (its tiles
    // spread over all CUs: ~0.16 ms for 1 808 queries) instead of a round of 16-query workgroups
    // (0.23 ms whatever its fill): 10 000 queries = 256 x 32 + a split tail of 113 tiles.
    auto main_plus_split_tail = [&](int64_t b_main) -> int {       // -1: not taken
        const int64_t tail = a.B - b_main;
        int S, KP;
        choose_encoder_split(c->n_cu, a.n_slots, a.H, (tail + 15) / 16, S, KP);
        if (S * KP <= 1 || !split_width_ok(a.H, S)) return -1;
        EncArgs m = a;
        m.B = b_main;
        int rc = launch_encoder(c, m, s);              // (whole rounds of equal workgroups)
        if (rc) return rc;
        EncArgs t = a;
        t.B = tail;
        t.lonlat = a.lonlat + 2 * b_main;
        t.ehat64 = a.ehat64 + ENC_EMBED * b_main;
        t.eraw64 = a.eraw64 ? a.eraw64 + ENC_EMBED * b_main : nullptr;
        t.ehat32 = a.ehat32 + ENC_EMBED * b_main;
        t.xq = a.xq + 4 * b_main;
        return launch_encoder_split(c, t, S, KP, s);
    };
    if (full_rounds > 0 && rem > 0 && rem <= 2048 && c->enc_split && c->enc_tail_split && !only16) {
        const int rc = main_plus_split_tail(a.B - rem);
        if (rc >= 0) return rc;
    }
    // A batch a little over one round of 16-query workgroups (4 097 .. 5 376 queries on 256 CUs: what a rank
    // of 2 encodes of BASELINE's batch) would fill 60 % of a round of 32-query workgroups and take that
    // round's whole time (0.40 ms): a full round of 16-query workgroups (0.24 ms) + a split tail (<= 0.13 ms)
    const int64_t round16 = (int64_t)16 * c->n_cu;
    // (the same behind full rounds of 32-query workgroups: 8 192 k + 4 097 .. 5 376 queries)
    if (rem > round16 && rem - round16 <= 1280 && c->enc_split && c->enc_tail_split && !only16) {
        const int rc = main_plus_split_tail(a.B - (rem - round16));
        if (rc >= 0) return rc;
    }
    int grid;
    if (only16 || a.B <= (int64_t)16 * c->n_cu) {
        // a batch that fits in one round either way: half-size workgroups on twice the CUs
        a.n_wg32 = 0;
        grid = (int)((a.B + 15) / 16);
    } else if (full_rounds > 0 && rem > 0 && rem <= (int64_t)16 * c->n_cu) {
        a.n_wg32 = (int32_t)(full_rounds * c->n_cu);
        grid = a.n_wg32 + (int)((rem + 15) / 16);
    } else {
        a.n_wg32 = (int32_t)wg32;
        grid = (int)wg32;
    }
    const size_t lds = c->enc_lds_bytes;
#define RANGE_ENC_CASE(NT, NW)                                                              \
    case NT: {                                                                              \
        int rc = set_dyn_lds(encoder_kernel<NT, NW>, lds);                                  \
        if (rc) return rc;                                                                  \
        hipLaunchKernelGGL((encoder_kernel<NT, NW>), dim3(grid), dim3(NW * 64), lds, s, a); \
        break;                                                                              \
    }
    ProfScope ps(c, RANGE_PROF_ENCODER, s);
    // 16 waves per workgroup where the hidden width allows (multiples of 256), else 4
    switch (a.H / 64) {
        RANGE_ENC_CASE(1, 4)
        RANGE_ENC_CASE(2, 4)
        RANGE_ENC_CASE(3, 4)
        RANGE_ENC_CASE(4, RANGE_ENC_WAVES)
        RANGE_ENC_CASE(5, 4)
        RANGE_ENC_CASE(6, 4)
        RANGE_ENC_CASE(7, 4)
        RANGE_ENC_CASE(8, RANGE_ENC_WAVES)
        RANGE_ENC_CASE(12, 16)
        RANGE_ENC_CASE(16, 16)
        default:
            return fail(RANGE_ERR_INVALID, "unsupported hidden width %d", a.H);
    }
#undef RANGE_ENC_CASE
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

// the three bf16 planes of the bank's values in MFMA fragment order (attend_bf16x3.h); 6 B per value
int build_vplanes(range_ctx* c) {
    const int64_t n_groups = (c->n_rows + 31) / 32;
    if (c->d_vplanes.ensure((size_t)n_groups * (PVB_GROUP_BYTES / 4)) != hipSuccess)
        return fail(RANGE_ERR_NOMEM, "out of device memory for the bf16 planes of the values (%lld MB)",
                    (long long)(n_groups * PVB_GROUP_BYTES >> 20));
    const int64_t threads = n_groups * 64 * 64;
    hipLaunchKernelGGL(vplanes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, c->d_values.p,
                       c->n_pad, n_groups, reinterpret_cast<u32x4*>(c->d_vplanes.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    c->vplanes_groups = n_groups;
    return RANGE_OK;
}

// preconditions of every kernel that forms softmax weights with the constant shift m = tau * log2(e)
int check_softmax_args(range_ctx* c, int64_t B, float tau_sem, float tau_geo) {
    if (!c->has_bank) return fail(RANGE_ERR_STATE, "bank not set (range_set_bank)");
    if (!c->has_values)
        return fail(RANGE_ERR_STATE, "keys-only bank (range_set_keys): only range_topk_stream runs on it");
    if (B <= 0) return fail(RANGE_ERR_INVALID, "B must be > 0");
    if (!(tau_sem > 0.f)) return fail(RANGE_ERR_INVALID, "tau_sem must be > 0");
    // the constant shift m = tau * log2(e) of the softmax statistics needs every logit <= 1:
    // unit keys (range/range.py:85-89).  A bank that skipped that preparation would overflow.
    // (written as !(x <= 1.001): a row holding NaN or infinity makes the largest norm NaN / inf and is
    // refused too - the reference would return NaN for EVERY query of every batch against such a bank,
    // range/range.py:213-215: one NaN logit poisons each softmax row)
    if (!(c->key_norm_max <= 1.001f))
        return fail(RANGE_ERR_INVALID, "bank keys are not L2-normalised or not finite (largest row norm %.4f): the softmax of "
                    "range_scan_stats / range_attend needs unit keys, as range/range.py:85-89 prepares them "
                    "(range_topk_stream accepts any norm)", (double)c->key_norm_max);
    if (tau_geo > 0.f && !(c->xyz_norm_max <= 1.001f))
        return fail(RANGE_ERR_INVALID, "bank locations are not unit vectors or not finite (largest row norm %.4f): the geographic "
                    "softmax needs them as range/utils/utils.py:11-16 computes them", (double)c->xyz_norm_max);
    return RANGE_OK;
}

int fill_scan_args(range_ctx* c, ScanArgs& a, const float* ehat32, const float* xq, int64_t B,
                   float tau_sem, float tau_geo, bool pass1, int p1_max_splits = 128) {
    int rc0 = check_softmax_args(c, B, tau_sem, tau_geo);
    if (rc0) return rc0;
    // the softmax statistics use the constant shift m = tau * log2(e) (scan_stats_kernel): the
    // smallest term 2^(-2m) must stay a normal float32
    if (tau_sem > RANGE_MAX_TAU || tau_geo > RANGE_MAX_TAU)
        return fail(RANGE_ERR_INVALID, "temperatures above %g are not supported (the reference uses 12, 15 and 40)",
                    (double)RANGE_MAX_TAU);
    const double LOG2E = 1.4426950408889634;
    a.keys = c->d_keys.p;
    a.xyz4 = c->d_xyz4.p;
    a.values = c->d_values.p;
    a.ehat = ehat32;
    a.xq = xq;
    a.B = B;
    a.n_valid = c->n_rows;
    a.n_blocks = (int32_t)((c->n_rows + BLK - 1) / BLK);
    a.n_qtiles = (int32_t)((B + QTILE - 1) / QTILE);
    // pass 1 writes 16 B per (query, split): many splits are free; pass 2 writes a 4 KB row
    // (a small batch may split pass 2 further, until every CU has a workgroup).  Every extra
    // split of pass 2 writes and re-reads a 4 KB row per query: 8 KB at ~4 TB/s against the
    // query's MFMA time n_rows * 2054 FLOP / 140 TFLOP/s, i.e. 140 / n_rows of the launch - small
    // for the whole bank on one GPU, 1 % per split for a 12 500-row shard.
    a.n_splits = pass1 ? choose_splits(a.n_qtiles, a.n_blocks, c->n_cu, RANGE_P1_WG_PER_CU, p1_max_splits)
                       : choose_splits(a.n_qtiles, a.n_blocks, c->n_cu, 1,
                                       std::max(32, std::min(512, (c->n_cu + a.n_qtiles - 1) / a.n_qtiles)),
                                       std::max(0.001, 140.0 / (double)c->n_rows));
    if (!pass1 && c->p2_splits_forced > 0) a.n_splits = std::max(1, std::min(c->p2_splits_forced, std::max(1, a.n_blocks / 4)));
    a.k_sem = (float)(tau_sem * LOG2E);
    a.k_geo = tau_geo > 0.f ? (float)(tau_geo * LOG2E) : 0.f;
    a.beta = 1.f;
    a.stats = nullptr;
    a.out = nullptr;
    a.cand_val = nullptr;
    a.cand_idx = nullptr;
    a.logits = nullptr;
    a.qt_offset = 0;
    a.rowmax = nullptr;
    a.sk_groups = 0;
    a.sk_cols = 1;
    if (!pass1) {
        c->last_qtiles = a.n_qtiles;
        c->last_splits = a.n_splits;
    }
    return RANGE_OK;
}

}  // namespace

extern "C" {

int range_abi_version(void) { return RANGE_ABI_VERSION; }
const char* range_last_error(void) { return g_err.c_str(); }
#ifndef RANGE_BUILD_FLAGS
#define RANGE_BUILD_FLAGS ""
#endif
const char* range_build_flags(void) { return RANGE_BUILD_FLAGS; }
#ifndef RANGE_SRC_SHA256
#define RANGE_SRC_SHA256 "unknown"
#endif
// (the literal is also what range_amd/_srchash.py: library_stamp() finds in the file without loading it)
static const char g_src_stamp[] = "RANGE_SRC_SHA256=" RANGE_SRC_SHA256;
const char* range_source_sha256(void) { return g_src_stamp + 17; }

int range_create(int device, range_ctx** out) {
    if (!out) return fail(RANGE_ERR_INVALID, "out is null");
    *out = nullptr;
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (device < 0 || device >= count)
        return fail(RANGE_ERR_INVALID, "device %d out of range (%d visible)", device, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RANGE_ERR_INVALID, "device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);
    range_ctx* c = new (std::nothrow) range_ctx();
    if (!c) return fail(RANGE_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const char* keep = std::getenv("RANGE_KEEP_LOGITS");
    c->allow_keep = !(keep && keep[0] == '0');
    if (const char* e = std::getenv("RANGE_HOST_TIMING")) c->host_timing = e[0] == '1';
    if (const char* e = std::getenv("RANGE_ENC_SPLIT")) c->enc_split = e[0] != '0';
    if (const char* e = std::getenv("RANGE_ENC_SPLIT2")) c->enc_split2 = e[0] != '0';
    if (const char* e = std::getenv("RANGE_ENC_SPLIT3")) c->enc_split3 = e[0] != '0';
    if (const char* e = std::getenv("RANGE_ENC_FUSED")) c->enc_fused = e[0] != '0';
    if (const char* e = std::getenv("RANGE_ENC_FUSED_MID")) c->enc_fused_mid = e[0] != '0';
    if (const char* e = std::getenv("RANGE_ENC_FUSED_MID_MIN_WG")) c->enc_fused_mid_min_wg = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("RANGE_ENC_TAIL")) c->enc_tail_split = e[0] != '0';
    if (const char* e = std::getenv("RANGE_TOPKS_GROUPS")) c->topks_groups = std::atoi(e);
    if (const char* e = std::getenv("RANGE_TOPKS_FORCE_EXACT")) c->topks_force_exact = e[0] == '1';
    if (const char* e = std::getenv("RANGE_TOPKS_KEYS")) c->topks_bf16 = std::strcmp(e, "f32") != 0;
    if (const char* e = std::getenv("RANGE_TOPKS_FUSED")) c->topks_fused = e[0] != '0';
    if (const char* e = std::getenv("RANGE_SMALL_FORWARD")) c->small_forward = e[0] != '0';
    if (const char* e = std::getenv("RANGE_TOPK_GEMM")) c->topk_gemm = e[0] != '0';
    if (const char* e = std::getenv("RANGE_TG_SAMPLE")) c->tg_sample = std::max(1, std::min(16, std::atoi(e)));
    if (const char* e = std::getenv("RANGE_P2_SPLITS")) c->p2_splits_forced = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("RANGE_P2_STREAMK")) c->p2_streamk = e[0] != '0';
    if (const char* e = std::getenv("RANGE_P2_COL_ROWS")) c->p2_col_rows = std::max(64, std::atoi(e));
    if (const char* e = std::getenv("RANGE_P2_STREAMK_ROWS")) c->p2_streamk_rows = std::max(0, std::atoi(e));
    {
        DeviceGuard g(device);
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) == hipSuccess) {
            std::memset(hp, 0, 64);
            if (hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
                c->h_async_err = (uint32_t*)hp;
                c->d_async_err = (uint32_t*)dp;
            } else {
                (void)hipHostFree(hp);
            }
        }
        (void)hipGetLastError();       // (without the word the kernels simply do not report)
    }
    *out = c;
    return RANGE_OK;
}

void range_destroy(range_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipDeviceSynchronize();
    delete ctx;
}

int64_t range_bank_rows(const range_ctx* ctx) { return ctx ? ctx->n_rows : 0; }

int range_set_encoder(range_ctx* c, const range_encoder_desc* d, const double* const* weights,
                      const double* const* biases) {
    if (!c || !d || !weights || !biases) return fail(RANGE_ERR_INVALID, "null argument");
    const int L = d->legendre_polys, Hc = d->hidden, NL = d->num_hidden_layers, E = d->embed_dim;
    if (L < 1 || L > 64) return fail(RANGE_ERR_INVALID, "legendre_polys %d unsupported (1..64)", L);
    // (`capacity` of the real checkpoint is unknown: a width no kernel exists for runs zero-padded as
    // the next one that has - host_plan.h: kernel_hidden_width - at the padded width's cost)
    const int H = kernel_hidden_width(Hc);
    if (H == 0) return fail(RANGE_ERR_INVALID, "hidden %d unsupported (1..1024)", Hc);
    if (NL < 1 || NL + 1 > ENC_MAX_LAYERS) return fail(RANGE_ERR_INVALID, "num_hidden_layers %d unsupported", NL);
    if (E != ENC_EMBED) return fail(RANGE_ERR_INVALID, "embed_dim %d unsupported (must be 256)", E);
    if (d->sh_mode != RANGE_SH_ANALYTIC && d->sh_mode != RANGE_SH_CLOSED_FORM)
        return fail(RANGE_ERR_INVALID, "unknown sh_mode %d", d->sh_mode);
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);

    // ---- slot plan, recurrence tables, weight packing: pure host arithmetic (host_plan.h, also
    //      built and run under sanitizers on the CPU: tests/native/host_sanitize.cpp)
    EncoderPlan plan;
    if (!build_encoder_plan(L, ENC_SLOTS_PER_ROUND, plan)) return fail(RANGE_ERR_INVALID, "internal: slot plan is not a permutation");
    const std::vector<int>& perm = plan.perm;
    const std::vector<int32_t>& slot_base = plan.slot_base;
    const int n_slots = plan.n_slots, n_rounds = plan.n_rounds, Kp = (int)perm.size();
    // (hidden widths beyond 512 run 16-query workgroups only, their activations packed densely)
    const int lds_main = (H > 512 ? 16 : ENC_QTILE) * std::max(plan.max_round, H);
    const size_t lds_bytes = (size_t)(lds_main + 16 * ENC_QTILE) * sizeof(double);   // + [<= 16 waves][32] partial norms
    if (lds_bytes > 160 * 1024) return fail(RANGE_ERR_INVALID, "encoder shape needs %zu B of LDS (>160 KiB)", lds_bytes);
    std::vector<double> coefA, coefB, seedc;
    recurrence_tables(L, d->sh_mode == RANGE_SH_ANALYTIC, coefA, coefB, seedc);
    auto pack = [](const double* W, int n_out, int k_in, const std::vector<int>* kperm, int Kpad) {
        return pack_weights(W, n_out, k_in, kperm, Kpad);
    };
    for (int i = 0; i <= NL; ++i) if (!weights[i] || !biases[i]) return fail(RANGE_ERR_INVALID, "weights[%d]/biases[%d] null", i, i);
    auto padded = [](const double* W, int n_out, int k_in, int n_pad, int k_pad) { return pad_weights(W, n_out, k_in, n_pad, k_pad); };
    if (H == Hc) {
        HIP_TRY(c->d_wp[0].upload(pack(weights[0], H, L * L, &perm, Kp)));
        for (int i = 1; i < NL; ++i) HIP_TRY(c->d_wp[i].upload(pack(weights[i], H, H, nullptr, H)));
        HIP_TRY(c->d_wp[NL].upload(pack(weights[NL], E, H, nullptr, H)));
    } else {
        HIP_TRY(c->d_wp[0].upload(pack(padded(weights[0], Hc, L * L, H, L * L).data(), H, L * L, &perm, Kp)));
        for (int i = 1; i < NL; ++i) HIP_TRY(c->d_wp[i].upload(pack(padded(weights[i], Hc, Hc, H, H).data(), H, H, nullptr, H)));
        HIP_TRY(c->d_wp[NL].upload(pack(padded(weights[NL], E, Hc, E, H).data(), E, H, nullptr, H)));
    }
    for (int i = 0; i <= NL; ++i) {
        std::vector<double> b((size_t)(i < NL ? H : E), 0.0);
        std::copy(biases[i], biases[i] + (i < NL ? Hc : E), b.begin());
        HIP_TRY(c->d_bias[i].upload(b));
    }
    HIP_TRY(c->d_slot_base.upload(slot_base));
    HIP_TRY(c->d_coefA.upload(coefA));
    HIP_TRY(c->d_coefB.upload(coefB));
    HIP_TRY(c->d_seedc.upload(seedc));

    EncArgs& a = c->enc;
    a = EncArgs{};
    a.L = L;
    a.n_slots = n_slots;
    a.n_rounds = n_rounds;
    a.n_layers = NL;
    a.H = H;
    a.kp0_total = Kp / 8;
    a.lds_main_doubles = lds_main;
    a.slot_base = c->d_slot_base.p;
    a.coefA = c->d_coefA.p;
    a.coefB = c->d_coefB.p;
    a.seedc = c->d_seedc.p;
    for (int i = 0; i <= NL; ++i) { a.wp[i] = c->d_wp[i].p; a.bias[i] = c->d_bias[i].p; }
    c->enc_lds_bytes = c->enc_lds_base = lds_bytes;
    c->desc = *d;
    c->has_encoder = true;
    return RANGE_OK;
}

int range_set_sh_table(range_ctx* c, int32_t L, const double* front, const double* a0, const double* a2,
                       const int32_t* p2, const int32_t* kx, const int32_t* off, const int32_t* cnt,
                       int64_t n_terms, const double* coef, const int32_t* pw) {
    if (!c || !front || !a0 || !a2 || !p2 || !kx || !off || !cnt || (n_terms > 0 && (!coef || !pw)))
        return fail(RANGE_ERR_INVALID, "null argument");
    if (!c->has_encoder) return fail(RANGE_ERR_STATE, "encoder not set (range_set_encoder)");
    if (c->desc.sh_mode != RANGE_SH_ANALYTIC)
        return fail(RANGE_ERR_INVALID, "the coefficient table belongs to the 'analytic' spherical harmonics");
    if (L != c->desc.legendre_polys) return fail(RANGE_ERR_INVALID, "table for L=%d, encoder has L=%d", L, c->desc.legendre_polys);
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    std::vector<SHDesc> desc((size_t)L * L);
    for (int l = 0; l < L; ++l)
        for (int m = 0; m <= l; ++m) {
            const int i = l * L + m;
            if (cnt[i] < 0 || off[i] < 0 || (int64_t)off[i] + cnt[i] > n_terms || p2[i] < 0 || p2[i] > 127 ||
                kx[i] < 0 || kx[i] >= L || cnt[i] > 32767)
                return fail(RANGE_ERR_INVALID, "bad table entry (l=%d, m=%d)", l, m);
            desc[i] = SHDesc{front[i], a0[i], a2[i], off[i], (int16_t)cnt[i], (int8_t)p2[i], (int8_t)kx[i]};
        }
    for (int64_t j = 0; j < n_terms; ++j)
        if (pw[j] < 0 || pw[j] >= L) return fail(RANGE_ERR_INVALID, "power %d out of range at term %lld", pw[j], (long long)j);
    const size_t lds = c->enc_lds_base + (size_t)ENC_QTILE * L * sizeof(double);
    if (lds > 160 * 1024) return fail(RANGE_ERR_INVALID, "encoder shape needs %zu B of LDS with the power table (>160 KiB)", lds);
    HIP_TRY(c->d_sh_desc.upload(desc));
    // (L <= 2: every function is a monomial and there are no terms - upload one zero, read nothing)
    HIP_TRY(c->d_sh_coef.upload(n_terms > 0 ? std::vector<double>(coef, coef + n_terms) : std::vector<double>(1, 0.0)));
    HIP_TRY(c->d_sh_pow.upload(n_terms > 0 ? std::vector<int32_t>(pw, pw + n_terms) : std::vector<int32_t>(1, 0)));
    c->enc.sh_desc = c->d_sh_desc.p;
    c->enc.sh_coef = c->d_sh_coef.p;
    c->enc.sh_pow = c->d_sh_pow.p;
    c->enc_lds_bytes = lds;
    return RANGE_OK;
}

// keys -> device (float32 rows + the bf16 fragment copy the prefilter of range_topk_stream
// streams) and the largest row norm, computed on the device.  `keys` may be a host or a device
// pointer (hipMemcpyDefault).
static int upload_keys(range_ctx* c, const float* keys, int64_t n_rows, int64_t n_pad) {
    c->tg_key_scale = 0.f;       // (the batch top-k rebuilds its fp16 copy for the new keys)
    HIP_TRY(c->d_keys.ensure((size_t)n_pad * KEY_DIM));
    if (n_pad > n_rows)
        HIP_TRY(hipMemset(c->d_keys.p + (size_t)n_rows * KEY_DIM, 0, (size_t)(n_pad - n_rows) * KEY_DIM * 4));
    HIP_TRY(hipMemcpy(c->d_keys.p, keys, (size_t)n_rows * KEY_DIM * 4, hipMemcpyDefault));
    const int64_t n_tiles = n_pad / BLK;
    HIP_TRY(c->d_keys_bf16.ensure((size_t)n_tiles * (TSB_TILE_BYTES / 4)));
    const int64_t threads = n_tiles * 8 * 64;
    hipLaunchKernelGGL(keyfrag_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, c->d_keys.p,
                       n_pad, n_tiles, reinterpret_cast<ts_u32x4*>(c->d_keys_bf16.p));
    HIP_TRY(hipGetLastError());
    if (!c->ws_topk_sync.p) {
        HIP_TRY(c->ws_topk_sync.ensure(TOPKS_SYNC_WORDS));
        HIP_TRY(hipMemset(c->ws_topk_sync.p, 0, TOPKS_SYNC_WORDS * 4));
        std::memset(c->topk_sync_base, 0, sizeof c->topk_sync_base);
    }
    uint32_t* scratch = c->ws_topk_sync.p + TOPKS_SYNC_SCRATCH;
    HIP_TRY(hipMemset(scratch, 0, 4));
    hipLaunchKernelGGL(key_norm_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, 0, c->d_keys.p, n_rows, scratch);
    HIP_TRY(hipGetLastError());
    float n2max = 0.f;
    HIP_TRY(hipMemcpy(&n2max, scratch, 4, hipMemcpyDeviceToHost));   // (synchronises)
    // (a bound: the float32 sum of squares is within 3e-5 of the exact one)
    c->key_norm_max = (float)(std::sqrt((double)n2max) * 1.0001);
    return RANGE_OK;
}

int range_set_bank(range_ctx* c, const float* keys, const float* values, const float* xyz,
                   int64_t n_rows, int64_t row_offset) {
    if (!c || !keys || !values || !xyz) return fail(RANGE_ERR_INVALID, "null argument");
    if (n_rows <= 0) return fail(RANGE_ERR_INVALID, "n_rows must be > 0");
    if (n_rows >= (int64_t)1 << 31) return fail(RANGE_ERR_INVALID, "n_rows too large");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const int64_t n_pad = (n_rows + BLK - 1) / BLK * BLK;
    c->has_bank = false;
    c->has_values = false;
    c->kept_B = 0;
    HIP_TRY(c->d_values.ensure((size_t)n_pad * VAL_DIM));
    HIP_TRY(c->d_xyz4.ensure((size_t)n_pad * 4));
    HIP_TRY(hipMemset(c->d_values.p, 0, (size_t)n_pad * VAL_DIM * 4));
    HIP_TRY(hipMemset(c->d_xyz4.p, 0, (size_t)n_pad * 16));
    HIP_TRY(hipMemcpy(c->d_values.p, values, (size_t)n_rows * VAL_DIM * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy2D(c->d_xyz4.p, 16, xyz, 12, 12, (size_t)n_rows, hipMemcpyHostToDevice));
    {
        double n2max = 0.0;
        for (int64_t r = 0; r < n_rows; ++r) {
            const float* x = xyz + 3 * r;
            n2max = std::max(n2max, (double)x[0] * x[0] + (double)x[1] * x[1] + (double)x[2] * x[2]);
        }
        c->xyz_norm_max = (float)std::sqrt(n2max);
    }
    int rc = upload_keys(c, keys, n_rows, n_pad);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    c->n_rows = n_rows;
    c->n_pad = n_pad;
    c->row_offset = row_offset;
    c->has_bank = true;
    c->has_values = true;
    c->vplanes_groups = 0;
    if (c->pv_mode == RANGE_PV_BF16X3) return build_vplanes(c);
    return RANGE_OK;
}

int range_set_keys(range_ctx* c, const float* keys, int64_t n_rows, int64_t row_offset) {
    if (!c || !keys) return fail(RANGE_ERR_INVALID, "null argument");
    if (n_rows <= 0) return fail(RANGE_ERR_INVALID, "n_rows must be > 0");
    if (n_rows >= (int64_t)1 << 31) return fail(RANGE_ERR_INVALID, "n_rows too large");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const int64_t n_pad = (n_rows + BLK - 1) / BLK * BLK;
    c->has_bank = false;
    c->has_values = false;
    c->kept_B = 0;
    c->d_values.release();
    c->d_xyz4.release();
    c->d_vplanes.release();
    c->vplanes_groups = 0;
    int rc = upload_keys(c, keys, n_rows, n_pad);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    c->n_rows = n_rows;
    c->n_pad = n_pad;
    c->row_offset = row_offset;
    c->has_bank = true;
    return RANGE_OK;
}

int range_set_pv_mode(range_ctx* c, int32_t mode) {
    if (!c) return fail(RANGE_ERR_INVALID, "null argument");
    if (mode != RANGE_PV_EXACT && mode != RANGE_PV_BF16X3) return fail(RANGE_ERR_INVALID, "unknown pv mode %d", mode);
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    c->pv_mode = mode;
    if (mode == RANGE_PV_BF16X3 && c->has_bank && c->vplanes_groups == 0) return build_vplanes(c);
    return RANGE_OK;
}
int32_t range_get_pv_mode(const range_ctx* c) { return c ? c->pv_mode : -1; }

// A persistent kernel that gave up waiting for its other workgroups (possible only when something
// else holds the GPU's CUs for seconds) has written NaN rows (encoder) / NaN values and -1 indices
// (top-k) for what it could not finish, and set a word of host memory.  That word is read here - in
// front of the next encoder / top-k call and behind every synchronising exit, without touching the
// stream - and the context takes the separate-launch form of that kernel from then on: co-residency
// of a persistent grid is a precondition the library cannot check, so after one failure it stops
// relying on it (the same branch RANGE_ENC_FUSED=0 / RANGE_TOPKS_FUSED=0 select).
static int check_async_error(range_ctx* c) {
    if (!c->h_async_err) return RANGE_OK;
    volatile uint32_t* w = c->h_async_err;
    const bool enc = w[RANGE_ASYNC_WORD_ENCODER] != 0, topk = w[RANGE_ASYNC_WORD_TOPK] != 0;
    if (!enc && !topk) return RANGE_OK;
    w[RANGE_ASYNC_WORD_ENCODER] = 0;
    w[RANGE_ASYNC_WORD_TOPK] = 0;
    if (enc) c->enc_fused = false;
    if (topk) c->topks_fused = false;
    return fail(RANGE_ERR_HIP, "a persistent %s launch of an earlier call gave up waiting for its workgroups (is another "
                               "process holding the GPU?); the rows it could not finish were written as NaN%s, and this "
                               "context runs that step as separate launches from now on: re-issue the call",
                enc && topk ? "encoder and a top-k" : enc ? "encoder" : "top-k", topk ? " / index -1" : "");
}

int range_check_async_error(range_ctx* c) {
    if (!c) return fail(RANGE_ERR_INVALID, "null argument");
    return check_async_error(c);
}

// test hook: the NEXT persistent launch of this context (the one-launch encoder of up to 512 queries,
// or the fused top-k) behaves as if its bounded in-launch wait had expired - the real give-up path:
// NaN / -1 outputs, the host-mapped word, the fall-back (tests/test_gpu_round2.py)
int range_debug_raise_async_error(range_ctx* c, range_stream_t) {
    if (!c) return fail(RANGE_ERR_INVALID, "null argument");
    if (!c->d_async_err) return fail(RANGE_ERR_STATE, "no host-mapped error word in this context");
    c->debug_giveup_next = true;
    return RANGE_OK;
}

// The give-up words as DATA, in stream order (range_hip.h: range_async_error_flag): a rank of a
// row-sharded job sends this with its rows, so that every rank learns from the WORD - not from NaN in
// the data, which a NaN coordinate produces as well - that a peer's persistent launch gave up.
__global__ void async_flag_kernel(const uint32_t* __restrict__ err, double* __restrict__ flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        flag[0] = (__hip_atomic_load(err + RANGE_ASYNC_WORD_ENCODER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) |
                   __hip_atomic_load(err + RANGE_ASYNC_WORD_TOPK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) ? 1.0 : 0.0;
}

int range_async_error_flag(range_ctx* c, double* flag_dev, range_stream_t stream) {
    if (!c || !flag_dev) return fail(RANGE_ERR_INVALID, "null argument");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    if (!c->d_async_err) {            // (no host-mapped words in this context: nothing can have been reported)
        HIP_TRY(hipMemsetAsync(flag_dev, 0, sizeof(double), (hipStream_t)stream));
        return RANGE_OK;
    }
    hipLaunchKernelGGL(async_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c->d_async_err, flag_dev);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

static int encode_impl(range_ctx* c, const double* lonlat, int64_t B, double* ehat64, float* ehat32,
                       float* xq32, double* eraw64, range_stream_t stream) {
    if (!c || !lonlat || !ehat64 || !ehat32 || !xq32) return fail(RANGE_ERR_INVALID, "null argument");
    if (!c->has_encoder) return fail(RANGE_ERR_STATE, "encoder not set (range_set_encoder)");
    if (B <= 0) return fail(RANGE_ERR_INVALID, "B must be > 0");
    if (int rc = check_async_error(c)) return rc;
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    EncArgs a = c->enc;
    a.lonlat = lonlat;
    a.ehat64 = ehat64;
    a.eraw64 = eraw64;
    a.ehat32 = ehat32;
    a.xq = xq32;
    a.B = B;
    return launch_encoder(c, a, (hipStream_t)stream);
}

int range_encode(range_ctx* c, const double* lonlat, int64_t B, double* ehat64, float* ehat32,
                 float* xq32, range_stream_t stream) {
    return encode_impl(c, lonlat, B, ehat64, ehat32, xq32, nullptr, stream);
}

int range_encode_raw(range_ctx* c, const double* lonlat, int64_t B, double* eraw64,
                     range_stream_t stream) {
    if (!c || !eraw64) return fail(RANGE_ERR_INVALID, "null argument");
    if (B <= 0) return fail(RANGE_ERR_INVALID, "B must be > 0");
    {
        DeviceGuard g(c->device);
        HIP_TRY(c->ws_ehat64.ensure((size_t)B * 256));
        HIP_TRY(c->ws_ehat32.ensure((size_t)B * 256));
        HIP_TRY(c->ws_xq.ensure((size_t)B * 4));
    }
    c->ws_queries = 0;
    int rc = encode_impl(c, lonlat, B, c->ws_ehat64.p, c->ws_ehat32.p, c->ws_xq.p, eraw64, stream);
    if (rc == RANGE_OK) c->ws_queries = B;
    return rc;
}

// Training-free coordinate encoders of the reference (range/range.py:262-272): one thread per
// location, float64.  mode 0 'Direct' (:262-264): (lon,lat)*pi/180, evaluated as (x*pi)/180 like
// the Python expression; mode 1 'Cartesian_3D' (:265-268, utils/utils.py:11-16): unit xyz of the
// radians above; mode 2 'Wrap' (positional_encoding/wrap.py:20-29): cos/sin of torch.deg2rad(x)
// = x * (pi/180) per column, ordered (cos lon, sin lon, cos lat, sin lat).
__global__ void coord_features_kernel(int mode, const double* __restrict__ lonlat, int64_t n,
                                      double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lon = lonlat[2 * i], lat = lonlat[2 * i + 1];
    constexpr double PI = 3.141592653589793;
    if (mode == 2) {
        constexpr double PI_180 = 0.017453292519943295;
        const double a = lon * PI_180, b = lat * PI_180;
        out[4 * i + 0] = cos(a);
        out[4 * i + 1] = sin(a);
        out[4 * i + 2] = cos(b);
        out[4 * i + 3] = sin(b);
        return;
    }
    const double a = (lon * PI) / 180.0, b = (lat * PI) / 180.0;
    if (mode == 0) {
        out[2 * i + 0] = a;
        out[2 * i + 1] = b;
    } else {
        const double cb = cos(b);
        out[3 * i + 0] = cb * cos(a);
        out[3 * i + 1] = cb * sin(a);
        out[3 * i + 2] = sin(b);
    }
}

int range_coord_features(range_ctx* c, int32_t mode, const double* lonlat, int64_t B, double* out,
                         range_stream_t stream) {
    if (!c || !lonlat || !out) return fail(RANGE_ERR_INVALID, "null argument");
    if (mode < 0 || mode > 2) return fail(RANGE_ERR_INVALID, "coordinate encoder mode %d", mode);
    if (B <= 0) return fail(RANGE_ERR_INVALID, "B must be > 0");
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(coord_features_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, mode, lonlat, B, out);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_blend(range_ctx* c, const float* G, const float* H, float beta, int64_t B, float* out,
                range_stream_t stream) {
    if (!c || !G || !H || !out) return fail(RANGE_ERR_INVALID, "null argument");
    if (B <= 0) return fail(RANGE_ERR_INVALID, "B must be > 0");
    DeviceGuard g(c->device);
    const int64_t n4 = B * (VAL_DIM / 4);
    hipLaunchKernelGGL(blend_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, G, H, beta, n4, out);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

// first_query / total_queries / force_splits: range_scan_stats_at (a scan in chunks whose
// kept logits share one workspace); a plain range_scan_stats is the chunk [0, B) of a scan of B.
static int scan_stats_impl(range_ctx* c, const float* ehat32, const float* xq32, int64_t B, float tau_sem,
                           float tau_geo, float* stats, int topk, float* topk_val, int64_t* topk_idx,
                           int32_t keep_logits, int64_t first_query, int64_t total_queries, int32_t force_splits,
                           range_stream_t stream) {
    if (!c || !ehat32 || !xq32 || !stats) return fail(RANGE_ERR_INVALID, "null argument");
    if (topk < 0 || topk > MAX_TOPK) return fail(RANGE_ERR_INVALID, "topk must be in [0,%d]", MAX_TOPK);
    if (topk > 0 && (!topk_val || !topk_idx)) return fail(RANGE_ERR_INVALID, "topk outputs null");
    if (first_query < 0 || first_query % QTILE != 0)
        return fail(RANGE_ERR_INVALID, "first_query must be a non-negative multiple of %d", QTILE);
    if (B > 0 && first_query + B > total_queries)
        return fail(RANGE_ERR_INVALID, "queries [%lld, %lld) exceed the scan's %lld", (long long)first_query,
                    (long long)(first_query + B), (long long)total_queries);
    if (force_splits < 0) return fail(RANGE_ERR_INVALID, "n_splits must be >= 0");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    ScanArgs a{};
    hipStream_t s = (hipStream_t)stream;
    // a later chunk extends the scan only in order, and only when the first chunk could keep
    const bool extends = first_query > 0 && c->kept_B == first_query && c->kept_total == total_queries;
    if (first_query > 0 && !extends) keep_logits = 0;
    if (!extends) c->kept_B = 0;
    // small batches are HBM-bound: many splits so that every CU streams a share of the keys
    const bool few = B <= 4 * QTILE;
    // The logits of this call are written to HBM when the caller asks for them (plain scan) or
    // when a larger batch wants its top-k: pass 1 then runs WITHOUT the in-loop list maintenance
    // (which costs more than its MFMAs) and a streaming selection over the kept logits follows.
    // 4 B per (query, bank row) of this context's shard; not done when that would take more than
    // half of the free device memory (pass 2 then recomputes, top-k uses the in-scan lists).
    bool write_logits = c->allow_keep && (topk > 0 ? B > RANGE_TOPK_INSCAN_MAX : keep_logits != 0);
    const int64_t n_qtiles = (total_queries + QTILE - 1) / QTILE, n_blocks = (c->n_rows + BLK - 1) / BLK;
    if (write_logits && extends) {
        // (the workspace was sized for the whole scan by its first chunk)
        if ((size_t)n_qtiles * n_blocks * 1024 > c->ws_logits.n || c->kept_blocks != n_blocks) { write_logits = false; c->kept_B = 0; }
    } else if (write_logits) {
        const size_t need = (size_t)n_qtiles * n_blocks * 1024;
        if (need > c->ws_logits.n) {        // (hipMemGetInfo is slow: only when growing)
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            if (need * sizeof(float) <= free_b / 2) {
                HIP_TRY(c->ws_logits.ensure(need));
            } else {
                write_logits = false;
                if (!c->warned_no_keep) {   // once per context: pass 2 will recompute the logits (25 % more MFMAs)
                    c->warned_no_keep = true;
                    std::fprintf(stderr, "librange_hip: the logits of %lld queries x %lld bank rows (%.1f GB) do not fit in "
                                 "half of the free device memory (%.1f GB free): not kept, pass 2 recomputes them\n",
                                 (long long)B, (long long)c->n_rows, need * 4e-9, free_b * 1e-9);
                }
            }
        }
    }
    const bool topk_from_kept = topk > 0 && write_logits;
    const bool topk_scan = topk > 0 && !topk_from_kept;
    // in-scan top-k candidates cost 512 B per (query, split) and are merged by one wave per
    // query: keep the split count moderate in that variant
    int rc = fill_scan_args(c, a, ehat32, xq32, B, tau_sem, tau_geo, true,
                            topk_scan ? (few ? 256 : 16) : (few ? 2048 : 128));
    if (rc) return rc;
    if (force_splits > 0) a.n_splits = std::max(1, std::min<int>(force_splits, std::max(1, a.n_blocks / 4)));
    if (write_logits) a.logits = c->ws_logits.p;
    a.qt_offset = (int32_t)(first_query / QTILE);
    if (topk_from_kept) {
        HIP_TRY(c->ws_rowmax.ensure((size_t)a.n_splits * B * 4));
        HIP_TRY(c->ws_theta.ensure((size_t)B));
        a.rowmax = c->ws_rowmax.p;
    }
    HIP_TRY(c->ws_stats_parts.ensure((size_t)a.n_splits * B * 4));
    a.out = c->ws_stats_parts.p;
    if (topk_scan) {
        HIP_TRY(c->ws_cand_val.ensure((size_t)a.n_splits * B * 4 * MAX_TOPK));
        HIP_TRY(c->ws_cand_idx.ensure((size_t)a.n_splits * B * 4 * MAX_TOPK));
        a.cand_val = c->ws_cand_val.p;
        a.cand_idx = c->ws_cand_idx.p;
    }
    const dim3 grid((unsigned)(a.n_splits * a.n_qtiles)), block(256);
    const bool geo = tau_geo > 0.f;
#define RANGE_SCAN_LAUNCH(G, T)                                                             \
    do {                                                                                    \
        rc = set_dyn_lds(scan_stats_kernel<G, T>, SCAN_LDS_BYTES);                          \
        if (rc) return rc;                                                                  \
        hipLaunchKernelGGL((scan_stats_kernel<G, T>), grid, block, SCAN_LDS_BYTES, s, a);   \
    } while (0)
    {
        ProfScope ps(c, RANGE_PROF_SCAN_STATS, s);
        if (geo && topk_scan) RANGE_SCAN_LAUNCH(true, true);
        else if (geo) RANGE_SCAN_LAUNCH(true, false);
        else if (topk_scan) RANGE_SCAN_LAUNCH(false, true);
        else RANGE_SCAN_LAUNCH(false, false);
    }
#undef RANGE_SCAN_LAUNCH
    HIP_TRY(hipGetLastError());
    if (a.logits && keep_logits) {
        c->kept_B = first_query + B;
        c->kept_total = total_queries;
        c->kept_blocks = a.n_blocks;
    }
    const int tpb = 256;
    if (a.n_splits > 32)
        hipLaunchKernelGGL(merge_stats_wave_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                           c->ws_stats_parts.p, a.n_splits, B, stats);
    else
        hipLaunchKernelGGL(merge_stats_kernel, dim3((unsigned)((B + tpb - 1) / tpb)), dim3(tpb), 0, s,
                           c->ws_stats_parts.p, a.n_splits, B, stats);
    HIP_TRY(hipGetLastError());
    if (topk_from_kept) {
        // enough waves to fill the chip: (B/16 wave slots) x chunks of bank blocks
        const int64_t n_slots = (B + 15) / 16;
        const int n_chunks = (int)std::max<int64_t>(1, std::min<int64_t>(
            std::min<int64_t>(64, a.n_blocks / 32 > 0 ? a.n_blocks / 32 : 1),
            ((int64_t)16 * c->n_cu + n_slots - 1) / n_slots));
        HIP_TRY(c->ws_cand_val.ensure((size_t)n_chunks * B * MAX_TOPK));
        HIP_TRY(c->ws_cand_idx.ensure((size_t)n_chunks * B * MAX_TOPK));
        hipLaunchKernelGGL(topk_threshold_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                           c->ws_rowmax.p, a.n_splits, B, c->ws_theta.p);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(topk_from_logits_kernel, dim3((unsigned)((n_slots + 3) / 4), (unsigned)n_chunks),
                           dim3(256), 0, s, c->ws_logits.p, a.n_blocks, B, c->n_rows, n_chunks,
                           c->ws_theta.p, c->ws_cand_val.p, c->ws_cand_idx.p);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(merge_topk_wave_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                           c->ws_cand_val.p, c->ws_cand_idx.p, n_chunks, B, MAX_TOPK, topk,
                           c->row_offset, topk_val, topk_idx);
        HIP_TRY(hipGetLastError());
    } else if (topk_scan) {
        hipLaunchKernelGGL(merge_topk_wave_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s,
                           c->ws_cand_val.p, c->ws_cand_idx.p, a.n_splits, B, 4 * MAX_TOPK, topk,
                           c->row_offset, topk_val, topk_idx);
        HIP_TRY(hipGetLastError());
    }
    return RANGE_OK;
}

int range_scan_stats(range_ctx* c, const float* ehat32, const float* xq32, int64_t B, float tau_sem,
                     float tau_geo, float* stats, int topk, float* topk_val, int64_t* topk_idx,
                     int32_t keep_logits, range_stream_t stream) {
    return scan_stats_impl(c, ehat32, xq32, B, tau_sem, tau_geo, stats, topk, topk_val, topk_idx, keep_logits,
                           0, B, 0, stream);
}

int range_scan_stats_at(range_ctx* c, const float* ehat32, const float* xq32, int64_t B, float tau_sem,
                        float tau_geo, float* stats, int64_t first_query, int64_t total_queries,
                        int32_t n_splits, range_stream_t stream) {
    return scan_stats_impl(c, ehat32, xq32, B, tau_sem, tau_geo, stats, 0, nullptr, nullptr, /*keep_logits=*/1,
                           first_query, total_queries, n_splits, stream);
}

// the bank splits a pass-1 launch of B queries chooses (fill_scan_args, no top-k)
int32_t range_p1_splits(const range_ctx* c, int64_t B) {
    if (!c || !c->has_bank || B <= 0) return 0;
    const int n_qtiles = (int)((B + QTILE - 1) / QTILE), n_blocks = (int)((c->n_rows + BLK - 1) / BLK);
    return choose_splits(n_qtiles, n_blocks, c->n_cu, RANGE_P1_WG_PER_CU, B <= 4 * QTILE ? 2048 : 128);
}

// repeats > 1 (range_topk_stream_timed): the whole call's launches are enqueued `repeats` times
// back to back between ONE pair of events (identical launches, identical results) and *avg_us
// receives the time per call.
static int topk_stream_impl(range_ctx* c, const float* ehat32, int64_t B, int32_t k, float* topk_val,
                            int64_t* topk_idx, int repeats, float* avg_us, range_stream_t stream) {
    if (!c || !ehat32 || !topk_val || !topk_idx) return fail(RANGE_ERR_INVALID, "null argument");
    if (!c->has_bank) return fail(RANGE_ERR_STATE, "bank not set (range_set_bank)");
    if (B <= 0 || k <= 0 || k > MAX_TOPK) return fail(RANGE_ERR_INVALID, "bad B or k");
    if (int rc0 = check_async_error(c)) return rc0;
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = (hipStream_t)stream;
    const int n_groups = (int)((B + 15) / 16);
    const int n_blocks = (int)((c->n_rows + BLK - 1) / BLK);
    // persistent grid: one workgroup per CU (its key tiles fill the LDS), 4 waves each streaming
    // its own tiles; one candidate list of 8 per (query, workgroup).  Query groups sharing one pass
    // over the keys: 2 groups (32 queries) are still at the ridge (16 FLOP per key byte) and take
    // the time of 1.3.
    constexpr int LIST = RANGE_TOPKS_LIST;
    constexpr int NWV = 4, DEP = 2;
    // (the bf16 prefilter with FOUR groups per pass was measured too: one pass for 64 queries takes
    // 27.0 us against 28.7 us for two passes of two groups - the list work per group, not the
    // stream, is what a pass costs by then - and needs 370 registers; not kept)
    const bool bf16 = c->topks_bf16;
    int G = c->topks_groups;
    if (G != 1 && G != 2) G = n_groups <= 1 ? 1 : 2;
    const int n_wg = std::max(1, std::min(std::min(c->n_cu, 256), (n_blocks + NWV - 1) / NWV));
    // the merge runs as the tail of the stream kernel while every query finds a workgroup of its own
    const bool fused = c->topks_fused && B <= n_wg;
    // (the candidate lists of a call are addressed by 32-bit byte offsets: B x lists x 256 B < 4 GB - the
    // Python layer calls in chunks of 16 384 queries = 1 GB at most)
    // (a bank whose largest key norm is so far from 1 that no power of two brings it into fp16's range - or
    // that is all zeros - keeps the streaming scan)
    int tg_e2 = 0;
    (void)std::frexp((double)c->key_norm_max, &tg_e2);                  // key_norm_max < 2^e2
    const bool tg_bank_ok = c->key_norm_max > 0.f && std::isfinite(c->key_norm_max) && std::abs(14 - tg_e2) <= 100;
    if (c->topk_gemm && bf16 && B > 256 && n_blocks >= 64 && !c->topks_force_exact && B <= 60000 && tg_bank_ok) {
        // batches beyond the one-launch regime: GEMM-shaped, list-free (topk_gemm.h): group maxima ->
        // per-query threshold -> candidates -> float32 re-rank.  Two workgroups per CU; the splits fill
        // one round of them (at least 4: 32 row groups for the threshold; at least 8 tiles each).
        TopkGemmArgs ga{};
        if (c->tg_key_scale == 0.f) {
            // the fp16 copy of the keys, scaled so that the largest row norm lies in [2^13, 2^14)
            const float ks = (float)std::ldexp(1.0, 14 - tg_e2);
            const int64_t n_tiles = c->n_pad / BLK;
            HIP_TRY(c->d_keys_f16.ensure((size_t)n_tiles * (TSB_TILE_BYTES / 4)));
            const int64_t threads = n_tiles * 8 * 64;
            hipLaunchKernelGGL(keyfrag_f16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, c->d_keys.p,
                               c->n_pad, n_tiles, ks, reinterpret_cast<ts_u32x4*>(c->d_keys_f16.p));
            HIP_TRY(hipGetLastError());
            c->tg_key_scale = ks;
        }
        ga.keys_f16 = c->d_keys_f16.p;
        ga.keys = c->d_keys.p;
        ga.ehat = ehat32;
        ga.B = B;
        ga.n_valid = c->n_rows;
        ga.n_blocks = n_blocks;
        ga.n_qblocks = (int32_t)((B + TG_QBLOCK - 1) / TG_QBLOCK);
        ga.n_splits = std::max(4, std::min(std::min(TG_WG_PER_CU * c->n_cu / ga.n_qblocks, n_blocks / 8), 64));
        ga.k = k;
        ga.row_offset = c->row_offset;
        ga.oval = topk_val;
        ga.oidx = topk_idx;
        if (!c->ws_exact_count.p) {
            HIP_TRY(c->ws_exact_count.ensure(2));
            HIP_TRY(hipMemsetAsync(c->ws_exact_count.p, 0, 2 * sizeof(int32_t), s));
        }
        ga.exact_count = c->ws_exact_count.p;
        HIP_TRY(c->ws_tg_gmax.ensure((size_t)ga.n_splits * 2 * B * 4));
        HIP_TRY(c->ws_tg_theta.ensure((size_t)B * 2));
        HIP_TRY(c->ws_tg_cnt.ensure((size_t)B * ga.n_splits * 4));
        HIP_TRY(c->ws_tg_cand.ensure((size_t)B * ga.n_splits * 4 * TG_CAP_L));
        HIP_TRY(c->ws_tg_ovf.ensure((size_t)B));
        ga.ovf = c->ws_tg_ovf.p;
        const int64_t n_qgroups = (B + 15) / 16;
        HIP_TRY(c->ws_tg_qfrag.ensure((size_t)n_qgroups * 8 * 64 * 4));
        HIP_TRY(c->ws_tg_qscale.ensure((size_t)B));
        ga.qfrag = c->ws_tg_qfrag.p;
        ga.gmax = c->ws_tg_gmax.p;
        ga.theta = c->ws_tg_theta.p;
        ga.cnt = c->ws_tg_cnt.p;
        ga.cand = c->ws_tg_cand.p;
        int rcg = set_dyn_lds(topk_gemm_kernel<0>, TG_LDS_BYTES);
        if (rcg) return rcg;
        rcg = set_dyn_lds(topk_gemm_kernel<1>, TG_LDS_BYTES);
        if (rcg) return rcg;
        hipEvent_t g0 = nullptr, g1 = nullptr;
        if (repeats > 1) {
            g0 = c->get_event();
            g1 = c->get_event();
            HIP_TRY(hipEventRecord(g0, s));
        }
        const dim3 ggrid((unsigned)(ga.n_qblocks * ga.n_splits));
        for (int rep = 0; rep < std::max(1, repeats); ++rep) {
            {
                ProfScope ps(c, RANGE_PROF_TOPK_STREAM, s);
                hipLaunchKernelGGL(qfrag_f16_kernel, dim3((unsigned)n_qgroups), dim3(256), 0, s, ehat32, B,
                                   reinterpret_cast<ts_u32x4*>(c->ws_tg_qfrag.p), c->ws_tg_qscale.p);
                ga.tile_stride = std::max(1, std::min(c->tg_sample, n_blocks / ga.n_splits / 4));
                hipLaunchKernelGGL(topk_gemm_kernel<0>, ggrid, dim3(TG_WAVES * 64), TG_LDS_BYTES, s, ga);
                ga.tile_stride = 1;
                hipLaunchKernelGGL(topk_gemm_threshold_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, ga.gmax,
                                   ga.n_splits * 2, B, ehat32, (float)((double)TG_EPS_REL * (double)c->key_norm_max * (double)c->tg_key_scale),
                                   c->key_norm_max > 0.f ? (float)std::log2((double)c->key_norm_max) : -INFINITY,
                                   c->ws_tg_qscale.p, c->ws_tg_theta.p);
#ifdef RANGE_EXP_TG_NOHIT       // timing experiment: pass B with a threshold nothing reaches (its MFMA + compare floor)
                HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->ws_tg_theta.p, 0x7f800000, (size_t)B * 2, s));
#endif
                hipLaunchKernelGGL(topk_gemm_kernel<1>, ggrid, dim3(TG_WAVES * 64), TG_LDS_BYTES, s, ga);
            }
            ProfScope ps(c, RANGE_PROF_TOPK_MERGE, s);
            hipLaunchKernelGGL(topk_gemm_rerank_kernel, dim3((unsigned)B), dim3(256), 0, s, ga);
            hipLaunchKernelGGL(topk_gemm_brute_kernel, dim3((unsigned)std::min<int64_t>(B, 2 * c->n_cu)), dim3(256), 0, s, ga);
            HIP_TRY(hipGetLastError());
        }
        if (repeats > 1) {
            HIP_TRY(hipEventRecord(g1, s));
            HIP_TRY(hipEventSynchronize(g1));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, g0, g1));
            if (avg_us) *avg_us = ms * 1e3f / (float)repeats;
            c->ev_pool.push_back(g0);
            c->ev_pool.push_back(g1);
        }
        return RANGE_OK;
    }
    HIP_TRY(c->ws_cand_keys.ensure((size_t)n_groups * 16 * n_wg * TOPKS_WL));
    HIP_TRY(c->ws_cand_dmax.ensure((size_t)n_groups * 16 * n_wg));
    if (!c->ws_exact_count.p) {
        HIP_TRY(c->ws_exact_count.ensure(2));    // [0] brute-force queries, [1] candidates ranked (diagnostic)
        HIP_TRY(hipMemsetAsync(c->ws_exact_count.p, 0, 2 * sizeof(int32_t), s));
    }
    if (!c->ws_topk_sync.p) {
        HIP_TRY(c->ws_topk_sync.ensure(TOPKS_SYNC_WORDS));
        HIP_TRY(hipMemsetAsync(c->ws_topk_sync.p, 0, TOPKS_SYNC_WORDS * 4, s));
        std::memset(c->topk_sync_base, 0, sizeof c->topk_sync_base);
    }
    TopkStreamArgs a{};
    a.keys = c->d_keys.p;
    a.ehat = ehat32;
    a.cand = c->ws_cand_keys.p;
    a.dmax = c->ws_cand_dmax.p;
    a.B = B;
    a.n_valid = c->n_rows;
    a.n_blocks = n_blocks;
    a.n_groups = n_groups;
    a.keys_bf16 = c->d_keys_bf16.p;
    a.sync = c->ws_topk_sync.p;
    a.err = c->d_async_err ? c->d_async_err + RANGE_ASYNC_WORD_TOPK : nullptr;
    a.fused = fused ? 1 : 0;
    a.k = k;
    a.row_offset = c->row_offset;
    a.force_exact = c->topks_force_exact ? 1 : 0;
    a.exact_count = c->ws_exact_count.p;
    a.eps_rel = bf16 ? TSB_EPS_REL : 0.f;
    a.kmax = c->key_norm_max;
    a.oval = topk_val;
    a.oidx = topk_idx;
#ifdef RANGE_EXP_TS_STAMPS
    static DevBuf<unsigned long long> stamps_buf;
    const int n_lists = n_wg * NWV;
    HIP_TRY(stamps_buf.ensure((size_t)n_lists * 8 + (size_t)n_wg * 16));
    HIP_TRY(hipMemsetAsync(stamps_buf.p, 0, ((size_t)n_lists * 8 + (size_t)n_wg * 16) * 8, s));
    a.stamps = stamps_buf.p;
#endif
    int rc = RANGE_OK;
#define RANGE_TOPKS_LAUNCH(GG)                                                                      \
    do {                                                                                            \
        rc = set_dyn_lds(topk_stream_kernel<GG, LIST, NWV, DEP>, TOPKS_LDS_BYTES);                  \
        if (rc) return rc;                                                                          \
        ProfScope ps(c, RANGE_PROF_TOPK_STREAM, s);                                                 \
        hipLaunchKernelGGL((topk_stream_kernel<GG, LIST, NWV, DEP>), dim3((unsigned)n_wg),          \
                           dim3(NWV * 64), TOPKS_LDS_BYTES, s, a);                                  \
    } while (0)
#define RANGE_TOPKS_LAUNCH_BF16(GG)                                                                 \
    do {                                                                                            \
        rc = set_dyn_lds(topk_stream_bf16_kernel<GG, LIST>, TOPKS_LDS_BYTES);                       \
        if (rc) return rc;                                                                          \
        ProfScope ps(c, RANGE_PROF_TOPK_STREAM, s);                                                 \
        hipLaunchKernelGGL((topk_stream_bf16_kernel<GG, LIST>), dim3((unsigned)n_wg), dim3(256),    \
                           TOPKS_LDS_BYTES, s, a);                                                  \
    } while (0)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (repeats > 1) {
        ev0 = c->get_event();
        ev1 = c->get_event();
        HIP_TRY(hipEventRecord(ev0, s));
    }
    for (int rep = 0; rep < std::max(1, repeats); ++rep) {
        for (int x = 0; x < 8; ++x) a.sync_base[x] = c->topk_sync_base[x];
        a.debug_giveup = fused && c->debug_giveup_next ? 1 : 0;
        if (bf16) {
            if (G == 1) RANGE_TOPKS_LAUNCH_BF16(1);
            else RANGE_TOPKS_LAUNCH_BF16(2);
        } else {
            if (G == 1) RANGE_TOPKS_LAUNCH(1);
            else RANGE_TOPKS_LAUNCH(2);
        }
        HIP_TRY(hipGetLastError());
        if (fused) {
            // every workgroup of the launch takes one ticket of its shard (blockIdx % 8)
            for (int x = 0; x < 8; ++x) c->topk_sync_base[x] += (uint32_t)((n_wg - x + 7) >> 3);
            c->debug_giveup_next = false;
        }
        if (!fused) {
            rc = set_dyn_lds(topk_merge_kernel<TOPKS_WL>, TOPKM_LDS_BYTES);
            if (rc) return rc;
            ProfScope ps(c, RANGE_PROF_TOPK_MERGE, s);
            hipLaunchKernelGGL(topk_merge_kernel<TOPKS_WL>, dim3((unsigned)B), dim3(256), TOPKM_LDS_BYTES, s, a, n_wg);
            HIP_TRY(hipGetLastError());
        }
    }
#undef RANGE_TOPKS_LAUNCH
#undef RANGE_TOPKS_LAUNCH_BF16
    if (repeats > 1) {
        HIP_TRY(hipEventRecord(ev1, s));
        HIP_TRY(hipEventSynchronize(ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        if (avg_us) *avg_us = ms * 1e3f / (float)repeats;
        c->ev_pool.push_back(ev0);
        c->ev_pool.push_back(ev1);
    }
#ifdef RANGE_EXP_TS_STAMPS
    if (std::getenv("RANGE_TOPKS_STAMPS")) {   // per stamp: earliest / median / latest wave, us after the first wave's start
        std::vector<unsigned long long> h((size_t)n_lists * 8 + (size_t)n_wg * 16);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(h.data(), stamps_buf.p, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < n_lists; ++w) t0 = std::min(t0, h[(size_t)w * 8]);
        std::fprintf(stderr, "topk_stream stamps (us: min med max)");
        for (int i = 0; i < 7; ++i) {
            if (i == 3) continue;                          // (slot 3 holds a sum, printed below)
            std::vector<double> v;
            for (int w = 0; w < n_lists; ++w) v.push_back((double)(h[(size_t)w * 8 + i] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            std::fprintf(stderr, " | %d: %.1f %.1f %.1f", i, v.front(), v[v.size() / 2], v.back());
        }
        {   // loop sums (us): waiting for tiles / issuing LDS-DMA / arithmetic
            std::vector<double> w, is, cp;
            for (int x = 0; x < n_lists; ++x) {
                w.push_back((double)h[(size_t)x * 8 + 3] * 0.01);
                is.push_back((double)(h[(size_t)x * 8 + 7] >> 32) * 0.01);
                cp.push_back((double)(h[(size_t)x * 8 + 7] & 0xFFFFFFFFull) * 0.01);
            }
            std::sort(w.begin(), w.end()); std::sort(is.begin(), is.end()); std::sort(cp.begin(), cp.end());
            std::fprintf(stderr, "\n  per wave over the tile loop (us, min med max): waiting %.1f %.1f %.1f | issuing DMA %.1f %.1f %.1f | arithmetic %.1f %.1f %.1f",
                         w.front(), w[w.size() / 2], w.back(), is.front(), is[is.size() / 2], is.back(), cp.front(), cp[cp.size() / 2], cp.back());
        }
        std::fprintf(stderr, "\n  end of tiles (stamp 4) by blockIdx %% 8:");
        for (int x = 0; x < 8; ++x) {
            std::vector<double> v;
            for (int w = 0; w < n_lists; ++w) if ((w % n_wg) % 8 == x) v.push_back((double)(h[(size_t)w * 8 + 4] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            std::fprintf(stderr, " %.1f/%.1f/%.1f", v.front(), v[v.size() / 2], v.back());
        }
        std::fprintf(stderr, "\n  by blockIdx / 32:");
        for (int x = 0; x < (n_wg + 31) / 32; ++x) {
            std::vector<double> v;
            for (int w = 0; w < n_lists; ++w) if ((w % n_wg) / 32 == x) v.push_back((double)(h[(size_t)w * 8 + 4] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            std::fprintf(stderr, " %.1f/%.1f/%.1f", v.front(), v[v.size() / 2], v.back());
        }
        std::fprintf(stderr, "\n  tail stamps (us after the first wave's start: min med max over the workgroups that have it)");
        for (int i = 0; i < 16; ++i) {
            std::vector<double> v;
            for (int w = 0; w < n_wg; ++w) {
                const unsigned long long t = h[(size_t)n_lists * 8 + (size_t)w * 16 + i];
                if (t && i == 14) {   // shader clocks of the merge / its real time -> MHz
                    const unsigned long long b0 = h[(size_t)n_lists * 8 + (size_t)w * 16 + 3], b1 = h[(size_t)n_lists * 8 + (size_t)w * 16 + 15];
                    if (b1 > b0) v.push_back((double)t / ((double)(b1 - b0) * 0.01));
                } else if (t) v.push_back((double)(t - t0) * 0.01);
            }
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            std::fprintf(stderr, " | %d (%zu): %.1f %.1f %.1f", i, v.size(), v.front(), v[v.size() / 2], v.back());
        }
        std::fprintf(stderr, "\n  by wave in workgroup:");
        for (int x = 0; x < 4; ++x) {
            std::vector<double> v;
            for (int w = 0; w < n_lists; ++w) if (w / n_wg == x) v.push_back((double)(h[(size_t)w * 8 + 4] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            std::fprintf(stderr, " %.1f/%.1f/%.1f", v.front(), v[v.size() / 2], v.back());
        }
        std::fprintf(stderr, "\n");
    }
#endif
    return RANGE_OK;
}

int range_topk_stream(range_ctx* c, const float* ehat32, int64_t B, int32_t k, float* topk_val,
                      int64_t* topk_idx, range_stream_t stream) {
    return topk_stream_impl(c, ehat32, B, k, topk_val, topk_idx, 1, nullptr, stream);
}

// forward(coords, return_topk=k): the top-k of the queries the context's last range_forward /
// range_forward_host call embedded - their e-hat (float32) is still in the workspace, so the side
// channel costs its scan only: no second encoder pass.
int range_topk_last(range_ctx* c, int64_t B, int32_t k, float* topk_val, int64_t* topk_idx,
                    range_stream_t stream) {
    if (!c || !topk_val || !topk_idx) return fail(RANGE_ERR_INVALID, "null argument");
    if (B <= 0 || c->ws_queries != B)
        return fail(RANGE_ERR_STATE, "range_topk_last(B=%lld): the workspace holds the e-hat of %lld queries (the last "
                    "range_forward / range_forward_host call of this context)", (long long)B, (long long)c->ws_queries);
    return topk_stream_impl(c, c->ws_ehat32.p, B, k, topk_val, topk_idx, 1, nullptr, stream);
}

int range_topk_stream_timed(range_ctx* c, const float* ehat32, int64_t B, int32_t k, float* topk_val,
                            int64_t* topk_idx, int32_t repeats, float* avg_us, range_stream_t stream) {
    if (repeats < 2 || !avg_us) return fail(RANGE_ERR_INVALID, "repeats must be >= 2 and avg_us non-null");
    return topk_stream_impl(c, ehat32, B, k, topk_val, topk_idx, repeats, avg_us, stream);
}

// A plain one-launch streaming read of the same bytes the top-k scan streams (`passes` times over
// the bf16 or the float32 copy of the keys): the ceiling a launch of that size can reach on this
// chip, measured the same way as range_topk_stream_timed.  16-byte non-temporal loads, 8 in flight
// per thread, 1 024 workgroups; the xor of everything read goes to one word per workgroup so that
// the loads stay.
__global__ __launch_bounds__(256) void stream_read_kernel(const ts_u32x4* __restrict__ p, int64_t n16, int passes,
                                                          uint32_t* __restrict__ sink) {
    // a workgroup reads 32 KB contiguous per step (8 loads of 16 bytes per thread, 4 KB apart)
    const int64_t step = (int64_t)gridDim.x * 2048;
    ts_u32x4 acc = {0u, 0u, 0u, 0u};
    for (int ps = 0; ps < passes; ++ps) {
        for (int64_t base = (int64_t)blockIdx.x * 2048; base < n16; base += step) {
            ts_u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = base + u * 256 + threadIdx.x;
                v[u] = __builtin_nontemporal_load(p + (i < n16 ? i : n16 - 1));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= v[u];
        }
    }
    uint32_t r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) r ^= __shfl_xor(r, off);
    if ((threadIdx.x & 63) == 0) atomicXor(sink + blockIdx.x, r);
}

int range_stream_read_timed(range_ctx* c, int32_t f32_keys, int32_t passes, int32_t repeats, float* avg_us,
                            range_stream_t stream) {
    if (!c || !avg_us) return fail(RANGE_ERR_INVALID, "null argument");
    if (!c->has_bank) return fail(RANGE_ERR_STATE, "bank not set (range_set_bank)");
    if (passes < 1 || repeats < 2) return fail(RANGE_ERR_INVALID, "passes must be >= 1 and repeats >= 2");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_tiles = c->n_pad / BLK;
    const ts_u32x4* src = f32_keys ? reinterpret_cast<const ts_u32x4*>(c->d_keys.p) : reinterpret_cast<const ts_u32x4*>(c->d_keys_bf16.p);
    const int64_t n16 = f32_keys ? c->n_pad * (KEY_DIM * 4 / 16) : n_tiles * (TSB_TILE_BYTES / 16);
    const int grid = 4 * c->n_cu;
    HIP_TRY(c->ws_read_sink.ensure((size_t)grid));
    hipEvent_t ev0 = c->get_event(), ev1 = c->get_event();
    hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, s, src, n16, passes, c->ws_read_sink.p);   // (warm-up)
    HIP_TRY(hipEventRecord(ev0, s));
    for (int rep = 0; rep < repeats; ++rep)
        hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, s, src, n16, passes, c->ws_read_sink.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev1, s));
    HIP_TRY(hipEventSynchronize(ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    *avg_us = ms * 1e3f / (float)repeats;
    c->ev_pool.push_back(ev0);
    c->ev_pool.push_back(ev1);
    return RANGE_OK;
}

int range_topk_stream_exact_count(range_ctx* c, int64_t* count) {
    if (!c || !count) return fail(RANGE_ERR_INVALID, "null argument");
    *count = 0;
    if (!c->ws_exact_count.p) return RANGE_OK;
    DeviceGuard g(c->device);
    int32_t v[2] = {0, 0};
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(v, c->ws_exact_count.p, sizeof v, hipMemcpyDeviceToHost));
    *count = v[0];
    if (std::getenv("RANGE_TOPKS_DIAG")) std::fprintf(stderr, "range_topk_stream: %d candidates ranked so far\n", v[1]);
    return check_async_error(c);     // (a merging workgroup of an earlier fused call that gave up)
}

int range_merge_stats(range_ctx* c, const float* parts, int32_t n_parts, int64_t B, float* out,
                      range_stream_t stream) {
    if (!c || !parts || !out) return fail(RANGE_ERR_INVALID, "null argument");
    if (n_parts <= 0 || B <= 0) return fail(RANGE_ERR_INVALID, "n_parts and B must be > 0");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const int tpb = 256;
    hipLaunchKernelGGL(merge_stats_kernel, dim3((unsigned)((B + tpb - 1) / tpb)), dim3(tpb), 0,
                       (hipStream_t)stream, parts, n_parts, B, out);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_merge_topk(range_ctx* c, const float* val_parts, const int64_t* idx_parts, int32_t n_parts,
                     int64_t B, int32_t k, float* val_out, int64_t* idx_out, range_stream_t stream) {
    if (!c || !val_parts || !idx_parts || !val_out || !idx_out) return fail(RANGE_ERR_INVALID, "null argument");
    if (n_parts <= 0 || B <= 0 || k <= 0 || k > MAX_TOPK) return fail(RANGE_ERR_INVALID, "bad n_parts/B/k");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipLaunchKernelGGL(merge_topk_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0,
                       (hipStream_t)stream, val_parts, (const int32_t*)nullptr, idx_parts, n_parts, B,
                       k, k, (int64_t)0, val_out, idx_out);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

// pass 2 into the context's split slabs; when `partial` is non-null the slabs are then summed
// (fixed order) into it, otherwise the caller consumes the slabs (n_splits_out of them) itself.
// kept_first >= 0: queries [kept_first, kept_first + B) of the last scan whose logits were kept
// (attend_stored_kernel; ehat32 is not read); kept_first < 0: recompute the logits.
static int attend_impl(range_ctx* c, const float* ehat32, const float* xq32, int64_t B, float tau_sem,
                       float tau_geo, float beta, const float* stats_global, float* partial,
                       SlabMap* map_out, int64_t kept_first, range_stream_t stream) {
    if (!c || (!ehat32 && kept_first < 0) || !xq32 || !stats_global)
        return fail(RANGE_ERR_INVALID, "null argument");
    if (kept_first >= 0) {
        if (c->kept_B <= 0) return fail(RANGE_ERR_STATE, "no kept logits (range_scan_stats with keep_logits)");
        if (kept_first % QTILE != 0) return fail(RANGE_ERR_INVALID, "first kept query must be a multiple of %d", QTILE);
        if (kept_first + B > c->kept_B)
            return fail(RANGE_ERR_INVALID, "queries [%lld, %lld) exceed the %lld kept", (long long)kept_first,
                        (long long)(kept_first + B), (long long)c->kept_B);
    }
    if (!(beta >= 0.f && beta <= 1.f)) return fail(RANGE_ERR_INVALID, "beta must be in [0,1]");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    ScanArgs a{};
    int rc = fill_scan_args(c, a, ehat32, xq32, B, tau_sem, tau_geo, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // Stream-K (attend_kernels.h: SlabMap): as many workgroups as CUs, each with the same number of
    // (query tile, bank block) units (at least 4: tiny launches take fewer workgroups), one slab per
    // query tile a workgroup touches.  For banks and SHARDS of up to 50 000 rows - measured, 10 000
    // queries, pass 2 + its reduction, against one workgroup per (split, query tile): 12 500 rows (a
    // rank of 8) -5.6 %, 25 000 -3.6 %, 50 000 -2.2 %; at 100 000 rows that scheme's 7.97 rounds of
    // workgroups are 98.5 % full already, the walk gains 0.5-1 % with two 50 000-row columns (and
    // loses 9 % with one: its workgroups re-read the values from HBM then, not from the Infinity
    // Cache) while its longer float32 accumulation chains cost accuracy (|sum of weights - 1| of the
    // worst of 10^5 queries 1.3e-5 instead of 0.5e-5): not taken there.  The exact kernels only;
    // RANGE_P2_STREAMK=0 / RANGE_P2_SPLITS=n restore the split scheme for A/B.
    const bool streamk = c->p2_streamk && c->p2_splits_forced == 0 && c->n_rows <= c->p2_streamk_rows &&
                         !(kept_first >= 0 && c->pv_mode == RANGE_PV_BF16X3);
    if (streamk) {
        // columns of at most 16 384 rows: an accumulation chain (one query tile's blocks of a column)
        // stays within ~2x the 481 blocks of the split scheme on the full bank
        a.sk_cols = (int32_t)std::max<int64_t>(1, std::min<int64_t>((c->n_rows + c->p2_col_rows - 1) / c->p2_col_rows,
                                                                    std::max(1, a.n_blocks / 4)));
        const int64_t Uc = (int64_t)a.n_qtiles * (a.n_blocks / a.sk_cols);     // (units of the shortest column)
        a.sk_groups = (int32_t)std::max<int64_t>(1, std::min<int64_t>(c->n_cu, Uc / 4));
        c->last_splits = a.sk_groups;
        HIP_TRY(c->ws_slabs.ensure((size_t)a.sk_cols * (a.sk_groups + a.n_qtiles) * QTILE * VAL_DIM));
    } else {
        HIP_TRY(c->ws_slabs.ensure((size_t)a.n_splits * B * VAL_DIM));
    }
    const SlabMap map{a.n_splits, a.sk_groups, a.n_blocks, a.n_qtiles, a.sk_cols};
    a.out = c->ws_slabs.p;
    a.stats = stats_global;
    const bool geo = tau_geo > 0.f;
    a.beta = geo ? beta : 1.f;
    const dim3 grid((unsigned)(streamk ? a.sk_groups : a.n_splits * a.n_qtiles)), block(256);
    if (kept_first >= 0) {
        if (a.n_blocks != c->kept_blocks) return fail(RANGE_ERR_STATE, "bank changed since the logits were kept");
        a.logits = c->ws_logits.p;
        a.qt_offset = (int32_t)(kept_first / QTILE);
        ProfScope ps(c, RANGE_PROF_ATTEND, s);
        if (c->pv_mode == RANGE_PV_BF16X3) {
            // opt-in: w @ V on three bf16 planes of both operands (attend_bf16x3.h)
            const int32_t n_groups = (a.n_blocks + 1) / 2;
            if (c->vplanes_groups != n_groups) return fail(RANGE_ERR_STATE, "bf16 planes of the values are missing");
            const char* planes = reinterpret_cast<const char*>(c->d_vplanes.p);
            if (geo) {
                rc = set_dyn_lds(attend_bf16x3_kernel<true>, PVB2_LDS_BYTES);
                if (rc) return rc;
                hipLaunchKernelGGL(attend_bf16x3_kernel<true>, grid, block, PVB2_LDS_BYTES, s, a, planes, n_groups);
            } else {
                rc = set_dyn_lds(attend_bf16x3_kernel<false>, PVB2_LDS_BYTES);
                if (rc) return rc;
                hipLaunchKernelGGL(attend_bf16x3_kernel<false>, grid, block, PVB2_LDS_BYTES, s, a, planes, n_groups);
            }
        } else if (geo) {
            rc = set_dyn_lds(attend_stored_kernel<true>, ATTEND_STORED_LDS_BYTES);
            if (rc) return rc;
#ifdef RANGE_EXP_P2_STAMPS
            static DevBuf<unsigned long long> p2_stamps;
            const char* stamp_file = std::getenv("RANGE_P2_STAMPS");
            if (stamp_file) {
                HIP_TRY(p2_stamps.ensure((size_t)grid.x * 16));
                a.diag = p2_stamps.p;
            }
#endif
            hipLaunchKernelGGL(attend_stored_kernel<true>, grid, block, ATTEND_STORED_LDS_BYTES, s, a);
#ifdef RANGE_EXP_P2_STAMPS
            if (stamp_file) {
                std::vector<unsigned long long> h((size_t)grid.x * 16);
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(h.data(), p2_stamps.p, h.size() * 8, hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen(stamp_file, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
            }
#endif
        } else {
            rc = set_dyn_lds(attend_stored_kernel<false>, ATTEND_STORED_LDS_BYTES);
            if (rc) return rc;
            hipLaunchKernelGGL(attend_stored_kernel<false>, grid, block, ATTEND_STORED_LDS_BYTES, s, a);
        }
    } else {
        ProfScope ps(c, RANGE_PROF_ATTEND, s);
        if (geo) {
            rc = set_dyn_lds(attend_kernel<true>, ATTEND_LDS_BYTES);
            if (rc) return rc;
            hipLaunchKernelGGL(attend_kernel<true>, grid, block, ATTEND_LDS_BYTES, s, a);
        } else {
            rc = set_dyn_lds(attend_kernel<false>, ATTEND_LDS_BYTES);
            if (rc) return rc;
            hipLaunchKernelGGL(attend_kernel<false>, grid, block, ATTEND_LDS_BYTES, s, a);
        }
    }
    HIP_TRY(hipGetLastError());
    if (map_out) *map_out = map;
    if (partial) {
        const int64_t total4 = B * (VAL_DIM / 4);
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s,
                           c->ws_slabs.p, map, B, partial);
        HIP_TRY(hipGetLastError());
    }
    return RANGE_OK;
}

int range_attend(range_ctx* c, const float* ehat32, const float* xq32, int64_t B, float tau_sem,
                 float tau_geo, float beta, const float* stats_global, float* partial,
                 range_stream_t stream) {
    if (!partial) return fail(RANGE_ERR_INVALID, "null argument");
    return attend_impl(c, ehat32, xq32, B, tau_sem, tau_geo, beta, stats_global, partial, nullptr, -1, stream);
}

int64_t range_kept_queries(const range_ctx* c) { return c ? c->kept_B : 0; }

int range_attend_kept(range_ctx* c, int64_t first_query, const float* xq32, int64_t B, float tau_sem,
                      float tau_geo, float beta, const float* stats_global, float* partial,
                      range_stream_t stream) {
    if (!partial) return fail(RANGE_ERR_INVALID, "null argument");
    if (first_query < 0) return fail(RANGE_ERR_INVALID, "first_query must be >= 0");
    return attend_impl(c, nullptr, xq32, B, tau_sem, tau_geo, beta, stats_global, partial, nullptr,
                       first_query, stream);
}

// Diagnostic (not part of the product path): same launch as range_attend with the instrumented
// kernel build; diag_dev receives 16 x uint64 per (workgroup, wave): cycles parked in vmcnt waits,
// barriers, and spent in the PV / QK phases.  Outputs go to the split slabs only.
int range_attend_diag(range_ctx* c, const float* ehat32, const float* xq32, int64_t B, float tau_sem,
                      float tau_geo, float beta, const float* stats_global,
                      unsigned long long* diag_dev, int64_t diag_capacity, range_stream_t stream) {
    if (!c || !ehat32 || !xq32 || !stats_global || !diag_dev) return fail(RANGE_ERR_INVALID, "null argument");
    DeviceGuard g(c->device);
    ScanArgs a{};
    int rc = fill_scan_args(c, a, ehat32, xq32, B, tau_sem, tau_geo, false);
    if (rc) return rc;
    if (!(tau_geo > 0.f)) return fail(RANGE_ERR_INVALID, "diagnostic build exists for the geo variant only");
    if ((int64_t)a.n_splits * a.n_qtiles * 4 * 16 > diag_capacity) return fail(RANGE_ERR_INVALID, "diag buffer too small");
    HIP_TRY(c->ws_slabs.ensure((size_t)a.n_splits * B * VAL_DIM));
    a.out = c->ws_slabs.p;
    a.stats = stats_global;
    a.beta = beta;
    a.diag = diag_dev;
    rc = set_dyn_lds(attend_kernel<true, true>, ATTEND_LDS_BYTES);
    if (rc) return rc;
    hipLaunchKernelGGL((attend_kernel<true, true>), dim3((unsigned)(a.n_splits * a.n_qtiles)), dim3(256),
                       ATTEND_LDS_BYTES, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_finalize(range_ctx* c, const float* partials, int32_t n_parts, const double* ehat64,
                   int64_t B, double* out, range_stream_t stream) {
    if (!c || !partials || !ehat64 || !out) return fail(RANGE_ERR_INVALID, "null argument");
    if (n_parts <= 0 || B <= 0) return fail(RANGE_ERR_INVALID, "n_parts and B must be > 0");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    const int64_t n = B * 320;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, partials, SlabMap{n_parts, 0, 0, 0, 1}, ehat64, B, (int64_t)0, B, out);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

// Up to 32 queries: the whole retrieval in ONE pass over the bank (attend_small.h) - every CU
// streams its share of keys, locations and values once and accumulates the un-normalised products
// of both heads; small_finalize_kernel sums the workgroups' partials, normalises, blends and packs.
// e-hat / xq of the B queries are in the context's workspace (range_encode ran on `stream`).
static int forward_small(range_ctx* c, int64_t B, float tau_sem, float tau_geo, float beta, double* out,
                         hipStream_t s) {
    int rc = check_softmax_args(c, B, tau_sem, tau_geo);
    if (rc) return rc;
    if (tau_sem > RANGE_MAX_TAU || tau_geo > RANGE_MAX_TAU)
        return fail(RANGE_ERR_INVALID, "temperatures above %g are not supported", (double)RANGE_MAX_TAU);
    const double LOG2E = 1.4426950408889634;
    const int n_blocks = (int)((c->n_rows + BLK - 1) / BLK);
    const int n_wg = std::max(1, std::min(c->n_cu, n_blocks));
    const int nq = B > 16 ? 2 : 1, qcap = 16 * nq;            // query tiles of a workgroup
    HIP_TRY(c->ws_small_o.ensure((size_t)n_wg * 2 * qcap * VAL_DIM));
    HIP_TRY(c->ws_small_z.ensure((size_t)n_wg * qcap * 2));
    SmallArgs a{};
    a.keys = c->d_keys.p;
    a.xyz4 = c->d_xyz4.p;
    a.values = c->d_values.p;
    a.ehat = c->ws_ehat32.p;
    a.xq = c->ws_xq.p;
    a.osum = c->ws_small_o.p;
    a.zsum = c->ws_small_z.p;
    a.B = B;
    a.n_valid = c->n_rows;
    a.n_blocks = n_blocks;
    a.k_sem = (float)(tau_sem * LOG2E);
    a.k_geo = tau_geo > 0.f ? (float)(tau_geo * LOG2E) : 0.f;
    const bool geo = tau_geo > 0.f;
    {
        ProfScope ps(c, RANGE_PROF_ATTEND, s);
#define RANGE_SMALL_LAUNCH(G, Q)                                                                   \
    do {                                                                                           \
        rc = set_dyn_lds(attend_small_kernel<G, Q>, as_lds_bytes(Q));                              \
        if (rc) return rc;                                                                         \
        hipLaunchKernelGGL((attend_small_kernel<G, Q>), dim3((unsigned)n_wg), dim3(256), as_lds_bytes(Q), s, a); \
    } while (0)
        if (geo && nq == 2) RANGE_SMALL_LAUNCH(true, 2);
        else if (geo) RANGE_SMALL_LAUNCH(true, 1);
        else if (nq == 2) RANGE_SMALL_LAUNCH(false, 2);
        else RANGE_SMALL_LAUNCH(false, 1);
#undef RANGE_SMALL_LAUNCH
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(small_finalize_kernel, dim3((unsigned)B, 8), dim3(1024), 0, s, c->ws_small_o.p, c->ws_small_z.p,
                       n_wg, qcap, geo ? 1 : 0, geo ? beta : 1.0f, c->ws_ehat64.p, out);
    HIP_TRY(hipGetLastError());
    c->last_qtiles = 1;
    c->last_splits = n_wg;
    return RANGE_OK;
}

static int encode_to_workspace(range_ctx* c, const double* lonlat, int64_t B, range_stream_t stream) {
    {
        DeviceGuard g(c->device);
        if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
        HIP_TRY(c->ws_ehat64.ensure((size_t)B * 256));
        HIP_TRY(c->ws_ehat32.ensure((size_t)B * 256));
        HIP_TRY(c->ws_xq.ensure((size_t)B * 4));
        HIP_TRY(c->ws_stats.ensure((size_t)B * 4));
    }
    return range_encode(c, lonlat, B, c->ws_ehat64.p, c->ws_ehat32.p, c->ws_xq.p, stream);
}

// encode -> pass 1 (keeping its logits) -> pass 2 into the context's split slabs; the caller
// finalizes (sums the slabs, packs with e-hat).  n_splits_out = number of slabs written.
static int forward_to_slabs(range_ctx* c, const double* lonlat, int64_t B, int32_t model, float beta,
                            SlabMap* map_out, range_stream_t stream) {
    if (model != RANGE_MODEL_RANGE && model != RANGE_MODEL_RANGE_PLUS)
        return fail(RANGE_ERR_INVALID, "unknown model %d", model);
    if (B <= 0) return fail(RANGE_ERR_INVALID, "B must be > 0");
    {
        DeviceGuard g(c->device);
        if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
        HIP_TRY(c->ws_ehat64.ensure((size_t)B * 256));
        HIP_TRY(c->ws_ehat32.ensure((size_t)B * 256));
        HIP_TRY(c->ws_xq.ensure((size_t)B * 4));
        HIP_TRY(c->ws_stats.ensure((size_t)B * 4));
    }
    const float tau_sem = model == RANGE_MODEL_RANGE ? 15.0f : 12.0f;   // range.py:103, 108
    const float tau_geo = model == RANGE_MODEL_RANGE ? 0.0f : 40.0f;    // range.py:109
    int rc = range_encode(c, lonlat, B, c->ws_ehat64.p, c->ws_ehat32.p, c->ws_xq.p, stream);
    if (rc) return rc;
    rc = range_scan_stats(c, c->ws_ehat32.p, c->ws_xq.p, B, tau_sem, tau_geo, c->ws_stats.p, 0,
                          nullptr, nullptr, /*keep_logits=*/1, stream);
    if (rc) return rc;
    // pass 2 consumes the logits pass 1 kept (recomputes them if they did not fit in memory)
    return attend_impl(c, c->ws_ehat32.p, c->ws_xq.p, B, tau_sem, tau_geo,
                       model == RANGE_MODEL_RANGE ? 1.0f : beta, c->ws_stats.p, nullptr, map_out,
                       c->kept_B == B ? 0 : -1, stream);
}

int range_forward(range_ctx* c, const double* lonlat, int64_t B, int32_t model, float beta,
                  double* out, range_stream_t stream) {
    if (!c || !lonlat || !out) return fail(RANGE_ERR_INVALID, "null argument");
    if (model != RANGE_MODEL_RANGE && model != RANGE_MODEL_RANGE_PLUS)
        return fail(RANGE_ERR_INVALID, "unknown model %d", model);
    c->ws_queries = 0;
    if (B > 0 && B <= 32 && c->small_forward) {
        // a handful of queries: one pass over the bank (attend_small.h)
        int rc = encode_to_workspace(c, lonlat, B, stream);
        if (rc) return rc;
        c->ws_queries = B;
        DeviceGuard g(c->device);
        return forward_small(c, B, model == RANGE_MODEL_RANGE ? 15.0f : 12.0f, model == RANGE_MODEL_RANGE ? 0.0f : 40.0f,
                             beta, out, (hipStream_t)stream);
    }
    // single GPU: the finalize kernel sums the split slabs itself (same fixed order as
    // reduce_parts_kernel, so the result is bit-identical to attend + finalize)
    SlabMap map{};
    int rc = forward_to_slabs(c, lonlat, B, model, beta, &map, stream);
    if (rc) return rc;
    c->ws_queries = B;
    DeviceGuard g(c->device);
    const int64_t n = B * 320;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       c->ws_slabs.p, map, c->ws_ehat64.p, B, (int64_t)0, B, out);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_host_copy(range_ctx* c, void* dst, const void* src, size_t bytes) {
    if (!c || !dst || !src) return fail(RANGE_ERR_INVALID, "null argument");
    if (!c->pool) c->pool.reset(new HostCopyPool(HostCopyPool::default_threads()));
    c->pool->copy(dst, src, bytes);
    return RANGE_OK;
}

// The reference's contract (range/range.py:240): the result is a host array.  The finalize kernel
// runs per slab of queries; each finished slab is copied to pinned staging memory on a copy stream
// while the next is finalized, and the host threads move it into the caller's array (first-touch
// page faults of a fresh array spread over the threads) while the DMA of the next is in flight.
int range_forward_host(range_ctx* c, const double* lonlat, int64_t B, int32_t model, float beta,
                       double* out_host, range_stream_t stream) {
    if (!c || !lonlat || !out_host) return fail(RANGE_ERR_INVALID, "null argument");
    if (model != RANGE_MODEL_RANGE && model != RANGE_MODEL_RANGE_PLUS)
        return fail(RANGE_ERR_INVALID, "unknown model %d", model);
    if (B <= 0) return fail(RANGE_ERR_INVALID, "B must be > 0");
    DeviceGuard g(c->device);
    if (!g.ok) return fail(RANGE_ERR_HIP, "hipSetDevice(%d) failed", c->device);
    hipStream_t s = (hipStream_t)stream;
    const auto t_begin = std::chrono::steady_clock::now();
    const size_t row_bytes = (size_t)RANGE_OUT_DIM * sizeof(double);
    HIP_TRY(c->ws_ehat64.ensure((size_t)B * 256));
    HIP_TRY(c->ws_ehat32.ensure((size_t)B * 256));
    HIP_TRY(c->ws_xq.ensure((size_t)B * 4));
    HIP_TRY(c->ws_stats.ensure((size_t)B * 4));
    HIP_TRY(c->ws_out64.ensure((size_t)B * RANGE_OUT_DIM));
    if (c->h_stage_bytes < (size_t)B * row_bytes) {
        if (c->h_stage) (void)hipHostFree(c->h_stage);
        c->h_stage = nullptr;
        c->h_stage_bytes = 0;
        HIP_TRY(hipHostMalloc(&c->h_stage, (size_t)B * row_bytes, hipHostMallocDefault));
        c->h_stage_bytes = (size_t)B * row_bytes;
    }
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->pool) c->pool.reset(new HostCopyPool(HostCopyPool::default_threads()));

    const float tau_sem = model == RANGE_MODEL_RANGE ? 15.0f : 12.0f;   // range.py:103, 108
    const float tau_geo = model == RANGE_MODEL_RANGE ? 0.0f : 40.0f;    // range.py:109
    const float bt = model == RANGE_MODEL_RANGE ? 1.0f : beta;
    c->ws_queries = 0;
    int rc = range_encode(c, lonlat, B, c->ws_ehat64.p, c->ws_ehat32.p, c->ws_xq.p, stream);
    if (rc) return rc;
    c->ws_queries = B;
    if (B <= 32 && c->small_forward) {
        // a handful of queries: one pass over the bank, one small copy
        rc = forward_small(c, B, tau_sem, tau_geo, bt, c->ws_out64.p, s);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->h_stage, c->ws_out64.p, (size_t)B * row_bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (int rc2 = check_async_error(c)) return rc2;
        std::memcpy(out_host, c->h_stage, (size_t)B * row_bytes);
        return RANGE_OK;
    }
    rc = range_scan_stats(c, c->ws_ehat32.p, c->ws_xq.p, B, tau_sem, tau_geo, c->ws_stats.p, 0,
                          nullptr, nullptr, /*keep_logits=*/1, stream);
    if (rc) return rc;
    const bool kept = c->kept_B == B;

    // Pass 2 runs in a few launches over consecutive query ranges (boundaries on query tiles): the
    // device->host copy and the host fill of a part overlap pass 2 of the next, so only the LAST
    // part's copy and fill are exposed - and the parts SHRINK: the rest, 4 096, 512 queries.  What
    // bounds the end is the copy queue (RANGE_HOST_TIMING=1 prints when each slab's copy was seen):
    // 0.205 ms per 1 000 queries (10 KB each at 52 GB/s) against 1.46 ms of pass 2 per 1 000 queries
    // against range_db_large, so a part up to seven times its successor is drained while the
    // successor computes; with (4 096, 1 024) the last slab was seen 0.35 ms after pass 2 ended, with
    // (4 096, 512) 0.15 ms (two equal halves: 1.4 ms).  Part sizes matter on the GPU side too: the
    // split count of pass 2 is chosen per launch so that its workgroups fill whole rounds of the
    // chip (8 query tiles x 32 bank splits, 64 x 4: exactly one round) - measured for 10 000 queries,
    // medians of 30 calls (tools/host_parts.py; the box wanders by +-0.1 ms): (4 096, 512) 19.71 ms,
    // (2 048, 256) 19.70, (3 072, 512) 19.87, (4 096, 1 024) 19.96, (1 792, 256) 19.98.
    std::vector<int64_t> cuts{0, B};
    if (B >= 4096) {
        // RANGE_HOST_PARTS="2048,512": sizes of the parts behind the first (tuning)
        std::vector<int64_t> tail{4096, 512};
        if (const char* e = std::getenv("RANGE_HOST_PARTS")) {
            tail.clear();
            for (const char* q = e; *q;) {
                char* end = nullptr;
                const long v = std::strtol(q, &end, 10);
                if (end == q) break;
                if (v > 0) tail.push_back(v);
                q = *end ? end + 1 : end;
            }
        }
        cuts = host_part_cuts(B, tail, QTILE);       // (host_plan.h: rounded up to a tile, clamped, monotonic)
    }
    constexpr int64_t SLAB = 1024;
    struct Slab { int64_t q0, nq; hipEvent_t fin, cop; };
    std::vector<Slab> slabs;
    auto give_back = [&]() { for (auto& sl : slabs) { c->ev_pool.push_back(sl.fin); c->ev_pool.push_back(sl.cop); } };
    hipError_t e = hipSuccess;
    for (size_t part = 0; part + 1 < cuts.size() && e == hipSuccess; ++part) {
        const int64_t p0 = cuts[part], p1 = cuts[part + 1];
        if (p1 <= p0) continue;
        SlabMap map{};
        rc = attend_impl(c, c->ws_ehat32.p + p0 * 256, c->ws_xq.p + p0 * 4, p1 - p0, tau_sem, tau_geo, bt,
                         c->ws_stats.p + p0 * 4, nullptr, &map, kept ? p0 : -1, stream);
        if (rc) { (void)hipDeviceSynchronize(); give_back(); return rc; }
        // finalize per slab (the split slabs of this part are overwritten by the next part's pass 2,
        // which is behind these kernels in stream order)
        // (the last part in slabs of 256 queries: its copies and fills are what the caller waits for,
        // and they pipeline per slab)
        const int64_t slab = part + 2 == cuts.size() && cuts.size() > 2 ? SLAB / 4 : SLAB;
        for (int64_t q0 = p0; q0 < p1 && e == hipSuccess; q0 += slab) {
            Slab sl{q0, std::min<int64_t>(slab, p1 - q0), c->get_event(), c->get_event()};
            slabs.push_back(sl);
            const int64_t n = sl.nq * 320;
            hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                               c->ws_slabs.p, map, c->ws_ehat64.p + p0 * 256, p1 - p0, q0 - p0, sl.nq,
                               c->ws_out64.p + p0 * RANGE_OUT_DIM);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(sl.fin, s);
            if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, sl.fin, 0);
            if (e == hipSuccess)
                e = hipMemcpyAsync((char*)c->h_stage + q0 * row_bytes, c->ws_out64.p + q0 * RANGE_OUT_DIM,
                                   (size_t)sl.nq * row_bytes, hipMemcpyDeviceToHost, c->copy_stream);
            if (e == hipSuccess) e = hipEventRecord(sl.cop, c->copy_stream);
        }
    }
    // Everything is enqueued and this thread would only wait now: the copy threads touch every page of
    // the caller's array meanwhile (one write per 4 KB page; the data follows).  A fresh array is
    // 25 000 untouched pages per 10 000 queries: faulted here, under the GPU's shadow, the fills
    // below are plain copies - also the last part's, which the caller waits for.
    if (e == hipSuccess && (size_t)B * row_bytes >= ((size_t)1 << 20)) {
        char* base = (char*)out_host;
        const size_t bytes = (size_t)B * row_bytes;
        c->pool->run([=](int t, int n) {
            const size_t per = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
            const size_t lo = per * (size_t)t;
            const size_t hi = lo + per < bytes ? lo + per : bytes;
            for (size_t o = lo; o < hi; o += 4096) *(volatile char*)(base + o) = 0;
        });
    }
    const auto t_enq = std::chrono::steady_clock::now();
    double wait_s = 0.0, copy_s = 0.0;
    std::vector<double> landed;                 // (host_timing) when each slab's copy was seen, ms after entry
    for (auto& sl : slabs) {
        if (e != hipSuccess) break;
        const auto w0 = std::chrono::steady_clock::now();
        e = hipEventSynchronize(sl.cop);
        const auto w1 = std::chrono::steady_clock::now();
        if (e != hipSuccess) break;
        c->pool->copy((char*)out_host + sl.q0 * row_bytes, (const char*)c->h_stage + sl.q0 * row_bytes,
                      (size_t)sl.nq * row_bytes);
        const auto w2 = std::chrono::steady_clock::now();
        wait_s += std::chrono::duration<double>(w1 - w0).count();
        copy_s += std::chrono::duration<double>(w2 - w1).count();
        if (c->host_timing) landed.push_back(std::chrono::duration<double>(w1 - t_begin).count() * 1e3);
    }
    if (e != hipSuccess) {
        (void)hipDeviceSynchronize();
        give_back();
        return fail(RANGE_ERR_HIP, "range_forward_host: %s", hipGetErrorString(e));
    }
    give_back();
    if (int rc2 = check_async_error(c)) return rc2;
    if (c->host_timing)
        std::fprintf(stderr, "range_forward_host B=%lld: enqueue %.2f ms, waiting for slabs %.2f ms, host fill %.2f ms "
                     "(%d threads), total %.2f ms\n", (long long)B,
                     std::chrono::duration<double>(t_enq - t_begin).count() * 1e3, wait_s * 1e3, copy_s * 1e3,
                     c->pool->size(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() * 1e3);
    if (c->host_timing) {
        std::fprintf(stderr, "  slab copies seen at (ms):");
        for (size_t i = 0; i < landed.size(); ++i) std::fprintf(stderr, " %.2f(%lld)", landed[i], (long long)slabs[i].nq);
        std::fprintf(stderr, "\n");
    }
    return RANGE_OK;
}

int range_profile_enable(range_ctx* c, int32_t on) {
    if (!c) return fail(RANGE_ERR_INVALID, "null argument");
    DeviceGuard g(c->device);
    for (auto& v : c->prof) {
        for (auto& p : v) { (void)hipEventSynchronize(p.second); c->ev_pool.push_back(p.first); c->ev_pool.push_back(p.second); }
        v.clear();
    }
    c->profile = on != 0;
    return RANGE_OK;
}

int range_profile_read(range_ctx* c, int32_t which, double* total_ms, int32_t* launches) {
    if (!c || which < 0 || which >= RANGE_PROF_KINDS || !total_ms || !launches) return fail(RANGE_ERR_INVALID, "bad argument");
    DeviceGuard g(c->device);
    double sum = 0.0;
    for (auto& p : c->prof[which]) {
        HIP_TRY(hipEventSynchronize(p.second));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int32_t)c->prof[which].size();
    return check_async_error(c);     // (the events above synchronised: a give-up of the profiled calls is known now)
}

int range_last_attend_geometry(const range_ctx* c, int32_t* n_query_tiles, int32_t* n_splits) {
    if (!c) return fail(RANGE_ERR_INVALID, "null argument");
    if (n_query_tiles) *n_query_tiles = c->last_qtiles;
    if (n_splits) *n_splits = c->last_splits;
    return RANGE_OK;
}

}  // extern "C"

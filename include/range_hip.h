/*
 * range_hip.h - C ABI of librange_hip.so: the MI355X (gfx950) engine behind
 * range_amd.load_model(...)(locs), the drop-in for the RANGE / RANGE+ forward path of mvrl/RANGE.
 *
 * The reference is pure Python / PyTorch and has NO plugin, operator or FFI layer for this path
 * (SURVEY.md section 8(b)); the "interface each entry point replaces" is therefore a span of the
 * reference's Python, cited per function as file:line relative to the reference root.
 * The Python host (range_amd/_native.py, ctypes) is the only intended caller, but nothing here
 * depends on Python or torch: plain pointers, sizes and a hipStream_t passed as void*.
 *
 * Conventions
 *   - every function returns RANGE_OK (0) or a negative error code; range_last_error() gives the
 *     message of the last failure on the calling thread.  Nothing throws across the ABI.
 *   - "dev" pointers are device memory owned by the caller (e.g. torch tensors' data_ptr());
 *     "host" pointers are ordinary host memory, read during the call (written by
 *     range_forward_host / range_host_copy, whose result IS host memory).
 *   - a ctx is bound to one GPU, is not thread-safe, owns its bank copy and workspace
 *     (hipMalloc), and may be used from any stream; calls are asynchronous on `stream` except
 *     where stated.  Several ctxs (one per GPU / per process) may coexist.
 *   - all kernels are hand-written HIP for gfx950; there is no CPU fallback.
 */
#ifndef RANGE_HIP_H
#define RANGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RANGE_ABI_VERSION 9

#define RANGE_KEY_DIM 256   /* satclip_embeddings width, range/range.py:85-86 */
#define RANGE_VAL_DIM 1024  /* image_embeddings width,   range/range.py:86, 90 */
#define RANGE_OUT_DIM 1280  /* location_feature_dim,     range/range.py:86     */
#define RANGE_MAX_TAU 43.0f /* largest temperature (the reference's: 12, 15, 40; range.py:103-109) */

enum {
    RANGE_OK = 0,
    RANGE_ERR_INVALID = -1, /* bad argument / unsupported shape */
    RANGE_ERR_HIP = -2,     /* a HIP runtime call failed */
    RANGE_ERR_STATE = -3,   /* encoder or bank not set */
    RANGE_ERR_NOMEM = -4
};

enum { RANGE_SH_ANALYTIC = 0, RANGE_SH_CLOSED_FORM = 1 };
enum { RANGE_MODEL_RANGE = 0, RANGE_MODEL_RANGE_PLUS = 1 };

typedef struct range_ctx range_ctx;
typedef void* range_stream_t; /* hipStream_t */

/* Hyper-parameters of the SatCLIP location encoder, as carried by the checkpoint's
 * hyper_parameters (satclip/main_old.py:15-37; satclip/model_old.py:326-330). */
typedef struct range_encoder_desc {
    int32_t legendre_polys;    /* L; SH feature count is L*L (spherical_harmonics.py:19-20) */
    int32_t hidden;            /* capacity H, 1..1024 (kernels exist for multiples of 64 up to 512, 768, 1024;
                                  other widths run zero-padded to the next of those: same result, bit for bit) */
    int32_t num_hidden_layers; /* SirenNet num_layers (>= 1) */
    int32_t embed_dim;         /* must be RANGE_KEY_DIM */
    int32_t sh_mode;           /* RANGE_SH_ANALYTIC | RANGE_SH_CLOSED_FORM (spherical_harmonics.py:22-25) */
} range_encoder_desc;

int range_abi_version(void);
const char* range_last_error(void);
/* The extra compiler flags this library was built with (build.sh records them); "" for the
 * default build.  A build carrying RANGE_EXP_* switches is a timing experiment whose results are
 * invalid: the Python binding refuses to load it. */
const char* range_build_flags(void);
/* SHA-256 (hex) of the sources this library was built from (range_amd/_srchash.py; build.sh embeds
 * it): the binding refuses a library whose stamp is not the checkout's, __graft_entry__.build() rebuilds. */
const char* range_source_sha256(void);

/* Create / destroy an engine context on GPU `device`. */
int range_create(int device, range_ctx** out);
void range_destroy(range_ctx* ctx);

/* Replaces get_satclip(...).double() + the module tree it builds
 * (satclip/load.py:3-18, satclip/location_encoder.py:73-112, 267-275; range/range.py:82-84).
 * weights[i] / biases[i], i = 0..num_hidden_layers: float64 HOST arrays in torch (out,in)
 * row-major layout: layers.{i}.weight (H x in), layers.{i}.bias (H), last entry = last_layer
 * (embed_dim x H).  The library re-packs them into MFMA fragment order on the device.
 * Synchronous. */
int range_set_encoder(range_ctx* ctx, const range_encoder_desc* desc,
                      const double* const* weights, const double* const* biases);

/* Reference-faithful "analytic" spherical harmonics.  range_set_encoder evaluates the basis by a
 * stable recurrence (the exact-math value).  The reference evaluates machine-generated, fully
 * expanded polynomials in cos(theta) whose coefficients (up to 1e14) are printed with 15 digits
 * (satclip/positional_encoding/spherical_harmonics_generate_ylms.py:19-40; used at
 * spherical_harmonics.py:35-42): towards the poles they cancel to values that differ from the
 * exact ones by up to 0.5, deterministically.  With this table (range_amd/sh_table.py: parsed from
 * the generated file or generated from scratch) the kernel walks the same sums in the same order.
 * Arrays are indexed l*L+m (0 <= m <= l < L); terms off[i] .. off[i]+cnt[i]-1 of coef/pow:
 *   Y_l^m = front * (a0 + a2 x^2)^(p2/2) * x^kx * sum_j coef_j x^pow_j * cos|sin(m phi), x = cos(theta).
 * Call after range_set_encoder (which clears a previous table); analytic sh_mode only. */
int range_set_sh_table(range_ctx* ctx, int32_t L, const double* front, const double* a0,
                       const double* a2, const int32_t* p2, const int32_t* kx, const int32_t* off,
                       const int32_t* cnt, int64_t n_terms, const double* coef, const int32_t* pow);

/* Replaces the bank upload of range/range.py:98-100 (and the per-forward re-upload of the values
 * at :217/:236).  Inputs are HOST arrays already prepared exactly as range/range.py:78-95 does:
 * keys (n_rows x 256) float32 rows L2-normalised in float32; values (n_rows x 1024) float32;
 * xyz (n_rows x 3) float32 unit vectors.  For a row-sharded bank pass this rank's rows and the
 * global index of its first row (used only to report top-k indices).  Synchronous. */
int range_set_bank(range_ctx* ctx, const float* keys, const float* values, const float* xyz,
                   int64_t n_rows, int64_t row_offset);
int64_t range_bank_rows(const range_ctx* ctx);

/* A keys-only bank for the top-k side channel (range_topk_stream): the satclip_embeddings column of
 * the bank alone (range/range.py:85-89), 1 KB per row instead of 5 KB.  `keys` (n_rows x 256 float32)
 * may be a HOST or a DEVICE pointer.  Replaces the bank of the context; every call that needs the
 * values or the locations (range_scan_stats, range_attend*, range_forward*) then returns
 * RANGE_ERR_STATE.  Rows need not be normalised (the top-k tolerates any norm).  Synchronous. */
int range_set_keys(range_ctx* ctx, const float* keys, int64_t n_rows, int64_t row_offset);

/* Opt-in arithmetic of the w @ V products of pass 2 (range/range.py:217, :236).  NOT in the
 * reference; the default is what the reference computes.
 *   RANGE_PV_EXACT  (default) exact float32 products (v_mfma_f32_16x16x4_f32)
 *   RANGE_PV_BF16X3 both operands split into three bf16 planes, the six largest cross products
 *                   accumulated in float32 (~2^-22 relative to the exact products); needs 6 B per
 *                   bank value of extra device memory, built here or at the next range_set_bank;
 *                   used by range_attend_kept / range_forward on kept logits only (a pass 2 that
 *                   recomputes its logits stays exact).  Synchronous. */
#define RANGE_PV_EXACT 0
#define RANGE_PV_BF16X3 1
int range_set_pv_mode(range_ctx* ctx, int32_t mode);
int32_t range_get_pv_mode(const range_ctx* ctx);

/* Kernel A.  Replaces self.loc_model(coords) and the normalisation of range/range.py:210-212
 * (spherical_harmonics.py:27-42 + location_encoder.py:98-112, fused, float64) and the query half
 * of range/range.py:225-229 (degrees -> unit xyz).
 *   lonlat_dev : (B,2) float64, (lon,lat) degrees
 *   ehat64_dev : (B,256) float64   normalised embedding (output columns 1024:1280)
 *   ehat32_dev : (B,256) float32   the .float() operand of range.py:213
 *   xq32_dev   : (B,4)   float32   (x,y,z,0), the .float() operand of range.py:231
 * Up to 512 queries run as ONE persistent launch whose workgroups wait for each other inside the
 * kernel (bounded: seconds).  Should such a wait ever give up - only when something else holds the
 * GPU's CUs that long - the rows of THAT call's outputs which could not be finished are written as
 * NaN (never stale memory), a word of host-mapped memory is set, and the context runs the encoder
 * as separate launches from then on.  The failure is reported as RANGE_ERR_HIP by the next call that
 * looks: range_encode / range_forward* / range_topk_stream at entry, range_forward_host /
 * range_profile_read / range_topk_stream_exact_count before they return (they synchronise), and
 * range_check_async_error, which a caller invokes behind its own synchronisation (the Python host
 * does behind every .cpu()).  Reported once; the re-issued call takes the fall-back path.  The
 * fused top-k (range_topk_stream) follows the same rules with NaN values / index -1. */
int range_check_async_error(range_ctx* ctx);
/* Test hook: the NEXT persistent launch on this context (one-launch encoder or fused top-k)
 * behaves as if its in-launch wait had expired.  `stream` is unused. */
int range_debug_raise_async_error(range_ctx* ctx, range_stream_t stream);
/* The give-up words as DATA, in stream order: *flag_dev (one float64 of device memory) = 1.0 when a
 * persistent launch enqueued before this call on this context has given up and the host has not yet
 * looked (range_check_async_error clears the words), else 0.0.  A rank of a row-sharded job sends the
 * flag with its rows (range_amd/save.py, range_amd/range.py): every rank then learns of a peer's
 * give-up from the WORD - never from NaN in the data, which a NaN coordinate produces as well
 * (a NaN / infinite coordinate gives a NaN row, as in the reference; rows are independent) - and all
 * ranks refuse the batch together.  No reference counterpart. */
int range_async_error_flag(range_ctx* ctx, double* flag_dev, range_stream_t stream);
int range_encode(range_ctx* ctx, const double* lonlat_dev, int64_t B, double* ehat64_dev,
                 float* ehat32_dev, float* xq32_dev, range_stream_t stream);

/* Kernel A, plain SatCLIP output: the un-normalised (B,256) float64 embedding that
 * LocationEncoder.forward returns for model_name == 'SatCLIP' (range/range.py:244-245). */
int range_encode_raw(range_ctx* ctx, const double* lonlat_dev, int64_t B, double* eraw64_dev,
                     range_stream_t stream);

/* The reference's training-free coordinate encoders (load_model names 'Direct', 'Cartesian_3D',
 * 'Wrap'; range/range.py:152-162, 170-173, 262-272).  float64, elementwise.
 *   mode 0 Direct       : out (B,2) = (lon,lat)*pi/180                       (range.py:262-264)
 *   mode 1 Cartesian_3D : out (B,3) = rad_to_cart of the above              (:265-268, utils/utils.py:11-16)
 *   mode 2 Wrap         : out (B,4) = (cos lon, sin lon, cos lat, sin lat) of deg2rad
 *                         (positional_encoding/wrap.py:20-29) */
#define RANGE_COORD_DIRECT 0
#define RANGE_COORD_CARTESIAN3D 1
#define RANGE_COORD_WRAP 2
int range_coord_features(range_ctx* ctx, int32_t mode, const double* lonlat_dev, int64_t B,
                         double* out_dev, range_stream_t stream);

/* Kernel B, pass 1.  Streaming log-sum-exp statistics of the temperature-scaled logits of
 * range/range.py:213-215 (semantic) and :231-234 (geographic) over THIS ctx's bank rows.
 *   tau_sem, tau_geo : temperatures (range.py:103, 108-109); tau_geo <= 0 disables the geo head
 *   ehat32_dev / xq32_dev rows must be UNIT vectors (what range_encode emits; range.py:212,
 *               utils.py:11-16) and the bank keys / xyz are (range.py:85-89, :93-95): every logit
 *               is then <= 1 and the statistics use the constant shift m = tau*log2(e) instead of
 *               a running maximum (no rescaling; partial statistics merge by plain sums).
 *               Temperatures above RANGE_MAX_TAU are rejected (2^(-2m) must stay a normal float),
 *               and so is a bank whose largest key or location row norm exceeds 1.001
 *               (RANGE_ERR_INVALID: the constant shift would overflow).
 *   stats_dev : (B,4) float32 = {m_sem, l_sem, m_geo, l_geo}, m = tau*log2(e),
 *               l = sum 2^(tau*log2(e)*logit - m) over the rows; log-sum-exp = (m + log2 l)/log2 e.
 *               Statistics of disjoint row sets (bank splits, bank shards) of the same query
 *               have the same m and their l add: range_merge_stats, or an all-reduce(sum).
 *   topk : 0, or k in [1,16]: also emit the k largest semantic similarities of each query
 *          (the "brute-force top-k" side channel), descending, ties -> lower row index:
 *          topk_val_dev (B,k) float32, topk_idx_dev (B,k) int64 (global row = row_offset + local).
 *          The similarities are written to HBM by pass 1 and the top-k is selected from them
 *          by a streaming kernel with a per-query threshold (list maintenance inside pass 1
 *          costs more than its MFMAs); a context that cannot keep logits (memory,
 *          RANGE_KEEP_LOGITS=0) keeps per-lane lists inside pass 1 instead.  Same result.
 *   keep_logits : non-zero: keep the raw semantic dot products of this call
 *          in the context (4 bytes per (query, bank row) of workspace) for range_attend_kept.  They
 *          stay valid until the next range_scan_stats / range_set_bank on this ctx.  Silently not
 *          kept when they would take more than half of the free device memory, or when the
 *          context was created with RANGE_KEEP_LOGITS=0 in the environment (A/B timing, tests). */
int range_scan_stats(range_ctx* ctx, const float* ehat32_dev, const float* xq32_dev, int64_t B,
                     float tau_sem, float tau_geo, float* stats_dev, int topk,
                     float* topk_val_dev, int64_t* topk_idx_dev, int32_t keep_logits,
                     range_stream_t stream);

/* Pass 1 in CHUNKS of one scan (row-sharded banks, range_amd/dist.py: the gather of chunk c + 1
 * and the exchange of chunk c's statistics travel while chunk c + 1 / c is being scanned; no
 * reference counterpart).  Queries [first_query, first_query + B) of a scan of `total_queries`
 * queries; their logits are kept at that offset of ONE workspace sized for the whole scan, so that
 * range_attend_kept(first, ...) addresses the scan's queries as if one range_scan_stats call had
 * kept them.  first_query must be a multiple of 64; the call with first_query == 0 starts a scan
 * (and decides, as range_scan_stats does, whether the logits fit), later calls extend it in order
 * (first_query == range_kept_queries(); otherwise - or when the first call could not keep - the
 * statistics are still computed and nothing is kept).
 *   n_splits   : bank splits of this launch (0: chosen from this call's geometry).  The float32 sums l
 *                depend on the split boundaries: a caller that wants the statistics of a query not to
 *                depend on how its scan was chunked passes the same value for every chunk
 *                (range_p1_splits(ctx, B) = what a call of B queries would choose). */
int range_scan_stats_at(range_ctx* ctx, const float* ehat32_dev, const float* xq32_dev, int64_t B,
                        float tau_sem, float tau_geo, float* stats_dev, int64_t first_query,
                        int64_t total_queries, int32_t n_splits, range_stream_t stream);
int32_t range_p1_splits(const range_ctx* ctx, int64_t B);

/* Number of queries whose logits the last range_scan_stats (or the range_scan_stats_at calls of the
 * current scan so far) kept (0: none). */
int64_t range_kept_queries(const range_ctx* ctx);

/* Small-batch top-k, the HBM-streaming form of the keys scan (no reference counterpart; north star:
 * "brute-force cosine-similarity top-k ... coalesced HBM-streaming kernel with per-wavefront
 * running top-k").  A persistent grid (one workgroup per CU): every wave streams its own 16-row
 * key tiles through a wave-private LDS ring against groups of 16 queries in registers; 1 or 2
 * groups share one pass over the keys, further groups take further passes inside the same
 * launch.  Candidate lists are merged on the way up (lane -> wave -> workgroup) and, for batches of
 * up to one query per workgroup (256), the final merge runs as the TAIL OF THE SAME LAUNCH: the last
 * workgroups to finish their tiles merge one query each (one launch answers the call); larger
 * batches run the same merge as a second launch.  Faster than range_scan_stats' top-k at every
 * batch size.  Same outputs and tie rule as the top-k of range_scan_stats.  By default the scan
 * reads a bf16 copy of the keys (built by range_set_bank / range_set_keys) and the candidates within
 * its error bound are re-ranked with the float32 similarity: the results are those of the float32
 * scan bit for bit (RANGE_TOPKS_KEYS=f32 in the environment selects that one).  Per-lane candidate
 * lists are short (4 entries); a query whose lists may have dropped a top-k member (detected
 * exactly) is recomputed by brute force inside the merge - range_topk_stream_exact_count reports
 * how many queries took that path since the context was created (synchronises the device; it also
 * returns RANGE_ERR_HIP if a merging workgroup of an earlier call gave up waiting for the stream
 * workgroups - possible only when the grid cannot be resident as a whole - in which case that
 * call's results for its query carry index -1 / NaN).
 * STREAM ORDER: the fused form's arrival counters are checked against a running base the HOST keeps
 * and passes by value with each launch - one fused launch of a context in flight at a time, in the
 * order the host issued them.  Calls on one context must therefore be issued on ONE stream (or on
 * streams the caller orders) and must NOT be captured into a HIP graph (a replay would hand the kernel
 * a stale base: workgroups would pick wrong merge slots or leave outputs unwritten).  A caller that
 * needs either uses separate contexts, or RANGE_TOPKS_FUSED=0 (merge as a second launch: no counters). */
int range_topk_stream(range_ctx* ctx, const float* ehat32_dev, int64_t B, int32_t k,
                      float* topk_val_dev, int64_t* topk_idx_dev, range_stream_t stream);
int range_topk_stream_exact_count(range_ctx* ctx, int64_t* count);
/* model(coords, return_topk=k) (SURVEY.md 8(b) "Call"; the reference only hints at it:
 * range/range.py:232): the top-k of the B queries the LAST range_forward / range_forward_host call of
 * this context embedded.  Their float32 e-hat is still in the context's workspace, so the side channel
 * costs its scan alone (range_topk_stream's kernels on that operand: the same values and indices as
 * range_topk_stream on range_encode's output, bit for bit) - no second encoder pass, no second call
 * into the Python layer's encode.  RANGE_ERR_STATE when B is not the last forward's query count. */
int range_topk_last(range_ctx* ctx, int64_t B, int32_t k, float* topk_val_dev, int64_t* topk_idx_dev,
                    range_stream_t stream);
/* Bench harness: the same call enqueued `repeats` (>= 2) times back to back between ONE pair of
 * HIP events on `stream` (a pair around a single ~15 us launch adds ~5 us of dispatch latency to
 * it); *avg_us = time per call (every launch of the call: the scan with its merge tail, or the
 * scan and the merge launch).  Synchronises with the stream. */
int range_topk_stream_timed(range_ctx* ctx, const float* ehat32_dev, int64_t B, int32_t k,
                            float* topk_val_dev, int64_t* topk_idx_dev, int32_t repeats,
                            float* avg_us, range_stream_t stream);

/* Bench harness: the ceiling of range_topk_stream's stream - a plain kernel that reads, in ONE
 * launch, the bytes a call streams (`passes` times the bf16 copy of the keys, or the float32 keys when
 * f32_keys != 0) and does nothing else; `repeats` (>= 2) launches back to back between one pair of
 * HIP events, *avg_us = time per launch.  bench.py reports a call's time against it (frac_of_copy).
 * Synchronises with the stream. */
int range_stream_read_timed(range_ctx* ctx, int32_t f32_keys, int32_t passes, int32_t repeats,
                            float* avg_us, range_stream_t stream);

/* Exact merge of per-shard statistics (row-sharded bank): parts_dev is (n_parts,B,4) as written
 * by range_scan_stats on each shard (e.g. after an all-gather); out_dev is (B,4). */
int range_merge_stats(range_ctx* ctx, const float* parts_dev, int32_t n_parts, int64_t B,
                      float* out_dev, range_stream_t stream);

/* Merge per-shard top-k candidate lists: (n_parts,B,k) values / global indices -> (B,k). */
int range_merge_topk(range_ctx* ctx, const float* val_parts_dev, const int64_t* idx_parts_dev,
                     int32_t n_parts, int64_t B, int32_t k, float* val_out_dev,
                     int64_t* idx_out_dev, range_stream_t stream);

/* Kernel B, pass 2.  Replaces range/range.py:213-217, 231-238: with the GLOBAL statistics of
 * pass 1 it recomputes the logits tile by tile, forms w = beta*p_sem + (1-beta)*p_geo and
 * accumulates w @ values over THIS ctx's rows (float32 MFMA, exact f32 products).
 *   beta : range.py:238 blend; with tau_geo <= 0 (plain RANGE, range.py:222) pass beta = 1
 *   partial_dev : (B,1024) float32.  Partials of different shards simply add.
 * ROUNDING AND POSITION.  The float32 sum over the bank rows is formed in pieces (bank splits, or - for
 * banks / shards of up to 50 000 rows, the default there - the segments of a persistent stream-K walk
 * over (query tile, bank block) units) that are added in a fixed order: a call is deterministic, bit for
 * bit, under repetition.  With the split scheme a query's pieces do not depend on where the query sits in
 * the batch; with the stream-K walk the cut points of a query TILE depend on the tile's index, so the
 * same query at another position of an equal-sized batch - or in a batch of another size - may differ in
 * the last bits (measured: < 2e-6 absolute on unit-scale values, tests/test_gpu_round6.py; the reference's
 * own sgemm blocking depends on the batch shape in the same way).  RANGE_P2_STREAMK=0 in the environment
 * at range_create selects the position-independent split scheme at every bank size. */
int range_attend(range_ctx* ctx, const float* ehat32_dev, const float* xq32_dev, int64_t B,
                 float tau_sem, float tau_geo, float beta, const float* stats_global_dev,
                 float* partial_dev, range_stream_t stream);

/* Kernel B, pass 2 on the logits KEPT by the last range_scan_stats(keep_logits = 1) of this ctx:
 * same result as range_attend, bit for bit, for the queries [first_query, first_query + B) of
 * that scan, without recomputing e . K^T (a fifth of pass 2's MFMA work; the kernel is MFMA-bound
 * and reads the kept tiles back at 4 bytes per (query, row)).  first_query must be a multiple of
 * 64; xq32_dev / stats_global_dev / partial_dev are those B queries' rows.  tau_sem, tau_geo and
 * beta are free (the kept values are un-scaled): one scan serves several attends (beta sweeps).
 * RANGE_ERR_STATE when nothing is kept. */
int range_attend_kept(range_ctx* ctx, int64_t first_query, const float* xq32_dev, int64_t B,
                      float tau_sem, float tau_geo, float beta, const float* stats_global_dev,
                      float* partial_dev, range_stream_t stream);

/* The blend of range/range.py:238 on its own, with the reference's float32 rounding:
 * out = (1-beta)*G + beta*H over (B,1024) float32.  For beta sweeps: G = range_attend(beta=0),
 * H = range_attend(beta=1) once, then one blend + finalize per beta. */
int range_blend(range_ctx* ctx, const float* G_dev, const float* H_dev, float beta, int64_t B,
                float* out_dev, range_stream_t stream);

/* Replaces the pack of range/range.py:222 / :240: out (B,1280) float64 =
 * [ sum over parts of partials (n_parts,B,1024) f32 widened | ehat64 (B,256) ]. */
int range_finalize(range_ctx* ctx, const float* partials_dev, int32_t n_parts,
                   const double* ehat64_dev, int64_t B, double* out_dev, range_stream_t stream);

/* Whole path on one GPU (bank not sharded): encode -> stats -> attend -> finalize.
 * Replaces LocationEncoder.forward for 'RANGE' / 'RANGE+' (range/range.py:206-240) up to the
 * final device->host copy, which stays in the Python shim.
 *   model : RANGE_MODEL_RANGE (tau 15) | RANGE_MODEL_RANGE_PLUS (tau 12 / 40, beta blend)
 *   out_dev : (B,1280) float64 */
int range_forward(range_ctx* ctx, const double* lonlat_dev, int64_t B, int32_t model, float beta,
                  double* out_dev, range_stream_t stream);

/* The same path with the reference's own output contract: the result lands in HOST memory
 * (range/range.py:240 returns a numpy array; range/utils/save.py:27-30 consumes it).  out_host is
 * the caller's (B,1280) float64 array, pageable and typically fresh (untouched pages).  The call
 * returns when out_host is complete (it is synchronous, like the reference's `.cpu()`): finalize
 * runs per slab of 1024 queries, each slab's device->host DMA (pinned staging, a copy stream of
 * the context) overlaps the next slab's, and a few host threads (RANGE_HOST_THREADS, default
 * min(8, cores)) move landed slabs into out_host so that its first-touch page faults are spread
 * over cores. */
int range_forward_host(range_ctx* ctx, const double* lonlat_dev, int64_t B, int32_t model,
                       float beta, double* out_host, range_stream_t stream);

/* dst[0,bytes) = src[0,bytes) on the host with the context's copy threads (used by the batch
 * driver to fill fresh result arrays from pinned staging memory). */
int range_host_copy(range_ctx* ctx, void* dst, const void* src, size_t bytes);

/* Introspection for the bench harness: launch geometry of the last scan/attend launch. */
int range_last_attend_geometry(const range_ctx* ctx, int32_t* n_query_tiles, int32_t* n_splits);

/* Per-kernel device timing with HIP events recorded on the launch stream, around the kernel
 * launch only (the reference's only timing hook is a host time.time() pair,
 * range/evaluation/visualize_embeddings.py:101-116).  range_profile_enable(ctx,1) starts
 * collecting (and clears earlier samples); range_profile_read synchronises with the recorded
 * events and returns the summed duration and the number of launches of one kernel. */
enum { RANGE_PROF_ENCODER = 0, RANGE_PROF_SCAN_STATS = 1, RANGE_PROF_ATTEND = 2,
       RANGE_PROF_TOPK_STREAM = 3, RANGE_PROF_TOPK_MERGE = 4, RANGE_PROF_KINDS = 5 };
int range_profile_enable(range_ctx* ctx, int32_t on);
int range_profile_read(range_ctx* ctx, int32_t which, double* total_ms, int32_t* launches);

/* Diagnostic only (kernel tuning): range_attend's launch with an instrumented kernel build that
 * sums, per (workgroup, wave), the shader cycles parked in vmcnt waits / barriers and spent in each
 * phase; 16 x uint64 per wave into diag_dev (capacity in uint64 words).  Results go to the
 * internal slabs only.  No reference counterpart. */
int range_attend_diag(range_ctx* ctx, const float* ehat32_dev, const float* xq32_dev, int64_t B,
                      float tau_sem, float tau_geo, float beta, const float* stats_global_dev,
                      unsigned long long* diag_dev, int64_t diag_capacity, range_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RANGE_HIP_H */

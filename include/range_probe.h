/* range_probe.h - C ABI of the downstream ridge probe in librange_hip.so (MI355X / gfx950).
 *
 * Replaces, for the embeddings this engine produces, the arithmetic of the reference's
 * evaluate_npz (range/utils/evaluate.py:14-47): MinMaxScaler (:38-42), RidgeCV(alphas, cv=3)
 * (:35) / RidgeClassifierCV(alphas, cv=10) (:30), fit (:44) and score (:45) - i.e. the
 * scikit-learn code those lines run (GridSearchCV over alpha with Ridge / RidgeClassifier per
 * fold, Cholesky normal equations, R^2 / accuracy).  Fold assignment, label encoding and the
 * choice of alpha are host logic (range_amd/evaluate.py); everything that touches the
 * (rows x features) data runs in the kernels behind these entry points, in float64.
 *
 * Conventions: as range_hip.h - plain pointers and sizes, *_dev = device memory owned by the
 * caller, *_host = host memory, row-major matrices with explicit leading dimensions, every call
 * enqueues on `stream` and returns RANGE_OK or a negative code (range_last_error() has the text).
 * A context is bound to one device and is not thread-safe.
 */
#ifndef RANGE_PROBE_H
#define RANGE_PROBE_H

#include "range_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct range_probe_ctx range_probe_ctx;

int range_probe_create(int device, range_probe_ctx** out);
void range_probe_destroy(range_probe_ctx* ctx);

/* Column minimum / maximum / sum of X (n x d); any output may be null.
 * MinMaxScaler.fit (evaluate.py:41: data_min_, data_max_) and the column means. */
int range_probe_colstats(range_probe_ctx* ctx, const double* X_dev, int64_t n, int32_t d,
                         int64_t ldx, double* min_dev, double* max_dev, double* sum_dev,
                         range_stream_t stream);

/* Z[i,:] = (X[perm[i],:] * scale + offset) - shift.  MinMaxScaler.transform (evaluate.py:41-42)
 * fused with the row gather that makes folds contiguous and with the centring Ridge applies
 * (sklearn _preprocess_data).  perm / scale+offset / shift may each be null (identity). */
int range_probe_scale_rows(range_probe_ctx* ctx, const double* X_dev, int64_t n, int32_t d,
                           int64_t ldx, const int64_t* perm_dev, const double* scale_dev,
                           const double* offset_dev, const double* shift_dev, double* Z_dev,
                           int64_t ldz, range_stream_t stream);

/* T[i,k] = (code[i] == first + k ? +1 : -1) - shift[k], T (n x c) dense.
 * RidgeClassifier's LabelBinarizer(pos_label=1, neg_label=-1) followed by the centring. */
int range_probe_onehot(range_probe_ctx* ctx, const int32_t* code_dev, int64_t n, int32_t c,
                       int32_t first, const double* shift_dev, double* T_dev,
                       range_stream_t stream);

/* C (M x N, ldc) = alpha * op(A) * op(B) + beta * C on the float64 matrix cores.
 * op(A) is M x K: A is (M x K, lda) when trans_a == 0, (K x M, lda) when trans_a == 1;
 * op(B) is K x N: B is (K x N, ldb) when trans_b == 0, (N x K, ldb) when trans_b == 1.
 * lower_only != 0 (needs M == N): only tiles on or below the diagonal are computed (syrk). */
int range_probe_gemm(range_probe_ctx* ctx, int32_t trans_a, int32_t trans_b, int32_t M, int32_t N,
                     int32_t K, double alpha, const double* A_dev, int64_t lda,
                     const double* B_dev, int64_t ldb, double beta, double* C_dev, int64_t ldc,
                     int32_t lower_only, range_stream_t stream);

/* Sufficient statistics of one block of rows (a fold): G = Z^T Z (d x d, lower triangle valid),
 * B = Z^T T (d x c), zsum (d) and tsum (c) = column sums.  These replace every X^T X / X^T y the
 * per-fold Ridge fits of GridSearchCV form (sklearn _solve_cholesky). */
int range_probe_gram(range_probe_ctx* ctx, const double* Z_dev, int64_t ldz, const double* T_dev,
                     int64_t ldt, int64_t rows, int32_t d, int32_t c, double* G_dev,
                     double* B_dev, double* zsum_dev, double* tsum_dev, range_stream_t stream);

/* out[i] = sum over p < n_parts of parts[p*count + i]  (statistics of all rows = sum over folds) */
int range_probe_sum_parts(range_probe_ctx* ctx, const double* parts_dev, int32_t n_parts,
                          int64_t count, double* out_dev, range_stream_t stream);

/* Ridge fits for groups x alphas systems at once (batched blocked Cholesky + triangular solves).
 * Group g trains on all rows except fold g (fold statistics Gf/Bf/zsumf/tsumf laid out
 * [g][...]), or on all rows when the fold pointers are null (groups must then be 1).
 *   ntr_host    : [groups] training rows of each group
 *   alphas_host : [n_alpha]
 *   W_dev       : [groups][d][n_alpha][c]  coefficients (in the coordinates of Z and T)
 *   c0_dev      : [groups][n_alpha][c]     intercepts   (prediction = z . W + c0)
 * Synchronises the stream once (to read the factorisation status); a non-positive pivot returns
 * RANGE_ERR_INVALID. */
int range_probe_solve(range_probe_ctx* ctx, const double* Gtot_dev, const double* Btot_dev,
                      const double* zsum_tot_dev, const double* tsum_tot_dev,
                      const double* Gf_dev, const double* Bf_dev, const double* zsumf_dev,
                      const double* tsumf_dev, const double* ntr_host, int32_t groups,
                      const double* alphas_host, int32_t n_alpha, int32_t d, int32_t c,
                      double* W_dev, double* c0_dev, range_stream_t stream);

/* Pieces of r2_score for P = Z W (rows x n_alpha*c), intercepts c0 (n_alpha*c), targets T
 * (rows x c), tsum = column sums of T over these rows:
 * out[(a*c+k)*2 + 0] = sum (t - p - c0)^2, [..+1] = sum (t - mean t)^2. */
int range_probe_r2_sums(range_probe_ctx* ctx, const double* P_dev, const double* c0_dev,
                        const double* T_dev, int64_t rows, int32_t c, int32_t n_alpha,
                        const double* tsum_dev, double* out_dev, range_stream_t stream);

/* hits[a] += number of rows whose predicted class equals code[i].  n_cls >= 3: arg-max over the
 * c = n_cls score columns with present[k] != 0; n_cls == 2: c == 1, class 1 iff score > 0
 * (LinearClassifierMixin.predict).  hits must be zeroed by the caller. */
int range_probe_accuracy(range_probe_ctx* ctx, const double* P_dev, const double* c0_dev,
                         const int32_t* code_dev, int64_t rows, int32_t c, int32_t n_alpha,
                         int32_t n_cls, const int32_t* present_dev, uint64_t* hits_dev,
                         range_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif

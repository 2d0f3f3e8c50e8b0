// Host side of the ridge probe (include/range_probe.h): launch geometry, workspace and the
// sequencing of the batched blocked Cholesky.  Part of librange_hip.so.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/range_probe.h"
#include "host_common.h"
#include "probe_kernels.h"

using namespace range_probe;
using namespace range_host;

struct range_probe_ctx {
    int device = 0;
    int n_cu = 256;
    DevBuf<double> ws_slabs, ws_part, ws_A, d_ntr, d_alphas;
    DevBuf<int32_t> d_info;
};

namespace {

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// One GEMM (batch == 1, split-K slabs when the tile count alone cannot fill the chip) or a batch
// of equally shaped GEMMs (blockIdx.z).
int launch_gemm(range_probe_ctx* c, GemmArgs g, bool a_kc, bool b_kc, int batch, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || batch <= 0) return RANGE_OK;
    const int tiles_m = cdiv(g.M, GEMM_TILE), tiles_n = cdiv(g.N, GEMM_TILE);
    const int tiles = g.lower_only ? tiles_m * (tiles_m + 1) / 2 : tiles_m * tiles_n;
    g.tiles_n = tiles_n;
    if (g.batch_inner <= 0) g.batch_inner = 1;
    int splits = 1;
    if (batch == 1 && g.K >= 2 * 512) {
        // two full rounds of the chip's 2 * n_cu workgroup slots, never a nearly empty third one
        const int slots = 2 * c->n_cu;
        splits = std::min(g.K / 512, std::max(1, 2 * slots / tiles));
    }
    g.k_chunk = cdiv(cdiv(g.K, splits), GEMM_KT) * GEMM_KT;
    splits = std::max(1, cdiv(g.K, g.k_chunk));
    double* dst = g.C;
    const int64_t ldd = g.ldc;
    const double beta = g.beta;
    if (splits > 1) {
        HIP_TRY(c->ws_slabs.ensure((size_t)splits * g.M * g.N));
        g.C = c->ws_slabs.p;
        g.ldc = g.N;
        g.c_ss = (int64_t)g.M * g.N;
    }
    const dim3 grid((unsigned)tiles, (unsigned)splits, (unsigned)batch);
    if (a_kc && b_kc) hipLaunchKernelGGL((dgemm_kernel<true, true>), grid, dim3(GEMM_THREADS), 0, s, g);
    else if (a_kc) hipLaunchKernelGGL((dgemm_kernel<true, false>), grid, dim3(GEMM_THREADS), 0, s, g);
    else if (b_kc) hipLaunchKernelGGL((dgemm_kernel<false, true>), grid, dim3(GEMM_THREADS), 0, s, g);
    else hipLaunchKernelGGL((dgemm_kernel<false, false>), grid, dim3(GEMM_THREADS), 0, s, g);
    HIP_TRY(hipGetLastError());
    if (splits > 1) {
        const int64_t cnt = (int64_t)g.M * g.N;
        hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv(cnt, 256)), dim3(256), 0, s, dst,
                           ldd, c->ws_slabs.p, (int64_t)g.N, cnt, splits, g.M, g.N, beta,
                           g.lower_only);
        HIP_TRY(hipGetLastError());
    }
    return RANGE_OK;
}

GemmArgs gemm_args(int M, int N, int K, double alpha, const double* A, int64_t a_rs, int64_t a_cs,
                   const double* B, int64_t b_rs, int64_t b_cs, double beta, double* C,
                   int64_t ldc) {
    GemmArgs g{};
    g.A = A; g.B = B; g.C = C;
    g.M = M; g.N = N; g.K = K;
    g.a_rs = a_rs; g.a_cs = a_cs; g.b_rs = b_rs; g.b_cs = b_cs; g.ldc = ldc;
    g.batch_inner = 1;
    g.alpha = alpha; g.beta = beta;
    return g;
}

int colstats(range_probe_ctx* c, const double* X, int64_t n, int32_t d, int64_t ldx, double* mn,
             double* mx, double* sm, hipStream_t s) {
    const int col_groups = cdiv(d, 64);
    const int splits = (int)std::max<int64_t>(1, std::min<int64_t>(n / 64, cdiv(4 * c->n_cu, col_groups)));
    HIP_TRY(c->ws_part.ensure((size_t)splits * 3 * d));
    hipLaunchKernelGGL(colstats_kernel, dim3(col_groups, splits), dim3(256), 0, s, X, n, d, ldx,
                       c->ws_part.p);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(colstats_fold_kernel, dim3(cdiv(d, 256)), dim3(256), 0, s, c->ws_part.p,
                       splits, d, mn, mx, sm);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

}  // namespace

extern "C" {

int range_probe_create(int device, range_probe_ctx** out) {
    if (!out) return fail(RANGE_ERR_INVALID, "null argument");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RANGE_ERR_INVALID, "device %d of %d", device, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RANGE_ERR_INVALID, "device %d is %s; this library is built for gfx950 only",
                    device, prop.gcnArchName);
    auto* c = new range_probe_ctx();
    c->device = device;
    c->n_cu = prop.multiProcessorCount;
    *out = c;
    return RANGE_OK;
}

void range_probe_destroy(range_probe_ctx* c) {
    if (!c) return;
    DeviceGuard g(c->device);
    delete c;
}

int range_probe_colstats(range_probe_ctx* c, const double* X, int64_t n, int32_t d, int64_t ldx,
                         double* mn, double* mx, double* sm, range_stream_t stream) {
    if (!c || !X) return fail(RANGE_ERR_INVALID, "null argument");
    if (n <= 0 || d <= 0 || ldx < d) return fail(RANGE_ERR_INVALID, "bad shape n=%lld d=%d ld=%lld",
                                                 (long long)n, d, (long long)ldx);
    DeviceGuard g(c->device);
    return colstats(c, X, n, d, ldx, mn, mx, sm, (hipStream_t)stream);
}

int range_probe_scale_rows(range_probe_ctx* c, const double* X, int64_t n, int32_t d, int64_t ldx,
                           const int64_t* perm, const double* scale, const double* offset,
                           const double* shift, double* Z, int64_t ldz, range_stream_t stream) {
    if (!c || !X || !Z) return fail(RANGE_ERR_INVALID, "null argument");
    if (n <= 0 || d <= 0 || ldx < d || ldz < d) return fail(RANGE_ERR_INVALID, "bad shape");
    if ((scale == nullptr) != (offset == nullptr))
        return fail(RANGE_ERR_INVALID, "scale and offset come together");
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)cdiv(n * d, 256)), dim3(256), 0,
                       (hipStream_t)stream, X, n, d, ldx, perm, scale, offset, shift, Z, ldz);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_probe_onehot(range_probe_ctx* c, const int32_t* code, int64_t n, int32_t cc,
                       int32_t first, const double* shift, double* T, range_stream_t stream) {
    if (!c || !code || !shift || !T) return fail(RANGE_ERR_INVALID, "null argument");
    if (n <= 0 || cc <= 0) return fail(RANGE_ERR_INVALID, "bad shape");
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(onehot_rows_kernel, dim3((unsigned)cdiv(n * cc, 256)), dim3(256), 0,
                       (hipStream_t)stream, code, n, cc, first, shift, T);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_probe_gemm(range_probe_ctx* c, int32_t trans_a, int32_t trans_b, int32_t M, int32_t N,
                     int32_t K, double alpha, const double* A, int64_t lda, const double* B,
                     int64_t ldb, double beta, double* C, int64_t ldc, int32_t lower_only,
                     range_stream_t stream) {
    if (!c || !A || !B || !C) return fail(RANGE_ERR_INVALID, "null argument");
    if (M <= 0 || N <= 0 || K <= 0 || ldc < N) return fail(RANGE_ERR_INVALID, "bad GEMM shape");
    if (lower_only && M != N) return fail(RANGE_ERR_INVALID, "lower_only needs M == N");
    if (lda < (trans_a ? M : K) || ldb < (trans_b ? K : N))
        return fail(RANGE_ERR_INVALID, "leading dimension too small");
    DeviceGuard g(c->device);
    GemmArgs a = gemm_args(M, N, K, alpha, A, trans_a ? 1 : lda, trans_a ? lda : 1, B,
                           trans_b ? 1 : ldb, trans_b ? ldb : 1, beta, C, ldc);
    a.lower_only = lower_only ? 1 : 0;
    return launch_gemm(c, a, !trans_a, trans_b != 0, 1, (hipStream_t)stream);
}

int range_probe_gram(range_probe_ctx* c, const double* Z, int64_t ldz, const double* T, int64_t ldt,
                     int64_t rows, int32_t d, int32_t cc, double* G, double* B, double* zsum,
                     double* tsum, range_stream_t stream) {
    if (!c || !Z || !T || !G || !B || !zsum || !tsum) return fail(RANGE_ERR_INVALID, "null argument");
    if (rows <= 0 || rows > INT32_MAX || d <= 0 || cc <= 0 || ldz < d || ldt < cc)
        return fail(RANGE_ERR_INVALID, "bad shape");
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)stream;
    // G = Z^T Z: op(A) = Z^T (element (i,k) = Z[k][i]), op(B) = Z
    GemmArgs a = gemm_args(d, d, (int)rows, 1.0, Z, 1, ldz, Z, ldz, 1, 0.0, G, d);
    a.lower_only = 1;
    int rc = launch_gemm(c, a, false, false, 1, s);
    if (rc) return rc;
    GemmArgs b = gemm_args(d, cc, (int)rows, 1.0, Z, 1, ldz, T, ldt, 1, 0.0, B, cc);
    rc = launch_gemm(c, b, false, false, 1, s);
    if (rc) return rc;
    rc = colstats(c, Z, rows, d, ldz, nullptr, nullptr, zsum, s);
    if (rc) return rc;
    return colstats(c, T, rows, cc, ldt, nullptr, nullptr, tsum, s);
}

int range_probe_sum_parts(range_probe_ctx* c, const double* parts, int32_t n_parts, int64_t count,
                          double* out, range_stream_t stream) {
    if (!c || !parts || !out) return fail(RANGE_ERR_INVALID, "null argument");
    if (n_parts <= 0 || count <= 0 || count > INT32_MAX) return fail(RANGE_ERR_INVALID, "bad shape");
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)cdiv(count, 256)), dim3(256), 0,
                       (hipStream_t)stream, out, count, parts, count, count, n_parts, 1,
                       (int32_t)count, 0.0, 0);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_probe_solve(range_probe_ctx* c, const double* Gtot, const double* Btot,
                      const double* zsum_tot, const double* tsum_tot, const double* Gf,
                      const double* Bf, const double* zsumf, const double* tsumf,
                      const double* ntr_host, int32_t groups, const double* alphas_host,
                      int32_t n_alpha, int32_t d, int32_t cc, double* W, double* c0,
                      range_stream_t stream) {
    if (!c || !Gtot || !Btot || !zsum_tot || !tsum_tot || !ntr_host || !alphas_host || !W || !c0)
        return fail(RANGE_ERR_INVALID, "null argument");
    const bool folds = Gf != nullptr;
    if (folds != (Bf != nullptr) || folds != (zsumf != nullptr) || folds != (tsumf != nullptr))
        return fail(RANGE_ERR_INVALID, "fold statistics come together");
    if (groups <= 0 || n_alpha <= 0 || d <= 0 || cc <= 0 || (!folds && groups != 1))
        return fail(RANGE_ERR_INVALID, "bad shape");
    for (int a = 0; a < n_alpha; ++a)
        if (!(alphas_host[a] >= 0.0)) return fail(RANGE_ERR_INVALID, "alpha must be >= 0");
    DeviceGuard g(c->device);
    hipStream_t s = (hipStream_t)stream;
    const int Q = groups * n_alpha;
    const int64_t dd = (int64_t)d * d;
    HIP_TRY(c->ws_A.ensure((size_t)Q * dd));
    HIP_TRY(c->d_ntr.ensure(groups));
    HIP_TRY(c->d_alphas.ensure(n_alpha));
    HIP_TRY(c->d_info.ensure(Q));
    HIP_TRY(hipMemcpyAsync(c->d_ntr.p, ntr_host, sizeof(double) * groups, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->d_alphas.p, alphas_host, sizeof(double) * n_alpha,
                           hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(c->d_info.p, 0, sizeof(int32_t) * Q, s));

    AssembleArgs p{};
    p.Gtot = Gtot; p.Btot = Btot; p.stot = zsum_tot; p.ttot = tsum_tot;
    p.Gf = Gf; p.Bf = Bf; p.sf = zsumf; p.tf = tsumf;
    p.ntr = c->d_ntr.p; p.alphas = c->d_alphas.p;
    p.d = d; p.c = cc; p.n_alpha = n_alpha;
    p.Aout = c->ws_A.p; p.Rout = W;
    const int64_t most = std::max(dd, (int64_t)d * cc);
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)cdiv(most, 256), (unsigned)groups),
                       dim3(256), 0, s, p);
    HIP_TRY(hipGetLastError());

    double* A = c->ws_A.p;
    const int64_t ldr = (int64_t)n_alpha * cc;          // row stride of W
    const int64_t r_grp = (int64_t)d * ldr, r_alpha = cc;
    // ---- factor: right-looking, 64-wide panels, all systems per launch -------------------
    for (int p0 = 0; p0 < d; p0 += PANEL) {
        const int nb = std::min(PANEL, d - p0);
        hipLaunchKernelGGL(potrf_diag_kernel, dim3(Q), dim3(64), 0, s, A, (int64_t)d, dd, p0, nb,
                           c->d_info.p);
        HIP_TRY(hipGetLastError());
        const int below = d - p0 - nb;
        if (below <= 0) break;
        hipLaunchKernelGGL(trsm_rows_kernel, dim3((unsigned)cdiv(below, 64), (unsigned)Q), dim3(64),
                           0, s, A, (int64_t)d, dd, p0, nb, d);
        HIP_TRY(hipGetLastError());
        // A22 -= L21 L21^T (lower tiles)
        const double* l21 = A + (int64_t)(p0 + nb) * d + p0;
        GemmArgs u = gemm_args(below, below, nb, -1.0, l21, d, 1, l21, 1, d, 1.0,
                               A + (int64_t)(p0 + nb) * d + (p0 + nb), d);
        u.lower_only = 1;
        u.a_bi = u.b_bi = u.c_bi = dd;
        u.batch_inner = Q;
        int rc = launch_gemm(c, u, true, true, Q, s);
        if (rc) return rc;
    }
    // ---- forward substitution  L Y = R ------------------------------------------------------
    for (int p0 = 0; p0 < d; p0 += PANEL) {
        const int nb = std::min(PANEL, d - p0);
        hipLaunchKernelGGL((trsv_cols_kernel<false>), dim3((unsigned)cdiv(cc, 64), (unsigned)Q),
                           dim3(64), 0, s, A, (int64_t)d, dd, p0, nb, W, ldr, n_alpha, r_grp,
                           r_alpha, cc);
        HIP_TRY(hipGetLastError());
        const int below = d - p0 - nb;
        if (below <= 0) break;
        // R[below] -= L[below, panel] * Y[panel]
        GemmArgs u = gemm_args(below, cc, nb, -1.0, A + (int64_t)(p0 + nb) * d + p0, d, 1,
                               W + (int64_t)p0 * ldr, ldr, 1, 1.0, W + (int64_t)(p0 + nb) * ldr, ldr);
        u.batch_inner = n_alpha;
        u.a_bo = (int64_t)n_alpha * dd; u.a_bi = dd;
        u.b_bo = r_grp; u.b_bi = r_alpha;
        u.c_bo = r_grp; u.c_bi = r_alpha;
        int rc = launch_gemm(c, u, true, false, Q, s);
        if (rc) return rc;
    }
    // ---- backward substitution  L^T W = Y ---------------------------------------------------
    const int last = ((d - 1) / PANEL) * PANEL;
    for (int p0 = last; p0 >= 0; p0 -= PANEL) {
        const int nb = std::min(PANEL, d - p0);
        hipLaunchKernelGGL((trsv_cols_kernel<true>), dim3((unsigned)cdiv(cc, 64), (unsigned)Q),
                           dim3(64), 0, s, A, (int64_t)d, dd, p0, nb, W, ldr, n_alpha, r_grp,
                           r_alpha, cc);
        HIP_TRY(hipGetLastError());
        if (p0 == 0) break;
        // R[0:p0] -= L[panel, 0:p0]^T * W[panel]
        GemmArgs u = gemm_args(p0, cc, nb, -1.0, A + (int64_t)p0 * d, 1, d, W + (int64_t)p0 * ldr,
                               ldr, 1, 1.0, W, ldr);
        u.batch_inner = n_alpha;
        u.a_bo = (int64_t)n_alpha * dd; u.a_bi = dd;
        u.b_bo = r_grp; u.b_bi = r_alpha;
        u.c_bo = r_grp; u.c_bi = r_alpha;
        int rc = launch_gemm(c, u, false, false, Q, s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(intercept_kernel, dim3((unsigned)cdiv((int64_t)n_alpha * cc, 64), (unsigned)groups),
                       dim3(64 * ICPT_SLICES), 0, s, p, W, c0);
    HIP_TRY(hipGetLastError());

    std::vector<int32_t> info(Q);
    HIP_TRY(hipMemcpyAsync(info.data(), c->d_info.p, sizeof(int32_t) * Q, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int q = 0; q < Q; ++q)
        if (info[q] != 0)
            return fail(RANGE_ERR_INVALID,
                        "ridge system (group %d, alpha %g) is not positive definite at pivot %d",
                        q / n_alpha, alphas_host[q % n_alpha], info[q]);
    return RANGE_OK;
}

int range_probe_r2_sums(range_probe_ctx* c, const double* P, const double* c0, const double* T,
                        int64_t rows, int32_t cc, int32_t n_alpha, const double* tsum, double* out,
                        range_stream_t stream) {
    if (!c || !P || !c0 || !T || !tsum || !out) return fail(RANGE_ERR_INVALID, "null argument");
    if (rows <= 0 || cc <= 0 || n_alpha <= 0) return fail(RANGE_ERR_INVALID, "bad shape");
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(r2_sums_kernel, dim3((unsigned)(n_alpha * cc)), dim3(256), 0,
                       (hipStream_t)stream, P, c0, T, rows, cc, n_alpha, tsum, out);
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

int range_probe_accuracy(range_probe_ctx* c, const double* P, const double* c0, const int32_t* code,
                         int64_t rows, int32_t cc, int32_t n_alpha, int32_t n_cls,
                         const int32_t* present, uint64_t* hits, range_stream_t stream) {
    if (!c || !P || !c0 || !code || !hits) return fail(RANGE_ERR_INVALID, "null argument");
    if (rows <= 0 || cc <= 0 || n_alpha <= 0 || n_cls < 2) return fail(RANGE_ERR_INVALID, "bad shape");
    if (n_cls == 2 ? cc != 1 : (cc != n_cls || !present))
        return fail(RANGE_ERR_INVALID, "score columns %d do not match %d classes", cc, n_cls);
    DeviceGuard g(c->device);
    hipLaunchKernelGGL(accuracy_kernel, dim3((unsigned)cdiv(rows, 256), (unsigned)n_alpha),
                       dim3(256), 0, (hipStream_t)stream, P, c0, code, rows, cc, n_alpha, n_cls,
                       present, reinterpret_cast<unsigned long long*>(hits));
    HIP_TRY(hipGetLastError());
    return RANGE_OK;
}

}  // extern "C"

// Device side of the downstream ridge probe (reference: range/utils/evaluate.py:14-47, i.e.
// MinMaxScaler + RidgeCV / RidgeClassifierCV of scikit-learn run with an integer cv).
//
// All arithmetic is float64 like scikit-learn's.  The pieces:
//   * column statistics, scaling + centring + row gather (HBM-bound, one pass each);
//   * dgemm_kernel: one LDS-tiled v_mfma_f64_16x16x4_f64 GEMM, C = alpha*op(A)*op(B) + beta*C with
//     arbitrary element strides, two-level batching, split-K slabs and a lower-triangle mode.  It
//     computes the per-fold Gram matrices Z^T Z and Z^T T (the dominant FLOPs, MFMA-bound), the
//     trailing updates of the blocked Cholesky factorisation, the block updates of the
//     triangular solves and the held-out predictions Z W;
//   * 64-wide panel kernels of a right-looking blocked Cholesky (diagonal block in LDS, one
//     thread per row for the panel solve), batched over all (fold, alpha) systems at once;
//   * scoring reductions (R^2 sums, arg-max accuracy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace range_probe {

typedef double pd4 __attribute__((ext_vector_type(4)));

constexpr int GEMM_TILE = 128;   // C tile per workgroup (8 waves, 32x64 each)
constexpr int GEMM_KT = 16;      // K extent of one LDS stage (4 MFMA k-steps)
constexpr int GEMM_LD = 130;     // LDS row stride in doubles (bank spread for both access patterns)
constexpr int PANEL = 64;        // Cholesky panel width

// ------------------------------------------------------------------------------------------
// column statistics of X (n x d, leading dimension ld): min, max and sum per column.
// grid.x covers the columns in groups of 64, grid.y splits the rows; partial results go to
// part[(split*3 + {0,1,2})*d + col] and are folded by colstats_fold_kernel.
__global__ __launch_bounds__(256) void colstats_kernel(const double* __restrict__ X, int64_t n,
                                                       int32_t d, int64_t ld,
                                                       double* __restrict__ part) {
    __shared__ double smin[4][64], smax[4][64], ssum[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int64_t rows_per = (n + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    const int64_t r1 = r0 + rows_per < n ? r0 + rows_per : n;
    double mn = INFINITY, mx = -INFINITY, sm = 0.0;
    if (col < d)
        for (int64_t r = r0 + w; r < r1; r += 4) {
            const double v = X[r * ld + col];
            mn = fmin(mn, v);
            mx = fmax(mx, v);
            sm += v;
        }
    smin[w][lane] = mn; smax[w][lane] = mx; ssum[w][lane] = sm;
    __syncthreads();
    if (w == 0 && col < d) {
        for (int i = 1; i < 4; ++i) {
            mn = fmin(mn, smin[i][lane]);
            mx = fmax(mx, smax[i][lane]);
            sm += ssum[i][lane];
        }
        double* o = part + (int64_t)blockIdx.y * 3 * d;
        o[col] = mn; o[d + col] = mx; o[2 * d + col] = sm;
    }
}

__global__ void colstats_fold_kernel(const double* __restrict__ part, int32_t n_splits, int32_t d,
                                     double* __restrict__ mn, double* __restrict__ mx,
                                     double* __restrict__ sm) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= d) return;
    double a = INFINITY, b = -INFINITY, s = 0.0;
    for (int i = 0; i < n_splits; ++i) {
        const double* p = part + (int64_t)i * 3 * d;
        a = fmin(a, p[col]);
        b = fmax(b, p[d + col]);
        s += p[2 * d + col];
    }
    if (mn) mn[col] = a;
    if (mx) mx[col] = b;
    if (sm) sm[col] = s;
}

// Z[i, :] = (X[perm ? perm[i] : i, :] * scale + offset) - shift.  The scaled value is formed with
// the two roundings of MinMaxScaler.transform (X *= scale; X += min_), no contraction.
__global__ __launch_bounds__(256) void scale_rows_kernel(const double* __restrict__ X, int64_t n,
                                                         int32_t d, int64_t ldx,
                                                         const int64_t* __restrict__ perm,
                                                         const double* __restrict__ scale,
                                                         const double* __restrict__ offset,
                                                         const double* __restrict__ shift,
                                                         double* __restrict__ Z, int64_t ldz) {
#pragma clang fp contract(off)   // hipcc fuses a*b+c by default, also through __dmul_rn/__dadd_rn
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * d) return;
    const int64_t i = e / d;
    const int col = (int)(e - i * d);
    const int64_t src = perm ? perm[i] : i;
    double v = X[src * ldx + col];
    if (scale) v = v * scale[col] + offset[col];
    if (shift) v -= shift[col];
    Z[i * ldz + col] = v;
}

// T[i, k] = (code[i] == first + k ? 1 : -1) - shift[k]   (LabelBinarizer(-1/+1), then centring)
__global__ __launch_bounds__(256) void onehot_rows_kernel(const int32_t* __restrict__ code, int64_t n,
                                                          int32_t c, int32_t first,
                                                          const double* __restrict__ shift,
                                                          double* __restrict__ T) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * c) return;
    const int64_t i = e / c;
    const int k = (int)(e - i * c);
    T[e] = (code[i] == first + k ? 1.0 : -1.0) - shift[k];
}

// ------------------------------------------------------------------------------------------
struct GemmArgs {
    const double* A;
    const double* B;
    double* C;
    int32_t M, N, K;
    int64_t a_rs, a_cs;     // A element (i,k) at A + i*a_rs + k*a_cs
    int64_t b_rs, b_cs;     // B element (k,j) at B + k*b_rs + j*b_cs
    int64_t ldc;            // C element (i,j) at C + i*ldc + j
    int32_t batch_inner;    // batch z -> (z / batch_inner, z % batch_inner)
    int64_t a_bo, a_bi, b_bo, b_bi, c_bo, c_bi;   // outer / inner batch strides (elements)
    double alpha, beta;
    int32_t lower_only;     // 1: only tiles on or below the diagonal (M == N)
    int32_t k_chunk;        // K range per blockIdx.y (multiple of GEMM_KT); slabs when gridDim.y > 1
    int64_t c_ss;           // slab stride (elements) when gridDim.y > 1: C + y*c_ss, alpha*acc only
    int32_t tiles_n;        // tile columns (ignored when lower_only)
};

// A_KC / B_KC: the K index is the contiguous one of that operand (selects the coalesced
// global -> LDS thread mapping; the strides above stay authoritative for addressing).
// 8 waves per workgroup, each owning a 32 x 64 part of the tile (8 accumulators, <= 128 VGPRs):
// two workgroups per CU = 4 waves per SIMD, which is what it takes to keep the float64 MFMA pipe
// busy (tools/micro/mfma_f64_peak.hip: 36 TFLOP/s with 1-2 waves per SIMD, 47-49 with >= 3).
constexpr int GEMM_THREADS = 512;
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(GEMM_THREADS, 4) void dgemm_kernel(GemmArgs g) {
    __shared__ double As[GEMM_KT * GEMM_LD];
    __shared__ double Bs[GEMM_KT * GEMM_LD];

    int tm, tn;
    if (g.lower_only) {
        // blockIdx.x enumerates (tm, tn <= tm) row by row
        int t = blockIdx.x;
        tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
        while (tm * (tm + 1) / 2 > t) --tm;
        tn = t - tm * (tm + 1) / 2;
    } else {
        tm = blockIdx.x / g.tiles_n;
        tn = blockIdx.x - tm * g.tiles_n;
    }
    const int m0 = tm * GEMM_TILE, n0 = tn * GEMM_TILE;
    const int zo = blockIdx.z / g.batch_inner, zi = blockIdx.z - zo * g.batch_inner;
    const double* __restrict__ A = g.A + zo * g.a_bo + zi * g.a_bi;
    const double* __restrict__ B = g.B + zo * g.b_bo + zi * g.b_bi;
    double* __restrict__ C = g.C + zo * g.c_bo + zi * g.c_bi + (int64_t)blockIdx.y * g.c_ss;
    const int kbeg = blockIdx.y * g.k_chunk;
    const int kend = kbeg + g.k_chunk < g.K ? kbeg + g.k_chunk : g.K;

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 64;
    const int r = lane & 15, q = lane >> 4;

    // global -> register staging: 4 doubles of A and 4 of B per thread and stage
    int a_m[4], a_k[4], b_n[4], b_k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (A_KC) { a_k[j] = t & 15; a_m[j] = (t >> 4) + 32 * j; }
        else      { a_m[j] = t & 127; a_k[j] = (t >> 7) + 4 * j; }
        if (B_KC) { b_k[j] = t & 15; b_n[j] = (t >> 4) + 32 * j; }
        else      { b_n[j] = t & 127; b_k[j] = (t >> 7) + 4 * j; }
    }
    double ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + a_m[j], ka = k0 + a_k[j];
            ra[j] = (m < g.M && ka < kend) ? A[(int64_t)m * g.a_rs + (int64_t)ka * g.a_cs] : 0.0;
            const int n = n0 + b_n[j], kb = k0 + b_k[j];
            rb[j] = (n < g.N && kb < kend) ? B[(int64_t)kb * g.b_rs + (int64_t)n * g.b_cs] : 0.0;
        }
    };

    pd4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = pd4{0.0, 0.0, 0.0, 0.0};

    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GEMM_KT) {
        __syncthreads();                       // the previous stage's operand reads are done
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[a_k[j] * GEMM_LD + a_m[j]] = ra[j];
            Bs[b_k[j] * GEMM_LD + b_n[j]] = rb[j];
        }
        __syncthreads();
        if (k0 + GEMM_KT < kend) fetch(k0 + GEMM_KT);   // overlaps the MFMAs below
#pragma unroll
        for (int ks = 0; ks < GEMM_KT / 4; ++ks) {
            const int kk = (ks * 4 + q) * GEMM_LD;
            double a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[kk + wm + i * 16 + r];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk + wn + j * 16 + r];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    const bool slab = gridDim.y > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // f64 16x16x4 accumulator layout: row = lane/16 + 4*reg, column = lane%16
                const int row = m0 + wm + i * 16 + q + 4 * e;
                const int col = n0 + wn + j * 16 + r;
                if (row < g.M && col < g.N) {
                    double* p = C + (int64_t)row * g.ldc + col;
                    const double v = g.alpha * acc[i][j][e];
                    *p = (slab || g.beta == 0.0) ? v : v + g.beta * *p;
                }
            }
}

// dst = beta*dst + sum over slabs (split-K epilogue), row-major M x N views
__global__ __launch_bounds__(256) void sum_slabs_kernel(double* __restrict__ dst, int64_t ldd,
                                                        const double* __restrict__ slabs,
                                                        int64_t lds_, int64_t slab_stride,
                                                        int32_t n_slabs, int32_t M, int32_t N,
                                                        double beta, int32_t lower_only) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)M * N) return;
    const int64_t i = e / N;
    const int j = (int)(e - i * N);
    if (lower_only && j / GEMM_TILE > i / GEMM_TILE) return;   // tiles the GEMM never wrote
    double s = 0.0;
    for (int k = 0; k < n_slabs; ++k) s += slabs[k * slab_stride + i * lds_ + j];
    double* p = dst + i * ldd + j;
    *p = beta == 0.0 ? s : s + beta * *p;
}

// ------------------------------------------------------------------------------------------
// Ridge systems.  System (g, a): train rows = all rows minus fold g (or all rows when Gf is
// null), alpha = alphas[a].  With Z centred by the global mean,
//   delta = (s_tot - s_g) / n_tr            (training mean of the features, relative to Z's origin)
//   tau   = (t_tot - t_g) / n_tr            (training mean of the targets)
//   A = (G_tot - G_g) - n_tr delta delta^T + alpha I        (lower triangle)
//   R = (B_tot - B_g) - n_tr delta tau^T
struct AssembleArgs {
    const double* Gtot; const double* Btot; const double* stot; const double* ttot;
    const double* Gf; const double* Bf; const double* sf; const double* tf;   // per fold, or null
    const double* ntr;        // [groups] training rows of each group
    const double* alphas;     // [n_alpha]
    int32_t d, c, n_alpha;
    double* Aout;             // [group][alpha][d][d]
    double* Rout;             // [group][d][alpha][c]
};

__global__ __launch_bounds__(256) void assemble_kernel(AssembleArgs p) {
    const int grp = blockIdx.y;
    const int64_t dd = (int64_t)p.d * p.d;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const double ntr = p.ntr[grp];
    if (e < dd) {
        const int i = (int)(e / p.d), j = (int)(e - (int64_t)i * p.d);
        if (j <= i) {
            double gij = p.Gtot[e], si = p.stot[i], sj = p.stot[j];
            if (p.Gf) {
                gij -= p.Gf[grp * dd + e];
                si -= p.sf[(int64_t)grp * p.d + i];
                sj -= p.sf[(int64_t)grp * p.d + j];
            }
            const double v = gij - (si / ntr) * sj;
            for (int a = 0; a < p.n_alpha; ++a)
                p.Aout[((int64_t)grp * p.n_alpha + a) * dd + e] = i == j ? v + p.alphas[a] : v;
        }
    }
    const int64_t dc = (int64_t)p.d * p.c;
    if (e < dc) {
        const int i = (int)(e / p.c), k = (int)(e - (int64_t)i * p.c);
        double b = p.Btot[e], si = p.stot[i], tk = p.ttot[k];
        if (p.Bf) {
            b -= p.Bf[grp * dc + e];
            si -= p.sf[(int64_t)grp * p.d + i];
            tk -= p.tf[(int64_t)grp * p.c + k];
        }
        const double v = b - (si / ntr) * tk;
        for (int a = 0; a < p.n_alpha; ++a)
            p.Rout[(((int64_t)grp * p.d + i) * p.n_alpha + a) * p.c + k] = v;
    }
}

// ---- 64-wide panel kernels ------------------------------------------------------------------
// All three keep one 64-vector per thread in registers (static indices, fully unrolled) and sweep
// it once: element j is finalised, then the remaining elements receive its rank-1 update - 2016
// independent FMAs per thread and no dependent chain longer than the 64 pivots.  The triangular
// factor comes from LDS as broadcast reads.

// Load the factored diagonal block L11 (nb x nb, lower) into LDS, padded to 64 x 64 with the
// identity so that a short last panel needs no special case.
__device__ __forceinline__ void load_l11(double (*L)[PANEL + 2], const double* __restrict__ l11,
                                         int64_t ld, int nb, int t, int nthreads) {
    for (int e = t; e < PANEL * PANEL; e += nthreads) {
        const int i = e >> 6, j = e & 63;
        L[i][j] = (i < nb && j <= i) ? l11[(int64_t)i * ld + j] : (i == j ? 1.0 : 0.0);
    }
}

// Triangular solve of one 64-vector per thread.  The vector lives in a lane-private LDS column
// X[k*XS + t] (XS = 65: conflict-free both for the thread's own accesses and for the transposed
// staging of row tiles); it is swept in 8 blocks of 8: the block's 8 unknowns are finalised in
// registers, then every remaining element receives their rank-8 update (8 independent-of-x FMAs
// per element, L11 entries as LDS broadcast reads).  Real loops over the blocks keep live ranges
// short - the fully unrolled register formulation makes hipcc hoist or sink 2016 operands and
// spill them.
// FWD:  x <- L^-1 x  (also x <- x L^-T for a row vector);  BWD:  x <- L^-T x.
constexpr int XS = PANEL + 1;
template <bool BWD>
__device__ __forceinline__ void tri_solve64(double* __restrict__ X, int t,
                                            const double (*L)[PANEL + 2]) {
    if (!BWD) {
        for (int jb = 0; jb < PANEL; jb += 8) {
            double xb[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) xb[a] = X[(jb + a) * XS + t];
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                xb[a] = xb[a] / L[jb + a][jb + a];
#pragma unroll
                for (int b = a + 1; b < 8; ++b) xb[b] -= L[jb + b][jb + a] * xb[a];
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) X[(jb + a) * XS + t] = xb[a];
#pragma unroll 4
            for (int m = jb + 8; m < PANEL; ++m) {
                double v = X[m * XS + t];
#pragma unroll
                for (int a = 0; a < 8; ++a) v -= L[m][jb + a] * xb[a];
                X[m * XS + t] = v;
            }
        }
    } else {
        for (int jb = PANEL - 8; jb >= 0; jb -= 8) {
            double xb[8];
#pragma unroll
            for (int a = 0; a < 8; ++a) xb[a] = X[(jb + a) * XS + t];
#pragma unroll
            for (int a = 7; a >= 0; --a) {
                xb[a] = xb[a] / L[jb + a][jb + a];
#pragma unroll
                for (int b = 0; b < a; ++b) xb[b] -= L[jb + a][jb + b] * xb[a];
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) X[(jb + a) * XS + t] = xb[a];
#pragma unroll 4
            for (int m = 0; m < jb; ++m) {
                double v = X[m * XS + t];
#pragma unroll
                for (int a = 0; a < 8; ++a) v -= L[jb + a][m] * xb[a];
                X[m * XS + t] = v;
            }
        }
    }
}

// Diagonal block of the blocked Cholesky: factor A[p0:p0+nb, p0:p0+nb] (lower) in place.
// One wave per system (blockIdx.x), lane t owns row t; the pivot column travels through LDS.
// info[system] = first non-positive pivot (1-based) or 0.
__global__ __launch_bounds__(64) void potrf_diag_kernel(double* __restrict__ Aall, int64_t ld,
                                                        int64_t sys_stride, int32_t p0, int32_t nb,
                                                        int32_t* __restrict__ info) {
    __shared__ double col[PANEL];
    double* a = Aall + (int64_t)blockIdx.x * sys_stride + (int64_t)p0 * ld + p0;
    const int t = threadIdx.x;
    const int tr = t < nb ? t : nb - 1;               // clamped row for the loads
    double x[PANEL];
#pragma unroll
    for (int j = 0; j < PANEL; ++j) {
        const double v = a[(int64_t)tr * ld + (j < nb ? j : nb - 1)];
        x[j] = (t < nb && j <= t) ? v : (j == t ? 1.0 : 0.0);
    }
    int bad = 0;
#pragma unroll
    for (int j = 0; j < PANEL; ++j) {
        double djj = __shfl(x[j], j);                 // lane j holds the diagonal element
        if (!(djj > 0.0)) { if (bad == 0) bad = p0 + j + 1; djj = 1.0; }
        const double piv = sqrt(djj);
        x[j] = t == j ? piv : x[j] / piv;
        __syncthreads();                              // (one wave) previous column fully consumed
        col[t] = x[j];
        __syncthreads();
#pragma unroll
        for (int m = j + 1; m < PANEL; ++m) x[m] -= x[j] * col[m];
        __builtin_amdgcn_sched_barrier(0);
    }
    if (t == 0 && bad && info[blockIdx.x] == 0) info[blockIdx.x] = bad;
#pragma unroll
    for (int j = 0; j < PANEL; ++j)
        if (t < nb && j <= t) a[(int64_t)t * ld + j] = x[j];
}

// Panel below the diagonal block: rows i >= p0+nb, A[i, p0:p0+nb] <- A[i, p0:p0+nb] * L11^-T.
// One wave per 64 rows (blockIdx.x), one thread per row; the 64 x 64 tile is read and written
// coalesced (lane = column) and transposed through LDS; blockIdx.y = system.
__global__ __launch_bounds__(64) void trsm_rows_kernel(double* __restrict__ Aall, int64_t ld,
                                                       int64_t sys_stride, int32_t p0, int32_t nb,
                                                       int32_t d) {
    __shared__ double L[PANEL][PANEL + 2];
    __shared__ double X[PANEL * XS];
    double* base = Aall + (int64_t)blockIdx.y * sys_stride;
    const int t = threadIdx.x;
    load_l11(L, base + (int64_t)p0 * ld + p0, ld, nb, t, 64);
    const int row0 = p0 + nb + blockIdx.x * 64;
#pragma unroll 8
    for (int i = 0; i < PANEL; ++i) {           // row i of the tile: element t -> X[t][i]
        const int row = row0 + i;
        X[t * XS + i] = (row < d && t < nb) ? base[(int64_t)row * ld + p0 + t] : 0.0;
    }
    __syncthreads();
    tri_solve64<false>(X, t, L);
    __syncthreads();
#pragma unroll 8
    for (int i = 0; i < PANEL; ++i) {
        const int row = row0 + i;
        if (row < d && t < nb) base[(int64_t)row * ld + p0 + t] = X[t * XS + i];
    }
}

// Triangular solve of one 64-row block of right-hand sides with the factored diagonal block:
// forward (TRANS = false):  Y = L11^-1 * R ;  backward (TRANS = true):  X = L11^-T * Y.
// One thread per RHS column (blockIdx.x * 64 + lane); blockIdx.y = system (group, alpha).
template <bool TRANS>
__global__ __launch_bounds__(64) void trsv_cols_kernel(const double* __restrict__ Aall, int64_t ld,
                                                       int64_t sys_stride, int32_t p0, int32_t nb,
                                                       double* __restrict__ Rall, int64_t ldr,
                                                       int32_t n_alpha, int64_t r_grp_stride,
                                                       int64_t r_alpha_stride, int32_t c) {
    __shared__ double L[PANEL][PANEL + 2];
    __shared__ double X[PANEL * XS];
    const int sys = blockIdx.y;
    const int t = threadIdx.x;
    load_l11(L, Aall + (int64_t)sys * sys_stride + (int64_t)p0 * ld + p0, ld, nb, t, 64);
    const int col = blockIdx.x * 64 + t;
    const bool active = col < c;
    double* rhs = Rall + (sys / n_alpha) * r_grp_stride + (sys % n_alpha) * r_alpha_stride +
                  (int64_t)p0 * ldr + (active ? col : 0);
#pragma unroll 8
    for (int k = 0; k < PANEL; ++k)
        X[k * XS + t] = (active && k < nb) ? rhs[(int64_t)k * ldr] : 0.0;
    __syncthreads();
    tri_solve64<TRANS>(X, t, L);
    if (active)
        for (int k = 0; k < nb; ++k) rhs[(int64_t)k * ldr] = X[k * XS + t];
}

// intercept in Z coordinates: c0[g][a][k] = tau_k - sum_i delta_i W[g][i][a][k].
// Block = 64 (alpha, target) columns x 16 slices of the feature index, reduced through LDS.
constexpr int ICPT_SLICES = 16;
__global__ __launch_bounds__(64 * ICPT_SLICES) void intercept_kernel(AssembleArgs p,
                                                                     const double* __restrict__ W,
                                                                     double* __restrict__ c0) {
    __shared__ double part[ICPT_SLICES][64];
    const int grp = blockIdx.y;
    const int ac = p.n_alpha * p.c;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const double ntr = p.ntr[grp];
    double s = 0.0;
    if (e < ac)
        for (int i = slice; i < p.d; i += ICPT_SLICES) {
            double si = p.stot[i];
            if (p.sf) si -= p.sf[(int64_t)grp * p.d + i];
            s += (si / ntr) * W[((int64_t)grp * p.d + i) * ac + e];
        }
    part[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && e < ac) {
        for (int k = 1; k < ICPT_SLICES; ++k) s += part[k][lane];
        const int k = e % p.c;
        double tk = p.ttot[k];
        if (p.tf) tk -= p.tf[(int64_t)grp * p.c + k];
        c0[(int64_t)grp * ac + e] = tk / ntr - s;
    }
}

// ------------------------------------------------------------------------------------------
// Scores.  P (rows x n_alpha*c) = Z W without the intercept; c0 (n_alpha*c).
// R^2 pieces per (alpha, target): out[(a*c+k)*2 + {0,1}] = {sum (t - p - c0)^2, sum (t - tbar)^2}
// with tbar the mean of T over these rows.  One workgroup per (alpha, target) pair.
__global__ __launch_bounds__(256) void r2_sums_kernel(const double* __restrict__ P,
                                                      const double* __restrict__ c0,
                                                      const double* __restrict__ T, int64_t rows,
                                                      int32_t c, int32_t n_alpha,
                                                      const double* __restrict__ tsum,
                                                      double* __restrict__ out) {
    __shared__ double sres[256], stot[256];
    const int a = blockIdx.x / c, k = blockIdx.x - a * c;
    const int ac = n_alpha * c;
    const double tbar = tsum[k] / (double)rows, b = c0[a * c + k];
    double res = 0.0, tot = 0.0;
    for (int64_t i = threadIdx.x; i < rows; i += 256) {
        const double tv = T[i * c + k];
        const double e1 = tv - (P[i * ac + a * c + k] + b), e2 = tv - tbar;
        res += e1 * e1;
        tot += e2 * e2;
    }
    sres[threadIdx.x] = res; stot[threadIdx.x] = tot;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sres[threadIdx.x] += sres[threadIdx.x + s];
            stot[threadIdx.x] += stot[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = sres[0]; out[blockIdx.x * 2 + 1] = stot[0]; }
}

// Accuracy counts per alpha.  n_cls >= 3: predicted class = arg-max over the classes with
// present[k] != 0 (first maximum wins, like numpy.argmax); n_cls == 2 (c == 1): class 1 iff the
// score is > 0.  hits[a] += #(predicted == code).
__global__ __launch_bounds__(256) void accuracy_kernel(const double* __restrict__ P,
                                                       const double* __restrict__ c0,
                                                       const int32_t* __restrict__ code,
                                                       int64_t rows, int32_t c, int32_t n_alpha,
                                                       int32_t n_cls,
                                                       const int32_t* __restrict__ present,
                                                       unsigned long long* __restrict__ hits) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int a = blockIdx.y;
    const int ac = n_alpha * c;
    int ok = 0;
    if (i < rows) {
        const double* p = P + i * ac + a * c;
        const double* b = c0 + a * c;
        int pred;
        if (n_cls == 2) {
            pred = (p[0] + b[0]) > 0.0 ? 1 : 0;
        } else {
            pred = -1;
            double best = -INFINITY;
            for (int k = 0; k < c; ++k) {
                const double v = p[k] + b[k];
                if (present[k] && (pred < 0 || v > best)) { best = v; pred = k; }
            }
        }
        ok = pred == code[i];
    }
    const unsigned long long m = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(hits + a, (unsigned long long)__popcll(m));
}

}  // namespace range_probe

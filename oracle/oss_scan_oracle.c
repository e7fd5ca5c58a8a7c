/*
 * oss_scan_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, sequential in time) of the selective-scan recurrence
 * that the reference's native module `selective_scan_cuda_core` computes.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every entry point
 * against tests/golden/g1_scan_*.npz, which were produced by running the
 * reference's own pure-PyTorch `selective_scan_ref`
 * (reference: Mamba/kernels/selective_scan/test_selective_scan.py:168-234) and
 * torch.autograd through it, in the build container (tests/golden/make_golden.py).
 *
 * What is restated (reference file:line):
 *   forward   Mamba/kernels/selective_scan/csrc/selective_scan/cus/selective_scan_fwd_kernel.cuh:111-162
 *             scan operator  .../selective_scan_common.h:89-96
 *   backward  .../cus/selective_scan_bwd_kernel.cuh:140-272
 *   host glue .../cus/selective_scan.cpp:157-349 (shapes, x layout, dB/dC fp32 accumulation)
 *
 * Layouts (all dense, row-major, caller converts 16-bit inputs to the real type):
 *   u, delta, out, dout, du, ddelta : (batch, dim, L)
 *   A, dA                           : (dim, N)
 *   B, C, dB, dC                    : (batch, G, N, L), row d uses group g = d / (dim / G)
 *   D, dD, bias, dbias              : (dim)  (NULL => absent)
 *   x                               : (batch, dim, n_chunks, 2N), n_chunks = ceil(L / chunk)
 *                                     x[...,2n]   = prod_{t <= end of chunk} a_{n,t}  (from t = 0)
 *                                     x[...,2n+1] = h_{n, end of chunk}
 *
 * Two instantiations: REAL = float (the arithmetic type of the reference kernel)
 * and REAL = double (arbiter when two fp32 implementations disagree).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SOFTPLUS_THRESHOLD 20.0 /* selective_scan_fwd_kernel.cuh:115-118 */

#define DEFINE_ORACLE(REAL, SUFFIX, EXP, LOG1P)                                                     \
                                                                                                    \
    static inline REAL softplus_##SUFFIX(REAL x) {                                                  \
        return (x <= (REAL)SOFTPLUS_THRESHOLD) ? LOG1P(EXP(x)) : x;                                 \
    }                                                                                               \
                                                                                                    \
    /* forward: fwd_kernel.cuh:111-162 */                                                           \
    void oss_oracle_scan_fwd_##SUFFIX(const REAL *u, const REAL *delta, const REAL *A,              \
                                      const REAL *B, const REAL *C, const REAL *D,                  \
                                      const REAL *bias, int batch, int dim, int L, int N, int G,    \
                                      int delta_softplus, int chunk, REAL *out, REAL *x) {          \
        const int rows_per_group = dim / G;                                                         \
        const int n_chunks = (L + chunk - 1) / chunk;                                               \
        _Pragma("omp parallel for collapse(2) schedule(static)")                                    \
        for (int b = 0; b < batch; ++b) {                                                           \
            for (int d = 0; d < dim; ++d) {                                                         \
                const int g = d / rows_per_group; /* fwd_kernel.cuh:82 */                          \
                const REAL *ur = u + ((size_t)b * dim + d) * L;                                     \
                const REAL *dr = delta + ((size_t)b * dim + d) * L;                                 \
                REAL *yr = out + ((size_t)b * dim + d) * L;                                         \
                REAL *xr = x ? x + ((size_t)b * dim + d) * n_chunks * 2 * N : NULL;                 \
                const REAL Dd = D ? D[d] : (REAL)0;                                                 \
                const REAL bd = bias ? bias[d] : (REAL)0;                                           \
                for (int t = 0; t < L; ++t) yr[t] = Dd * ur[t]; /* :120 */                          \
                for (int n = 0; n < N; ++n) {                                                       \
                    const REAL An = A[(size_t)d * N + n];                                           \
                    const REAL *Bn = B + (((size_t)b * G + g) * N + n) * L;                         \
                    const REAL *Cn = C + (((size_t)b * G + g) * N + n) * L;                         \
                    REAL h = 0, p = 1;                                                              \
                    for (int t = 0; t < L; ++t) {                                                   \
                        REAL dt = dr[t] + bd;                                                       \
                        if (delta_softplus) dt = softplus_##SUFFIX(dt);                             \
                        const REAL a = EXP(dt * An);          /* :125-127,137 */                    \
                        const REAL bb = Bn[t] * (dt * ur[t]); /* :119,137 */                        \
                        h = a * h + bb;                       /* common.h:91-96 */                  \
                        p = p * a;                                                                  \
                        yr[t] += Cn[t] * h; /* :160-162 */                                          \
                        if (xr && (((t + 1) % chunk) == 0 || t == L - 1)) { /* :155-158 */          \
                            const int c = t / chunk;                                                \
                            xr[(size_t)c * 2 * N + 2 * n] = p;                                      \
                            xr[(size_t)c * 2 * N + 2 * n + 1] = h;                                  \
                        }                                                                           \
                    }                                                                               \
                }                                                                                   \
            }                                                                                       \
        }                                                                                           \
    }                                                                                               \
                                                                                                    \
    /* backward: bwd_kernel.cuh:140-272.  dB/dC are accumulated in REAL over the rows of a group   \
       in increasing d (the reference uses fp32 atomics, cus/selective_scan.cpp:319-321), dA / dD / \
       dbias over batch in increasing b.  Outputs dA,dB,dC,dD,dbias are overwritten. */            \
    void oss_oracle_scan_bwd_##SUFFIX(const REAL *u, const REAL *delta, const REAL *A,              \
                                      const REAL *B, const REAL *C, const REAL *D,                  \
                                      const REAL *bias, const REAL *dout, int batch, int dim,       \
                                      int L, int N, int G, int delta_softplus, REAL *du,            \
                                      REAL *ddelta, REAL *dA, REAL *dB, REAL *dC, REAL *dD,         \
                                      REAL *dbias) {                                                \
        const int rows_per_group = dim / G;                                                         \
        memset(dA, 0, sizeof(REAL) * (size_t)dim * N);                                              \
        memset(dB, 0, sizeof(REAL) * (size_t)batch * G * N * L);                                    \
        memset(dC, 0, sizeof(REAL) * (size_t)batch * G * N * L);                                    \
        if (dD) memset(dD, 0, sizeof(REAL) * (size_t)dim);                                          \
        if (dbias) memset(dbias, 0, sizeof(REAL) * (size_t)dim);                                    \
        /* parallel over (batch, group): rows of one group are summed in order by one thread */     \
        REAL *dA_part = (REAL *)calloc((size_t)batch * dim * N, sizeof(REAL));                      \
        REAL *dD_part = (REAL *)calloc((size_t)batch * dim, sizeof(REAL));                          \
        REAL *db_part = (REAL *)calloc((size_t)batch * dim, sizeof(REAL));                          \
        _Pragma("omp parallel")                                                                     \
        {                                                                                           \
            REAL *hbuf = (REAL *)malloc(sizeof(REAL) * (size_t)L);                                  \
            REAL *dtb = (REAL *)malloc(sizeof(REAL) * (size_t)L);                                   \
            REAL *abuf = (REAL *)malloc(sizeof(REAL) * (size_t)L);                                  \
            REAL *ddt = (REAL *)malloc(sizeof(REAL) * (size_t)L);                                   \
            _Pragma("omp for collapse(2) schedule(dynamic)")                                        \
            for (int b = 0; b < batch; ++b) {                                                       \
                for (int g = 0; g < G; ++g) {                                                       \
                    for (int d = g * rows_per_group; d < (g + 1) * rows_per_group; ++d) {           \
                        const REAL *ur = u + ((size_t)b * dim + d) * L;                             \
                        const REAL *dr = delta + ((size_t)b * dim + d) * L;                         \
                        const REAL *gr = dout + ((size_t)b * dim + d) * L;                          \
                        REAL *dur = du + ((size_t)b * dim + d) * L;                                 \
                        REAL *ddr = ddelta + ((size_t)b * dim + d) * L;                             \
                        const REAL Dd = D ? D[d] : (REAL)0;                                         \
                        const REAL bd = bias ? bias[d] : (REAL)0;                                   \
                        REAL dD_acc = 0;                                                            \
                        for (int t = 0; t < L; ++t) {                                               \
                            REAL dt = dr[t] + bd;                                                   \
                            if (delta_softplus) dt = softplus_##SUFFIX(dt);                         \
                            dtb[t] = dt;                                                            \
                            dur[t] = Dd * gr[t]; /* :151 */                                         \
                            ddt[t] = 0;                                                             \
                            dD_acc += gr[t] * ur[t]; /* :153 */                                     \
                        }                                                                           \
                        for (int n = 0; n < N; ++n) {                                               \
                            const REAL An = A[(size_t)d * N + n];                                   \
                            const REAL *Bn = B + (((size_t)b * G + g) * N + n) * L;                 \
                            const REAL *Cn = C + (((size_t)b * G + g) * N + n) * L;                 \
                            REAL *dBn = dB + (((size_t)b * G + g) * N + n) * L;                     \
                            REAL *dCn = dC + (((size_t)b * G + g) * N + n) * L;                     \
                            REAL h = 0;                                                             \
                            for (int t = 0; t < L; ++t) { /* recompute fwd, :140-169 */             \
                                const REAL a = EXP(dtb[t] * An);                                    \
                                abuf[t] = a;                                                        \
                                h = a * h + Bn[t] * (dtb[t] * ur[t]);                               \
                                hbuf[t] = h;                                                        \
                            }                                                                       \
                            REAL dh = 0, dA_acc = 0;                                                \
                            for (int t = L - 1; t >= 0; --t) { /* reverse scan, :170-193 */         \
                                const REAL a_next = (t + 1 < L) ? abuf[t + 1] : (REAL)0;            \
                                dh = Cn[t] * gr[t] + a_next * dh;                                   \
                                const REAL bb = Bn[t] * (dtb[t] * ur[t]);                           \
                                const REAL pp = hbuf[t] - bb; /* a_t h_{t-1}, :202 */               \
                                dur[t] += dh * Bn[t] * dtb[t];             /* :200-201 */           \
                                ddt[t] += dh * Bn[t] * ur[t] + dh * An * pp; /* :203 */             \
                                dA_acc += dh * dtb[t] * pp;                /* :204 */               \
                                dBn[t] += dh * dtb[t] * ur[t];             /* :205 */               \
                                dCn[t] += gr[t] * hbuf[t];                 /* :206 */               \
                            }                                                                       \
                            dA_part[((size_t)b * dim + d) * N + n] = dA_acc;                        \
                        }                                                                           \
                        REAL db_acc = 0;                                                            \
                        for (int t = 0; t < L; ++t) { /* :228-245 */                               \
                            REAL v = ddt[t];                                                        \
                            if (delta_softplus) {                                                   \
                                const REAL raw = dr[t] + bd;                                        \
                                if (raw <= (REAL)SOFTPLUS_THRESHOLD) v = v / ((REAL)1 + EXP(-raw)); \
                            }                                                                       \
                            ddr[t] = v;                                                             \
                            db_acc += v;                                                            \
                        }                                                                           \
                        dD_part[(size_t)b * dim + d] = dD_acc;                                      \
                        db_part[(size_t)b * dim + d] = db_acc;                                      \
                    }                                                                               \
                }                                                                                   \
            }                                                                                       \
            free(hbuf); free(dtb); free(abuf); free(ddt);                                           \
        }                                                                                           \
        for (int b = 0; b < batch; ++b) {                                                           \
            for (int d = 0; d < dim; ++d) {                                                         \
                for (int n = 0; n < N; ++n)                                                         \
                    dA[(size_t)d * N + n] += dA_part[((size_t)b * dim + d) * N + n];                \
                if (dD) dD[d] += dD_part[(size_t)b * dim + d];                                      \
                if (dbias) dbias[d] += db_part[(size_t)b * dim + d];                                \
            }                                                                                       \
        }                                                                                           \
        free(dA_part); free(dD_part); free(db_part);                                                \
    }

DEFINE_ORACLE(float, f32, expf, log1pf)
DEFINE_ORACLE(double, f64, exp, log1p)

int oss_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oss_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

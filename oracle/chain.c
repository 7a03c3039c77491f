/* CPU oracle, C part: bit-exact statement of the accumulation order the HIP
 * similarity kernel commits to, plus a stable full ranking.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/oracle.py header): linked/loaded only by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * The reference computes scores with BLAS sgemm (np.dot,
 * mdir/components/optim/score/cirscore.py:69), whose fp32 summation order is
 * unspecified.  gfx950's fp32 MFMA is bit-for-bit a k-ordered fmaf chain, so the
 * build fixes ONE legal order -- k = 0,1,...,D-1, one fused multiply-add per k,
 * starting from +0 -- and this file states it in portable C.  The numpy oracle
 * (oracle.scores) bounds the distance to the BLAS result; this one is what the
 * GPU must match to the last bit.
 *
 * Build: oracle/Makefile  ->  oracle/liboracle_chain.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* scores[q*n + i] = sum_k qv[k*nq + q] * db[k*n + i], k ascending, fmaf chain.
 * db  : [d][n]  fp32, the reference's dimension-major layout
 *       (cirtorch/networks/imageretrievalnet.py:291)
 * qv  : [d][nq] fp32
 * out : [nq][n] fp32 (row per query; the transpose of cirscore.py:69's [N,Q]) */
void oracle_scores_chain(const float *db, const float *qv, int64_t n, int64_t d,
                         int64_t nq, float *out)
{
    enum { BLK = 512 };
    #pragma omp parallel for schedule(static)
    for (int64_t i0 = 0; i0 < n; i0 += BLK) {
        int64_t w = n - i0 < BLK ? n - i0 : BLK;
        float acc[BLK];
        for (int64_t q = 0; q < nq; ++q) {
            for (int64_t i = 0; i < w; ++i) acc[i] = 0.0f;
            for (int64_t k = 0; k < d; ++k) {
                const float a = qv[k * nq + q];
                const float *row = db + k * n + i0;
                for (int64_t i = 0; i < w; ++i) acc[i] = fmaf(a, row[i], acc[i]);
            }
            memcpy(out + q * n + i0, acc, (size_t)w * sizeof(float));
        }
    }
}

/* Same chain for row-major operands: a [m][d], b [n][d] -> out [m][n];
 * out[i][j] = chain_k a[i][k]*b[j][k] (+ bias[i] added LAST, if given).
 * States the order of the whitening projection kernel (rows of P against
 * centred descriptors, mdir/components/data/wrapper.py:193-195). */
void oracle_gemm_nt_chain(const float *a, const float *b, int64_t m, int64_t n,
                          int64_t d, float *out)
{
    #pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) {
            float acc = 0.0f;
            for (int64_t k = 0; k < d; ++k) acc = fmaf(a[i * d + k], b[j * d + k], acc);
            out[i * n + j] = acc;
        }
}

/* Sort key: larger score first; -0 == +0; NaN after everything (numpy's
 * argsort(-s) also puts NaN last, cirscore.py:70); ties by ascending id. */
static inline uint32_t desc_key(float s)
{
    uint32_t u;
    if (s != s) return 0xFFFFFFFFu;
    if (s == 0.0f) s = 0.0f;
    memcpy(&u, &s, 4);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u); /* ascending order-preserving */
    return ~u;                                      /* descending */
}

uint32_t oracle_desc_key(float s) { return desc_key(s); }

/* Full ranking of one query row: ranks[r] = database id at position r.
 * LSD radix on (key) is stable, so equal keys keep ascending id. */
static void rank_row(const float *s, int64_t n, int64_t *ranks, uint32_t *k0, uint32_t *k1,
                     int64_t *i1)
{
    int64_t *src_i = ranks, *dst_i = i1;
    uint32_t *src_k = k0, *dst_k = k1;
    for (int64_t i = 0; i < n; ++i) { k0[i] = desc_key(s[i]); ranks[i] = i; }
    for (int pass = 0; pass < 4; ++pass) {
        int64_t cnt[257];
        memset(cnt, 0, sizeof cnt);
        const int sh = 8 * pass;
        for (int64_t i = 0; i < n; ++i) cnt[((src_k[i] >> sh) & 255u) + 1]++;
        for (int b = 0; b < 256; ++b) cnt[b + 1] += cnt[b];
        for (int64_t i = 0; i < n; ++i) {
            int64_t p = cnt[(src_k[i] >> sh) & 255u]++;
            dst_k[p] = src_k[i];
            dst_i[p] = src_i[i];
        }
        uint32_t *tk = src_k; src_k = dst_k; dst_k = tk;
        int64_t *ti = src_i; src_i = dst_i; dst_i = ti;
    }
    /* 4 passes = even number of swaps: result is back in `ranks`. */
}

/* scores [nq][n] -> ranks [nq][n] int64 (row per query). */
void oracle_rank_full(const float *scores, int64_t n, int64_t nq, int64_t *ranks)
{
    #pragma omp parallel
    {
        uint32_t *k0 = malloc((size_t)n * 4), *k1 = malloc((size_t)n * 4);
        int64_t *i1 = malloc((size_t)n * 8);
        #pragma omp for schedule(dynamic)
        for (int64_t q = 0; q < nq; ++q) rank_row(scores + q * n, n, ranks + q * n, k0, k1, i1);
        free(k0); free(k1); free(i1);
    }
}

/* Position of given ids in the ranking of one query row, without sorting:
 * #items with a smaller key + #equal-key items with a smaller id. */
void oracle_rank_of(const float *s, int64_t n, const int64_t *ids, int64_t nids, int64_t *pos)
{
    for (int64_t t = 0; t < nids; ++t) {
        const int64_t id = ids[t];
        const uint32_t kid = desc_key(s[id]);
        int64_t c = 0;
        for (int64_t i = 0; i < n; ++i) {
            const uint32_t k = desc_key(s[i]);
            c += (k < kid) || (k == kid && i < id);
        }
        pos[t] = c;
    }
}

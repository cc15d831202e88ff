/*
 * rm2_oracle.c -- CPU restatement of filmyou-core's RM2 MapReduce job.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product library (filmyou-core_amd/csrc) never does.
 *
 * It follows the reference's Java source stage by stage, keeping its loop nest, its double arithmetic and
 * its observable quirks.  M/ = /root/reference/src/main/java/es/udc/fi/dc/irlab/
 *
 *   job RM2-1  M/rm/RM2Job.java:110-151, M/rm/SimpleScoreByUserHDFSMapper.java:34-42 (score > 0 filter),
 *              M/rm/DoubleSumAndCountReducer.java:31-45 (s_u, and counter += (long) s_u * 100  -- quirk Q1)
 *   job RM2-2  M/rm/RM2Job.java:164-205 (totalSum = counter / OFFSET, :95), M/rm/SimpleScoreByItemHDFSMapper.java:34-42,
 *              M/rm/DoubleSumAndDividerReducer.java:31-46 (p(i|C) = itemsum / totalSum)
 *   job RM2-3  M/rm/RM2Job.java:214-270; routing M/common/AbstractByClusterMapper.java:77-79 (unknown user -> cluster 0, Q2)
 *              and M/common/AbstractByClusterAndCountMapper.java:86-102 (splits: they partition the target users of a
 *              cluster and never change a score, so one pass per cluster is made here);
 *              reducer M/rm/AbstractRM2Reducer.java:129-233 (reduce), :238-272 (item set = items rated inside the cluster),
 *              :281-303 (p(i|C) lookup), :185-190 + :384-389 (dense cache of p(i|u)), :321-371 (buildRecommendations),
 *              ordering M/util/IntDouble.java:31-34, output cast M/rm/RM2HDFSReducer.java:48.
 *
 * Pinned against the reference's own golden vectors (tests/golden/rm_test_data.json, transcribed from
 * src/test/java/.../testdata/RMTestData.java:234-464 and ClusteringTestData.java:90-93) by tests/test_oracle_golden.py.
 *
 * Where the reference leaves an order unspecified (hash-map iteration, PriorityQueue ties) the oracle uses
 * ascending raw ids; that changes nothing but the order of exactly tied scores.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "oracle.h"

static char g_err[256];
const char* rm2o_last_error(void) { return g_err; }

/* Test-side selection: when set, job RM2-3 runs only the listed reduce groups (clusters); jobs RM2-1/2 (user sums, totalSum,
 * p(i|C)) still see ALL ratings, as in the reference, where each cluster is its own reduce group (RM2Job.java:251).  Used to run
 * whole clusters of a full-size job through the fp64 scorers in bounded time. */
static int g_n_sel = 0;
static int32_t g_sel[64];
void rm2o_select_clusters(int32_t n, const int32_t* list) {
    g_n_sel = n < 0 ? 0 : (n > 64 ? 64 : n);
    for (int k = 0; k < g_n_sel; k++) g_sel[k] = list[k];
}
static int cluster_selected(int c) {
    if (g_n_sel == 0) return 1;
    for (int k = 0; k < g_n_sel; k++)
        if (g_sel[k] == c) return 1;
    return 0;
}

#define FAIL(code, ...)                              \
    do {                                             \
        snprintf(g_err, sizeof g_err, __VA_ARGS__);  \
        rc = (code);                                 \
        goto done;                                   \
    } while (0)

typedef struct {
    int32_t id;
    int32_t aux;
} id_pair;

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}

/* index of `key` in sorted unique array, or -1 */
static int64_t find_sorted(const int32_t* a, int64_t n, int32_t key) {
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] == key) return mid;
        if (a[mid] < key) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

static int64_t sort_unique(int32_t* a, int64_t n) {
    if (n == 0) return 0;
    qsort(a, (size_t)n, sizeof(int32_t), cmp_i32);
    int64_t m = 1;
    for (int64_t i = 1; i < n; i++)
        if (a[i] != a[m - 1]) a[m++] = a[i];
    return m;
}

/* java.lang.Double.compare(a, b): NaN is greater than everything, -0.0 < 0.0 */
static int java_double_compare(double a, double b) {
    if (a < b) return -1;
    if (a > b) return 1;
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an == bn ? 0 : (an ? 1 : -1);
    int as = signbit(a) != 0, bs = signbit(b) != 0; /* both equal in value: order -0.0 before 0.0 */
    return as == bs ? 0 : (as ? -1 : 1);
}

typedef struct {
    int32_t item_local;
    int32_t item_raw;
    double score;
} cand;

/* IntDouble.compareTo is Double.compare(other.value, this.value): the queue polls the LARGEST value first
 * (M/util/IntDouble.java:31-34); ties are broken here by ascending raw item id. */
static int cmp_cand(const void* pa, const void* pb) {
    const cand* a = (const cand*)pa;
    const cand* b = (const cand*)pb;
    int c = java_double_compare(b->score, a->score);
    if (c) return c;
    return (a->item_raw > b->item_raw) - (a->item_raw < b->item_raw);
}

struct rm2o_result {
    int64_t n_recs;
    int32_t *rec_user, *rec_item, *rec_cluster;
    float* rec_score;
    int64_t n_users;
    int32_t* user_id;
    double* user_sum;
    int64_t n_items;
    int32_t* item_id;
    double* item_coll;
    double* item_sum;
    double total_sum;
    int64_t log_terms;   /* (u, i, j) terms evaluated: sum over scored users of n_u * |unrated(u)| */
    int64_t fma_terms;   /* log_terms * (U_c - 1) */
    double seconds_scoring;
};

int64_t rm2o_n_recs(const rm2o_result* r) { return r->n_recs; }
const int32_t* rm2o_rec_user(const rm2o_result* r) { return r->rec_user; }
const int32_t* rm2o_rec_item(const rm2o_result* r) { return r->rec_item; }
const int32_t* rm2o_rec_cluster(const rm2o_result* r) { return r->rec_cluster; }
const float* rm2o_rec_score(const rm2o_result* r) { return r->rec_score; }
int64_t rm2o_n_users(const rm2o_result* r) { return r->n_users; }
const int32_t* rm2o_user_id(const rm2o_result* r) { return r->user_id; }
const double* rm2o_user_sum(const rm2o_result* r) { return r->user_sum; }
int64_t rm2o_n_items(const rm2o_result* r) { return r->n_items; }
const int32_t* rm2o_item_id(const rm2o_result* r) { return r->item_id; }
const double* rm2o_item_coll(const rm2o_result* r) { return r->item_coll; }
const double* rm2o_item_sum(const rm2o_result* r) { return r->item_sum; }
double rm2o_total_sum(const rm2o_result* r) { return r->total_sum; }
int64_t rm2o_log_terms(const rm2o_result* r) { return r->log_terms; }
int64_t rm2o_fma_terms(const rm2o_result* r) { return r->fma_terms; }
double rm2o_seconds_scoring(const rm2o_result* r) { return r->seconds_scoring; }

void rm2o_free(rm2o_result* r) {
    if (!r) return;
    free(r->rec_user); free(r->rec_item); free(r->rec_cluster); free(r->rec_score);
    free(r->user_id); free(r->user_sum); free(r->item_id); free(r->item_coll); free(r->item_sum);
    free(r);
}

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

/* `gram` != 0 selects the restructured scorer of rm2o_run_gram (below); everything in front of the per-user loop is shared */
static int run_impl(const rm2o_params* P, int64_t nnz_in, const int32_t* user, const int32_t* item, const float* score,
                    int64_t n_map, const int32_t* map_user, const int32_t* map_cluster, const int32_t* cluster_count,
                    rm2o_result** out, int gram) {
    int rc = 0;
    rm2o_result* R = (rm2o_result*)calloc(1, sizeof *R);
    int32_t *uid = NULL, *iid = NULL, *du = NULL, *di = NULL, *ucl = NULL, *csize = NULL, *cstart = NULL, *cusers = NULL;
    int64_t *rowptr = NULL, *fill = NULL;
    int32_t* col = NULL;
    float* val = NULL;
    double *usum = NULL, *isum = NULL, *icoll = NULL;
    int64_t cap = 0;
    g_err[0] = 0;
    if (!R) return -1;
    const int K = P->number_of_clusters;
    if (K <= 0) FAIL(-2, "numberOfClusters must be > 0");

    /* ---- R1: keep score > 0 only (SimpleScoreBy*Mapper.map) ---- */
    int64_t nnz = 0;
    for (int64_t t = 0; t < nnz_in; t++) nnz += score[t] > 0;
    uid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    iid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    {
        int64_t k = 0;
        for (int64_t t = 0; t < nnz_in; t++)
            if (score[t] > 0) { uid[k] = user[t]; iid[k] = item[t]; k++; }
    }
    int64_t nU = sort_unique(uid, nnz), nI = sort_unique(iid, nnz);

    /* ---- job RM2-1: s_u = sum of (double) score, counter += (long) s_u * OFFSET  (Q1) ---- */
    usum = (double*)calloc((size_t)nU + 1, sizeof(double));
    isum = (double*)calloc((size_t)nI + 1, sizeof(double));
    icoll = (double*)calloc((size_t)nI + 1, sizeof(double));
    du = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    di = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    rowptr = (int64_t*)calloc((size_t)nU + 2, sizeof(int64_t));
    {
        int64_t k = 0;
        for (int64_t t = 0; t < nnz_in; t++) {
            if (!(score[t] > 0)) continue;
            int64_t u = find_sorted(uid, nU, user[t]), i = find_sorted(iid, nI, item[t]);
            du[k] = (int32_t)u; di[k] = (int32_t)i; k++;
            usum[u] += (double)score[t];
            isum[i] += (double)score[t];
            rowptr[u + 1]++;
        }
    }
    long long counter = 0;
    for (int64_t u = 0; u < nU; u++) counter += (long long)usum[u] * 100LL; /* cast binds to the sum first */
    const double total_sum = (double)counter / (double)100LL;                /* RM2Job.java:95 "sum / OFFSET" */

    /* ---- job RM2-2: p(i|C) = itemsum / totalSum ---- */
    for (int64_t i = 0; i < nI; i++) icoll[i] = isum[i] / total_sum;

    /* user-major copy of the ratings (the reducer's sparsePreferences) */
    for (int64_t u = 0; u < nU; u++) rowptr[u + 1] += rowptr[u];
    col = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    val = (float*)malloc(sizeof(float) * (size_t)(nnz + 1));
    fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nU + 1));
    memcpy(fill, rowptr, sizeof(int64_t) * (size_t)nU);
    {
        int64_t k = 0;
        for (int64_t t = 0; t < nnz_in; t++) {
            if (!(score[t] > 0)) continue;
            int64_t pos = fill[du[k]]++;
            col[pos] = di[k]; val[pos] = score[t]; k++;
        }
    }

    /* ---- routing: cluster of every user (AbstractByClusterMapper.getCluster; missing -> 0) ---- */
    ucl = (int32_t*)calloc((size_t)nU + 1, sizeof(int32_t));
    for (int64_t m = 0; m < n_map; m++) {
        int64_t u = find_sorted(uid, nU, map_user[m]);
        if (u >= 0) ucl[u] = map_cluster[m];
    }
    csize = (int32_t*)calloc((size_t)K + 1, sizeof(int32_t));
    cstart = (int32_t*)calloc((size_t)K + 2, sizeof(int32_t));
    for (int64_t u = 0; u < nU; u++) {
        if (ucl[u] < 0 || ucl[u] >= K) FAIL(-3, "user %d is mapped to cluster %d outside [0,%d)", uid[u], ucl[u], K);
        csize[ucl[u]]++;
    }
    if (cluster_count)
        for (int c = 0; c < K; c++)
            if (csize[c] && cluster_count[c] != csize[c])
                /* AbstractRM2Reducer.java:153-160 reads exactly clusterSizes[c] user-sum records first */
                FAIL(-4, "clusteringCount[%d]=%d but %d rated users are routed to that cluster", c, cluster_count[c], csize[c]);
    for (int c = 0; c < K; c++) cstart[c + 1] = cstart[c] + csize[c];
    cusers = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nU + 1));
    {
        int32_t* f = (int32_t*)malloc(sizeof(int32_t) * (size_t)(K + 1));
        memcpy(f, cstart, sizeof(int32_t) * (size_t)K);
        for (int64_t u = 0; u < nU; u++) cusers[f[ucl[u]]++] = (int32_t)u;
        free(f);
    }

    /* ---- job RM2-3: one reduce group per cluster ---- */
    const double lambda = P->lambda;
    const int topn = P->number_of_recommendations;
    int nthreads = P->n_threads > 0 ? P->n_threads : 1;
#ifndef _OPENMP
    nthreads = 1;
#endif
    double t_score = 0.0;
    for (int c = 0; c < K; c++) {
        const int Uc = csize[c];
        if (Uc == 0 || !cluster_selected(c)) continue;
        const int32_t* cu = cusers + cstart[c];
        /* createUserAndItemMappings: items rated by somebody of this cluster, dense re-index */
        int32_t* loc = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nI + 1));
        for (int64_t i = 0; i < nI; i++) loc[i] = -1;
        for (int a = 0; a < Uc; a++)
            for (int64_t e = rowptr[cu[a]]; e < rowptr[cu[a] + 1]; e++) loc[col[e]] = 0;
        int Ic = 0;
        for (int64_t i = 0; i < nI; i++)
            if (loc[i] == 0) loc[i] = Ic++;
        int32_t* items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(Ic + 1));
        for (int64_t i = 0; i < nI; i++)
            if (loc[i] >= 0) items[loc[i]] = (int32_t)i;

        /* cache[v][i] = probItemGivenUser(i, v) = (1-lambda)*(rating/sum) + lambda*p(i|C), rating = 0.0 if absent */
        double* cache = NULL;
        double *G = NULL, *bvec = NULL;   /* gram scorer only */
        if (gram) {
            /* "Best CPU" restructuring (NOT the reference's loop nest; the same identity the GPU path uses, in fp64):
             *   sum_{v != u} c_vi c_vj = G[j][i] + (l p_i) e_uj   for i not rated by u,
             *   G[j][i] = (1-l)^2 (X^T X)_ji + l (1-l) p_j b_i,  X_vi = r_vi / s_v,  b_i = sum_v X_vi,
             *   e_uj    = (1-l) (b_j - x_uj) + l (U_c - 1) p_j.
             * One dense Ic x Ic matrix per cluster replaces the U_c - 1 multiply-adds per log term. */
            const char* bud = getenv("RM2O_GRAM_BUDGET_GB");   /* host memory the dense fp64 Gram may take (default 12 GB) */
            if ((double)Ic * (double)Ic * 8.0 > (bud ? atof(bud) : 12.0) * 1e9) { free(loc); free(items); FAIL(-6, "cluster %d: the %d x %d Gram does not fit the CPU baseline's budget", c, Ic, Ic); }
            G = (double*)calloc((size_t)Ic * (size_t)Ic, sizeof(double));
            bvec = (double*)calloc((size_t)Ic + 1, sizeof(double));
            if (!G || !bvec) { free(G); free(bvec); free(loc); free(items); FAIL(-5, "cluster %d: cannot allocate the %d x %d Gram", c, Ic, Ic); }
            for (int a = 0; a < Uc; a++) {
                const double sum = usum[cu[a]];
                for (int64_t e = rowptr[cu[a]]; e < rowptr[cu[a] + 1]; e++) bvec[loc[col[e]]] += (double)val[e] / sum;
            }
            /* X^T X: every user adds the outer product of its row; rows of G are independent -> parallel over j */
            const double w2 = (1 - lambda) * (1 - lambda), w1 = lambda * (1 - lambda);
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads)
            for (int j = 0; j < Ic; j++) {
                double* row = G + (size_t)j * Ic;
                const double pj = icoll[items[j]];
                for (int i = 0; i < Ic; i++) row[i] = w1 * pj * bvec[i];
            }
            /* scatter by user (sequential over users, parallel over the user's rated rows: distinct rows of G) */
            for (int a = 0; a < Uc; a++) {
                const double sum = usum[cu[a]];
                const int64_t r0 = rowptr[cu[a]], r1 = rowptr[cu[a] + 1];
#pragma omp parallel for schedule(static) num_threads(nthreads) if (r1 - r0 > 256)
                for (int64_t e = r0; e < r1; e++) {
                    double* row = G + (size_t)loc[col[e]] * Ic;
                    const double xj = w2 * (double)val[e] / sum;
                    for (int64_t f = r0; f < r1; f++) row[loc[col[f]]] += xj * ((double)val[f] / sum);
                }
            }
        } else {
            cache = (double*)malloc(sizeof(double) * (size_t)Uc * (size_t)Ic);
            if (!cache) { free(loc); free(items); FAIL(-5, "cluster %d: cannot allocate the %d x %d cache", c, Uc, Ic); }
        }
        for (int a = 0; a < Uc && !gram; a++) {
            double* row = cache + (size_t)a * Ic;
            const double sum = usum[cu[a]];
            for (int i = 0; i < Ic; i++) row[i] = (1 - lambda) * (0.0 / sum) + lambda * icoll[items[i]];
            for (int64_t e = rowptr[cu[a]]; e < rowptr[cu[a] + 1]; e++) {
                int i = loc[col[e]];
                row[i] = (1 - lambda) * ((double)val[e] / sum) + lambda * icoll[items[i]];
            }
        }

        /* per-user output slots so that the parallel loop writes in a fixed order */
        cand** ulist = (cand**)calloc((size_t)Uc, sizeof(cand*));
        int* ucount = (int*)calloc((size_t)Uc, sizeof(int));
        int64_t c_terms = 0;
        double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(+ : c_terms)
        for (int a = 0; a < Uc; a++) {
            const int32_t u = cu[a];
            const int n = (int)(rowptr[u + 1] - rowptr[u]);
            const int n_unrated = Ic - n;
            if (n_unrated == 0) continue;              /* "does not have any unrated item in the cluster" */
            if (uid[u] < P->filter_users) continue;    /* AbstractRM2Reducer.java:221-223 */
            char* is_rated = (char*)calloc((size_t)Ic, 1);
            int32_t* rated = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n + 1));
            for (int k = 0; k < n; k++) { rated[k] = loc[col[rowptr[u] + k]]; is_rated[rated[k]] = 1; }
            cand* list = (cand*)malloc(sizeof(cand) * (size_t)n_unrated);
            int m = 0;
            const double pvpi = (n - 1) * log((double)P->number_of_items) - n * log((double)Uc);
            if (gram) {
                /* rows of G of the user's rated items are streamed once each; acc[i] collects the log terms */
                double* acc = (double*)calloc((size_t)Ic, sizeof(double));
                const double sum_u = usum[u];
                for (int k = 0; k < n; k++) {
                    const int j = rated[k];
                    const double x = (double)val[rowptr[u] + k] / sum_u;
                    const double e = (1 - lambda) * (bvec[j] - x) + lambda * (double)(Uc - 1) * icoll[items[j]];
                    const double* row = G + (size_t)j * Ic;
                    for (int i = 0; i < Ic; i++) acc[i] += log(row[i] + lambda * icoll[items[i]] * e);
                }
                for (int i = 0; i < Ic; i++) {
                    if (is_rated[i]) continue;
                    list[m].item_local = i; list[m].item_raw = iid[items[i]]; list[m].score = acc[i] + pvpi; m++;
                }
                free(acc);
            }
            for (int i = 0; i < Ic && !gram; i++) {
                if (is_rated[i]) continue;
                double log_result = 0.0;
                for (int k = 0; k < n; k++) {
                    const int j = rated[k];
                    double sum = 0.0;
                    for (int v = 0; v < Uc; v++) {       /* neighbours = all users of the cluster but u */
                        if (v == a) continue;
                        sum += cache[(size_t)v * Ic + i] * cache[(size_t)v * Ic + j];
                    }
                    log_result += log(sum);
                }
                log_result += pvpi;
                list[m].item_local = i; list[m].item_raw = iid[items[i]]; list[m].score = log_result; m++;
            }
            c_terms += (int64_t)n * n_unrated;
            qsort(list, (size_t)m, sizeof(cand), cmp_cand);
            int keep = topn < m ? topn : m;
            if (keep < 0) keep = 0;
            ulist[a] = list; ucount[a] = keep;
            free(is_rated); free(rated);
        }
        t_score += now_s() - t0;
        R->log_terms += c_terms;
        R->fma_terms += gram ? 0 : c_terms * (int64_t)(Uc - 1);

        for (int a = 0; a < Uc; a++) {
            if (!ulist[a]) continue;
            if (R->n_recs + ucount[a] > cap) {
                cap = (R->n_recs + ucount[a]) * 2 + 1024;
                R->rec_user = (int32_t*)realloc(R->rec_user, sizeof(int32_t) * (size_t)cap);
                R->rec_item = (int32_t*)realloc(R->rec_item, sizeof(int32_t) * (size_t)cap);
                R->rec_cluster = (int32_t*)realloc(R->rec_cluster, sizeof(int32_t) * (size_t)cap);
                R->rec_score = (float*)realloc(R->rec_score, sizeof(float) * (size_t)cap);
            }
            for (int k = 0; k < ucount[a]; k++) {
                int64_t o = R->n_recs++;
                R->rec_user[o] = uid[cu[a]];
                R->rec_item[o] = ulist[a][k].item_raw;
                R->rec_score[o] = (float)ulist[a][k].score;  /* RM2HDFSReducer.java:48 */
                R->rec_cluster[o] = c;
            }
            free(ulist[a]);
        }
        free(ulist); free(ucount); free(cache); free(G); free(bvec); free(items); free(loc);
    }

    R->n_users = nU; R->user_id = uid; R->user_sum = usum; uid = NULL; usum = NULL;
    R->n_items = nI; R->item_id = iid; R->item_coll = icoll; R->item_sum = isum; iid = NULL; icoll = NULL; isum = NULL;
    R->total_sum = total_sum;
    R->seconds_scoring = t_score;
done:
    free(uid); free(iid); free(du); free(di); free(ucl); free(csize); free(cstart); free(cusers);
    free(rowptr); free(fill); free(col); free(val); free(usum); free(isum); free(icoll);
    if (rc) { rm2o_free(R); R = NULL; }
    *out = R;
    return rc;
}

int rm2o_run(const rm2o_params* P, int64_t nnz_in, const int32_t* user, const int32_t* item, const float* score,
             int64_t n_map, const int32_t* map_user, const int32_t* map_cluster, const int32_t* cluster_count,
             rm2o_result** out) {
    return run_impl(P, nnz_in, user, item, score, n_map, map_user, map_cluster, cluster_count, out, 0);
}

/* The same job with the scoring loop restructured around a per-cluster Gram matrix (see run_impl): bench.py's
 * "cpu_baseline_gram" line, so that the GPU / CPU ratio is not inflated by the reference's O(U_c) inner loop.
 * Not the reference's algorithm: checked against rm2o_run (tests/test_oracle_golden.py), never used as the parity oracle. */
int rm2o_run_gram(const rm2o_params* P, int64_t nnz_in, const int32_t* user, const int32_t* item, const float* score,
                  int64_t n_map, const int32_t* map_user, const int32_t* map_cluster, const int32_t* cluster_count,
                  rm2o_result** out) {
    return run_impl(P, nnz_in, user, item, score, n_map, map_user, map_cluster, cluster_count, out, 1);
}

/*
 * itemsim_oracle.c -- CPU statement of the item-item similarity build.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The arithmetic of this path is NOT in /root/reference: the reference only calls
 * org.apache.mahout:mahout-core:0.8 `RowSimilarityJob` (pom.xml:15-19) at
 * M/baselinerecommender/BaselineRecommenderJob.java:241-253 with --similarityClassname, --maxSimilaritiesPerRow,
 * --excludeSelfSimilarity true and --threshold; the package is excluded from compilation (pom.xml:81-83) and no
 * reference test or golden vector covers it.  This file therefore restates Mahout 0.8's published algorithm:
 *   normsAndTranspose : cosine L2-normalises every item row (CosineSimilarity.normalize);
 *   pairwiseSimilarity: for every user column, every pair of its items contributes aggregate(a,b)
 *                       (cosine: a*b of the normalised values; co-occurrence: 1); the sums per (i,j) are the
 *                       similarity; j == i dropped when excludeSelfSimilarity; values < threshold dropped
 *                       (NO_THRESHOLD = Double.MIN_VALUE: only non-positive values are dropped);
 *   asMatrix          : mirror to the full matrix, keep the top maxSimilaritiesPerRow per item.
 * Mahout's RANDOM down-sampling of users with more than maxPrefsPerUserInItemSimilarity preferences
 * (BaselinePreparePreferenceMatrixJob.java:126-129) has no reproducible output: the option is modelled by a deterministic
 * systematic sample instead (filter_prefs below), and minPrefsPerUser (:104) as a plain drop.
 * Ties in the top-K (unspecified in Mahout's TopElementsQueue) are broken by ascending item id.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "oracle.h"

struct isimo_result {
    int64_t n;
    int32_t *item, *other;
    double* sim;
    int64_t pairs;
    double seconds;
};
int64_t isimo_n(const isimo_result* r) { return r->n; }
const int32_t* isimo_item(const isimo_result* r) { return r->item; }
const int32_t* isimo_other(const isimo_result* r) { return r->other; }
const double* isimo_sim(const isimo_result* r) { return r->sim; }
int64_t isimo_pairs(const isimo_result* r) { return r->pairs; }
double isimo_seconds(const isimo_result* r) { return r->seconds; }
void isimo_free(isimo_result* r) {
    if (!r) return;
    free(r->item); free(r->other); free(r->sim); free(r);
}

static int cmp_i32(const void* a, const void* b) {
    int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return (x > y) - (x < y);
}
static int64_t sort_unique(int32_t* a, int64_t n) {
    if (n == 0) return 0;
    qsort(a, (size_t)n, sizeof(int32_t), cmp_i32);
    int64_t m = 1;
    for (int64_t i = 1; i < n; i++)
        if (a[i] != a[m - 1]) a[m++] = a[i];
    return m;
}
static int64_t find_sorted(const int32_t* a, int64_t n, int32_t key) {
    int64_t lo = 0, hi = n - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] == key) return mid;
        if (a[mid] < key) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
typedef struct { int32_t j; double s; } ent;
static int cmp_ent(const void* pa, const void* pb) {
    const ent* a = (const ent*)pa; const ent* b = (const ent*)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    return (a->j > b->j) - (a->j < b->j);
}

/* input preparation: drop users with fewer than min_prefs preferences, cut users with more than max_prefs down to max_prefs by
 * the deterministic systematic sample of include/filmyou.h (preference k of n, in ascending item id, is kept iff
 * floor((k+1) m / n) > floor(k m / n)).  Mahout's own sampling is random: this models the OPTION, not a Mahout run. */
typedef struct { int32_t u, i; float s; } pref;
static int cmp_pref(const void* a, const void* b) {
    const pref* x = (const pref*)a; const pref* y = (const pref*)b;
    if (x->u != y->u) return (x->u > y->u) - (x->u < y->u);
    return (x->i > y->i) - (x->i < y->i);
}
static int64_t filter_prefs(int64_t nnz, const int32_t* user, const int32_t* item, const float* score, int min_prefs, int max_prefs,
                            int32_t* ou, int32_t* oi, float* os) {
    pref* p = (pref*)malloc(sizeof(pref) * (size_t)(nnz + 1));
    for (int64_t t = 0; t < nnz; t++) { p[t].u = user[t]; p[t].i = item[t]; p[t].s = score[t]; }
    qsort(p, (size_t)nnz, sizeof(pref), cmp_pref);
    int64_t o = 0;
    for (int64_t a = 0; a < nnz;) {
        int64_t b = a;
        while (b < nnz && p[b].u == p[a].u) b++;
        const int64_t deg = b - a;
        for (int64_t k = 0; k < deg; k++) {
            int ok = deg >= min_prefs;
            if (ok && max_prefs > 0 && deg > max_prefs) ok = ((k + 1) * max_prefs) / deg > (k * max_prefs) / deg;
            if (ok) { ou[o] = p[a + k].u; oi[o] = p[a + k].i; os[o] = p[a + k].s; o++; }
        }
        a = b;
    }
    free(p);
    return o;
}

int isimo_run(const isimo_params* P, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
              isimo_result** out) {
    int32_t *fu = NULL, *fi = NULL;
    float* fs = NULL;
    if (P->min_prefs_per_user > 1 || P->max_prefs_per_user > 0) {
        fu = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
        fi = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
        fs = (float*)malloc(sizeof(float) * (size_t)(nnz + 1));
        nnz = filter_prefs(nnz, user, item, score, P->min_prefs_per_user, P->max_prefs_per_user, fu, fi, fs);
        user = fu; item = fi; score = fs;
    }
    isimo_result* R = (isimo_result*)calloc(1, sizeof *R);
    int32_t* uid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    int32_t* iid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    memcpy(uid, user, sizeof(int32_t) * (size_t)nnz);
    memcpy(iid, item, sizeof(int32_t) * (size_t)nnz);
    const int64_t nU = sort_unique(uid, nnz), nI = sort_unique(iid, nnz);
    int64_t* uptr = (int64_t*)calloc((size_t)nU + 2, sizeof(int64_t));
    int64_t* iptr = (int64_t*)calloc((size_t)nI + 2, sizeof(int64_t));
    int32_t* du = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    int32_t* di = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    double* norm2 = (double*)calloc((size_t)nI + 1, sizeof(double));
    for (int64_t t = 0; t < nnz; t++) {
        du[t] = (int32_t)find_sorted(uid, nU, user[t]);
        di[t] = (int32_t)find_sorted(iid, nI, item[t]);
        uptr[du[t] + 1]++; iptr[di[t] + 1]++;
        norm2[di[t]] += (double)score[t] * (double)score[t];
    }
    for (int64_t u = 0; u < nU; u++) { R->pairs += uptr[u + 1] * (uptr[u + 1] - 1) / 2; uptr[u + 1] += uptr[u]; }
    for (int64_t i = 0; i < nI; i++) iptr[i + 1] += iptr[i];
    /* user-major (item, normalised value) and item-major (user) copies */
    int32_t* ucol = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    double* uval = (double*)malloc(sizeof(double) * (size_t)(nnz + 1));
    int32_t* irow = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    double* ival = (double*)malloc(sizeof(double) * (size_t)(nnz + 1));
    int64_t* uf = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nU + 1));
    int64_t* itf = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nI + 1));
    memcpy(uf, uptr, sizeof(int64_t) * (size_t)nU);
    memcpy(itf, iptr, sizeof(int64_t) * (size_t)nI);
    for (int64_t t = 0; t < nnz; t++) {
        double x = P->similarity == ISIM_COSINE ? (double)score[t] / sqrt(norm2[di[t]]) : 1.0;
        int64_t a = uf[du[t]]++, b = itf[di[t]]++;
        ucol[a] = di[t]; uval[a] = x;
        irow[b] = du[t]; ival[b] = x;
    }
    const int K = P->max_similarities_per_item;
    ent** rows = (ent**)calloc((size_t)nI + 1, sizeof(ent*));
    int* rown = (int*)calloc((size_t)nI + 1, sizeof(int));
    int nthreads = P->n_threads > 0 ? P->n_threads : 1;
    double t0 = 0.0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        double* acc = (double*)calloc((size_t)nI + 1, sizeof(double));
        char* hit = (char*)calloc((size_t)nI + 1, 1);
        int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nI + 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t i = 0; i < nI; i++) {
            int nt = 0;
            for (int64_t e = iptr[i]; e < iptr[i + 1]; e++) {
                const int32_t u = irow[e];
                const double xi = ival[e];
                for (int64_t f = uptr[u]; f < uptr[u + 1]; f++) {
                    const int32_t j = ucol[f];
                    if (!hit[j]) { hit[j] = 1; touched[nt++] = j; }
                    acc[j] += xi * uval[f];
                }
            }
            ent* list = (ent*)malloc(sizeof(ent) * (size_t)(nt + 1));
            int m = 0;
            for (int k = 0; k < nt; k++) {
                const int32_t j = touched[k];
                const double s = acc[j];
                acc[j] = 0.0; hit[j] = 0;
                if (P->exclude_self && j == i) continue;
                if (P->has_threshold ? !(s >= P->threshold) : !(s > 0.0)) continue;
                list[m].j = iid[j]; list[m].s = s; m++;
            }
            qsort(list, (size_t)m, sizeof(ent), cmp_ent);
            rows[i] = list; rown[i] = m < K ? m : K;
        }
        free(acc); free(hit); free(touched);
    }
#ifdef _OPENMP
    R->seconds = omp_get_wtime() - t0;
#endif
    for (int64_t i = 0; i < nI; i++) R->n += rown[i];
    R->item = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R->n + 1));
    R->other = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R->n + 1));
    R->sim = (double*)malloc(sizeof(double) * (size_t)(R->n + 1));
    int64_t o = 0;
    for (int64_t i = 0; i < nI; i++) {
        for (int k = 0; k < rown[i]; k++) { R->item[o] = iid[i]; R->other[o] = rows[i][k].j; R->sim[o] = rows[i][k].s; o++; }
        free(rows[i]);
    }
    free(rows); free(rown); free(uid); free(iid); free(uptr); free(iptr); free(du); free(di); free(norm2);
    free(ucol); free(uval); free(irow); free(ival); free(uf); free(itf);
    free(fu); free(fi); free(fs);
    *out = R;
    return 0;
}

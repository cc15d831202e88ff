/*
 * itemcf_oracle.c -- CPU statement of the item-based CF recommendation phases (partialMultiply + aggregateAndRecommend)
 * that consume the similarity matrix.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED (no reference test or fixture covers the baselinerecommender package; it is excluded from
 * compilation, pom.xml:81-83).  What is restated, and from where:
 *   in the reference tree (followed line by line)
 *     M/baselinerecommender/BaselineAggregateAndRecommendReducer.java:97-161  reduceNonBooleanData: per similarity column
 *         count[i]++ for every stored entry, denominators += |sim|, numerators += pref * sim; prediction = num / den only
 *         where count > 1 ("at least 2 datapoints", :140-146)
 *     :80-96   reduceBooleanData: predictions = sum of the similarity columns
 *     :195-235 writeRecommendedItems: (float) value, NaN skipped, the numRecommendations largest
 *     M/baselinerecommender/BaselineRecommenderJob.java:70, 146-148, 309  only the maxPrefsPerUser (default 50) strongest
 *         preferences of a user are considered
 *   Mahout 0.8 classes wired at BaselineRecommenderJob.java:289-298, 347-348 (third-party, NOT in the tree; restated
 *   from the published algorithm, unverified):
 *     UserVectorSplitterMapper: a user with more than maxPrefsPerUser preferences keeps those >= the
 *         maxPrefsPerUser-th largest value (ties at the threshold are all kept);
 *     SimilarityMatrixRowWrapperMapper: the column of item j is its similarity row plus the entry (j, NaN), which is how
 *         items the user already rated drop out (their prediction is NaN);
 *     ToVectorAndPrefReducer: an item without a similarity row contributes nothing;
 *     TopItemsQueue: ties in unspecified order (here: ascending item id).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

struct icfo_result {
    int64_t n;
    int32_t *user, *item;
    float* score;
};
int64_t icfo_n(const icfo_result* r) { return r->n; }
const int32_t* icfo_user(const icfo_result* r) { return r->user; }
const int32_t* icfo_item(const icfo_result* r) { return r->item; }
const float* icfo_score(const icfo_result* r) { return r->score; }
void icfo_free(icfo_result* r) {
    if (!r) return;
    free(r->user); free(r->item); free(r->score); free(r);
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
static int cmp_float_desc(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x < y) - (x > y);
}
typedef struct { int32_t item; float v; } rec;
static int cmp_rec(const void* pa, const void* pb) {
    const rec* a = (const rec*)pa; const rec* b = (const rec*)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->item > b->item) - (a->item < b->item);
}

int icfo_run(const icfo_params* P, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
             int64_t n_sim, const int32_t* sim_item, const int32_t* sim_other, const double* sim_value, icfo_result** out) {
    icfo_result* R = (icfo_result*)calloc(1, sizeof *R);
    int32_t* uid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    int32_t* iid = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    memcpy(uid, user, sizeof(int32_t) * (size_t)nnz);
    memcpy(iid, item, sizeof(int32_t) * (size_t)nnz);
    const int64_t nU = sort_unique(uid, nnz), nI = sort_unique(iid, nnz);
    /* user-major preferences */
    int64_t* uptr = (int64_t*)calloc((size_t)nU + 2, sizeof(int64_t));
    int32_t* du = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    for (int64_t t = 0; t < nnz; t++) { du[t] = (int32_t)find_sorted(uid, nU, user[t]); uptr[du[t] + 1]++; }
    for (int64_t u = 0; u < nU; u++) uptr[u + 1] += uptr[u];
    int32_t* pj = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    float* pv = (float*)malloc(sizeof(float) * (size_t)(nnz + 1));
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nU + 1));
    memcpy(fill, uptr, sizeof(int64_t) * (size_t)nU);
    for (int64_t t = 0; t < nnz; t++) { int64_t p = fill[du[t]]++; pj[p] = (int32_t)find_sorted(iid, nI, item[t]); pv[p] = score[t]; }
    /* similarity rows by dense item index (rows arrive grouped by item) */
    int64_t* sstart = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nI + 1));
    int64_t* scnt = (int64_t*)calloc((size_t)nI + 1, sizeof(int64_t));
    for (int64_t i = 0; i < nI; i++) sstart[i] = -1;
    for (int64_t t = 0; t < n_sim; t++) {
        int64_t i = find_sorted(iid, nI, sim_item[t]);
        if (i < 0) continue;
        if (sstart[i] < 0) sstart[i] = t;
        scnt[i]++;
    }
    double* num = (double*)calloc((size_t)nI + 1, sizeof(double));
    double* den = (double*)calloc((size_t)nI + 1, sizeof(double));
    int32_t* cnt = (int32_t*)calloc((size_t)nI + 1, sizeof(int32_t));
    int32_t* touched = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nI + 1));
    rec* list = (rec*)malloc(sizeof(rec) * (size_t)(nI + 1));
    float* tmp = (float*)malloc(sizeof(float) * (size_t)(nnz + 1));
    int64_t cap = 0;
    for (int64_t u = 0; u < nU; u++) {
        const int64_t a = uptr[u], b = uptr[u + 1];
        float threshold = -INFINITY;
        if (b - a > P->max_prefs_per_user) {   /* keep the maxPrefsPerUser strongest preferences (ties at the cut kept) */
            memcpy(tmp, pv + a, sizeof(float) * (size_t)(b - a));
            qsort(tmp, (size_t)(b - a), sizeof(float), cmp_float_desc);
            threshold = tmp[P->max_prefs_per_user - 1];
        }
        int nt = 0;
        for (int64_t e = a; e < b; e++) {
            if (pv[e] < threshold) continue;
            const int32_t j = pj[e];
            if (sstart[j] < 0) continue;                       /* no similarity row: the preference contributes nothing */
            const double p = P->boolean_data ? 1.0 : (double)pv[e];
            for (int64_t t = sstart[j]; t < sstart[j] + scnt[j]; t++) {
                const int64_t i = find_sorted(iid, nI, sim_other[t]);
                if (i < 0) continue;
                if (!cnt[i]) touched[nt++] = (int32_t)i;
                cnt[i]++;
                den[i] += fabs(sim_value[t]);
                num[i] += p * sim_value[t];
            }
            if (!cnt[j]) touched[nt++] = j;                    /* the (j, NaN) entry of the wrapped column */
            cnt[j]++;
            den[j] = NAN;
            num[j] = NAN;
        }
        int m = 0;
        for (int k = 0; k < nt; k++) {
            const int32_t i = touched[k];
            double pred = NAN;
            /* numerators.nonZeroes() / recommendationVector.nonZeroes() (BaselineAggregateAndRecommendReducer.java:148, 195):
             * an exactly-zero numerator never becomes a candidate */
            if (num[i] == 0.0) pred = NAN;
            else if (P->boolean_data) pred = num[i];
            else if (cnt[i] > 1) pred = num[i] / den[i];
            cnt[i] = 0; num[i] = 0.0; den[i] = 0.0;
            const float v = (float)pred;
            if (isnan(v)) continue;
            list[m].item = iid[i]; list[m].v = v; m++;
        }
        qsort(list, (size_t)m, sizeof(rec), cmp_rec);
        int keep = m < P->num_recommendations ? m : P->num_recommendations;
        if (R->n + keep > cap) {
            cap = (R->n + keep) * 2 + 1024;
            R->user = (int32_t*)realloc(R->user, sizeof(int32_t) * (size_t)cap);
            R->item = (int32_t*)realloc(R->item, sizeof(int32_t) * (size_t)cap);
            R->score = (float*)realloc(R->score, sizeof(float) * (size_t)cap);
        }
        for (int k = 0; k < keep; k++) { R->user[R->n] = uid[u]; R->item[R->n] = list[k].item; R->score[R->n] = list[k].v; R->n++; }
    }
    free(uid); free(iid); free(uptr); free(du); free(pj); free(pv); free(fill); free(sstart); free(scnt);
    free(num); free(den); free(cnt); free(touched); free(list); free(tmp);
    *out = R;
    return 0;
}

/*
 * oracle.h -- C interface of the CPU oracles (TEST INFRASTRUCTURE ONLY, see rm2_oracle.c / itemsim_oracle.c).
 * Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product path.
 */
#ifndef FILMYOU_ORACLE_H
#define FILMYOU_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* keys mirror the Hadoop Configuration keys of M/rmrecommender/RMRecommenderDriver.java:49-120 */
typedef struct {
    double lambda;                      /* "lambda" */
    int32_t number_of_items;            /* "numberOfItems" (global, used only in pvpi -- quirk Q6) */
    int32_t number_of_recommendations;  /* "numberOfRecommendations" */
    int32_t filter_users;               /* "filterUsers": users with id < this get no list */
    int32_t number_of_clusters;         /* "numberOfClusters" */
    int32_t n_threads;                  /* OpenMP threads over the target users of a cluster (1 = serial reducer) */
} rm2o_params;

typedef struct rm2o_result rm2o_result;

/* COO ratings (raw ids), clustering pairs (may be empty: every user -> cluster 0), optional clusteringCount[K]. */
int rm2o_run(const rm2o_params* P, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
             int64_t n_map, const int32_t* map_user, const int32_t* map_cluster, const int32_t* cluster_count,
             rm2o_result** out);
/* the same job, scoring restructured around a per-cluster Gram matrix ("best CPU" baseline; not the parity oracle) */
int rm2o_run_gram(const rm2o_params* P, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
                  int64_t n_map, const int32_t* map_user, const int32_t* map_cluster, const int32_t* cluster_count,
                  rm2o_result** out);
/* test-side: run only these clusters in job RM2-3 (n = 0: all); the statistics jobs always see every rating */
void rm2o_select_clusters(int32_t n, const int32_t* list);
const char* rm2o_last_error(void);
void rm2o_free(rm2o_result*);
/* recommendations: grouped by cluster asc, user id asc, then best first (ties: ascending item id) */
int64_t rm2o_n_recs(const rm2o_result*);
const int32_t* rm2o_rec_user(const rm2o_result*);
const int32_t* rm2o_rec_item(const rm2o_result*);
const int32_t* rm2o_rec_cluster(const rm2o_result*);
const float* rm2o_rec_score(const rm2o_result*);
/* rm2/userSum and rm2/itemColl equivalents, ascending raw id */
int64_t rm2o_n_users(const rm2o_result*);
const int32_t* rm2o_user_id(const rm2o_result*);
const double* rm2o_user_sum(const rm2o_result*);
int64_t rm2o_n_items(const rm2o_result*);
const int32_t* rm2o_item_id(const rm2o_result*);
const double* rm2o_item_coll(const rm2o_result*);
const double* rm2o_item_sum(const rm2o_result*);
double rm2o_total_sum(const rm2o_result*);
int64_t rm2o_log_terms(const rm2o_result*);
int64_t rm2o_fma_terms(const rm2o_result*);
double rm2o_seconds_scoring(const rm2o_result*);

/* ---- item-item similarity (Mahout 0.8 RowSimilarityJob as called at M/baselinerecommender/BaselineRecommenderJob.java:241-253) ---- */
enum { ISIM_COSINE = 0, ISIM_COOCCURRENCE = 1 };
typedef struct {
    int32_t similarity;                 /* --similarityClassname: SIMILARITY_COSINE | SIMILARITY_COOCCURRENCE */
    int32_t max_similarities_per_item;  /* --maxSimilaritiesPerRow (default 100, BaselineRecommenderJob.java:67) */
    int32_t exclude_self;               /* --excludeSelfSimilarity (the call site passes true) */
    int32_t has_threshold;              /* 0 = RowSimilarityJob.NO_THRESHOLD */
    double threshold;                   /* --threshold */
    int32_t n_threads;
    int32_t min_prefs_per_user;         /* users with fewer preferences are dropped (BaselinePreparePreferenceMatrixJob.java:104); <= 1: nobody */
    int32_t max_prefs_per_user;         /* 0 = no cap; else the deterministic systematic sample of include/filmyou.h (Mahout samples at RANDOM) */
} isimo_params;
typedef struct isimo_result isimo_result;
int isimo_run(const isimo_params* P, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
              isimo_result** out);
void isimo_free(isimo_result*);
/* rows grouped by item id asc, best first (ties: ascending other-item id) */
int64_t isimo_n(const isimo_result*);
const int32_t* isimo_item(const isimo_result*);
const int32_t* isimo_other(const isimo_result*);
const double* isimo_sim(const isimo_result*);
int64_t isimo_pairs(const isimo_result*);   /* sum_u n_u (n_u - 1) / 2 */
double isimo_seconds(const isimo_result*);

/* ---- item-based CF recommendation phases 3-4 (M/baselinerecommender/BaselineAggregateAndRecommendReducer.java) ---- */
typedef struct {
    int32_t num_recommendations;   /* --numRecommendations (default 100, BaselineRecommenderJob.java:66) */
    int32_t max_prefs_per_user;    /* --maxPrefsPerUser: strongest preferences considered per user (default 50, :70) */
    int32_t boolean_data;          /* --booleanData */
} icfo_params;
typedef struct icfo_result icfo_result;
/* similarity rows (item, other, value) grouped by item, as produced by the similarity build */
int icfo_run(const icfo_params* P, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
             int64_t n_sim, const int32_t* sim_item, const int32_t* sim_other, const double* sim_value, icfo_result** out);
void icfo_free(icfo_result*);
int64_t icfo_n(const icfo_result*);          /* rows grouped by user id asc, best first (ties: ascending item id) */
const int32_t* icfo_user(const icfo_result*);
const int32_t* icfo_item(const icfo_result*);
const float* icfo_score(const icfo_result*);

/* ---- cluster assignment (cluster_oracle.c): FindClusterMapper / FindSubClusterMapper / CountReducer */
int clo_assign(int32_t n_rows, int32_t k, const double* H /* row-major n_rows x k */, int32_t first_user, int32_t cluster_offset,
               int32_t* user, int32_t* cluster);
int clo_count(int64_t n, const int32_t* cluster, int32_t n_clusters, int32_t* count);

/* ---- NMF / PPC factorisation (nmf_oracle.c).  Returns 0, -2 id out of range, -3 a user without ratings, -4 an item without */
int nmfo_run(int32_t n_users, int32_t n_items, int32_t k, int32_t iterations, int32_t ppc, int32_t norm_freq, int64_t nnz,
             const int32_t* user, const int32_t* item, const float* score, double* H, double* W);

#ifdef __cplusplus
}
#endif

#endif

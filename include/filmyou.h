/*
 * filmyou.h -- C ABI of the MI355X-native replacement for filmyou-core's two recommendation jobs.
 *
 * This is the drop-in boundary: the entry points below are what a JNI shim binds in place of the reference's
 * MapReduce Driver/Mapper/Reducer classes (binding shown in INTEGRATION.md).  Plain pointers and sizes only.
 * M/ = src/main/java/es/udc/fi/dc/irlab/ in dvalcarce/filmyou-core.
 *
 *   fy_rm2_*      replaces  ToolRunner.run(conf, new RM2Job(), args)   M/rmrecommender/RMRecommenderDriver.java:200-201
 *                 i.e. M/rm/RM2Job.java:76-100 (jobs RM2-1, RM2-2, RM2-3) and everything they run:
 *                 the M/rm mappers, the M/rm DoubleSum reducers, M/rm/AbstractRM2Reducer.java:129-389,
 *                 M/common/AbstractByCluster*Mapper.java (routing), M/util/IntDouble.java (ordering).
 *   fy_itemsim_*  replaces  ToolRunner.run(getConf(), new RowSimilarityJob(), {...})
 *                 M/baselinerecommender/BaselineRecommenderJob.java:241-253 (Mahout 0.8 RowSimilarityJob).
 *
 * Conventions: no exceptions cross the ABI; every function returns FY_OK (0) or a negative fy_status and leaves a
 * message for fy_last_error() (thread-local).  Input arrays are caller-owned and may be freed as soon as the call
 * returns.  Results are library-owned until fy_result_free.  One fy_context drives one GPU; a process uses one
 * context per device (one process per GPU under torch.distributed / a Hadoop task per GPU).  A context is not
 * re-entrant: calls on one context must not overlap; distinct contexts may be used from distinct threads.
 * Every entry point that takes a context, or a ratings / job / result object made on one, selects that context's
 * device for the calling thread (hipSetDevice) before it touches the GPU, and leaves it selected: a host may drive
 * several devices from one process and hand an object to another thread.
 * There is NO CPU fallback: without a gfx950 device every compute entry point fails with FY_ERR_NO_DEVICE.
 */
#ifndef FILMYOU_H
#define FILMYOU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FY_ABI_VERSION 5

typedef enum {
    FY_OK = 0,
    FY_ERR_INVALID_ARGUMENT = -1,
    FY_ERR_NO_DEVICE = -2,       /* no HIP device / not gfx950 */
    FY_ERR_HIP = -3,             /* a HIP runtime call failed */
    FY_ERR_OUT_OF_MEMORY = -4,
    FY_ERR_CLUSTER_RANGE = -5,   /* a user is routed to a cluster outside [0, numberOfClusters): the reference would
                                    throw ArrayIndexOutOfBounds at M/common/AbstractByClusterAndCountMapper.java:88 */
    FY_ERR_CLUSTER_COUNT = -6,   /* clusteringCount disagrees with the rated users routed to a cluster: the reference
                                    reducer reads exactly clusterSizes[c] user-sum records first (AbstractRM2Reducer.java:153-160) */
    FY_ERR_DUPLICATE_RATING = -7,/* two ratings for one (user, item): Cassandra's PRIMARY KEY (user, item) forbids it */
    FY_ERR_NEGATIVE_ID = -8,     /* user / item ids must be >= 0 */
    FY_ERR_STATE = -9,           /* calls made out of order */
    FY_ERR_UNSUPPORTED = -10,
    FY_ERR_COLLECTIVE = -11,     /* a fy_collectives callback returned non-zero */
    FY_ERR_IO = -12              /* fy_seqfile_* / fy_mapfile_*: missing, truncated, compressed or foreign-typed file */
} fy_status;

typedef struct fy_context fy_context;
typedef struct fy_ratings fy_ratings;
typedef struct fy_rm2_job fy_rm2_job;
typedef struct fy_result fy_result;

/* ------------------------------------------------------------------ context */
int fy_abi_version(void);
const char* fy_last_error(void);
/* Binds the calling process to GPU `device_ordinal`, creates the HIP stream every kernel of this context runs on. */
int fy_context_create(int device_ordinal, fy_context** out);
void fy_context_destroy(fy_context*);
int fy_context_synchronize(fy_context*);
/* Fault injection for the error-path tests (the reference has none): the nth HBM request of this
 * context from now on (1 = the next one) fails with FY_ERR_OUT_OF_MEMORY; 0 disarms.  A failed job leaves the context
 * usable: every stream is drained before any buffer of the job is released. */
int fy_context_inject_alloc_failure(fy_context*, int64_t nth);
/* The library's launch-shape knobs (FY_* environment variables: test and measurement hooks) are read from
 * the environment once, by fy_context_create; this call reads them again.  No job reads the environment. */
int fy_context_reload_tuning(fy_context*);
/* The hipStream_t of the context (as void*), e.g. to order caller-side copies against the job. */
void* fy_context_stream(fy_context*);

/* ------------------------------------------------------------------ ratings
 * COO triples with the caller's raw int32 ids (1-based in the reference's data) exactly as the reference's
 * mappers receive them: table ratings(user int, item int, score float) (test/.../util/CassandraUtils.java:93-94) or
 * SequenceFile<IntPairWritable(user,item), FloatWritable> (M/util/DataInitialization.java:164-172).
 * `location`: FY_HOST pointers are copied over PCIe; FY_DEVICE pointers (HBM of the context's GPU) are copied
 * device-to-device, so the caller keeps ownership either way. */
enum { FY_HOST = 0, FY_DEVICE = 1 };
int fy_ratings_create(fy_context*, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
                      int location, fy_ratings** out);
void fy_ratings_destroy(fy_ratings*);
int64_t fy_ratings_nnz(const fy_ratings*);
/* Releases what the jobs keep on the ratings object between calls (see fy_rm2_params::flags): the next job builds everything again.
 * A caching job whose clustering, rank or world differs from the kept state's releases it itself before it builds its own; a
 * FY_RM2_NO_CACHE job neither uses nor touches it (call this first when its memory is wanted back: the kept state of an ML-25M-shaped
 * job is ~2.5 GB).  Jobs over ONE ratings object must not run concurrently; distinct ratings objects (and contexts) may. */
void fy_ratings_drop_cache(fy_ratings*);

/* ------------------------------------------------------------------ RM2 job
 * Field names follow the Hadoop Configuration keys of M/rmrecommender/RMRecommenderDriver.java:49-120. */
typedef struct {
    double lambda;                      /* "lambda" (default 0.1), Jelinek-Mercer smoothing */
    int32_t number_of_items;            /* "numberOfItems": global item count, used only in pvpi (AbstractRM2Reducer.java:327-329) */
    int32_t number_of_recommendations;  /* "numberOfRecommendations" (default 1000) */
    int32_t filter_users;               /* "filterUsers": users with id < this get no list (AbstractRM2Reducer.java:221-223) */
    int32_t number_of_clusters;         /* "numberOfClusters" */
    int32_t rank;                       /* this process scores user shard `rank` of `world` (1 GPU: 0 of 1) */
    int32_t world;
    uint32_t flags;                     /* FY_RM2_* below, 0 = defaults */
    int64_t workspace_bytes;            /* cap for the per-batch score scratch in HBM; 0 = default (16 GiB) */
} fy_rm2_params;

/* flags: by default a job keeps what it built from the ratings and the clustering alone (CSR / CSC, per-item statistics, the row
 * kernel's tables) on the fy_ratings object, and a later job over the same ratings AND the same clustering (and rank / world)
 * starts from it -- fy_stats::prepared_from_cache / tables_from_cache.  The reference has no such state: every RM2Job.run re-reads
 * and re-shuffles the ratings (RM2Job.java:130-258).  FY_RM2_NO_CACHE: build everything in this job, keep nothing (the "cold" job). */
#define FY_RM2_NO_CACHE 1u

/* Stage 1 (jobs RM2-1/RM2-2 up to the exchange): score > 0 filter, CSR/CSC in HBM, cluster routing, user sums,
 * and this rank's PARTIAL per-item rating sums + partial floor-sum total.
 * Clustering = the reference's `clustering` file as (user, cluster) pairs; n_map = 0 routes every user to
 * cluster 0 (Trove default, quirk Q2).  cluster_count (numberOfClusters ints, the `clusteringCount` file) may be
 * NULL; when given it is validated.  map_* / cluster_count are HOST pointers. */
int fy_rm2_prepare(fy_context*, const fy_rm2_params*, const fy_ratings*, int64_t n_map, const int32_t* map_user,
                   const int32_t* map_cluster, const int32_t* cluster_count, fy_rm2_job** out);
/* The exchange buffer of this rank, in HBM: `*len` doubles, the same `*len` on every rank (the ratings are replicated):
 * all-gather it (RCCL) into world * len doubles.  Two layouts (fy_rm2_stats_layout tells which):
 *   replicated prep : [one partial rating sum per rated item, ascending raw item id][this rank's partial sum of floor(s_u), quirk Q1]
 *   sharded prep    : a job of several ranks with at least as many non-empty clusters as ranks gives every rank WHOLE clusters and
 *                     preps the ratings of its own clusters alone (the reference's map-side partitioning by cluster,
 *                     IntKeyPartitioner.java:15); its buffer is indexed by RAW id:
 *                     [max item id + 1 item sums][the floor-sum][max user id + 1 user sums s_u][failure flag of this rank's share]. */
int fy_rm2_partial_stats(fy_rm2_job*, double** device_buf, int64_t* len);
/* n_item_slots item sums, one floor-sum, n_user_slots user sums (0: replicated prep, no user sums and no flag), the flag. */
int fy_rm2_stats_layout(fy_rm2_job*, int64_t* n_item_slots, int64_t* n_user_slots);
/* `gathered` = world * len doubles in HBM, rank-major.  Summed in rank order (bit-reproducible).  Optional when world == 1. */
int fy_rm2_set_global_stats(fy_rm2_job*, const double* gathered_device, int32_t world);
/* Collectives of the process group the `world` ranks form (one process per GPU).  The signatures are RCCL's
 * (ncclAllGather / ncclReduceScatter with ncclSum on float): device pointers, the operation is enqueued in order on
 * `stream` (a hipStream_t) -- no host synchronisation is implied.  Return 0 on success.  `user` is handed back verbatim.
 *   all_gather:      recv[k * bytes .. (k + 1) * bytes) = rank k's send[0 .. bytes)
 *   reduce_scatter:  recv[i] = sum over ranks k of send_k[rank * count + i],  i < count   (send holds world * count floats)
 * With collectives installed, a cluster whose users span ALL ranks is scored cooperatively: every rank
 * builds only its row range of the cluster's co-rating matrix, evaluates every user's partial log-sums over those rows, and
 * the partial sums of the seed columns, of the block bounds and of the surviving blocks are reduce-scattered to the rank
 * that owns the user -- about 1/world of the single-GPU work and matrix per rank.  Without them (or when a cluster lives on
 * fewer ranks) every rank builds the whole matrix of the clusters it holds users of, and only the user loop is sharded.
 * If fy_rm2_set_global_stats was not called, fy_rm2_score all-gathers the partial statistics through `all_gather` itself.
 * Every rank must install collectives (or none), and all ranks must call fy_rm2_score together. */
typedef struct {
    void* user;
    int (*all_gather)(void* user, const void* send, void* recv, int64_t bytes, void* stream);
    int (*reduce_scatter_f32)(void* user, const float* send, float* recv, int64_t count, void* stream);
} fy_collectives;
int fy_rm2_set_collectives(fy_rm2_job*, const fy_collectives*);

/* The compiled transport for fy_collectives: RCCL (ncclAllGather / ncclReduceScatter over xGMI) queued on the context's own
 * stream, no host synchronisation.  librccl.so is opened at run time, a single-GPU host does not need it.  One process per
 * GPU: rank 0 makes the 128-byte id (fy_rccl_unique_id) and hands it to the other ranks by whatever channel the host has
 * (the Hadoop Configuration / a file / torch.distributed's store); fy_rccl_create is collective over all ranks
 * (ncclCommInitRank).  Replaces the shuffle + DistributedCache + counter exchange of M/rm/RM2Job.java:130-149, 184-198,
 * 260-263 for hosts that are not Python (the C++ mirror, the JNI shim). */
typedef struct fy_rccl fy_rccl;
int fy_rccl_unique_id(char* out128);
int fy_rccl_create(fy_context*, int rank, int world, const char* id128, fy_rccl** out);
int fy_rccl_collectives(fy_rccl*, fy_collectives* out);   /* fills the callbacks; `out->user` is the fy_rccl, which must outlive the job */
int fy_rccl_counters(const fy_rccl*, int64_t* all_gathers, int64_t* reduce_scatters, int64_t* payload_bytes);
void fy_rccl_destroy(fy_rccl*);           /* the context must still exist (its stream is drained first) ... */
void fy_rccl_detach_context(fy_rccl*);    /* ... unless this was called: the context is already gone, only the communicator is released */
/* Stage 2 (job RM2-3): per-cluster co-rating matrix, p(i|u) scoring of this rank's users, top-N. */
int fy_rm2_score(fy_rm2_job*, fy_result** out);
void fy_rm2_job_destroy(fy_rm2_job*);

/* One call = one complete single-GPU job on host buffers (what the JNI shim calls): context on device 0. */
int fy_rm2_run(const fy_rm2_params*, int64_t nnz, const int32_t* user, const int32_t* item, const float* score,
               int64_t n_map, const int32_t* map_user, const int32_t* map_cluster, const int32_t* cluster_count,
               fy_result** out);

/* ------------------------------------------------------------------ item-item similarity build */
enum { FY_SIMILARITY_COSINE = 0, FY_SIMILARITY_COOCCURRENCE = 1 };
typedef struct {
    int32_t similarity;                 /* --similarityClassname SIMILARITY_COSINE | SIMILARITY_COOCCURRENCE */
    int32_t max_similarities_per_item;  /* --maxSimilaritiesPerRow (default 100, BaselineRecommenderJob.java:67) */
    int32_t exclude_self;               /* --excludeSelfSimilarity (call site passes true) */
    int32_t has_threshold;              /* 0 = RowSimilarityJob.NO_THRESHOLD */
    double threshold;                   /* --threshold */
    int32_t rank;                       /* this process builds item-row shard `rank` of `world` */
    int32_t world;
    uint32_t flags;
    /* input preparation of the preference matrix (M/baselinerecommender/BaselinePreparePreferenceMatrixJob.java:104, 126-129):
     * users with fewer than min_prefs_per_user preferences are dropped (Mahout ToUserVectorsReducer.MIN_PREFERENCES_PER_USER;
     * reference default 1 = nobody); users with more than max_prefs_per_user preferences are cut down to that many.  Mahout's
     * ToItemVectorsMapper draws a RANDOM sample there (no reproducible output exists), this library a DETERMINISTIC systematic
     * one over the user's preferences in ascending item id: preference k of n is kept iff floor((k+1) m / n) > floor(k m / n)
     * -- NOT parity with any particular Mahout run, documented as such.  0 = option off. */
    int32_t min_prefs_per_user;
    int32_t max_prefs_per_user;
} fy_itemsim_params;
int fy_itemsim_build(fy_context*, const fy_itemsim_params*, const fy_ratings*, fy_result** out);
int fy_itemsim_run(const fy_itemsim_params*, int64_t nnz, const int32_t* user, const int32_t* item,
                   const float* score, fy_result** out);

/* ------------------------------------------------------------------ item-based CF recommendation (phases 3-4)
 * Replaces the partialMultiply and aggregateAndRecommend jobs of M/baselinerecommender/BaselineRecommenderJob.java:285-328,
 * 340-393 (reducer M/baselinerecommender/BaselineAggregateAndRecommendReducer.java:97-161, 195-235): prediction(u, i) =
 * sum_j sim(j, i) pref(u, j) / sum_j |sim(j, i)| over the user's maxPrefsPerUser strongest preferences j whose similarity
 * row holds i, kept only where at least two preferences contribute; items of those preferences are excluded; the
 * numRecommendations largest predictions per user.  `similarities` is the result of fy_itemsim_build with world == 1 on
 * the same context (the whole matrix).  Users are sharded by (rank, world).  Output rows: (user, item, (float) prediction, 0). */
typedef struct {
    int32_t num_recommendations;   /* --numRecommendations (default 100, BaselineRecommenderJob.java:66) */
    int32_t max_prefs_per_user;    /* --maxPrefsPerUser (default 50, :70) */
    int32_t boolean_data;          /* --booleanData */
    int32_t rank;
    int32_t world;
    uint32_t flags;
} fy_itemcf_params;
int fy_itemcf_recommend(fy_context*, const fy_itemcf_params*, const fy_ratings*, fy_result* similarities, fy_result** out);

/* ------------------------------------------------------------------ cluster assignment (the stage in front of the RM2 job)
 * Replaces ClusterAssignmentJob's map-only jobs and CountClustersJob (M/nmf/clustering/ClusterAssignmentJob.java:60-135,
 * FindClusterMapper.java:37-45, FindSubClusterMapper.java:46-76, CountReducer.java:31-45): for every row j of H (n_rows x k
 * doubles, row-major; the factor matrix the NMF/PPC stage leaves behind)
 *     user[j] = first_user + j          (the SequenceFile key of DataInitialization.createDoubleMatrix / the H files)
 *     cluster[j] = cluster_offset + first index of the largest value of the row   (Vector.maxValueIndex(); -1 + no offset if
 *                  no value exceeds -infinity)
 * Sub-clustering calls it once per parent cluster c with cluster_offset = c * ceil(numberOfUsers / numberOfClusters).
 * count_inout (host, n_clusters ints, may be NULL) is INCREMENTED per routed user = the `clusteringCount` file; a cluster id
 * outside [0, n_clusters) then fails with FY_ERR_CLUSTER_RANGE.  H: FY_HOST or FY_DEVICE; user_out / cluster_out: host arrays of
 * n_rows ints -- exactly the (map_user, map_cluster, cluster_count) arguments of fy_rm2_prepare / fy_rm2_run. */
int fy_cluster_assign(fy_context*, int32_t n_rows, int32_t k, const double* H, int location, int32_t first_user,
                      int32_t cluster_offset, int32_t n_clusters, int32_t* user_out, int32_t* cluster_out, int32_t* count_inout);

/* ------------------------------------------------------------------ results
 * Rows as the reference writes them: RM2 (user, item, (float) relevance, cluster) -- RM2HDFSReducer.java:48 /
 * RM2CassandraReducer.java:49-63 -- grouped by user, best first; item-sim (item, other item, similarity) grouped by
 * item, best first.  Accessors return HOST pointers (the first accessor call downloads from HBM and synchronises). */
int64_t fy_result_size(fy_result*);
const int32_t* fy_result_key0(fy_result*);      /* user (RM2) | item (item-sim) */
const int32_t* fy_result_key1(fy_result*);      /* item (RM2) | other item (item-sim) */
const float* fy_result_value(fy_result*);       /* relevance | similarity */
const int32_t* fy_result_aux(fy_result*);       /* cluster (RM2) | 0 */
/* rm2/userSum and rm2/itemColl (what TestHDFSRM2.java:70-71 asserts), ascending raw id; RM2 only */
int64_t fy_result_n_users(fy_result*);
const int32_t* fy_result_user_id(fy_result*);
const double* fy_result_user_sum(fy_result*);
int64_t fy_result_n_items(fy_result*);
const int32_t* fy_result_item_id(fy_result*);
const double* fy_result_item_coll(fy_result*);
double fy_result_total_sum(fy_result*);
void fy_result_free(fy_result*);

typedef struct {
    int64_t nnz;               /* ratings kept (score > 0) */
    int64_t n_users, n_items, n_clusters_nonempty;
    int64_t users_scored;      /* users that received a list (this rank) */
    int64_t recs;              /* rows in the result (this rank) */
    int64_t log_terms;         /* RM2: (u, i unrated, j rated) terms evaluated by the scoring kernel (this rank) */
    int64_t pair_contribs;     /* ordered co-rating pair contributions accumulated by the row kernel (= sum n_u^2 walked) */
    int64_t unordered_pairs;   /* item-sim unit: sum_u n_u (n_u - 1) / 2 over the rows this rank builds */
    double ms_prepare;         /* HIP-event milliseconds on the context's stream, per phase */
    double ms_cooc;            /* co-rating row kernel (RM2: dense M build; item-sim: whole build) */
    double ms_score;           /* RM2 scoring kernel, summed over launches */
    double ms_topn;
    double ms_total;
    int64_t score_launches;    /* launches of the scoring kernels */
    int64_t cooc_launches;
    int64_t blocks_total;        /* RM2 branch and bound: (user, 256-column block) pairs behind the seed columns ... */
    int64_t blocks_survived;     /* ... and how many of them had to be scored exactly */
    int64_t log_terms_evaluated; /* log terms actually evaluated (seed + bound + survivor passes); 0 = no pruning: log_terms */
    int64_t prune_fallbacks;     /* user batches whose bound did not bite (e.g. lambda = 0) and that were redone with the full pass */
    double ms_tables;            /* RM2: p(i|C), per-rating values, packed CSR, chunk offsets and segment tables (between prepare and the M build) */
    double ms_mirror;            /* RM2: mirror pass of the symmetric walk (lower triangle of the co-rating matrix + its block maxima) */
    int64_t topn_select_users;   /* users whose list needed the radix-select fallback of the top-N kernel (more than 2048 candidates reached the lower bound) */
    int64_t panel_clusters;      /* clusters built in column-panel mode (many clusters: only the popular columns of the co-rating matrix are stored) */
    int64_t stray_blocks;        /* panel mode: surviving (user, block) pairs behind the panel, scored exactly from the sparse data */
    int64_t bound_repairs;       /* panel mode: 64-column sub-blocks dropped by the second bound (without the user's own co-ratings) */
    int64_t isim_candidates;     /* item similarity, symmetric build: candidates the band sweep appended to the rows' lists (of I^2 / 2 elements x 2 rows) */
    int64_t isim_redone_rows;    /* ... rows whose list overflowed and were redone exactly from the matrix */
    int64_t prepared_from_cache; /* RM2: 1 = fy_rm2_prepare found its structures on the fy_ratings object (same clustering): nothing was sorted */
    int64_t tables_from_cache;   /* RM2: 1 = the row kernel's tables (packed CSR, segment tables) were re-used */
    /* (ABI 4) what the row kernels of the job read and write besides the packed CSR entries, for the kernel's OWN byte model
     * (bench.py roofline.frac_own_bytes) and its LDS-atomic floor (one ds_add wave instruction per segment): */
    int64_t cooc_segments;       /* <= 64-entry segments in the job's segment tables (12 B of descriptor each) */
    int64_t cooc_matrix_bytes;   /* bytes of co-rating matrix / panel / block-bound rows the row kernels store */
    int64_t rows_refined;        /* list rows whose score nearly cancels (|score| < c sqrt(n)) and that were scored again in fp64 from the fp32 head rows */
} fy_stats;
int fy_result_stats(fy_result*, fy_stats* out);

/* ------------------------------------------------------------------ NMF / PPC factorisation (produces the H of fy_cluster_assign)
 * Replaces NMFDriver / PPCDriver (M/nmf/AbstractNMFDriver.java:88-150): numberOfIterations rounds of the multiplicative
 * updates, each computing H2 and W2 from the same old (H, W) -- ComputeHJob + HComputationReducer.java:57-75 (or
 * PPCComputeHJob + PPCHComputationReducer.java:61-96) and ComputeWJob + WComputationMapper.java:100-118 -- in fp64 with
 * eps = 1e-12 (MatrixComputationJob.java:41).  Only ratings with score > 0 enter (VectorByItemHDFSMapper.java:37-40).
 * H: number_of_users x k, W: number_of_items x k doubles, row-major, HOST memory, updated in place; row r belongs to id
 * r + 1 (the H / W files are keyed from 1, DataInitialization.createMatrix).  A user (item) in [1, n] without a kept rating
 * fails like the reference's NoSuchElementException ("User %d has not rated any item" / "Item %d has not been rated by
 * anybody"); an id outside the ranges fails too (the reference would hit a null vector). */
typedef struct {
    int32_t number_of_users;           /* "numberOfUsers" */
    int32_t number_of_items;           /* "numberOfItems" */
    int32_t number_of_clusters;        /* "numberOfClusters" = k (<= 256) */
    int32_t number_of_iterations;      /* "numberOfIterations" */
    int32_t ppc;                       /* 0 = NMFDriver, 1 = PPCDriver */
    int32_t normalization_frequency;   /* PPC: the rows of H are L1-normalised when iteration % f == 0 (Java's %, so the
                                          reference's unset key, -1, normalises every iteration); 0 = never */
} fy_nmf_params;
/* stats: nnz / n_users / n_items, ms_prepare (the sorted copies of the ratings), ms_cooc = the iterations alone (H / W resident in
 * HBM), ms_total = everything including the host transfers of H and W in both directions. */
int fy_nmf_factorize(fy_context*, const fy_nmf_params*, const fy_ratings*, double* H_inout, double* W_inout, fy_stats* stats_or_null);

/* ------------------------------------------------------------------ the Hadoop files on either side of the RM2 job
 * (csrc/fy_seqfile.cpp).  Hadoop 1.2.1 SequenceFile, version 6, uncompressed record format, as the
 * reference's jobs and fixture writers produce it (M/util/DataInitialization.java:155-222, M/rm/RM2HDFSReducer.java:44-50,
 * M/rm/RM2Job.java:110-205).  `path` may be a file, a job output directory (every part file, hidden files skipped) or a
 * MapFile directory.  Readers return malloc'ed arrays: release them with fy_buffer_free.  Host-only: no GPU is touched.
 * PARITY UNPINNED at the byte level (the reference holds no binary fixture); IntPairWritable is Mahout 0.8's, restated as two
 * big-endian int32 -- see the header of csrc/fy_seqfile.cpp. */
int fy_seqfile_read_int_int(const char* path, int64_t* n, int32_t** key, int32_t** value);           /* clustering, clusteringCount */
int fy_seqfile_read_int_double(const char* path, int64_t* n, int32_t** key, double** value);         /* rm2/userSum, rm2/itemColl */
int fy_seqfile_read_intpair_float(const char* path, int64_t* n, int32_t** first, int32_t** second, float** value);   /* ratings, recommendations */
int fy_seqfile_write_int_int(const char* file, int64_t n, const int32_t* key, const int32_t* value);
int fy_seqfile_write_int_double(const char* file, int64_t n, const int32_t* key, const double* value);
int fy_seqfile_write_intpair_float(const char* file, int64_t n, const int32_t* first, const int32_t* second, const float* value);
int fy_mapfile_write_int_double(const char* dir, int64_t n, const int32_t* key, const double* value);   /* rm2/itemColl: data + index */
void fy_buffer_free(void*);

#ifdef __cplusplus
}
#endif
#endif /* FILMYOU_H */

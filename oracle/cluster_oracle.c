/* cluster_oracle.c -- CPU restatement of the cluster-assignment stage that feeds the RM2 job.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the product path never links this.
 *
 * Follows  M/nmf/clustering/FindClusterMapper.java:37-45      cluster(user j) = h_j.maxValueIndex()
 *          M/nmf/clustering/FindSubClusterMapper.java:46-76   cluster = parent * ceil(numberOfUsers / numberOfClusters) + argmax
 *          M/nmf/clustering/CountReducer.java:31-45           clusteringCount[c] = number of users routed to c
 * (M = /root/reference/src/main/java/es/udc/fi/dc/irlab).  Pinned by the reference's own vectors: ClusteringTestData.H ->
 * clustering / clusteringCount and SubClusteringTestData.H0/H1 -> clustering (tests/golden/clustering_test_data.json,
 * tests/test_oracle_golden.py).  Vector.maxValueIndex() itself is Mahout 0.8 (not in the tree): restated as "first index of
 * the strictly largest value, -1 if no value is greater than -infinity", which is what the fixtures exercise (positive rows). */
#include <math.h>
#include <stdint.h>

#include "oracle.h"

int clo_assign(int32_t n_rows, int32_t k, const double* H, int32_t first_user, int32_t cluster_offset, int32_t* user, int32_t* cluster) {
    if (n_rows < 0 || k < 0 || (n_rows > 0 && (!H || !user || !cluster))) return -1;
    for (int32_t r = 0; r < n_rows; r++) {
        int32_t best = -1;
        double max = -INFINITY;
        for (int32_t j = 0; j < k; j++) {
            const double v = H[(int64_t)r * k + j];
            if (v > max) { max = v; best = j; }
        }
        user[r] = first_user + r;                       /* DataInitialization.createDoubleMatrix: key = row + start */
        cluster[r] = best < 0 ? -1 : cluster_offset + best;
    }
    return 0;
}

int clo_count(int64_t n, const int32_t* cluster, int32_t n_clusters, int32_t* count) {
    for (int32_t c = 0; c < n_clusters; c++) count[c] = 0;
    for (int64_t i = 0; i < n; i++) {
        if (cluster[i] < 0 || cluster[i] >= n_clusters) return -1;
        count[cluster[i]]++;
    }
    return 0;
}

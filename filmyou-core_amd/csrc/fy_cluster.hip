// fy_cluster.hip -- cluster assignment: argmax over the rows of the factor matrix H + the per-cluster user count.
//
// Replaces (first half):
//   M/nmf/clustering/FindClusterMapper.java:37-45       emit (j, argmax h_j)
//   M/nmf/clustering/FindSubClusterMapper.java:46-76    emit (j, parent * ceil(numberOfUsers / numberOfClusters) + argmax h_j)
//   M/nmf/clustering/CountReducer.java:31-45            clusteringCount
// HBM-bound: 8 k bytes read per user; one thread per row for the small k of the reference (5 .. 50 clusters), one wave per row
// beyond 64 columns.
#include <algorithm>
#include <vector>

#include "fy_common.hpp"
#include "fy_rm2.hpp"

namespace fy {

// first index of the strictly largest value (Vector.maxValueIndex()): NaN never wins, -1 if nothing exceeds -infinity
__global__ void k_argmax_rows(int32_t n_rows, int32_t k, const double* __restrict__ H, int32_t first_user, int32_t offset,
                              int32_t n_clusters, int32_t* __restrict__ user, int32_t* __restrict__ cluster,
                              int32_t* __restrict__ count, int* __restrict__ err) {
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += gridDim.x * blockDim.x) {
        const double* __restrict__ h = H + (int64_t)r * k;
        int32_t best = -1;
        double mx = -INFINITY;
        for (int32_t j = 0; j < k; j++) {
            const double v = h[j];
            if (v > mx) { mx = v; best = j; }
        }
        const int32_t c = best < 0 ? -1 : offset + best;
        user[r] = first_user + r;
        cluster[r] = c;
        if (count) {
            if (c < 0 || c >= n_clusters) atomicOr(err, 1);
            else atomicAdd(&count[c], 1);
        }
    }
}

// wide rows: one wave per row, lanes stride the columns; ties resolved towards the smaller index
__global__ void k_argmax_rows_wide(int32_t n_rows, int32_t k, const double* __restrict__ H, int32_t first_user, int32_t offset,
                                   int32_t n_clusters, int32_t* __restrict__ user, int32_t* __restrict__ cluster,
                                   int32_t* __restrict__ count, int* __restrict__ err) {
    const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    for (int32_t r = blockIdx.x * wpb + (threadIdx.x >> 6); r < n_rows; r += gridDim.x * wpb) {
        const double* __restrict__ h = H + (int64_t)r * k;
        int32_t best = -1;
        double mx = -INFINITY;
        for (int32_t j = lane; j < k; j += 64) {
            const double v = h[j];
            if (v > mx) { mx = v; best = j; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double om = __shfl_xor(mx, o, 64);
            const int32_t ob = __shfl_xor(best, o, 64);
            if (ob >= 0 && (om > mx || (om == mx && (best < 0 || ob < best)))) { mx = om; best = ob; }
        }
        if (lane == 0) {
            const int32_t c = best < 0 ? -1 : offset + best;
            user[r] = first_user + r;
            cluster[r] = c;
            if (count) {
                if (c < 0 || c >= n_clusters) atomicOr(err, 1);
                else atomicAdd(&count[c], 1);
            }
        }
    }
}

void cluster_assign(Context* ctx, int32_t n_rows, int32_t k, const double* H, int location, int32_t first_user, int32_t cluster_offset,
                    int32_t n_clusters, int32_t* user_out, int32_t* cluster_out, int32_t* count_inout) {
    if (n_rows < 0 || k < 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "n_rows and k must be >= 0");
    if (n_rows > 0 && (!H || !user_out || !cluster_out)) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "H / output arrays are NULL");
    if (count_inout && n_clusters <= 0) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "n_clusters must be > 0 when counts are requested");
    if (location != FY_HOST && location != FY_DEVICE) FY_FAIL(FY_ERR_INVALID_ARGUMENT, "location must be FY_HOST or FY_DEVICE");
    if (n_rows == 0) return;
    hipStream_t st = ctx->stream;
    DevBuf<double> dH;
    const double* h = H;
    if (location == FY_HOST) {
        dH.alloc(ctx, (size_t)n_rows * std::max(1, k));
        h2d(ctx, dH.get(), H, (size_t)n_rows * k);
        h = dH.get();
    }
    DevBuf<int32_t> d_user(ctx, (size_t)n_rows), d_cluster(ctx, (size_t)n_rows), d_count(ctx, (size_t)std::max(1, n_clusters));
    DevBuf<int> err(ctx, 1);
    err.zero();
    d_count.zero();
    int32_t* cnt = count_inout ? d_count.get() : nullptr;
    const int grid_thread = (int)std::max<int64_t>(1, std::min<int64_t>(4096, ((int64_t)n_rows + 255) / 256));
    const int grid_wave = (int)std::max<int64_t>(1, std::min<int64_t>(4096, ((int64_t)n_rows + 3) / 4));
    if (k <= 64) k_argmax_rows<<<grid_thread, 256, 0, st>>>(n_rows, k, h, first_user, cluster_offset, n_clusters, d_user.get(), d_cluster.get(), cnt, err.get());
    else k_argmax_rows_wide<<<grid_wave, 256, 0, st>>>(n_rows, k, h, first_user, cluster_offset, n_clusters, d_user.get(), d_cluster.get(), cnt, err.get());
    FY_KERNEL_CHECK();
    d2h(ctx, user_out, d_user.get(), (size_t)n_rows);
    d2h(ctx, cluster_out, d_cluster.get(), (size_t)n_rows);
    std::vector<int32_t> hc;
    if (count_inout) {
        hc.resize((size_t)n_clusters);
        d2h(ctx, hc.data(), d_count.get(), (size_t)n_clusters);
    }
    int herr = 0;
    d2h(ctx, &herr, err.get(), 1);
    sync(ctx);
    if (herr) FY_FAIL(FY_ERR_CLUSTER_RANGE, "a user is routed to a cluster outside [0, %d)", n_clusters);
    if (count_inout)
        for (int32_t c = 0; c < n_clusters; c++) count_inout[c] += hc[c];
}

}  // namespace fy

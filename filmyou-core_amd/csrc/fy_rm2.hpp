// fy_rm2.hpp -- internal interface between the C ABI (fy_api.hip) and the job implementations.
#pragma once
#include <memory>

#include "fy_common.hpp"

// result rows live in HBM until an accessor asks for them
struct fy_result {
    fy::Context* ctx = nullptr;
    int kind = 0;   // 0 = RM2, 1 = item-sim
    int64_t n = 0;
    fy::DevBuf<int32_t> d_key0, d_key1, d_aux;
    fy::DevBuf<float> d_value;
    fy::DevBuf<int32_t> d_user_id, d_item_id;
    fy::DevBuf<double> d_user_sum, d_icoll;
    double total_sum = 0.0;
    fy_stats st{};
    // host mirrors, filled on first access
    bool rows_on_host = false, sums_on_host = false;
    std::vector<int32_t> h_key0, h_key1, h_aux, h_user_id, h_item_id;
    std::vector<float> h_value;
    std::vector<double> h_user_sum, h_icoll;
};

namespace fy {
fy_rm2_job* rm2_prepare(Context*, const fy_rm2_params*, const fy_ratings*, int64_t n_map, const int32_t* map_user,
                        const int32_t* map_cluster, const int32_t* cluster_count);
void rm2_partial_stats(fy_rm2_job*, double** buf, int64_t* len);
void rm2_stats_layout(fy_rm2_job*, int64_t* n_item_slots, int64_t* n_user_slots);
void rm2_set_global_stats(fy_rm2_job*, const double* gathered, int32_t world);
void rm2_set_collectives(fy_rm2_job*, const fy_collectives*);
fy_result* rm2_score(fy_rm2_job*);
void rm2_job_destroy(fy_rm2_job*);
fy_result* itemsim_build(Context*, const fy_itemsim_params*, const fy_ratings*);
struct Prepared;
// upper triangle of the weighted co-rating Gram of a one-cluster structure as fp32 (fy_rm2.hip; used by fy_itemsim.hip)
bool gram_half_build(Context* ctx, const Prepared& P, const float* csc_w, const float* bounds3, float* G, int64_t ldm, double* ms_tables,
                     double* ms_walk);
void cluster_assign(Context*, int32_t n_rows, int32_t k, const double* H, int location, int32_t first_user, int32_t cluster_offset,
                    int32_t n_clusters, int32_t* user_out, int32_t* cluster_out, int32_t* count_inout);
void nmf_factorize(Context*, const fy_nmf_params*, const fy_ratings*, double* H, double* W, fy_stats* st);
fy_result* itemcf_recommend(Context*, const fy_itemcf_params*, const fy_ratings*, fy_result* similarities);

// top-N over rows of a dense score matrix (NaN = not a candidate): k_topn_fast + k_topn_select of fy_rm2.hip.
// n_out[u] rows are written at out_off[u] for u in [0, n_rows); item ids come from rank_item_raw[column].
void launch_topn_rows(Context* ctx, hipStream_t st, const float* S, int64_t ldS, int32_t n_cols, int32_t n_rows,
                      const int32_t* n_out, const int32_t* out_off, const int32_t* rank_item_raw, const int32_t* slot2du,
                      const int32_t* uid, int32_t slot0, int32_t aux_value, int32_t* out_user, int32_t* out_item,
                      float* out_score, int32_t* out_aux, int32_t* overflow, int32_t* any_overflow, int32_t top_n_hint = 0);
}  // namespace fy

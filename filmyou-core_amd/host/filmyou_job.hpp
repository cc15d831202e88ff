// filmyou_job.hpp -- C++ host side of the drop-in (header only, C++17), a mirror of the reference's job interface over
// the C ABI of include/filmyou.h.  The reference's host language is Java; this image has no JVM, so the host layer is
// written in C++ (and mirrored in Python, filmyou-core_amd/host.py).  Names, argument meaning and error behaviour
// follow the reference:
//   fy::host::Configuration   <- org.apache.hadoop.conf.Configuration with the option names of
//                                M/rmrecommender/RMRecommenderDriver.java:49-120 (string values, typed getters)
//   fy::host::RM2Job::run     <- es.udc.fi.dc.irlab.rm.RM2Job.run (M/rm/RM2Job.java:76-100): returns 0 on success,
//                                throws std::runtime_error("RM2 failed!: ...") like RM2Job.java:144-147
//   fy::host::RowSimilarityJob::run <- Mahout RowSimilarityJob as invoked at
//                                M/baselinerecommender/BaselineRecommenderJob.java:241-253 (same option names)
// Input/output stay in the caller's hands (the reference's Cassandra / HDFS readers and writers are unchanged): the
// job takes rating triples and hands back rows through a sink callback shaped like writePreference
// (M/rm/AbstractRM2Reducer.java:404-406).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/filmyou.h"

namespace fy {
namespace host {

class Configuration {
   public:
    void set(const std::string& k, const std::string& v) { kv_[k] = v; }
    void setInt(const std::string& k, long v) { kv_[k] = std::to_string(v); }
    void setBoolean(const std::string& k, bool v) { kv_[k] = v ? "true" : "false"; }
    bool has(const std::string& k) const { return kv_.count(k) || defaults().count(k); }
    std::string get(const std::string& k, const std::string& dflt = "") const {
        auto it = kv_.find(k);
        if (it != kv_.end()) return it->second;
        auto d = defaults().find(k);
        return d != defaults().end() ? d->second : dflt;
    }
    long getInt(const std::string& k, long dflt) const { return has(k) ? std::stol(get(k)) : dflt; }
    double getDouble(const std::string& k, double dflt) const { return has(k) ? std::stod(get(k)) : dflt; }
    bool getBoolean(const std::string& k, bool dflt) const { return has(k) ? get(k) == "true" : dflt; }

   private:
    // RMRecommenderDriver.loadDefaultSetup (M/rmrecommender/RMRecommenderDriver.java:89-120)
    static const std::map<std::string, std::string>& defaults() {
        static const std::map<std::string, std::string> d = {
            {"lambda", "0.1"}, {"numberOfRecommendations", "1000"}, {"clusterSplit", "400"}, {"splitSize", "100"},
            {"filterUsers", "0"}, {"directory", "recommendation"}, {"clustering", "clustering"},
            {"clusteringCount", "clusteringCount"}};
        return d;
    }
    std::map<std::string, std::string> kv_;
};

struct Ratings {   // what the reference's mappers receive: (user, item, score) records
    std::vector<int32_t> user, item;
    std::vector<float> score;
    void add(int32_t u, int32_t i, float s) { user.push_back(u); item.push_back(i); score.push_back(s); }
};

struct Clustering {   // the `clustering` and `clusteringCount` side files
    std::vector<int32_t> user, cluster;
    std::vector<int32_t> count;   // optional, numberOfClusters entries
};

// writePreference(context, userId, itemId, score, cluster)
using PreferenceSink = std::function<void(int32_t user, int32_t item, float score, int32_t cluster)>;

class RM2Job {
   public:
    explicit RM2Job(const Configuration& conf) : conf_(conf) {}

    // rm2/userSum and rm2/itemColl of the last run (what TestHDFSRM2.java:70-71 asserts)
    std::vector<int32_t> userSumKeys, itemCollKeys;
    std::vector<double> userSum, itemColl;
    double totalSum = 0.0;
    fy_stats stats{};

    int run(const Ratings& r, const Clustering& c, const PreferenceSink& sink) {
        fy_rm2_params p{};
        p.lambda = std::stod(conf_.get("lambda"));   // Double.valueOf(conf.get("lambda")), AbstractRM2Reducer.java:108
        p.number_of_items = (int32_t)conf_.getInt("numberOfItems", -1);
        p.number_of_clusters = (int32_t)conf_.getInt("numberOfClusters", -1);
        if (p.number_of_items <= 0 || p.number_of_clusters <= 0)
            throw std::invalid_argument("numberOfItems and numberOfClusters are required");
        p.number_of_recommendations = (int32_t)conf_.getInt("numberOfRecommendations", 1000);
        p.filter_users = (int32_t)conf_.getInt("filterUsers", 0);
        p.rank = 0;
        p.world = 1;
        std::vector<int32_t> cc;
        if (!c.count.empty()) {
            cc.assign((size_t)p.number_of_clusters, 0);
            for (size_t k = 0; k < c.count.size() && k < cc.size(); k++) cc[k] = c.count[k];
        }
        fy_result* res = nullptr;
        const int rc = fy_rm2_run(&p, (int64_t)r.user.size(), r.user.data(), r.item.data(), r.score.data(),
                                  (int64_t)c.user.size(), c.user.data(), c.cluster.data(), cc.empty() ? nullptr : cc.data(), &res);
        if (rc != FY_OK) throw std::runtime_error(std::string("RM2 failed!: ") + fy_last_error());
        const int64_t n = fy_result_size(res);
        const int32_t *u = fy_result_key0(res), *i = fy_result_key1(res), *cl = fy_result_aux(res);
        const float* s = fy_result_value(res);
        for (int64_t k = 0; k < n; k++) sink(u[k], i[k], s[k], cl[k]);
        const int64_t nu = fy_result_n_users(res), ni = fy_result_n_items(res);
        userSumKeys.assign(fy_result_user_id(res), fy_result_user_id(res) + nu);
        userSum.assign(fy_result_user_sum(res), fy_result_user_sum(res) + nu);
        itemCollKeys.assign(fy_result_item_id(res), fy_result_item_id(res) + ni);
        itemColl.assign(fy_result_item_coll(res), fy_result_item_coll(res) + ni);
        totalSum = fy_result_total_sum(res);
        fy_result_stats(res, &stats);
        fy_result_free(res);
        return 0;
    }

   private:
    Configuration conf_;
};

class RowSimilarityJob {
   public:
    using SimilaritySink = std::function<void(int32_t item, int32_t other, float similarity)>;
    // args as passed by BaselineRecommenderJob: --similarityClassname, --maxSimilaritiesPerRow,
    // --excludeSelfSimilarity, --threshold
    int run(const Ratings& r, const std::string& similarityClassname, int maxSimilaritiesPerRow, bool excludeSelfSimilarity,
            const double* threshold, const SimilaritySink& sink) {
        fy_itemsim_params p{};
        if (similarityClassname == "SIMILARITY_COSINE") p.similarity = FY_SIMILARITY_COSINE;
        else if (similarityClassname == "SIMILARITY_COOCCURRENCE") p.similarity = FY_SIMILARITY_COOCCURRENCE;
        else throw std::invalid_argument("similarityClassname must be SIMILARITY_COSINE or SIMILARITY_COOCCURRENCE");
        p.max_similarities_per_item = maxSimilaritiesPerRow;
        p.exclude_self = excludeSelfSimilarity ? 1 : 0;
        p.has_threshold = threshold ? 1 : 0;
        p.threshold = threshold ? *threshold : 0.0;
        p.rank = 0;
        p.world = 1;
        fy_result* res = nullptr;
        const int rc = fy_itemsim_run(&p, (int64_t)r.user.size(), r.user.data(), r.item.data(), r.score.data(), &res);
        if (rc != FY_OK) throw std::runtime_error(std::string("RowSimilarityJob failed!: ") + fy_last_error());
        const int64_t n = fy_result_size(res);
        const int32_t *a = fy_result_key0(res), *b = fy_result_key1(res);
        const float* s = fy_result_value(res);
        for (int64_t k = 0; k < n; k++) sink(a[k], b[k], s[k]);
        fy_result_free(res);
        return 0;
    }
};

}  // namespace host
}  // namespace fy

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

    // One rank of a multi-GPU job (one process per GPU): the staged entry points with the library's compiled RCCL transport
    // (fy_rccl_*, csrc/fy_rccl.hip).  `rccl_id` = the 128 bytes rank 0 got from fy_rccl_unique_id and handed to the other
    // ranks through the host's own channel (a Hadoop Configuration entry, a file, ...).  Emits this rank's users only.
    int runRank(const Ratings& r, const Clustering& c, const PreferenceSink& sink, int device, int rank, int world, const char* rccl_id) {
        fy_rm2_params p = params();
        p.rank = rank;
        p.world = world;
        fy_context* ctx = nullptr;
        fy_ratings* rt = nullptr;
        fy_rm2_job* job = nullptr;
        fy_rccl* comm = nullptr;
        fy_result* res = nullptr;
        auto fail = [&](const char* what) {
            const std::string msg = std::string("RM2 failed!: ") + what + ": " + fy_last_error();
            if (job) fy_rm2_job_destroy(job);
            if (comm) fy_rccl_destroy(comm);
            if (rt) fy_ratings_destroy(rt);
            if (ctx) fy_context_destroy(ctx);
            throw std::runtime_error(msg);
        };
        std::vector<int32_t> cc = counts(c, p.number_of_clusters);
        if (fy_context_create(device, &ctx) != FY_OK) fail("context");
        if (fy_ratings_create(ctx, (int64_t)r.user.size(), r.user.data(), r.item.data(), r.score.data(), FY_HOST, &rt) != FY_OK) fail("ratings");
        if (fy_rm2_prepare(ctx, &p, rt, (int64_t)c.user.size(), c.user.data(), c.cluster.data(), cc.empty() ? nullptr : cc.data(), &job) != FY_OK) fail("prepare");
        if (fy_rccl_create(ctx, rank, world, rccl_id, &comm) != FY_OK) fail("rccl");
        fy_collectives coll{};
        if (fy_rccl_collectives(comm, &coll) != FY_OK || fy_rm2_set_collectives(job, &coll) != FY_OK) fail("collectives");
        if (fy_rm2_score(job, &res) != FY_OK) fail("score");
        collect(res, sink);
        fy_rccl_counters(comm, &allGathers, &reduceScatters, &collectiveBytes);
        fy_result_free(res);
        fy_rm2_job_destroy(job);
        fy_rccl_destroy(comm);
        fy_ratings_destroy(rt);
        fy_context_destroy(ctx);
        return 0;
    }
    int64_t allGathers = 0, reduceScatters = 0, collectiveBytes = 0;   // of the last runRank

    // RM2Job.run on the reference's own files (M/rm/RM2Job.java:76-100 with useCassandraInput / Output = false): ratings from
    // <mapred.input.dir>, <directory>/<clustering>, <directory>/<clusteringCount>; writes <directory>/rm2/userSum,
    // <directory>/rm2/itemColl (MapFile) and the recommendations under <mapred.output.dir>.  Files: fy_seqfile_* (layout
    // parity unpinned, see csrc/fy_seqfile.cpp).  <directory>/rm2 is NOT wiped here (no filesystem walk in this header).
    int runFiles() {
        const std::string in = conf_.get("mapred.input.dir"), out = conf_.get("mapred.output.dir"), base = conf_.get("directory");
        if (in.empty() || out.empty()) throw std::invalid_argument("mapred.input.dir and mapred.output.dir are required");
        Ratings r;
        Clustering c;
        int64_t n = 0;
        int32_t *a = nullptr, *b = nullptr;
        float* v = nullptr;
        if (fy_seqfile_read_intpair_float(in.c_str(), &n, &a, &b, &v) != FY_OK) throw std::runtime_error(std::string("RM2 failed!: ") + fy_last_error());
        r.user.assign(a, a + n); r.item.assign(b, b + n); r.score.assign(v, v + n);
        fy_buffer_free(a); fy_buffer_free(b); fy_buffer_free(v);
        if (fy_seqfile_read_int_int((base + "/" + conf_.get("clustering")).c_str(), &n, &a, &b) != FY_OK)
            throw std::runtime_error(std::string("RM2 failed!: ") + fy_last_error());
        c.user.assign(a, a + n); c.cluster.assign(b, b + n);
        fy_buffer_free(a); fy_buffer_free(b);
        if (fy_seqfile_read_int_int((base + "/" + conf_.get("clusteringCount")).c_str(), &n, &a, &b) != FY_OK)
            throw std::runtime_error(std::string("RM2 failed!: ") + fy_last_error());
        const long K = conf_.getInt("numberOfClusters", -1);
        c.count.assign((size_t)(K > 0 ? K : 1), 0);
        for (int64_t k = 0; k < n; k++)
            if (a[k] >= 0 && a[k] < (int32_t)c.count.size()) c.count[(size_t)a[k]] = b[k];
        fy_buffer_free(a); fy_buffer_free(b);
        std::vector<int32_t> ou, oi;
        std::vector<float> os;
        run(r, c, [&](int32_t user, int32_t item, float score, int32_t) { ou.push_back(user); oi.push_back(item); os.push_back(score); });
        if (fy_seqfile_write_int_double((base + "/rm2/userSum/part-r-00000").c_str(), (int64_t)userSum.size(), userSumKeys.data(), userSum.data()) != FY_OK ||
            fy_mapfile_write_int_double((base + "/rm2/itemColl/part-r-00000").c_str(), (int64_t)itemColl.size(), itemCollKeys.data(), itemColl.data()) != FY_OK ||
            fy_seqfile_write_intpair_float((out + "/part-r-00000").c_str(), (int64_t)ou.size(), ou.data(), oi.data(), os.data()) != FY_OK)
            throw std::runtime_error(std::string("RM2 failed!: ") + fy_last_error());
        return 0;
    }

   private:
    fy_rm2_params params() const {
        fy_rm2_params p{};
        p.lambda = std::stod(conf_.get("lambda"));
        p.number_of_items = (int32_t)conf_.getInt("numberOfItems", -1);
        p.number_of_clusters = (int32_t)conf_.getInt("numberOfClusters", -1);
        if (p.number_of_items <= 0 || p.number_of_clusters <= 0) throw std::invalid_argument("numberOfItems and numberOfClusters are required");
        p.number_of_recommendations = (int32_t)conf_.getInt("numberOfRecommendations", 1000);
        p.filter_users = (int32_t)conf_.getInt("filterUsers", 0);
        p.world = 1;
        return p;
    }
    static std::vector<int32_t> counts(const Clustering& c, int32_t K) {
        std::vector<int32_t> cc;
        if (!c.count.empty()) {
            cc.assign((size_t)K, 0);
            for (size_t k = 0; k < c.count.size() && k < cc.size(); k++) cc[k] = c.count[k];
        }
        return cc;
    }
    void collect(fy_result* res, const PreferenceSink& sink) {
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
    }
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

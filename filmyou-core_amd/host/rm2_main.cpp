// rm2_main.cpp -- a tiny C++ driver over fy::host::RM2Job, used by tests/test_cpp_host_gpu.py.
//   rm2_main <ratings.txt> <clustering.txt|-> <lambda> <numberOfItems> <numberOfClusters> <numberOfRecommendations>
// ratings.txt: "user item score" per line; clustering.txt: "user cluster" per line.  Prints "user item score cluster".
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "filmyou_job.hpp"

// rm2_main --files <mapred.input.dir> <mapred.output.dir> <directory> <lambda> <numberOfItems> <numberOfClusters> <numberOfRecommendations>
//   : RM2Job::runFiles on the reference's SequenceFile layout.
// rm2_main --rccl <ratings.txt> <clustering.txt|-> <lambda> <numberOfItems> <numberOfClusters> <numberOfRecommendations>
//   : RM2Job::runRank as rank 0 of a world of 1 through the compiled RCCL transport (fy_rccl_*).
static int main_files(char** a) {
    fy::host::Configuration conf;
    conf.set("mapred.input.dir", a[0]); conf.set("mapred.output.dir", a[1]); conf.set("directory", a[2]);
    conf.set("lambda", a[3]); conf.set("numberOfItems", a[4]); conf.set("numberOfClusters", a[5]); conf.set("numberOfRecommendations", a[6]);
    try {
        fy::host::RM2Job job(conf);
        job.runFiles();
        fprintf(stderr, "totalSum %.17g recs %lld\n", job.totalSum, (long long)job.stats.recs);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 9 && strcmp(argv[1], "--files") == 0) return main_files(argv + 2);
    bool rccl = false;
    if (argc == 8 && strcmp(argv[1], "--rccl") == 0) { rccl = true; argv++; argc--; }
    if (argc != 7) { fprintf(stderr, "usage: %s ratings clustering lambda numberOfItems numberOfClusters numberOfRecommendations\n", argv[0]); return 2; }
    fy::host::Ratings r;
    fy::host::Clustering c;
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    int u, i; float s;
    while (fscanf(f, "%d %d %f", &u, &i, &s) == 3) r.add(u, i, s);
    fclose(f);
    if (strcmp(argv[2], "-") != 0) {
        f = fopen(argv[2], "r");
        if (!f) { perror(argv[2]); return 2; }
        int cl;
        while (fscanf(f, "%d %d", &u, &cl) == 2) { c.user.push_back(u); c.cluster.push_back(cl); }
        fclose(f);
    }
    fy::host::Configuration conf;
    conf.set("lambda", argv[3]);
    conf.set("numberOfItems", argv[4]);
    conf.set("numberOfClusters", argv[5]);
    conf.set("numberOfRecommendations", argv[6]);
    try {
        fy::host::RM2Job job(conf);
        auto sink = [](int32_t user, int32_t item, float score, int32_t cluster) { printf("%d %d %.9g %d\n", user, item, score, cluster); };
        if (rccl) {
            char id[128];
            if (fy_rccl_unique_id(id) != FY_OK) { fprintf(stderr, "RM2 failed!: %s\n", fy_last_error()); return 1; }
            job.runRank(r, c, sink, 0, 0, 1, id);
            fprintf(stderr, "rccl all_gathers %lld reduce_scatters %lld bytes %lld\n", (long long)job.allGathers, (long long)job.reduceScatters, (long long)job.collectiveBytes);
        } else {
            job.run(r, c, sink);
        }
        fprintf(stderr, "totalSum %.17g users %zu items %zu recs %lld\n", job.totalSum, job.userSum.size(), job.itemColl.size(), (long long)job.stats.recs);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}

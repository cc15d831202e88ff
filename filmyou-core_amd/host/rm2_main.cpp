// rm2_main.cpp -- a tiny C++ driver over fy::host::RM2Job, used by tests/test_cpp_host_gpu.py.
//   rm2_main <ratings.txt> <clustering.txt|-> <lambda> <numberOfItems> <numberOfClusters> <numberOfRecommendations>
// ratings.txt: "user item score" per line; clustering.txt: "user cluster" per line.  Prints "user item score cluster".
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "filmyou_job.hpp"

int main(int argc, char** argv) {
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
        job.run(r, c, [](int32_t user, int32_t item, float score, int32_t cluster) { printf("%d %d %.9g %d\n", user, item, score, cluster); });
        fprintf(stderr, "totalSum %.17g users %zu items %zu recs %lld\n", job.totalSum, job.userSum.size(), job.itemColl.size(), (long long)job.stats.recs);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}

import sys, importlib, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
P = importlib.import_module("filmyou-core_amd"); S = importlib.import_module("filmyou-core_amd.synth")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
topn = int(sys.argv[2]) if len(sys.argv) > 2 else 50
u, i, s, facts = S.generate("ml25m", device="cuda:0")
uu = np.arange(1, facts["n_users"] + 1, dtype=np.int32)
cl = (uu, S.hash_clustering(uu, K)) if K > 1 else None
conf = P.Configuration(); conf.set("lambda", "0.1"); conf.setInt("numberOfItems", facts["n_items"]); conf.setInt("numberOfClusters", K); conf.setInt("numberOfRecommendations", topn)
ctx = P.Context(0); r = P.Ratings(ctx, u, i, s)
for rep in range(2):
    rec = P.RM2Job(conf, ctx).run(r, clustering=cl)
    st = rec.stats
    rec.close()
print(json.dumps(st))
print("survived frac", st["blocks_survived"] / max(1, st["blocks_total"]), "evaluated/ref terms", st["log_terms_evaluated"] / st["log_terms"])

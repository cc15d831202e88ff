import importlib, sys, time, torch
sys.path.insert(0, '/root/repo')
P = importlib.import_module("filmyou-core_amd"); S = importlib.import_module("filmyou-core_amd.synth")
u,i,s,f = S.generate("ml25m", device=torch.device("cuda",0))
ctx = P.Context(0); r = P.Ratings(ctx,u,i,s); job = P.RowSimilarityJob(ctx)
for ex in (True, 2, 3):
    for _ in range(2):
        res = job.run(r, maxSimilaritiesPerRow=100, excludeSelfSimilarity=ex); st = res.stats; res.close()
    print(ex, st["ms_cooc"], st["isim_candidates"])

"""filmyou-core_amd -- MI355X-native RM2 scorer and item-item similarity build behind filmyou-core's job interface.

Host-side mirror (Python, ctypes over the C ABI of include/filmyou.h) of the two reference entry points:

  RM2Job.run(conf, ...)                 <- es.udc.fi.dc.irlab.rm.RM2Job.run            (M/rm/RM2Job.java:76-100)
  RowSimilarityJob.run(args, ...)       <- org.apache.mahout...RowSimilarityJob as called at
                                           M/baselinerecommender/BaselineRecommenderJob.java:241-253

The directory name contains a hyphen (it is fixed by the project layout), so import it with
``importlib.import_module("filmyou-core_amd")`` or through the ``filmyou_core_amd`` alias module at the repo root.
All compute happens in libfilmyou_hip.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
from . import _native
from ._native import build, LIB_PATH
from .host import (BaselineRecommenderJob, ClusterAssignmentJob, Configuration, Context, FilmYouError, ItemRecommendations, ItemSimilarities, NMFDriver, PreparedRM2, Ratings, Recommendations, RM2Job,
                   RowSimilarityJob, SIMILARITY_COSINE, SIMILARITY_COOCCURRENCE)

__all__ = ["build", "BaselineRecommenderJob", "ClusterAssignmentJob", "ItemRecommendations", "NMFDriver", "LIB_PATH", "Configuration", "Context", "FilmYouError", "ItemSimilarities", "PreparedRM2", "Ratings",
           "Recommendations", "RM2Job", "RowSimilarityJob", "SIMILARITY_COSINE", "SIMILARITY_COOCCURRENCE", "_native"]

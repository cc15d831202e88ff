"""Host side of the drop-in: the reference's job interface re-stated over the C ABI.

Mirrors (same names, argument meaning, error behaviour):
  * Hadoop ``Configuration`` with the keys of M/rmrecommender/RMRecommenderDriver.java:49-120
  * ``RM2Job.run`` (M/rm/RM2Job.java:76-100): returns 0-equivalent (a result object) or raises
    ``RuntimeError("<job> failed!: ...")`` like RM2Job.java:144-147
  * ``RowSimilarityJob.run`` with the option names passed at M/baselinerecommender/BaselineRecommenderJob.java:241-253
The reference's Cassandra / HDFS readers and writers stay on the Java side (north_star: unchanged); this layer takes
the rating triples they produce and returns the rows they consume.
"""
import ctypes as C
import os

import numpy as np

from . import _native

SIMILARITY_COSINE = "SIMILARITY_COSINE"
SIMILARITY_COOCCURRENCE = "SIMILARITY_COOCCURRENCE"
_SIMILARITY = {SIMILARITY_COSINE: 0, SIMILARITY_COOCCURRENCE: 1}


class FilmYouError(RuntimeError):
    """A non-zero fy_status from the native library."""

    def __init__(self, code, message):
        super().__init__("fy_status %d: %s" % (code, message))
        self.code = code
        self.message = message


def _check(rc):
    if rc != 0:
        raise FilmYouError(rc, _native.load().fy_last_error().decode(errors="replace"))


class Configuration(dict):
    """The slice of org.apache.hadoop.conf.Configuration the jobs read: string values, typed getters with defaults."""

    # option names of RMRecommenderDriver (M/rmrecommender/RMRecommenderDriver.java:49-120) and their defaults
    DEFAULTS = {"lambda": "0.1", "numberOfRecommendations": "1000", "clusterSplit": "400", "splitSize": "100",
                "filterUsers": "0", "directory": "recommendation", "clustering": "clustering",
                "clusteringCount": "clusteringCount"}

    def set(self, key, value):
        self[key] = str(value)

    def setInt(self, key, value):
        self[key] = str(int(value))

    def setFloat(self, key, value):
        # Configuration.setFloat stores Float.toString(value): 0.5f -> "0.5" (quirk Q3)
        self[key] = str(np.float32(value))

    def setBoolean(self, key, value):
        self[key] = "true" if value else "false"

    def get(self, key, default=None):
        if key in self:
            return dict.get(self, key)
        return self.DEFAULTS.get(key, default)

    def getInt(self, key, default):
        v = self.get(key)
        return int(v) if v is not None else default

    def getDouble(self, key, default):
        v = self.get(key)
        return float(v) if v is not None else default

    def getBoolean(self, key, default):
        v = self.get(key)
        return (str(v).lower() == "true") if v is not None else default


class Context:
    """One GPU, one HIP stream (fy_context)."""

    def __init__(self, device=0):
        self._lib = _native.load()
        h = C.c_void_p()
        _check(self._lib.fy_context_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        self._tuning_env = self._fy_env()

    @staticmethod
    def _fy_env():
        return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("FY_")))

    def sync_tuning(self):
        """The library reads its FY_* knobs (test / measurement hooks) from the environment once, at fy_context_create.  This
        host mirror re-reads them (fy_context_reload_tuning) when the process environment has changed since -- the parity tests
        switch paths per test on one context; a production job never gets here with a changed environment."""
        env = self._fy_env()
        if env != self._tuning_env:
            _check(self._lib.fy_context_reload_tuning(self._h))
            self._tuning_env = env

    def synchronize(self):
        _check(self._lib.fy_context_synchronize(self._h))

    def inject_alloc_failure(self, nth):
        """Fault injection (tests): the nth HBM request from now fails with FY_ERR_OUT_OF_MEMORY; 0 disarms."""
        _check(self._lib.fy_context_inject_alloc_failure(self._h, int(nth)))

    @property
    def stream(self):
        return self._lib.fy_context_stream(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.fy_context_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _is_torch_tensor(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class Ratings:
    """Rating triples in HBM.  Accepts numpy arrays (copied over PCIe) or torch tensors already on the context's GPU."""

    def __init__(self, ctx, user, item, score):
        self._lib = _native.load()
        self.ctx = ctx
        if _is_torch_tensor(user):
            import torch
            assert user.is_cuda and item.is_cuda and score.is_cuda, "device tensors expected"
            assert user.dtype == torch.int32 and item.dtype == torch.int32 and score.dtype == torch.float32
            user, item, score = user.contiguous(), item.contiguous(), score.contiguous()
            torch.cuda.current_stream(user.device).synchronize()   # producer stream != the context's stream
            n, ptrs, loc = user.numel(), (user.data_ptr(), item.data_ptr(), score.data_ptr()), 1
        else:
            user = np.ascontiguousarray(user, dtype=np.int32)
            item = np.ascontiguousarray(item, dtype=np.int32)
            score = np.ascontiguousarray(score, dtype=np.float32)
            n, ptrs, loc = len(user), (user.ctypes.data, item.ctypes.data, score.ctypes.data), 0
        assert n == len(item) == len(score)
        self._keep = (user, item, score)
        h = C.c_void_p()
        _check(self._lib.fy_ratings_create(ctx._h, n, ptrs[0], ptrs[1], ptrs[2], loc, C.byref(h)))
        self._h = h
        self._keep = None
        self.nnz = n

    def drop_cache(self):
        """fy_ratings_drop_cache: releases what earlier jobs kept on this object (CSR / CSC, statistics, row-kernel tables)."""
        if getattr(self, "_h", None) and self.ctx._h:
            self._lib.fy_ratings_drop_cache(self._h)

    def close(self):
        if getattr(self, "_h", None):
            if self.ctx._h:
                self._lib.fy_ratings_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _np(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).copy()


class _Result:
    def __init__(self, handle, ctx=None):
        self._lib = _native.load()
        self._h = handle
        self._ctx = ctx   # rows are downloaded through the context's stream: keep it alive

    def _stats(self):
        st = _native.Stats()
        _check(self._lib.fy_result_stats(self._h, C.byref(st)))
        return st.as_dict()

    def close(self):
        if getattr(self, "_h", None):
            if self._ctx is None or self._ctx._h:   # a result must not outlive its context's stream
                self._lib.fy_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Recommendations(_Result):
    """Rows of the `recommendations` table / output SequenceFile: (user, item, relevance float32, cluster),
    grouped by user, best first -- plus the job's side outputs rm2/userSum and rm2/itemColl."""

    def __init__(self, handle, ctx=None):
        super().__init__(handle, ctx)
        L = self._lib
        self.stats = self._stats()
        self.size = L.fy_result_size(handle)
        self._rows = None
        self._sums = None

    def rows(self):
        if self._rows is None:
            L, h, n = self._lib, self._h, self.size
            self._rows = {"user": _np(L.fy_result_key0(h), n, np.int32), "item": _np(L.fy_result_key1(h), n, np.int32),
                          "score": _np(L.fy_result_value(h), n, np.float32),
                          "cluster": _np(L.fy_result_aux(h), n, np.int32)}
        return self._rows

    def sums(self):
        if self._sums is None:
            L, h = self._lib, self._h
            nu, ni = L.fy_result_n_users(h), L.fy_result_n_items(h)
            self._sums = {"user_id": _np(L.fy_result_user_id(h), nu, np.int32),
                          "user_sum": _np(L.fy_result_user_sum(h), nu, np.float64),
                          "item_id": _np(L.fy_result_item_id(h), ni, np.int32),
                          "item_coll": _np(L.fy_result_item_coll(h), ni, np.float64),
                          "total_sum": L.fy_result_total_sum(h)}
        return self._sums


class ItemSimilarities(_Result):
    """Rows of the similarity matrix: (item, other item, similarity), grouped by item, best first."""

    def __init__(self, handle, ctx=None):
        super().__init__(handle, ctx)
        self.stats = self._stats()
        self.size = self._lib.fy_result_size(handle)
        self._rows = None

    def rows(self):
        if self._rows is None:
            L, h, n = self._lib, self._h, self.size
            self._rows = {"item": _np(L.fy_result_key0(h), n, np.int32), "other": _np(L.fy_result_key1(h), n, np.int32),
                          "sim": _np(L.fy_result_value(h), n, np.float32)}
        return self._rows


def _i32(a):
    return np.ascontiguousarray(a if a is not None else [], dtype=np.int32)


class RM2Job:
    """Relevance Model 2 job (M/rm/RM2Job.java:54-272) on one MI355X (or one rank of several).

    ``conf`` carries the reference's keys: lambda, numberOfItems, numberOfClusters, numberOfRecommendations,
    filterUsers (clusterSplit / splitSize are accepted and ignored: they only partition the reference's reduce groups
    and never change a score -- M/common/AbstractByClusterAndCountMapper.java:86-102)."""

    JOB_NAME = "RM2"

    def __init__(self, conf, ctx=None):
        self.conf = conf
        self.ctx = ctx

    def _params(self, rank, world, workspace_bytes):
        conf = self.conf
        n_clusters = conf.getInt("numberOfClusters", -1)
        n_items = conf.getInt("numberOfItems", -1)
        if n_clusters is None or n_clusters <= 0 or n_items is None or n_items <= 0:
            # AbstractJob.parseArguments rejects a missing required option (TestRMRecommenderJob.java:39-74)
            raise ValueError("numberOfClusters and numberOfItems are required")
        lam = float(conf.get("lambda"))    # Double.valueOf(conf.get("lambda")), AbstractRM2Reducer.java:108
        return _native.RM2Params(lam, n_items, conf.getInt("numberOfRecommendations", 1000),
                                 conf.getInt("filterUsers", 0), n_clusters, int(rank), int(world), 0,
                                 int(workspace_bytes))

    def prepare(self, ratings, clustering=None, clustering_count=None, rank=0, world=1, workspace_bytes=0, cache=True):
        """Stage 1 (jobs RM2-1 / RM2-2 up to the exchange).  Returns a PreparedRM2 holding this rank's partial item
        statistics in HBM; see RM2Job.run for the arguments."""
        lib = _native.load()
        p = self._params(rank, world, workspace_bytes)
        if not cache:
            p.flags |= 1            # FY_RM2_NO_CACHE
        ctx = self.ctx or Context(0)
        self.ctx = ctx
        own_ratings = not isinstance(ratings, Ratings)
        r = Ratings(ctx, *ratings) if own_ratings else ratings
        try:
            mu, mc = (_i32(clustering[0]), _i32(clustering[1])) if clustering is not None else (_i32(None), _i32(None))
            if len(mu) != len(mc):
                raise ValueError("clustering users / clusters differ in length")
            cc = None
            if clustering_count is not None:
                cc = np.zeros(p.number_of_clusters, dtype=np.int32)
                k = min(len(clustering_count), p.number_of_clusters)
                cc[:k] = np.asarray(clustering_count, dtype=np.int32)[:k]
            job = C.c_void_p()
            try:
                ctx.sync_tuning()
                _check(lib.fy_rm2_prepare(ctx._h, C.byref(p), r._h, len(mu), mu.ctypes.data, mc.ctypes.data,
                                          cc.ctypes.data if cc is not None else None, C.byref(job)))
            except FilmYouError as e:
                raise RuntimeError("%s failed!: %s" % (self.JOB_NAME, e.message)) from e
            return PreparedRM2(job, ctx, world)
        finally:
            if own_ratings:
                r.close()

    def run(self, ratings, clustering=None, clustering_count=None, rank=0, world=1, exchange=None,
            workspace_bytes=0, collectives=None, cache=True):
        """ratings: a ``Ratings`` or a (user, item, score) triple of arrays.
        clustering: (users, clusters) arrays = the reference's `clustering` file; None routes everyone to cluster 0.
        clustering_count: array of numberOfClusters sizes = the `clusteringCount` file (validated when given).
        world > 1 needs one of
          collectives: an object with ``all_gather(send_ptr, recv_ptr, nbytes, stream_ptr)`` and
            ``reduce_scatter_f32(send_ptr, recv_ptr, count, stream_ptr)`` on device pointers (parallel.TorchCollectives =
            RCCL through torch.distributed): the statistics are all-gathered inside the library and clusters that span
            all ranks are scored cooperatively (every rank builds 1/world of the co-rating matrix);
          exchange(device_ptr, length) -> device_ptr of world*length doubles: only the all-gather of the per-item
            statistics (parallel.StatsExchange); every rank then builds the whole matrix of the clusters it holds users of.
        cache: a ``Ratings`` object keeps what the job built from the ratings and the clustering alone (CSR / CSC, per-item
          statistics, the row kernel's tables); a later job over the same object and the same clustering starts from it
          (stats["prepared_from_cache"]).  cache=False = FY_RM2_NO_CACHE: build everything, keep nothing (the cold job).
        Raises RuntimeError("RM2 failed!: ...") on any failure, like RM2Job.java:144-147."""
        if world > 1 and exchange is None and collectives is None:
            raise ValueError("world > 1 needs collectives (or at least an exchange for the item statistics)")
        prepared = self.prepare(ratings, clustering, clustering_count, rank, world, workspace_bytes, cache=cache)
        try:
            if collectives is not None:
                prepared.set_collectives(collectives)
            elif world > 1:
                ptr, n = prepared.partial_stats()
                prepared.set_global_stats(exchange(ptr, n))
            return prepared.score()
        finally:
            prepared.close()


    def run_from_files(self, rank=0, world=1, **kw):
        """RM2Job.run on the reference's own file layout (M/rm/RM2Job.java:76-100, useCassandraInput/Output = false):
        ratings         <mapred.input.dir>                     SequenceFile<IntPairWritable(user,item), FloatWritable>
        clustering      <directory>/<clustering>               SequenceFile<IntWritable, IntWritable>
        clusteringCount <directory>/<clusteringCount>          SequenceFile<IntWritable, IntWritable>
        and writes      <directory>/rm2/userSum/part-r-00000   SequenceFile<IntWritable, DoubleWritable>
                        <directory>/rm2/itemColl/part-r-00000  MapFile<IntWritable, DoubleWritable> (data + index)
                        <mapred.output.dir>/part-r-00000       SequenceFile<IntPairWritable(user,item), FloatWritable>
        <directory>/rm2 is wiped first, like HadoopUtils.removeData (RM2Job.java:84).  Returns 0 (the Tool contract); a
        failure raises RuntimeError("RM2 failed!: ...").  Files: filmyou-core_amd/seqfile.py (layout parity unpinned)."""
        import shutil

        from . import seqfile
        conf = self.conf
        inp, outp = conf.get("mapred.input.dir"), conf.get("mapred.output.dir")
        base = conf.get("directory", "recommendation")
        if not inp or not outp:
            raise ValueError("mapred.input.dir and mapred.output.dir must be set (HadoopUtils.getInputPath / getOutputPath)")
        try:
            user, item, score = seqfile.read_intpair_float(inp)
            cu, cc = seqfile.read_int_int(os.path.join(base, conf.get("clustering", "clustering")))
            ck, cv = seqfile.read_int_int(os.path.join(base, conf.get("clusteringCount", "clusteringCount")))
        except seqfile.SeqFileError as e:
            raise RuntimeError("%s failed!: %s" % (self.JOB_NAME, e)) from e
        K = conf.getInt("numberOfClusters", -1)
        count = np.zeros(max(K, 1), dtype=np.int32)
        ok = (ck >= 0) & (ck < len(count))
        count[ck[ok]] = cv[ok]
        rm2 = os.path.join(base, "rm2")
        suffix = "part-r-%05d" % rank
        # Rank 0 alone wipes <directory>/rm2 (RM2Job.java:84), BEFORE its first collective: every other rank writes only after the
        # job, i.e. after a collective rank 0 has joined.  (The wipe comes before the output check, as in the reference: RM2Job.run
        # removes <directory>/rm2 at :84 and only the submission of job RM2-3 runs checkOutputSpecs -- a refused job has already lost
        # the previous run's statistics there too.)
        # mapred.output.dir is never deleted (the reference does not: Hadoop's FileOutputFormat.checkOutputSpecs refuses an existing
        # directory and the job fails).  One rank: an existing directory fails the job.  Several ranks share the directory, so it may
        # exist -- but it must not hold ANY part-r-* file, whoever wrote it (a rerun into the directory of a job with a larger world would
        # leave that job's other part files beside the new ones, and a reader of the directory would get stale rows without any
        # error).  No rank writes before it has passed a collective inside the job, which every rank joins only after this check: all
        # ranks see the same directory and fail together instead of hanging in a collective.
        if rank == 0:
            shutil.rmtree(rm2, ignore_errors=True)
        if world == 1:
            stale = [outp] if os.path.exists(outp) else []
        else:
            stale = sorted(f for f in (os.listdir(outp) if os.path.isdir(outp) else []) if f.startswith("part-r-"))
            if os.path.exists(outp) and not os.path.isdir(outp):
                stale = [outp]
        if stale:
            raise RuntimeError("%s failed!: output directory %s already exists%s" % (self.JOB_NAME, outp, "" if world == 1 else " and holds " + ", ".join(stale[:4])))
        rec = self.run((user, item, score), clustering=(cu, cc), clustering_count=count, rank=rank, world=world, **kw)
        try:
            rows = rec.rows()
            if rank == 0:       # rm2/userSum and rm2/itemColl are the GLOBAL statistics (jobs RM2-1 / RM2-2): one copy
                sums = rec.sums()
                seqfile.write_int_double(os.path.join(rm2, "userSum", "part-r-00000"), sums["user_id"], sums["user_sum"])
                seqfile.write_mapfile_int_double(os.path.join(rm2, "itemColl", "part-r-00000"), sums["item_id"], sums["item_coll"])
            seqfile.write_intpair_float(os.path.join(outp, suffix), rows["user"], rows["item"], rows["score"])
        finally:
            rec.close()
        return 0


class PreparedRM2:
    """A job between its two stages (fy_rm2_job): the CSR/CSC and this rank's partial statistics are in HBM."""

    def __init__(self, handle, ctx, world):
        self._lib = _native.load()
        self._h = handle
        self._ctx = ctx
        self.world = world

    def partial_stats(self):
        """(device pointer, length in doubles) of the exchange buffer: per-item partial rating sums in ascending raw
        item id order + this rank's partial of the floor-sum counter (x100) (+ user sums and a flag: stats_layout)."""
        buf, n = C.c_void_p(), C.c_int64()
        _check(self._lib.fy_rm2_partial_stats(self._h, C.byref(buf), C.byref(n)))
        return buf.value, n.value

    def stats_layout(self):
        """(n_item_slots, n_user_slots) of the exchange buffer: [n_item_slots item sums][floor-sum][n_user_slots user sums][flag if any user
        slots].  Replicated prep: one slot per rated item (ascending raw id), no user slots.  Sharded prep (several ranks, whole clusters per
        rank, a rank preps its own clusters' ratings alone): slots by RAW id, user sums and a failure flag behind the floor-sum."""
        ni, nu = C.c_int64(), C.c_int64()
        _check(self._lib.fy_rm2_stats_layout(self._h, C.byref(ni), C.byref(nu)))
        return ni.value, nu.value

    def set_global_stats(self, gathered_device_ptr):
        _check(self._lib.fy_rm2_set_global_stats(self._h, gathered_device_ptr, self.world))

    def set_collectives(self, comm):
        """Installs the process group's collectives (fy_rm2_set_collectives).  `comm.all_gather(send, recv, nbytes,
        stream)` / `comm.reduce_scatter_f32(send, recv, count, stream)` get raw device pointers and the hipStream_t the
        operation has to be ordered on; an exception inside them fails the job (FY_ERR_COLLECTIVE)."""
        self._comm_error = None
        if hasattr(comm, "native_struct"):       # compiled transport (parallel.RcclCollectives): C callbacks, no Python in the path
            self._coll = comm.native_struct
            self._keep_comm = comm
            _check(self._lib.fy_rm2_set_collectives(self._h, C.byref(self._coll)))
            return

        def guard(fn):
            def call(_user, send, recv, n, stream):
                try:
                    fn(send, recv, n, stream)
                    return 0
                except BaseException as e:      # must not unwind through the C frames
                    self._comm_error = e
                    return 1
            return call

        self._callbacks = (_native.ALL_GATHER_FN(guard(comm.all_gather)), _native.REDUCE_SCATTER_FN(guard(comm.reduce_scatter_f32)))
        self._coll = _native.Collectives(None, self._callbacks[0], self._callbacks[1])
        _check(self._lib.fy_rm2_set_collectives(self._h, C.byref(self._coll)))

    def score(self):
        res = C.c_void_p()
        try:
            try:
                self._ctx.sync_tuning()
                _check(self._lib.fy_rm2_score(self._h, C.byref(res)))
            except FilmYouError as e:
                if getattr(self, "_comm_error", None) is not None:
                    raise RuntimeError("%s failed!: collective: %r" % (RM2Job.JOB_NAME, self._comm_error)) from self._comm_error
                raise
        except FilmYouError as e:
            raise RuntimeError("%s failed!: %s" % (RM2Job.JOB_NAME, e.message)) from e
        return Recommendations(res, self._ctx)

    def close(self):
        if getattr(self, "_h", None):
            if self._ctx._h:
                self._lib.fy_rm2_job_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ItemRecommendations(_Result):
    """Rows of the item-based recommender's output table: (user, item, score), grouped by user, best first."""

    def __init__(self, handle, ctx=None):
        super().__init__(handle, ctx)
        self.stats = self._stats()
        self.size = self._lib.fy_result_size(handle)
        self._rows = None

    def rows(self):
        if self._rows is None:
            L, h, n = self._lib, self._h, self.size
            self._rows = {"user": _np(L.fy_result_key0(h), n, np.int32), "item": _np(L.fy_result_key1(h), n, np.int32),
                          "score": _np(L.fy_result_value(h), n, np.float32)}
        return self._rows


class BaselineRecommenderJob:
    """Item-based CF (M/baselinerecommender/BaselineRecommenderJob.java:179-328) from the similarity phase on: phase 2
    (RowSimilarityJob), phase 3 (partialMultiply) and phase 4 (aggregateAndRecommend) on the GPU, with the job's
    option names and defaults (:140-176)."""

    JOB_NAME = "BaselineRecommenderJob"

    def __init__(self, ctx=None):
        self.ctx = ctx

    def run(self, ratings, numRecommendations=100, maxPrefsPerUser=50, maxSimilaritiesPerItem=100,
            similarityClassname=SIMILARITY_COSINE, threshold=None, booleanData=False, rank=0, world=1,
            similarities=None):
        """Returns (ItemRecommendations, ItemSimilarities).  `similarities` may be passed to skip phase 2
        (Mahout's --startPhase)."""
        lib = _native.load()
        ctx = self.ctx or Context(0)
        self.ctx = ctx
        own_ratings = not isinstance(ratings, Ratings)
        r = Ratings(ctx, *ratings) if own_ratings else ratings
        try:
            sims = similarities
            if sims is None:
                # every rank needs the whole matrix for phase 3: built with world = 1
                sims = RowSimilarityJob(ctx).run(r, similarityClassname, maxSimilaritiesPerItem, True, threshold)
            p = _native.ItemCFParams(int(numRecommendations), int(maxPrefsPerUser), 1 if booleanData else 0, int(rank),
                                     int(world), 0)
            res = C.c_void_p()
            try:
                ctx.sync_tuning()
                _check(lib.fy_itemcf_recommend(ctx._h, C.byref(p), r._h, sims._h, C.byref(res)))
            except FilmYouError as e:
                raise RuntimeError("%s failed!: %s" % (self.JOB_NAME, e.message)) from e
            return ItemRecommendations(res, ctx), sims
        finally:
            if own_ratings:
                r.close()


class RowSimilarityJob:
    """Item-item similarity build with the options the reference passes to Mahout's RowSimilarityJob
    (M/baselinerecommender/BaselineRecommenderJob.java:241-253)."""

    JOB_NAME = "RowSimilarityJob"

    def __init__(self, ctx=None):
        self.ctx = ctx

    def run(self, ratings, similarityClassname=SIMILARITY_COSINE, maxSimilaritiesPerRow=100,
            excludeSelfSimilarity=True, threshold=None, rank=0, world=1, minPrefsPerUser=1, maxPrefsPerUser=None):
        """minPrefsPerUser / maxPrefsPerUser: the input preparation in front of the similarity job
        (BaselinePreparePreferenceMatrixJob.java:104, 126-129).  Users with fewer preferences than minPrefsPerUser are
        dropped (reference default 1).  maxPrefsPerUser = the reference's maxPrefsPerUserInItemSimilarity (its default 1000 draws
        a RANDOM sample in Mahout's ToItemVectorsMapper); here None = no cap, a number = a DETERMINISTIC systematic sample
        (include/filmyou.h) -- no parity with any particular Mahout run."""
        lib = _native.load()
        if similarityClassname not in _SIMILARITY:
            raise ValueError("similarityClassname must be SIMILARITY_COSINE or SIMILARITY_COOCCURRENCE")
        p = _native.ItemSimParams(_SIMILARITY[similarityClassname], int(maxSimilaritiesPerRow),
                                  1 if excludeSelfSimilarity else 0, 0 if threshold is None else 1,
                                  0.0 if threshold is None else float(threshold), int(rank), int(world), 0,
                                  int(minPrefsPerUser), 0 if maxPrefsPerUser is None else int(maxPrefsPerUser))
        ctx = self.ctx or Context(0)
        self.ctx = ctx
        own_ratings = not isinstance(ratings, Ratings)
        r = Ratings(ctx, *ratings) if own_ratings else ratings
        res = C.c_void_p()
        try:
            try:
                ctx.sync_tuning()
                _check(lib.fy_itemsim_build(ctx._h, C.byref(p), r._h, C.byref(res)))
            except FilmYouError as e:
                raise RuntimeError("%s failed!: %s" % (self.JOB_NAME, e.message)) from e
            return ItemSimilarities(res, ctx)
        finally:
            if own_ratings:
                r.close()


class ClusterAssignmentJob:
    """Cluster assignment in front of the RM2 job (M/nmf/clustering/ClusterAssignmentJob.java:60-135 + CountClustersJob):
    every user goes to the cluster of the largest entry of its row of H (FindClusterMapper.java:37-45), or -- sub-clustering,
    FindSubClusterMapper.java:46-76 -- to  parent * ceil(numberOfUsers / numberOfClusters) + argmax  of its row of that
    parent cluster's H.  ``run`` returns ``(users, clusters, counts)``: the reference's `clustering` and `clusteringCount`
    files as arrays, i.e. the ``clustering=(users, clusters), clustering_count=counts`` arguments of ``RM2Job.run``."""

    def __init__(self, context):
        self._ctx = context
        self._lib = _native.load()

    def _assign(self, H, first_user, offset, counts):
        import numpy as np
        device = hasattr(H, "is_cuda") and H.is_cuda
        if device:
            import torch
            assert H.dtype == torch.float64 and H.is_contiguous() and H.dim() == 2
            torch.cuda.current_stream(H.device).synchronize()
            n, k, ptr = int(H.shape[0]), int(H.shape[1]), H.data_ptr()
        else:
            H = np.ascontiguousarray(H, dtype=np.float64)
            if H.ndim != 2:
                raise ValueError("H must be a (users x clusters) matrix")
            n, k, ptr = H.shape[0], H.shape[1], H.ctypes.data
        users, clusters = np.zeros(n, np.int32), np.zeros(n, np.int32)
        _check(self._lib.fy_cluster_assign(self._ctx._h, n, k, ptr, 1 if device else 0, int(first_user),
                                           int(offset), len(counts), users.ctypes.data, clusters.ctypes.data,
                                           counts.ctypes.data if len(counts) else None))
        return users, clusters

    def run(self, H, first_user=1, number_of_clusters=None):
        """H: (users x numberOfClusters) float64, numpy or a CUDA torch tensor.  first_user: id of row 0 (the reference's H
        files are keyed from 1)."""
        import numpy as np
        if len(H.shape) != 2:
            raise ValueError("H must be a (users x clusters) matrix")
        k = int(H.shape[1]) if number_of_clusters is None else int(number_of_clusters)
        counts = np.zeros(k, np.int32)
        users, clusters = self._assign(H, first_user, 0, counts)
        return users, clusters, counts

    def run_sub(self, parts, number_of_users, number_of_clusters):
        """Sub-clustering: parts = [(parent_cluster, H_parent, first_user), ...] (one H per `cluster<c>` directory of the
        reference); the total number of clusters becomes numberOfClusters * ceil(numberOfUsers / numberOfClusters)."""
        import numpy as np
        n_sub = -(-int(number_of_users) // int(number_of_clusters))
        counts = np.zeros(n_sub * int(number_of_clusters), np.int32)
        us, cs = [], []
        for parent, H, first_user in parts:
            u, c = self._assign(H, first_user, int(parent) * n_sub, counts)
            us.append(u)
            cs.append(c)
        return np.concatenate(us), np.concatenate(cs), counts


class NMFDriver:
    """NMF (``ppc=False``, M/nmf/NMFDriver.java) or PPC (``ppc=True``, M/nmf/ppc/PPCDriver.java) factorisation of the rating
    matrix: ``run`` performs numberOfIterations multiplicative updates of (H, W) (M/nmf/AbstractNMFDriver.java:118-146) on the
    GPU in fp64 and returns the new ``(H, W)``.  H is (numberOfUsers x numberOfClusters), W (numberOfItems x numberOfClusters);
    row r belongs to id r + 1.  Configuration keys: numberOfUsers, numberOfItems, numberOfClusters, numberOfIterations,
    normalizationFrequency (PPC; Java's ``iteration % f``; 0 = never)."""

    def __init__(self, conf, context, ppc=False):
        self._conf, self._ctx, self._ppc = conf, context, bool(ppc)
        self._lib = _native.load()
        self.stats = None

    def run(self, ratings, H, W):
        import numpy as np
        conf = self._conf
        H = np.array(H, dtype=np.float64, order="C")
        W = np.array(W, dtype=np.float64, order="C")
        n_users = conf.getInt("numberOfUsers", H.shape[0])
        n_items = conf.getInt("numberOfItems", W.shape[0])
        k = conf.getInt("numberOfClusters", H.shape[1])
        if H.shape != (n_users, k) or W.shape != (n_items, k):
            raise ValueError("H must be numberOfUsers x numberOfClusters and W numberOfItems x numberOfClusters")
        p = _native.NMFParams(n_users, n_items, k, conf.getInt("numberOfIterations", 1), 1 if self._ppc else 0,
                              conf.getInt("normalizationFrequency", -1))
        own = not isinstance(ratings, Ratings)
        r = Ratings(self._ctx, *ratings) if own else ratings
        try:
            st = _native.Stats()
            try:
                _check(self._lib.fy_nmf_factorize(self._ctx._h, C.byref(p), r._h, H.ctypes.data, W.ctypes.data, C.byref(st)))
            except FilmYouError as e:
                raise RuntimeError("%s failed!: %s" % ("PPC" if self._ppc else "NMF", e.message)) from e
            self.stats = st.as_dict()
        finally:
            if own:
                r.close()
        return H, W

/*
 * NativeRM2Job.java -- the reference-side class a maintainer adds next to es.udc.fi.dc.irlab.rm.RM2Job.
 * NOT COMPILED IN THIS IMAGE (no JDK / Hadoop / Mahout jars).  Same contract as RM2Job (AbstractJob, int run(String[]),
 * parameters through the Hadoop Configuration), so RMRecommenderDriver.run only changes
 *     ToolRunner.run(conf, new RM2Job(), args)        -- RMRecommenderDriver.java:200-201
 * into
 *     ToolRunner.run(conf, new NativeRM2Job(), args)
 * The rating triples are read with the UNCHANGED input formats (SequenceFileInputFormat / CqlInputFormat record readers
 * used by the SimpleScoreBy*Mapper classes), buffered in direct ByteBuffers, scored on the GPU, and written with the
 * UNCHANGED sinks (the same IntPairWritable/FloatWritable rows as RM2HDFSReducer.java:48, or the same
 * keys/values as RM2CassandraReducer.java:49-63 through CqlOutputFormat's RecordWriter).
 */
package es.udc.fi.dc.irlab.rm;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

import org.apache.hadoop.conf.Configuration;
import org.apache.mahout.common.AbstractJob;

import es.udc.fi.dc.irlab.rmrecommender.RMRecommenderDriver;

public class NativeRM2Job extends AbstractJob {

    static {
        System.loadLibrary("filmyou_jni");   // libfilmyou_jni.so -> libfilmyou_hip.so
    }

    /** Hadoop keys of the multi-GPU launch (one JVM per GPU): which shard this process scores, and the 128-byte RCCL id that rank 0
     *  made with {@link #rcclUniqueId()} and published (hex) through the job Configuration.  Unset = one GPU. */
    public static final String RANK_NAME = "filmyou.rank", WORLD_NAME = "filmyou.world", RCCL_ID_NAME = "filmyou.rcclId";
    /** rows of a result window handed to the sink at once (a direct ByteBuffer is limited to 2^31 - 1 bytes) */
    private static final int WINDOW_ROWS = 1 << 24;

    /** One GPU: fy_rm2_run.  Several: fy_context_create(localDevice) + fy_rm2_prepare + fy_rccl_create + fy_rm2_set_collectives +
     *  fy_rm2_score (the item statistics are all-gathered over xGMI inside the library).  clusterCount = the clusteringCount file
     *  as numberOfClusters ints: the reducer relies on it being exact (AbstractRM2Reducer.java:143-160), the library validates it. */
    private static native long run(double lambda, int numberOfItems, int numberOfRecommendations, int filterUsers,
            int numberOfClusters, long nnz, ByteBuffer user, ByteBuffer item, ByteBuffer score, long nMap,
            ByteBuffer mapUser, ByteBuffer mapCluster, ByteBuffer clusterCount, int rank, int world, int localDevice, byte[] rcclId);

    /** fy_rccl_unique_id: called by rank 0 only */
    public static native byte[] rcclUniqueId();

    private static native long size(long handle);

    /** rows [first, first + rows) of column 0 = user, 1 = item, 2 = score (float), 3 = cluster: a view of library-owned memory */
    private static native ByteBuffer window(long handle, int column, long first, int rows);

    private static native void free(long handle);

    /** Rating source and recommendation sink: thin adapters over the reference's existing readers / writers. */
    public interface RatingSource {
        /** fills user/item/score (native order, direct) and returns nnz */
        long read(Configuration conf, ByteBuffer[] out) throws Exception;
    }

    public interface PreferenceSink {
        /** same arguments as AbstractRM2Reducer.writePreference */
        void write(int userId, int itemId, float score, int cluster) throws Exception;
    }

    private final RatingSource source;
    private final RatingSource clusteringSource;        // (user, cluster) pairs of <directory>/<clustering>
    private final RatingSource clusteringCountSource;   // (cluster, count) pairs of <directory>/<clusteringCount>
    private final PreferenceSink sink;

    public NativeRM2Job(final RatingSource source, final RatingSource clusteringSource, final RatingSource clusteringCountSource,
            final PreferenceSink sink) {
        this.source = source;
        this.clusteringSource = clusteringSource;
        this.clusteringCountSource = clusteringCountSource;
        this.sink = sink;
    }

    private static byte[] unhex(final String s) {
        if (s == null) {
            return null;
        }
        final byte[] out = new byte[s.length() / 2];
        for (int k = 0; k < out.length; k++) {
            out[k] = (byte) Integer.parseInt(s.substring(2 * k, 2 * k + 2), 16);
        }
        return out;
    }

    @Override
    public int run(final String[] args) throws Exception {
        final Configuration conf = getConf();
        final int numberOfClusters = conf.getInt(RMRecommenderDriver.numberOfClusters, -1);
        final ByteBuffer[] coo = new ByteBuffer[3];
        final long nnz = source.read(conf, coo);
        final ByteBuffer[] map = new ByteBuffer[3];
        final long nMap = clusteringSource.read(conf, map);
        // clusteringCount: (cluster, count) pairs -> a dense array of numberOfClusters ints, like AbstractRM2Reducer.setup's clusterSizes[]
        final ByteBuffer[] cnt = new ByteBuffer[3];
        final long nCnt = clusteringCountSource.read(conf, cnt);
        final ByteBuffer clusterCount = ByteBuffer.allocateDirect(4 * Math.max(1, numberOfClusters)).order(ByteOrder.nativeOrder());
        final ByteBuffer ck = cnt[0].order(ByteOrder.nativeOrder()), cv = cnt[1].order(ByteOrder.nativeOrder());
        for (long k = 0; k < nCnt; k++) {
            final int c = ck.getInt((int) (4 * k));
            if (c >= 0 && c < numberOfClusters) {
                clusterCount.putInt(4 * c, cv.getInt((int) (4 * k)));
            }
        }
        final int rank = conf.getInt(RANK_NAME, 0), world = conf.getInt(WORLD_NAME, 1);
        // run() throws RuntimeException("RM2 failed!: ...") exactly where RM2Job threw "<jobName> failed!"
        final long h = run(Double.valueOf(conf.get(RM2Job.LAMBDA_NAME)), conf.getInt(RMRecommenderDriver.numberOfItems, -1),
                conf.getInt(RMRecommenderDriver.numberOfRecommendations, -1), conf.getInt(RMRecommenderDriver.filterUsers, 0),
                numberOfClusters, nnz, coo[0], coo[1], coo[2], nMap, map[0], map[1], clusterCount, rank, world,
                conf.getInt("filmyou.localDevice", rank), world > 1 ? unhex(conf.get(RCCL_ID_NAME)) : null);
        try {
            final long n = size(h);      // (rows of THIS rank: its shard of the users; a long -- 480 189 users x 1000 rows is 4.8e8)
            for (long first = 0; first < n; first += WINDOW_ROWS) {
                final int rows = (int) Math.min((long) WINDOW_ROWS, n - first);
                final ByteBuffer u = window(h, 0, first, rows).order(ByteOrder.nativeOrder());
                final ByteBuffer i = window(h, 1, first, rows).order(ByteOrder.nativeOrder());
                final ByteBuffer s = window(h, 2, first, rows).order(ByteOrder.nativeOrder());
                final ByteBuffer c = window(h, 3, first, rows).order(ByteOrder.nativeOrder());
                for (int k = 0; k < rows; k++) {
                    sink.write(u.getInt(4 * k), i.getInt(4 * k), s.getFloat(4 * k), c.getInt(4 * k));
                }
            }
        } finally {
            free(h);
        }
        return 0;
    }
}

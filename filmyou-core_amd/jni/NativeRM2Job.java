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

    private static native long run(double lambda, int numberOfItems, int numberOfRecommendations, int filterUsers,
            int numberOfClusters, long nnz, ByteBuffer user, ByteBuffer item, ByteBuffer score, long nMap,
            ByteBuffer mapUser, ByteBuffer mapCluster, ByteBuffer clusterCount);

    private static native long size(long handle);

    private static native ByteBuffer users(long handle);

    private static native ByteBuffer items(long handle);

    private static native ByteBuffer scores(long handle);

    private static native ByteBuffer clusters(long handle);

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
    private final RatingSource clusteringSource;   // (user, cluster) pairs of <directory>/<clustering>
    private final PreferenceSink sink;

    public NativeRM2Job(final RatingSource source, final RatingSource clusteringSource, final PreferenceSink sink) {
        this.source = source;
        this.clusteringSource = clusteringSource;
        this.sink = sink;
    }

    @Override
    public int run(final String[] args) throws Exception {
        final Configuration conf = getConf();
        final ByteBuffer[] coo = new ByteBuffer[3];
        final long nnz = source.read(conf, coo);
        final ByteBuffer[] map = new ByteBuffer[3];
        final long nMap = clusteringSource.read(conf, map);
        // run() throws RuntimeException("RM2 failed!: ...") exactly where RM2Job threw "<jobName> failed!"
        final long h = run(Double.valueOf(conf.get(RM2Job.LAMBDA_NAME)), conf.getInt(RMRecommenderDriver.numberOfItems, -1),
                conf.getInt(RMRecommenderDriver.numberOfRecommendations, -1), conf.getInt(RMRecommenderDriver.filterUsers, 0),
                conf.getInt(RMRecommenderDriver.numberOfClusters, -1), nnz, coo[0], coo[1], coo[2], nMap, map[0], map[1], null);
        try {
            final int n = (int) size(h);
            final ByteBuffer u = users(h).order(ByteOrder.nativeOrder());
            final ByteBuffer i = items(h).order(ByteOrder.nativeOrder());
            final ByteBuffer s = scores(h).order(ByteOrder.nativeOrder());
            final ByteBuffer c = clusters(h).order(ByteOrder.nativeOrder());
            for (int k = 0; k < n; k++) {
                sink.write(u.getInt(4 * k), i.getInt(4 * k), s.getFloat(4 * k), c.getInt(4 * k));
            }
        } finally {
            free(h);
        }
        return 0;
    }
}

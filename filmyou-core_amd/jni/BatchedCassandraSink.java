/*
 * BatchedCassandraSink.java -- SURVEY.md section 8f row 4: the Cassandra sink, batched, on the JAVA side.
 * NOT COMPILED IN THIS IMAGE (no JDK, no cassandra-driver-core / cassandra-all jars) and not exercised by any test here:
 * source only, written against cassandra-driver-core 2.1.5, which the reference already depends on (pom.xml:43-46).
 *
 * What it replaces.  RM2CassandraReducer.writePreference (M/rm/RM2CassandraReducer.java:49-63) emits ONE
 *     UPDATE <keyspace>.<table> USING TTL <ttl> SET cluster = ?      (M/util/CassandraSetup.java:92-94)
 * per recommendation through CqlOutputFormat's record writer, allocating a LinkedHashMap, a LinkedList and four
 * ByteBuffers per row.  With the scoring on the GPU the sink is the slow half of the job: ML-25M shape, top-50 = 8.1 M rows
 * per job, produced in ~30 ms.
 *
 * What it does.  NativeRM2Job hands its PreferenceSink the rows grouped by user (fy_result: "rows grouped by user,
 * descending score"), and `user` is the partition key of the output table (keys user, item, relevance:
 * RM2CassandraReducer.java:52-54).  All rows of one user therefore go to one replica set: they are sent as ONE
 * UNLOGGED batch of bound statements of one prepared UPDATE (same statement, same TTL, same values as the reference
 * writes), asynchronously, with a bounded number of batches in flight.  N = numberOfRecommendations rows per batch; a
 * user's list longer than MAX_BATCH rows is cut into several batches (the server warns above ~5 kB per batch by default:
 * 12 bytes of keys per row -> 256 rows stay below it).
 */
package es.udc.fi.dc.irlab.rm;

import java.util.ArrayDeque;

import org.apache.hadoop.conf.Configuration;

import com.datastax.driver.core.BatchStatement;
import com.datastax.driver.core.Cluster;
import com.datastax.driver.core.PreparedStatement;
import com.datastax.driver.core.ResultSetFuture;
import com.datastax.driver.core.Session;

import es.udc.fi.dc.irlab.rmrecommender.RMRecommenderDriver;

public class BatchedCassandraSink implements NativeRM2Job.PreferenceSink, AutoCloseable {

    private static final int MAX_BATCH = 256;      // rows per batch (one partition)
    private static final int MAX_IN_FLIGHT = 128;  // batches on the wire before the producer waits for the oldest

    private final Cluster cluster;
    private final Session session;
    private final PreparedStatement update;
    private final ArrayDeque<ResultSetFuture> inFlight = new ArrayDeque<ResultSetFuture>();
    private BatchStatement batch = new BatchStatement(BatchStatement.Type.UNLOGGED);
    private int batchUser = Integer.MIN_VALUE;

    /** Same configuration keys as CassandraSetup.updateConfForOutput (M/util/CassandraSetup.java:79-97). */
    public BatchedCassandraSink(final Configuration conf) {
        final String host = conf.get(RMRecommenderDriver.cassandraHost);
        final String keyspace = conf.get(RMRecommenderDriver.cassandraKeyspace);
        final String table = conf.get(RMRecommenderDriver.cassandraTableOut);
        final String ttl = conf.get(RMRecommenderDriver.cassandraTTL);
        cluster = Cluster.builder().addContactPoint(host).build();       // native protocol port (9042), not the Thrift rpc port
        session = cluster.connect();
        // the reference's statement, with the key columns CqlRecordWriter appends from the `keys` map of writePreference
        update = session.prepare(String.format(
                "UPDATE %s.%s USING TTL %s SET cluster = ? WHERE user = ? AND item = ? AND relevance = ?", keyspace, table, ttl));
    }

    /** One recommendation: same arguments, same stored values as RM2CassandraReducer.writePreference. */
    @Override
    public void write(final int userId, final int itemId, final float score, final int clusterId) throws Exception {
        if (userId != batchUser || batch.size() >= MAX_BATCH) {
            flush();
            batchUser = userId;
        }
        batch.add(update.bind(clusterId, userId, itemId, score));
    }

    private void flush() throws Exception {
        if (batch.size() == 0) {
            return;
        }
        while (inFlight.size() >= MAX_IN_FLIGHT) {
            inFlight.removeFirst().getUninterruptibly();   // a failed batch throws here: the job fails like a failed reducer
        }
        inFlight.addLast(session.executeAsync(batch));
        batch = new BatchStatement(BatchStatement.Type.UNLOGGED);
    }

    /** Sends the last batch and waits for every acknowledgement; the job is complete when this returns. */
    @Override
    public void close() throws Exception {
        try {
            flush();
            while (!inFlight.isEmpty()) {
                inFlight.removeFirst().getUninterruptibly();
            }
        } finally {
            session.close();
            cluster.close();
        }
    }
}

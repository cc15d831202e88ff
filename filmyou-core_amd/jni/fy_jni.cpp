// fy_jni.cpp -- JNI shim: es.udc.fi.dc.irlab.rm.NativeRM2Job / ...baselinerecommender.NativeRowSimilarity -> C ABI.
//
// NOT COMPILED IN THIS IMAGE (no JDK: jni.h is absent); kept as the reference-side binding a maintainer adds.
// Build on a host with a JDK and ROCm:
//   g++ -std=c++17 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I<repo>/include
//       fy_jni.cpp -L<repo>/filmyou-core_amd/lib -lfilmyou_hip -o libfilmyou_jni.so     (one command line)
// All buffers are direct java.nio ByteBuffers in native byte order, so no copy is made on the Java side; the library
// copies them to HBM once (fy_ratings_create, FY_HOST).
#include <jni.h>

#include <cstring>
#include <string>

#include "filmyou.h"

static void throw_runtime(JNIEnv* env, const char* job) {
    std::string msg = std::string(job) + " failed!: " + fy_last_error();   // RM2Job.java:144-147 "<job> failed!"
    env->ThrowNew(env->FindClass("java/lang/RuntimeException"), msg.c_str());
}

template <class T>
static const T* direct(JNIEnv* env, jobject buf) {
    return buf ? static_cast<const T*>(env->GetDirectBufferAddress(buf)) : nullptr;
}

extern "C" {

// A finished job as the Java side holds it: the rows, and -- for a rank of several -- the objects that must outlive them.
struct JniResult {
    fy_result* res = nullptr;
    fy_rccl* comm = nullptr;
    fy_ratings* ratings = nullptr;
    fy_context* ctx = nullptr;
};
static void release(JniResult* r) {
    if (!r) return;
    if (r->res) fy_result_free(r->res);
    if (r->comm) fy_rccl_destroy(r->comm);          // (before the context: the communicator drains the context's stream)
    if (r->ratings) fy_ratings_destroy(r->ratings);
    if (r->ctx) fy_context_destroy(r->ctx);
    delete r;
}

// private static native long run(double lambda, int numberOfItems, int numberOfRecommendations, int filterUsers,
//     int numberOfClusters, long nnz, ByteBuffer user, ByteBuffer item, ByteBuffer score,
//     long nMap, ByteBuffer mapUser, ByteBuffer mapCluster, ByteBuffer clusterCount, int rank, int world, int localDevice, byte[] rcclId);
JNIEXPORT jlong JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_run(JNIEnv* env, jclass, jdouble lambda, jint numberOfItems,
                                                                      jint numberOfRecommendations, jint filterUsers,
                                                                      jint numberOfClusters, jlong nnz, jobject user, jobject item,
                                                                      jobject score, jlong nMap, jobject mapUser,
                                                                      jobject mapCluster, jobject clusterCount, jint rank, jint world,
                                                                      jint localDevice, jbyteArray rcclId) {
    fy_rm2_params p;
    std::memset(&p, 0, sizeof p);
    p.lambda = lambda;
    p.number_of_items = numberOfItems;
    p.number_of_recommendations = numberOfRecommendations;
    p.filter_users = filterUsers;
    p.number_of_clusters = numberOfClusters;
    p.rank = rank;
    p.world = world;
    JniResult* out = new JniResult;
    if (world <= 1) {       // one GPU: one call
        p.rank = 0;
        p.world = 1;
        const int rc = fy_rm2_run(&p, nnz, direct<int32_t>(env, user), direct<int32_t>(env, item), direct<float>(env, score), nMap,
                                  direct<int32_t>(env, mapUser), direct<int32_t>(env, mapCluster), direct<int32_t>(env, clusterCount), &out->res);
        if (rc != FY_OK) { release(out); throw_runtime(env, "RM2"); return 0; }
        return reinterpret_cast<jlong>(out);
    }
    // one JVM per GPU: context on this process's device, RCCL communicator from the id rank 0 published, the staged entry points
    char id[128];
    if (!rcclId || env->GetArrayLength(rcclId) != 128) {
        release(out);
        env->ThrowNew(env->FindClass("java/lang/RuntimeException"), "RM2 failed!: filmyou.world > 1 needs the 128-byte filmyou.rcclId of rank 0");
        return 0;
    }
    env->GetByteArrayRegion(rcclId, 0, 128, reinterpret_cast<jbyte*>(id));
    fy_rm2_job* job = nullptr;
    int rc = fy_context_create(localDevice, &out->ctx);
    if (rc == FY_OK) rc = fy_ratings_create(out->ctx, nnz, direct<int32_t>(env, user), direct<int32_t>(env, item), direct<float>(env, score), FY_HOST, &out->ratings);
    if (rc == FY_OK) rc = fy_rm2_prepare(out->ctx, &p, out->ratings, nMap, direct<int32_t>(env, mapUser), direct<int32_t>(env, mapCluster),
                                         direct<int32_t>(env, clusterCount), &job);
    if (rc == FY_OK) rc = fy_rccl_create(out->ctx, rank, world, id, &out->comm);        // collective over all ranks
    fy_collectives coll;
    if (rc == FY_OK) rc = fy_rccl_collectives(out->comm, &coll);
    if (rc == FY_OK) rc = fy_rm2_set_collectives(job, &coll);
    if (rc == FY_OK) rc = fy_rm2_score(job, &out->res);
    if (job) fy_rm2_job_destroy(job);
    if (rc != FY_OK) { release(out); throw_runtime(env, "RM2"); return 0; }
    return reinterpret_cast<jlong>(out);
}

// public static native byte[] rcclUniqueId();
JNIEXPORT jbyteArray JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_rcclUniqueId(JNIEnv* env, jclass) {
    char id[128];
    if (fy_rccl_unique_id(id) != FY_OK) { throw_runtime(env, "RM2"); return nullptr; }
    jbyteArray a = env->NewByteArray(128);
    env->SetByteArrayRegion(a, 0, 128, reinterpret_cast<const jbyte*>(id));
    return a;
}

JNIEXPORT jlong JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_size(JNIEnv*, jclass, jlong h) {
    return fy_result_size(reinterpret_cast<JniResult*>(h)->res);
}
// private static native ByteBuffer window(long handle, int column, long first, int rows): a view of rows [first, first + rows) of one
// of the library-owned result arrays (valid until free); a direct ByteBuffer holds at most 2^31 - 1 bytes, the result may be larger
JNIEXPORT jobject JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_window(JNIEnv* env, jclass, jlong h, jint column, jlong first, jint rows) {
    fy_result* r = reinterpret_cast<JniResult*>(h)->res;
    const int64_t n = fy_result_size(r);
    if (first < 0 || rows < 0 || first + rows > n || column < 0 || column > 3) {
        env->ThrowNew(env->FindClass("java/lang/IndexOutOfBoundsException"), "result window outside [0, size)");
        return nullptr;
    }
    const void* base = column == 0 ? static_cast<const void*>(fy_result_key0(r))
                     : column == 1 ? static_cast<const void*>(fy_result_key1(r))
                     : column == 2 ? static_cast<const void*>(fy_result_value(r))
                                   : static_cast<const void*>(fy_result_aux(r));
    return env->NewDirectByteBuffer(const_cast<char*>(static_cast<const char*>(base)) + 4 * first, (jlong)rows * 4);
}
JNIEXPORT void JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_free(JNIEnv*, jclass, jlong h) {
    release(reinterpret_cast<JniResult*>(h));
}

// private static native long run(int similarity, int maxSimilaritiesPerRow, boolean excludeSelf, boolean hasThreshold,
//     double threshold, long nnz, ByteBuffer user, ByteBuffer item, ByteBuffer score);
JNIEXPORT jlong JNICALL Java_es_udc_fi_dc_irlab_baselinerecommender_NativeRowSimilarity_run(
    JNIEnv* env, jclass, jint similarity, jint maxSimilaritiesPerRow, jboolean excludeSelf, jboolean hasThreshold, jdouble threshold,
    jlong nnz, jobject user, jobject item, jobject score) {
    fy_itemsim_params p;
    std::memset(&p, 0, sizeof p);
    p.similarity = similarity;
    p.max_similarities_per_item = maxSimilaritiesPerRow;
    p.exclude_self = excludeSelf ? 1 : 0;
    p.has_threshold = hasThreshold ? 1 : 0;
    p.threshold = threshold;
    p.rank = 0;
    p.world = 1;
    JniResult* out = new JniResult;
    const int rc = fy_itemsim_run(&p, nnz, direct<int32_t>(env, user), direct<int32_t>(env, item), direct<float>(env, score), &out->res);
    if (rc != FY_OK) { release(out); throw_runtime(env, "RowSimilarityJob"); return 0; }
    return reinterpret_cast<jlong>(out);
}
// rows (item, other item, similarity): the same handle type and accessors as the RM2 job's (columns 0, 1, 2)
JNIEXPORT jlong JNICALL Java_es_udc_fi_dc_irlab_baselinerecommender_NativeRowSimilarity_size(JNIEnv* env, jclass c, jlong h) {
    return Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_size(env, c, h);
}
JNIEXPORT jobject JNICALL Java_es_udc_fi_dc_irlab_baselinerecommender_NativeRowSimilarity_window(JNIEnv* env, jclass c, jlong h, jint column, jlong first, jint rows) {
    return Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_window(env, c, h, column, first, rows);
}
JNIEXPORT void JNICALL Java_es_udc_fi_dc_irlab_baselinerecommender_NativeRowSimilarity_free(JNIEnv* env, jclass c, jlong h) {
    Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_free(env, c, h);
}

}  // extern "C"

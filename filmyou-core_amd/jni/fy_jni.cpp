// fy_jni.cpp -- JNI shim: es.udc.fi.dc.irlab.rm.NativeRM2Job / ...baselinerecommender.NativeRowSimilarity -> C ABI.
//
// NOT COMPILED IN THIS IMAGE (no JDK: jni.h is absent); kept as the reference-side binding a maintainer adds.
// Build on a host with a JDK and ROCm:
//   g++ -std=c++17 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I<repo>/include \
//       fy_jni.cpp -L<repo>/filmyou-core_amd/lib -lfilmyou_hip -o libfilmyou_jni.so
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

// private static native long run(double lambda, int numberOfItems, int numberOfRecommendations, int filterUsers,
//     int numberOfClusters, long nnz, ByteBuffer user, ByteBuffer item, ByteBuffer score,
//     long nMap, ByteBuffer mapUser, ByteBuffer mapCluster, ByteBuffer clusterCount);
JNIEXPORT jlong JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_run(JNIEnv* env, jclass, jdouble lambda, jint numberOfItems,
                                                                      jint numberOfRecommendations, jint filterUsers,
                                                                      jint numberOfClusters, jlong nnz, jobject user, jobject item,
                                                                      jobject score, jlong nMap, jobject mapUser,
                                                                      jobject mapCluster, jobject clusterCount) {
    fy_rm2_params p;
    std::memset(&p, 0, sizeof p);
    p.lambda = lambda;
    p.number_of_items = numberOfItems;
    p.number_of_recommendations = numberOfRecommendations;
    p.filter_users = filterUsers;
    p.number_of_clusters = numberOfClusters;
    p.rank = 0;
    p.world = 1;
    fy_result* res = nullptr;
    const int rc = fy_rm2_run(&p, nnz, direct<int32_t>(env, user), direct<int32_t>(env, item), direct<float>(env, score), nMap,
                              direct<int32_t>(env, mapUser), direct<int32_t>(env, mapCluster), direct<int32_t>(env, clusterCount), &res);
    if (rc != FY_OK) { throw_runtime(env, "RM2"); return 0; }
    return reinterpret_cast<jlong>(res);
}

JNIEXPORT jlong JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_size(JNIEnv*, jclass, jlong h) {
    return fy_result_size(reinterpret_cast<fy_result*>(h));
}
// ByteBuffer views over the library-owned result arrays (valid until free)
JNIEXPORT jobject JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_users(JNIEnv* env, jclass, jlong h) {
    fy_result* r = reinterpret_cast<fy_result*>(h);
    return env->NewDirectByteBuffer(const_cast<int32_t*>(fy_result_key0(r)), fy_result_size(r) * 4);
}
JNIEXPORT jobject JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_items(JNIEnv* env, jclass, jlong h) {
    fy_result* r = reinterpret_cast<fy_result*>(h);
    return env->NewDirectByteBuffer(const_cast<int32_t*>(fy_result_key1(r)), fy_result_size(r) * 4);
}
JNIEXPORT jobject JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_scores(JNIEnv* env, jclass, jlong h) {
    fy_result* r = reinterpret_cast<fy_result*>(h);
    return env->NewDirectByteBuffer(const_cast<float*>(fy_result_value(r)), fy_result_size(r) * 4);
}
JNIEXPORT jobject JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_clusters(JNIEnv* env, jclass, jlong h) {
    fy_result* r = reinterpret_cast<fy_result*>(h);
    return env->NewDirectByteBuffer(const_cast<int32_t*>(fy_result_aux(r)), fy_result_size(r) * 4);
}
JNIEXPORT void JNICALL Java_es_udc_fi_dc_irlab_rm_NativeRM2Job_free(JNIEnv*, jclass, jlong h) {
    fy_result_free(reinterpret_cast<fy_result*>(h));
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
    fy_result* res = nullptr;
    const int rc = fy_itemsim_run(&p, nnz, direct<int32_t>(env, user), direct<int32_t>(env, item), direct<float>(env, score), &res);
    if (rc != FY_OK) { throw_runtime(env, "RowSimilarityJob"); return 0; }
    return reinterpret_cast<jlong>(res);
}

}  // extern "C"

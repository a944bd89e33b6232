/* jtk_jni.c -- JNI glue between com.knuddels.jtokkit.hip.HipEncoding and the C ABI (include/jtokkit_amd.h).
 * NOT COMPILED in the build image (no JDK / jni.h there).  Build where a JDK exists:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include jtk_jni.c \
 *       -L.. -ljtokkit_amd -o libjtokkit_amd_jni.so
 * Status codes map 1:1 to the exceptions the reference throws (see the enum in jtokkit_amd.h). */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "jtokkit_amd.h"

static void throw_for(JNIEnv* env, int rc) {
    const char* cls;
    const char* msg = jtk_last_error();
    switch (rc) {
        case JTK_ERR_UNSUPPORTED_SPECIAL: cls = "java/lang/UnsupportedOperationException";
            msg = "Encoding special tokens is not supported yet."; break;          /* GptBytePairEncoding.java:54 */
        case JTK_ERR_UNKNOWN_TOKEN:
        case JTK_ERR_INVALID_ARGUMENT: cls = "java/lang/IllegalArgumentException"; break;
        case JTK_ERR_OUT_OF_MEMORY: cls = "java/lang/OutOfMemoryError"; break;
        default: cls = "java/lang/IllegalStateException"; break;                  /* EncodingFactory.java:142,151,162 */
    }
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), msg);
}

JNIEXPORT jlong JNICALL Java_com_knuddels_jtokkit_hip_HipEncoding_nativeCreate(
        JNIEnv* env, jclass c, jstring name, jint kind, jbyteArray tiktoken, jobjectArray lits, jintArray ids, jint device) {
    (void)c;
    const char* cname = (*env)->GetStringUTFChars(env, name, NULL);   /* ASCII encoding name only */
    jsize tlen = (*env)->GetArrayLength(env, tiktoken);
    jbyte* tbytes = (*env)->GetByteArrayElements(env, tiktoken, NULL);
    jsize ns = (*env)->GetArrayLength(env, lits);
    const char** clits = (const char**)calloc((size_t)ns + 1, sizeof(char*));
    jint* cids = (*env)->GetIntArrayElements(env, ids, NULL);
    for (jsize i = 0; i < ns; i++)
        clits[i] = (*env)->GetStringUTFChars(env, (jstring)(*env)->GetObjectArrayElement(env, lits, i), NULL);
    jtk_encoding* enc = NULL;
    int rc = jtk_encoding_create(cname, kind, (const uint8_t*)tbytes, (size_t)tlen, clits, (const int32_t*)cids, ns, device, &enc);
    for (jsize i = 0; i < ns; i++)
        (*env)->ReleaseStringUTFChars(env, (jstring)(*env)->GetObjectArrayElement(env, lits, i), clits[i]);
    free(clits);
    (*env)->ReleaseIntArrayElements(env, ids, cids, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, tiktoken, tbytes, JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, name, cname);
    if (rc != JTK_OK) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)enc;
}

JNIEXPORT void JNICALL Java_com_knuddels_jtokkit_hip_HipEncoding_nativeDestroy(JNIEnv* env, jclass c, jlong h) {
    (void)env; (void)c;
    jtk_encoding_destroy((jtk_encoding*)(intptr_t)h);
}

JNIEXPORT jlong JNICALL Java_com_knuddels_jtokkit_hip_HipEncoding_nativeBatchCreate(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    jtk_batch* b = NULL;
    int rc = jtk_batch_create((const jtk_encoding*)(intptr_t)h, &b);
    if (rc != JTK_OK) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)b;
}

JNIEXPORT jintArray JNICALL Java_com_knuddels_jtokkit_hip_HipEncoding_nativeEncode(
        JNIEnv* env, jclass c, jlong batch, jbyteArray utf8, jint flags, jint maxTokens, jbooleanArray truncated) {
    (void)c;
    jsize len = (*env)->GetArrayLength(env, utf8);
    jbyte* bytes = (*env)->GetByteArrayElements(env, utf8, NULL);
    int32_t* toks = (int32_t*)malloc(((size_t)len + 1) * sizeof(int32_t));     /* tokens <= bytes */
    int64_t n = 0; int tr = 0;
    int rc = jtk_encode((jtk_batch*)(intptr_t)batch, (const uint8_t*)bytes, len, (uint32_t)flags, maxTokens,
                        toks, (int64_t)len + 1, &n, &tr);
    (*env)->ReleaseByteArrayElements(env, utf8, bytes, JNI_ABORT);
    if (rc != JTK_OK) { free(toks); throw_for(env, rc); return NULL; }
    jintArray out = (*env)->NewIntArray(env, (jsize)n);
    (*env)->SetIntArrayRegion(env, out, 0, (jsize)n, (const jint*)toks);
    free(toks);
    jboolean jt = tr ? JNI_TRUE : JNI_FALSE;
    (*env)->SetBooleanArrayRegion(env, truncated, 0, 1, &jt);
    return out;
}

JNIEXPORT jobject JNICALL Java_com_knuddels_jtokkit_hip_HipEncoding_nativeEncodeBatch(
        JNIEnv* env, jclass c, jlong batch, jobject utf8, jlongArray docOff, jint flags) {
    (void)c;
    jtk_batch* b = (jtk_batch*)(intptr_t)batch;
    const uint8_t* text = (const uint8_t*)(*env)->GetDirectBufferAddress(env, utf8);
    jsize n1 = (*env)->GetArrayLength(env, docOff);
    jlong* off = (*env)->GetLongArrayElements(env, docOff, NULL);
    int64_t nt = 0;
    int rc = jtk_batch_encode(b, text, (const int64_t*)off, n1 - 1, (uint32_t)flags, &nt);
    (*env)->ReleaseLongArrayElements(env, docOff, off, JNI_ABORT);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    jintArray toks = (*env)->NewIntArray(env, (jsize)nt);
    jlongArray toff = (*env)->NewLongArray(env, n1);
    jintArray stat = (*env)->NewIntArray(env, n1 - 1);
    jint* ptoks = (*env)->GetPrimitiveArrayCritical(env, toks, NULL);
    jlong* ptoff = (*env)->GetPrimitiveArrayCritical(env, toff, NULL);
    jint* pstat = (*env)->GetPrimitiveArrayCritical(env, stat, NULL);
    rc = jtk_batch_fetch(b, (int32_t*)ptoks, nt, (int64_t*)ptoff, (int32_t*)pstat);
    (*env)->ReleasePrimitiveArrayCritical(env, stat, pstat, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, toff, ptoff, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, toks, ptoks, 0);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    jclass rcCls = (*env)->FindClass(env, "com/knuddels/jtokkit/hip/HipEncoding$BatchResult");
    jmethodID ctor = (*env)->GetMethodID(env, rcCls, "<init>", "([I[J[I)V");
    return (*env)->NewObject(env, rcCls, ctor, toks, toff, stat);
}

JNIEXPORT jbyteArray JNICALL Java_com_knuddels_jtokkit_hip_HipEncoding_nativeDecode(JNIEnv* env, jclass c, jlong h, jintArray ids) {
    (void)c;
    jsize n = (*env)->GetArrayLength(env, ids);
    jint* p = (*env)->GetIntArrayElements(env, ids, NULL);
    int64_t len = 0;
    int rc = jtk_decode((const jtk_encoding*)(intptr_t)h, (const int32_t*)p, n, NULL, 0, &len);
    uint8_t* buf = rc == JTK_OK ? (uint8_t*)malloc((size_t)len + 1) : NULL;
    if (rc == JTK_OK) rc = jtk_decode((const jtk_encoding*)(intptr_t)h, (const int32_t*)p, n, buf, len, &len);
    (*env)->ReleaseIntArrayElements(env, ids, p, JNI_ABORT);
    if (rc != JTK_OK) { free(buf); throw_for(env, rc); return NULL; }
    jbyteArray out = (*env)->NewByteArray(env, (jsize)len);
    (*env)->SetByteArrayRegion(env, out, 0, (jsize)len, (const jbyte*)buf);
    free(buf);
    return out;
}

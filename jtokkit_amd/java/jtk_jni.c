/* jtk_jni.c -- JNI glue between com.knuddels.jtokkit.hip.HipEncoding and the C ABI (include/jtokkit_amd.h).
 * NOT COMPILED in the build image (no JDK / jni.h there).  Build where a JDK exists:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include jtk_jni.c \
 *       -L.. -ljtokkit_amd -o libjtokkit_amd_jni.so
 * Status codes map 1:1 to the exceptions the reference throws (see the enum in jtokkit_amd.h).
 * No JNI critical section is held across a device call (a blocking HIP call inside GetPrimitiveArrayCritical can stall the
 * collector): results are read from the batch's pinned host buffers (JTK_ENCODE_TO_HOST + jtk_batch_host_result) and
 * copied with Set*ArrayRegion. */
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
        case JTK_ERR_UNENCODABLE: cls = "java/lang/IllegalArgumentException";
            msg = "Unknown token for encoding"; break;                              /* TokenEncoder.java:66-68 */
        case JTK_ERR_OUT_OF_MEMORY: cls = "java/lang/OutOfMemoryError"; break;
        default: cls = "java/lang/IllegalStateException"; break;                  /* EncodingFactory.java:142,151,162 */
    }
    jclass c = (*env)->FindClass(env, cls);
    (*env)->ThrowNew(env, c, msg);
    (*env)->DeleteLocalRef(env, c);
}

#define ENC(h) ((jtk_encoding*)(intptr_t)(h))
#define BATCH(h) ((jtk_batch*)(intptr_t)(h))
#define SVC(h) ((jtk_service*)(intptr_t)(h))
#define FN(name) Java_com_knuddels_jtokkit_hip_HipEncoding_##name

JNIEXPORT jlong JNICALL FN(nativeCreate)(JNIEnv* env, jclass c, jstring name, jint kind, jbyteArray tiktoken, jobjectArray lits,
                                         jintArray ids, jint device) {
    (void)c;
    const char* cname = (*env)->GetStringUTFChars(env, name, NULL);   /* ASCII encoding name only */
    jsize tlen = (*env)->GetArrayLength(env, tiktoken);
    jbyte* tbytes = (*env)->GetByteArrayElements(env, tiktoken, NULL);
    jsize ns = (*env)->GetArrayLength(env, lits);
    const char** clits = (const char**)calloc((size_t)ns + 1, sizeof(char*));
    jstring* jlits = (jstring*)calloc((size_t)ns + 1, sizeof(jstring));          /* the elements are fetched ONCE and kept */
    jint* cids = (*env)->GetIntArrayElements(env, ids, NULL);
    for (jsize i = 0; i < ns; i++) {
        jlits[i] = (jstring)(*env)->GetObjectArrayElement(env, lits, i);
        clits[i] = (*env)->GetStringUTFChars(env, jlits[i], NULL);               /* special literals are ASCII */
    }
    jtk_encoding* enc = NULL;
    int rc = jtk_encoding_create(cname, kind, (const uint8_t*)tbytes, (size_t)tlen, clits, (const int32_t*)cids, ns, device, &enc);
    for (jsize i = 0; i < ns; i++) {
        (*env)->ReleaseStringUTFChars(env, jlits[i], clits[i]);
        (*env)->DeleteLocalRef(env, jlits[i]);
    }
    free(jlits);
    free(clits);
    (*env)->ReleaseIntArrayElements(env, ids, cids, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, tiktoken, tbytes, JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, name, cname);
    if (rc != JTK_OK) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)enc;
}

JNIEXPORT void JNICALL FN(nativeDestroy)(JNIEnv* env, jclass c, jlong h) { (void)env; (void)c; jtk_encoding_destroy(ENC(h)); }

JNIEXPORT jlong JNICALL FN(nativeServiceCreate)(JNIEnv* env, jclass c, jlong h, jint workers) {
    (void)c;
    jtk_service* s = NULL;
    int rc = jtk_service_create(ENC(h), workers, &s);
    if (rc != JTK_OK) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)s;
}
JNIEXPORT void JNICALL FN(nativeServiceDestroy)(JNIEnv* env, jclass c, jlong s) { (void)env; (void)c; jtk_service_destroy(SVC(s)); }

/* Encoding.encode / encodeOrdinary (with or without maxTokens) for one String: blocks in the service, which coalesces the
 * concurrent callers into one device batch */
JNIEXPORT jintArray JNICALL FN(nativeServiceEncode)(JNIEnv* env, jclass c, jlong svc, jbyteArray utf8, jint flags, jint maxTokens,
                                                    jbooleanArray truncated) {
    (void)c;
    jsize len = (*env)->GetArrayLength(env, utf8);
    jbyte* bytes = (*env)->GetByteArrayElements(env, utf8, NULL);
    int32_t* toks = (int32_t*)malloc(((size_t)len + 1) * sizeof(int32_t));     /* tokens <= bytes */
    int64_t n = 0; int tr = 0;
    /* (only JTK_ENCODE_ORDINARY passes: a count-only call would leave `toks` unwritten) */
    int rc = jtk_service_encode(SVC(svc), (const uint8_t*)bytes, len, (uint32_t)flags & JTK_ENCODE_ORDINARY, maxTokens, toks, (int64_t)len + 1, &n, &tr);
    (*env)->ReleaseByteArrayElements(env, utf8, bytes, JNI_ABORT);
    if (rc != JTK_OK) { free(toks); throw_for(env, rc); return NULL; }
    jintArray out = (*env)->NewIntArray(env, (jsize)n);
    (*env)->SetIntArrayRegion(env, out, 0, (jsize)n, (const jint*)toks);
    free(toks);
    jboolean jt = tr ? JNI_TRUE : JNI_FALSE;
    (*env)->SetBooleanArrayRegion(env, truncated, 0, 1, &jt);
    return out;
}

/* encodeAsync: the two halves of the above.  The document's bytes and the token buffer live in one native block until the
 * ticket has been waited for (the service reads and writes them from its worker threads). */
typedef struct { jtk_ticket* ticket; int64_t len; int32_t* toks; uint8_t bytes[]; } jtk_jni_pending;

JNIEXPORT jlong JNICALL FN(nativeServiceSubmit)(JNIEnv* env, jclass c, jlong svc, jbyteArray utf8, jint flags) {
    (void)c;
    jsize len = (*env)->GetArrayLength(env, utf8);
    const size_t text_bytes = ((size_t)len + 7) & ~(size_t)7;
    jtk_jni_pending* p = (jtk_jni_pending*)malloc(sizeof(jtk_jni_pending) + text_bytes + ((size_t)len + 1) * sizeof(int32_t));
    if (!p) { throw_for(env, JTK_ERR_OUT_OF_MEMORY); return 0; }
    (*env)->GetByteArrayRegion(env, utf8, 0, len, (jbyte*)p->bytes);
    p->len = len;
    p->toks = (int32_t*)(p->bytes + text_bytes);
    int rc = jtk_service_submit(SVC(svc), p->bytes, len, (uint32_t)flags & JTK_ENCODE_ORDINARY, -1, p->toks, (int64_t)len + 1, &p->ticket);
    if (rc != JTK_OK) { free(p); throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)p;
}

JNIEXPORT jintArray JNICALL FN(nativeServiceWait)(JNIEnv* env, jclass c, jlong svc, jlong pending) {
    (void)c;
    jtk_jni_pending* p = (jtk_jni_pending*)(intptr_t)pending;
    int64_t n = 0;
    int rc = jtk_service_wait(SVC(svc), p->ticket, &n, NULL);
    if (rc != JTK_OK) { free(p); throw_for(env, rc); return NULL; }
    jintArray out = (*env)->NewIntArray(env, (jsize)n);
    (*env)->SetIntArrayRegion(env, out, 0, (jsize)n, (const jint*)p->toks);
    free(p);
    return out;
}

JNIEXPORT jlong JNICALL FN(nativeBatchCreate)(JNIEnv* env, jclass c, jlong h) {
    (void)c;
    jtk_batch* b = NULL;
    int rc = jtk_batch_create(ENC(h), &b);
    if (rc != JTK_OK) { throw_for(env, rc); return 0; }
    return (jlong)(intptr_t)b;
}
JNIEXPORT void JNICALL FN(nativeBatchDestroy)(JNIEnv* env, jclass c, jlong b) { (void)env; (void)c; jtk_batch_destroy(BATCH(b)); }

/* the batch's pinned host result -> HipEncoding.BatchResult(int[] tokens, long[] tokOff, int[] status) */
static jobject batch_result(JNIEnv* env, jtk_batch* b, jsize n_docs) {
    const int32_t* toks = NULL; const int64_t* toff = NULL; const int32_t* stat = NULL;
    int64_t nt = 0;
    int rc = jtk_batch_result(b, &nt, NULL, NULL);
    if (rc == JTK_OK) rc = jtk_batch_host_result(b, &toks, &toff, &stat);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    jintArray jt = (*env)->NewIntArray(env, toks ? (jsize)nt : 0);
    jlongArray jo = (*env)->NewLongArray(env, n_docs + 1);
    jintArray js = (*env)->NewIntArray(env, n_docs);
    if (toks && nt > 0) (*env)->SetIntArrayRegion(env, jt, 0, (jsize)nt, (const jint*)toks);
    (*env)->SetLongArrayRegion(env, jo, 0, n_docs + 1, (const jlong*)toff);
    if (n_docs > 0) (*env)->SetIntArrayRegion(env, js, 0, n_docs, (const jint*)stat);
    jclass cls = (*env)->FindClass(env, "com/knuddels/jtokkit/hip/HipEncoding$BatchResult");
    jmethodID ctor = (*env)->GetMethodID(env, cls, "<init>", "([I[J[I)V");
    jobject out = (*env)->NewObject(env, cls, ctor, jt, jo, js);
    (*env)->DeleteLocalRef(env, cls);
    return out;
}

/* a loop of Encoding.encode / encodeOrdinary / countTokens (flags bit 2) over a batch: chunked H2D, kernels and D2H overlap */
JNIEXPORT jobject JNICALL FN(nativeEncodeBatch)(JNIEnv* env, jclass c, jlong batch, jobject utf8, jlongArray docOff, jint flags) {
    (void)c;
    const uint8_t* text = (const uint8_t*)(*env)->GetDirectBufferAddress(env, utf8);
    jsize n1 = (*env)->GetArrayLength(env, docOff);
    jlong* off = (*env)->GetLongArrayElements(env, docOff, NULL);                /* a copy or a pin, no critical section */
    int64_t nt = 0;
    int rc = jtk_batch_encode(BATCH(batch), text, (const int64_t*)off, n1 - 1, (uint32_t)flags | JTK_ENCODE_TO_HOST, &nt);
    (*env)->ReleaseLongArrayElements(env, docOff, off, JNI_ABORT);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    return batch_result(env, BATCH(batch), n1 - 1);
}

/* custom java.util.regex.Pattern: the matches found on the JVM, encoded on the device */
JNIEXPORT jobject JNICALL FN(nativeEncodeBatchPieces)(JNIEnv* env, jclass c, jlong batch, jobject utf8, jlongArray docOff,
                                                      jlongArray pieceBegin, jlongArray pieceEnd, jint flags) {
    (void)c;
    const uint8_t* text = (const uint8_t*)(*env)->GetDirectBufferAddress(env, utf8);
    jsize n1 = (*env)->GetArrayLength(env, docOff), np = (*env)->GetArrayLength(env, pieceBegin);
    jlong* off = (*env)->GetLongArrayElements(env, docOff, NULL);
    jlong* pb = (*env)->GetLongArrayElements(env, pieceBegin, NULL);
    jlong* pe = (*env)->GetLongArrayElements(env, pieceEnd, NULL);
    int64_t nt = 0;
    int rc = jtk_batch_encode_pieces(BATCH(batch), text, (const int64_t*)off, n1 - 1, (const int64_t*)pb, (const int64_t*)pe, np,
                                     (uint32_t)flags | JTK_ENCODE_TO_HOST, &nt);
    (*env)->ReleaseLongArrayElements(env, pieceEnd, pe, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, pieceBegin, pb, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, docOff, off, JNI_ABORT);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    return batch_result(env, BATCH(batch), n1 - 1);
}

/* Encoding.encode(text, maxTokens) for every document of the batch's last encode (GptBytePairEncoding.java:90-100 on the device) */
JNIEXPORT void JNICALL FN(nativeTruncateBatch)(JNIEnv* env, jclass c, jlong batch, jlong maxTokens, jlongArray kept, jbooleanArray truncated) {
    (void)c;
    jsize n = (*env)->GetArrayLength(env, kept);
    int rc = jtk_batch_truncate(BATCH(batch), maxTokens);
    int64_t* k = (int64_t*)malloc(((size_t)n + 1) * sizeof(int64_t));
    uint8_t* t = (uint8_t*)malloc((size_t)n + 1);
    if (rc == JTK_OK) rc = jtk_batch_fetch_truncated(BATCH(batch), k, t);
    if (rc == JTK_OK) {
        (*env)->SetLongArrayRegion(env, kept, 0, n, (const jlong*)k);
        (*env)->SetBooleanArrayRegion(env, truncated, 0, n, (const jboolean*)t);
    }
    free(k); free(t);
    if (rc != JTK_OK) throw_for(env, rc);
}

/* Encoding.encode(text, maxTokens) for every document WITHOUT encoding the documents whole (the reference stops matching at
 * maxTokens, GptBytePairEncoding.java:83-88): ids[d * maxTokens ..] hold kept[d] ids of document d */
JNIEXPORT void JNICALL FN(nativeEncodeBatchMaxTokens)(JNIEnv* env, jclass c, jlong batch, jobject utf8, jlongArray docOff, jint flags,
                                                       jint maxTokens, jintArray ids, jlongArray kept, jbooleanArray truncated) {
    (void)c;
    const uint8_t* text = (const uint8_t*)(*env)->GetDirectBufferAddress(env, utf8);
    jsize n1 = (*env)->GetArrayLength(env, docOff);
    const size_t n = (size_t)(n1 - 1), mt = (size_t)(maxTokens > 0 ? maxTokens : 0);
    jlong* off = (*env)->GetLongArrayElements(env, docOff, NULL);
    int32_t* tk = (int32_t*)malloc((n * mt + 1) * sizeof(int32_t));
    int64_t* k = (int64_t*)malloc((n + 1) * sizeof(int64_t));
    uint8_t* t = (uint8_t*)malloc(n + 1);
    int32_t* st = (int32_t*)malloc((n + 1) * sizeof(int32_t));
    int rc = jtk_batch_encode_max_tokens(BATCH(batch), text, (const int64_t*)off, (int64_t)n, (uint32_t)flags & JTK_ENCODE_ORDINARY,
                                         (int64_t)mt, tk, k, t, st);
    (*env)->ReleaseLongArrayElements(env, docOff, off, JNI_ABORT);
    for (size_t d = 0; d < n && rc == JTK_OK; d++) if (st[d] != JTK_OK) rc = st[d];          /* :52-56 special token in the text */
    if (rc == JTK_OK) {
        (*env)->SetIntArrayRegion(env, ids, 0, (jsize)(n * mt), (const jint*)tk);
        (*env)->SetLongArrayRegion(env, kept, 0, (jsize)n, (const jlong*)k);
        (*env)->SetBooleanArrayRegion(env, truncated, 0, (jsize)n, (const jboolean*)t);
    }
    free(tk); free(k); free(t); free(st);
    if (rc != JTK_OK) throw_for(env, rc);
}

/* a loop of Encoding.decodeBytes over many token lists: one device pass */
JNIEXPORT jobjectArray JNICALL FN(nativeDecodeBatch)(JNIEnv* env, jclass c, jlong batch, jintArray ids, jlongArray seqOff) {
    (void)c;
    jsize n1 = (*env)->GetArrayLength(env, seqOff);
    jint* pid = (*env)->GetIntArrayElements(env, ids, NULL);
    jlong* poff = (*env)->GetLongArrayElements(env, seqOff, NULL);
    int64_t nb = 0;
    int rc = jtk_batch_decode(BATCH(batch), (const int32_t*)pid, (const int64_t*)poff, n1 - 1, &nb);
    (*env)->ReleaseLongArrayElements(env, seqOff, poff, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, ids, pid, JNI_ABORT);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    uint8_t* bytes = (uint8_t*)malloc((size_t)nb + 1);
    int64_t* boff = (int64_t*)malloc((size_t)n1 * sizeof(int64_t));
    int32_t* stat = (int32_t*)malloc((size_t)n1 * sizeof(int32_t));
    rc = jtk_batch_decode_fetch(BATCH(batch), bytes, nb, boff, stat);
    jobjectArray out = NULL;
    if (rc == JTK_OK) {
        for (jsize q = 0; q + 1 < n1 && rc == JTK_OK; q++) if (stat[q] != JTK_OK) rc = stat[q];     /* :313 unknown token */
    }
    if (rc == JTK_OK) {
        jclass ba = (*env)->FindClass(env, "[B");
        out = (*env)->NewObjectArray(env, n1 - 1, ba, NULL);
        for (jsize q = 0; q + 1 < n1; q++) {
            jbyteArray one = (*env)->NewByteArray(env, (jsize)(boff[q + 1] - boff[q]));
            (*env)->SetByteArrayRegion(env, one, 0, (jsize)(boff[q + 1] - boff[q]), (const jbyte*)(bytes + boff[q]));
            (*env)->SetObjectArrayElement(env, out, q, one);
            (*env)->DeleteLocalRef(env, one);
        }
        (*env)->DeleteLocalRef(env, ba);
    }
    free(bytes); free(boff); free(stat);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    return out;
}

JNIEXPORT jbyteArray JNICALL FN(nativeDecode)(JNIEnv* env, jclass c, jlong h, jintArray ids) {
    (void)c;
    jsize n = (*env)->GetArrayLength(env, ids);
    jint* p = (*env)->GetIntArrayElements(env, ids, NULL);
    int64_t len = 0;
    int rc = jtk_decode(ENC(h), (const int32_t*)p, n, NULL, 0, &len);
    uint8_t* buf = rc == JTK_OK ? (uint8_t*)malloc((size_t)len + 1) : NULL;
    if (rc == JTK_OK) rc = jtk_decode(ENC(h), (const int32_t*)p, n, buf, len, &len);
    (*env)->ReleaseIntArrayElements(env, ids, p, JNI_ABORT);
    if (rc != JTK_OK) { free(buf); throw_for(env, rc); return NULL; }
    jbyteArray out = (*env)->NewByteArray(env, (jsize)len);
    (*env)->SetByteArrayRegion(env, out, 0, (jsize)len, (const jbyte*)buf);
    free(buf);
    return out;
}

/* page-locked direct buffers: the device reads the documents by DMA from where the JVM put them */
JNIEXPORT jobject JNICALL FN(nativeHostAlloc)(JNIEnv* env, jclass c, jlong bytes) {
    (void)c;
    void* p = NULL;
    int rc = jtk_host_alloc((size_t)bytes, &p);
    if (rc != JTK_OK) { throw_for(env, rc); return NULL; }
    return (*env)->NewDirectByteBuffer(env, p, bytes);
}
JNIEXPORT void JNICALL FN(nativeHostFree)(JNIEnv* env, jclass c, jobject buffer) {
    (void)c;
    jtk_host_free((*env)->GetDirectBufferAddress(env, buffer));
}

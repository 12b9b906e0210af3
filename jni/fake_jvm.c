/* A JVM stand-in that can RUN jni/pinot_gpu_jni.c where no JDK exists: an implementation of exactly the JNIEnv functions declared in
 * jni/stub/jni.h (the ones that file uses), with the JNI specification's semantics for them -- arrays, strings, object arrays, local
 * references, pinned elements, pending exceptions.  Test infrastructure only (tests/test_jni_harness_cpu.py, tests/test_gpu_jni_harness.py
 * through pinot_amd/jni_harness.py): the native half of the binding -- array pinning, exception mapping, reference discipline, the
 * batch call -- executes here against the real libpinot_gpu.so; what remains untested without a JDK is the Java half.
 *
 * It is NOT binary-compatible with a JVM (the stub's function table has neither the real table's order nor its size): it only ever
 * meets pinot_gpu_jni.c compiled against the same stub header, inside libpinot_gpu_jni_fake.so (jni/Makefile, target `fake`).
 *
 * Reference model: every object carries a count of references (local references handed to native code or to the harness, and slots of
 * object arrays that hold it); DeleteLocalRef drops one.  fj_live_refs / fj_peak_refs count LOCAL references, which is what a JVM limits
 * (16 guaranteed without EnsureLocalCapacity): a native method that creates them in a loop has to give them back in the loop. */
#include <jni.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { FJ_INT_ARRAY = 1, FJ_LONG_ARRAY = 2, FJ_DOUBLE_ARRAY = 3, FJ_OBJECT_ARRAY = 4, FJ_STRING = 5, FJ_CLASS = 6, FJ_BUFFER = 7 };

typedef struct fj_object {
  int kind;
  jsize len;          /* elements (arrays), bytes without the terminator (strings / class names) */
  void* data;         /* elements; fj_object*[] for object arrays; char[] for strings and class names; the address for buffers */
  int refs;
  int pins;           /* Get<Type>ArrayElements / GetStringUTFChars not yet released */
} fj_object;

static long g_live_refs, g_peak_refs, g_live_objects, g_pins;
static int g_exception_pending;
static char g_exception_class[128], g_exception_message[1024];

/* The local references of the native call in progress (fj_push_frame .. fj_pop_frame): a JVM frees whatever a native method leaves behind
 * when it returns -- FindClass results, the elements it fetched -- and so does fj_pop_frame; the PEAK inside the frame is what a method
 * that loops over many objects has to keep small. */
static fj_object** g_frame;
static long g_frame_len, g_frame_cap;
static int g_frame_active;

static void local_ref_created_for(fj_object* o) {
  if (++g_live_refs > g_peak_refs) g_peak_refs = g_live_refs;
  if (g_frame_active) {
    if (g_frame_len == g_frame_cap) {
      g_frame_cap = g_frame_cap ? 2 * g_frame_cap : 64;
      g_frame = (fj_object**)realloc(g_frame, (size_t)g_frame_cap * sizeof(fj_object*));
    }
    g_frame[g_frame_len++] = o;
  }
}

static fj_object* new_object(int kind, jsize len, size_t elem) {
  fj_object* o = (fj_object*)calloc(1, sizeof(fj_object));
  if (!o) return NULL;
  o->kind = kind;
  o->len = len;
  o->data = calloc((size_t)(len > 0 ? len : 0) + 1, elem);
  if (!o->data) { free(o); return NULL; }
  o->refs = 1;
  g_live_objects++;
  local_ref_created_for(o);
  return o;
}

static void drop(fj_object* o) {
  if (!o || --o->refs > 0) return;
  if (o->kind == FJ_OBJECT_ARRAY) for (jsize i = 0; i < o->len; i++) drop(((fj_object**)o->data)[i]);
  if (o->kind != FJ_BUFFER) free(o->data);
  free(o);
  g_live_objects--;
}

static fj_object* obj(jobject j) { return (fj_object*)(void*)j; }
static jobject ref(fj_object* o) { return (jobject)(void*)o; }

static jclass fj_FindClass(JNIEnv* env, const char* name) {
  (void)env;
  fj_object* o = new_object(FJ_CLASS, (jsize)strlen(name), 1);
  if (o) memcpy(o->data, name, strlen(name));
  return ref(o);
}
static jint fj_ThrowNew(JNIEnv* env, jclass clazz, const char* message) {
  (void)env;
  g_exception_pending = 1;
  snprintf(g_exception_class, sizeof(g_exception_class), "%s", clazz ? (const char*)obj(clazz)->data : "?");
  snprintf(g_exception_message, sizeof(g_exception_message), "%s", message ? message : "");
  return 0;
}
static jboolean fj_ExceptionCheck(JNIEnv* env) { (void)env; return (jboolean)g_exception_pending; }
static void forget_in_frame(fj_object* o) {
  for (long i = g_frame_len - 1; i >= 0; i--) if (g_frame[i] == o) { g_frame[i] = g_frame[--g_frame_len]; return; }
}
static void fj_DeleteLocalRef(JNIEnv* env, jobject r) { (void)env; if (r) { if (g_frame_active) forget_in_frame(obj(r)); g_live_refs--; drop(obj(r)); } }
static jstring fj_NewStringUTF(JNIEnv* env, const char* utf) {
  (void)env;
  fj_object* o = new_object(FJ_STRING, (jsize)strlen(utf), 1);
  if (o) memcpy(o->data, utf, strlen(utf));
  return ref(o);
}
static const char* fj_GetStringUTFChars(JNIEnv* env, jstring s, jboolean* is_copy) {
  (void)env;
  if (is_copy) *is_copy = 0;
  obj(s)->pins++; g_pins++;
  return (const char*)obj(s)->data;
}
static void fj_ReleaseStringUTFChars(JNIEnv* env, jstring s, const char* chars) { (void)env; (void)chars; obj(s)->pins--; g_pins--; }
static jsize fj_GetArrayLength(JNIEnv* env, jarray a) { (void)env; return obj(a)->len; }
static jobjectArray fj_NewObjectArray(JNIEnv* env, jsize len, jclass clazz, jobject init) {
  (void)env; (void)clazz;
  fj_object* o = new_object(FJ_OBJECT_ARRAY, len, sizeof(fj_object*));
  if (o && init) for (jsize i = 0; i < len; i++) { ((fj_object**)o->data)[i] = obj(init); obj(init)->refs++; }
  return ref(o);
}
static jobject fj_GetObjectArrayElement(JNIEnv* env, jobjectArray a, jsize index) {
  (void)env;
  fj_object* e = (index >= 0 && index < obj(a)->len) ? ((fj_object**)obj(a)->data)[index] : NULL;
  if (e) { e->refs++; local_ref_created_for(e); }          /* a NEW local reference, as in a JVM */
  return ref(e);
}
static void fj_SetObjectArrayElement(JNIEnv* env, jobjectArray a, jsize index, jobject value) {
  (void)env;
  if (index < 0 || index >= obj(a)->len) return;
  fj_object** slot = &((fj_object**)obj(a)->data)[index];
  if (value) obj(value)->refs++;
  drop(*slot);
  *slot = obj(value);
}
static jintArray fj_NewIntArray(JNIEnv* env, jsize len) { (void)env; return ref(new_object(FJ_INT_ARRAY, len, sizeof(jint))); }
static jlongArray fj_NewLongArray(JNIEnv* env, jsize len) { (void)env; return ref(new_object(FJ_LONG_ARRAY, len, sizeof(jlong))); }
static jdoubleArray fj_NewDoubleArray(JNIEnv* env, jsize len) { (void)env; return ref(new_object(FJ_DOUBLE_ARRAY, len, sizeof(jdouble))); }
static void* pin(jarray a, jboolean* is_copy) { if (is_copy) *is_copy = 0; obj(a)->pins++; g_pins++; return obj(a)->data; }
static jint* fj_GetIntArrayElements(JNIEnv* env, jintArray a, jboolean* c) { (void)env; return (jint*)pin(a, c); }
static jlong* fj_GetLongArrayElements(JNIEnv* env, jlongArray a, jboolean* c) { (void)env; return (jlong*)pin(a, c); }
static jdouble* fj_GetDoubleArrayElements(JNIEnv* env, jdoubleArray a, jboolean* c) { (void)env; return (jdouble*)pin(a, c); }
static void unpin(jarray a) { obj(a)->pins--; g_pins--; }
static void fj_ReleaseIntArrayElements(JNIEnv* env, jintArray a, jint* e, jint mode) { (void)env; (void)e; (void)mode; unpin(a); }
static void fj_ReleaseLongArrayElements(JNIEnv* env, jlongArray a, jlong* e, jint mode) { (void)env; (void)e; (void)mode; unpin(a); }
static void fj_ReleaseDoubleArrayElements(JNIEnv* env, jdoubleArray a, jdouble* e, jint mode) { (void)env; (void)e; (void)mode; unpin(a); }
static void* fj_GetDirectBufferAddress(JNIEnv* env, jobject buffer) { (void)env; return buffer ? obj(buffer)->data : NULL; }
static void fj_SetLongArrayRegion(JNIEnv* env, jlongArray a, jsize start, jsize len, const jlong* buf) {
  (void)env;
  if (start >= 0 && len >= 0 && start + len <= obj(a)->len) memcpy((jlong*)obj(a)->data + start, buf, (size_t)len * sizeof(jlong));
}

static const struct JNINativeInterface_ g_table = {
  fj_FindClass, fj_ThrowNew, fj_ExceptionCheck, fj_DeleteLocalRef, fj_NewStringUTF, fj_GetStringUTFChars, fj_ReleaseStringUTFChars,
  fj_GetArrayLength, fj_NewObjectArray, fj_GetObjectArrayElement, fj_SetObjectArrayElement, fj_NewIntArray, fj_NewLongArray,
  fj_NewDoubleArray, fj_GetIntArrayElements, fj_GetLongArrayElements, fj_GetDoubleArrayElements, fj_ReleaseIntArrayElements,
  fj_ReleaseLongArrayElements, fj_ReleaseDoubleArrayElements, fj_GetDirectBufferAddress, fj_SetLongArrayRegion};
static JNIEnv g_env = &g_table;

/* ---- what the harness (ctypes) calls ---- */
#define FJ_API __attribute__((visibility("default")))
FJ_API JNIEnv* fj_env(void) { return &g_env; }
/* around every native call: the references the method leaves behind are freed when it returns, except the one it returns */
FJ_API void fj_push_frame(void) { g_frame_active = 1; g_frame_len = 0; }
FJ_API void fj_pop_frame(void* result) {
  int kept = 0;
  g_frame_active = 0;
  for (long i = 0; i < g_frame_len; i++) {
    if (!kept && g_frame[i] == (fj_object*)result) { kept = 1; continue; }
    g_live_refs--;
    drop(g_frame[i]);
  }
  g_frame_len = 0;
}
FJ_API void* fj_int_array(const int32_t* values, int32_t n) { fj_object* o = new_object(FJ_INT_ARRAY, n, sizeof(jint)); if (o && n) memcpy(o->data, values, (size_t)n * sizeof(jint)); return o; }
FJ_API void* fj_long_array(const int64_t* values, int32_t n) { fj_object* o = new_object(FJ_LONG_ARRAY, n, sizeof(jlong)); if (o && n) memcpy(o->data, values, (size_t)n * sizeof(jlong)); return o; }
FJ_API void* fj_object_array(int32_t n) { return new_object(FJ_OBJECT_ARRAY, n, sizeof(fj_object*)); }
FJ_API void* fj_string(const char* utf) { return obj(fj_NewStringUTF(&g_env, utf)); }
FJ_API void* fj_buffer(void* address) { fj_object* o = new_object(FJ_BUFFER, 0, 1); if (o) { free(o->data); o->data = address; } return o; }
FJ_API void fj_set(void* array, int32_t index, void* value) { fj_SetObjectArrayElement(&g_env, ref((fj_object*)array), index, ref((fj_object*)value)); }
FJ_API void* fj_get(void* array, int32_t index) { fj_object* a = (fj_object*)array; return (index >= 0 && index < a->len) ? ((fj_object**)a->data)[index] : NULL; }   /* borrowed */
FJ_API int32_t fj_kind(void* o) { return o ? ((fj_object*)o)->kind : 0; }
FJ_API int32_t fj_len(void* o) { return o ? ((fj_object*)o)->len : -1; }
FJ_API void* fj_data(void* o) { return o ? ((fj_object*)o)->data : NULL; }
FJ_API void fj_release(void* o) { fj_DeleteLocalRef(&g_env, ref((fj_object*)o)); }          /* a reference the harness holds */
FJ_API int32_t fj_exception_pending(void) { return g_exception_pending; }
FJ_API const char* fj_exception_class(void) { return g_exception_class; }
FJ_API const char* fj_exception_message(void) { return g_exception_message; }
FJ_API void fj_exception_clear(void) { g_exception_pending = 0; g_exception_class[0] = 0; g_exception_message[0] = 0; }
FJ_API int64_t fj_live_refs(void) { return g_live_refs; }
FJ_API int64_t fj_peak_refs(void) { return g_peak_refs; }
FJ_API void fj_reset_peak(void) { g_peak_refs = g_live_refs; }
FJ_API int64_t fj_live_objects(void) { return g_live_objects; }
FJ_API int64_t fj_pins(void) { return g_pins; }

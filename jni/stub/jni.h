/* NOT the JDK's jni.h.
 *
 * This environment has no JDK, so there is no jni.h to compile jni/pinot_gpu_jni.c against.  This file declares, with the JNI
 * specification's names and signatures, exactly the types and JNIEnv functions that file uses, so that `gcc -fsyntax-only -Wall -Werror`
 * can type-check it (tests/test_marshal.py does).  The function table below has neither the real table's order nor its size: nothing
 * built against this header may ever meet a real JVM.  The one thing that is linked against it is the JVM stand-in of the tests,
 * jni/fake_jvm.c, which implements exactly this table (libpinot_gpu_jni_fake.so: pinot_gpu_jni.c and fake_jvm.c compiled against the same
 * header, so the two agree by construction).  On a box with a JDK the real <jni.h> is found first (jni/Makefile puts
 * $(JAVA_HOME)/include on the include path and never this directory). */
#ifndef PINOT_GPU_JNI_STUB_H
#define PINOT_GPU_JNI_STUB_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;

struct _jobject;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jobjectArray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jdoubleArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;

struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv* env, const char* name);
  jint (*ThrowNew)(JNIEnv* env, jclass clazz, const char* message);
  jboolean (*ExceptionCheck)(JNIEnv* env);
  void (*DeleteLocalRef)(JNIEnv* env, jobject ref);
  jstring (*NewStringUTF)(JNIEnv* env, const char* utf);
  const char* (*GetStringUTFChars)(JNIEnv* env, jstring str, jboolean* is_copy);
  void (*ReleaseStringUTFChars)(JNIEnv* env, jstring str, const char* chars);
  jsize (*GetArrayLength)(JNIEnv* env, jarray array);
  jobjectArray (*NewObjectArray)(JNIEnv* env, jsize len, jclass clazz, jobject init);
  jobject (*GetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index);
  void (*SetObjectArrayElement)(JNIEnv* env, jobjectArray array, jsize index, jobject value);
  jintArray (*NewIntArray)(JNIEnv* env, jsize len);
  jlongArray (*NewLongArray)(JNIEnv* env, jsize len);
  jdoubleArray (*NewDoubleArray)(JNIEnv* env, jsize len);
  jint* (*GetIntArrayElements)(JNIEnv* env, jintArray array, jboolean* is_copy);
  jlong* (*GetLongArrayElements)(JNIEnv* env, jlongArray array, jboolean* is_copy);
  jdouble* (*GetDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jboolean* is_copy);
  void (*ReleaseIntArrayElements)(JNIEnv* env, jintArray array, jint* elems, jint mode);
  void (*ReleaseLongArrayElements)(JNIEnv* env, jlongArray array, jlong* elems, jint mode);
  void (*ReleaseDoubleArrayElements)(JNIEnv* env, jdoubleArray array, jdouble* elems, jint mode);
  void* (*GetDirectBufferAddress)(JNIEnv* env, jobject buffer);
  void (*SetLongArrayRegion)(JNIEnv* env, jlongArray array, jsize start, jsize len, const jlong* buf);
};

#endif

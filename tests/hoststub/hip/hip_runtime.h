// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets the HOST side of libflagstats_hip.so (engines, block pipeline,
// sessions, multi-device entry, C-ABI routing) compile as plain C++ and run under ThreadSanitizer on a box without a GPU.
// "Device memory" is host memory; a stream is a real FIFO worker thread, so copies and "kernels" run asynchronously to the
// caller exactly where the real runtime would run them, and TSan sees every buffer they touch.  Implemented in
// tests/hoststub/hip_stub.cpp.  Never part of the product.
#ifndef FLAGSTATS_TEST_HIP_STUB_H_
#define FLAGSTATS_TEST_HIP_STUB_H_

#include <stddef.h>
#include <stdint.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorInvalidDevice = 101, hipErrorNotReady = 600, hipErrorStreamCaptureUnsupported = 900 };

struct StubStream;
struct StubEvent;
typedef StubStream* hipStream_t;
typedef StubEvent* hipEvent_t;
typedef int hipDevice_t;

enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
enum hipMemoryType { hipMemoryTypeHost = 0, hipMemoryTypeDevice = 1, hipMemoryTypeManaged = 3, hipMemoryTypeUnregistered = 4 };
enum {
    hipStreamNonBlocking = 1,
    hipEventDisableTiming = 2,
    hipEventBlockingSync = 1,
    hipEventDisableSystemFence = 0x20000000,
    hipHostMallocDefault = 0,
    hipHostMallocNumaUser = 0x20000000
};

struct hipDeviceProp_t {
    char gcnArchName[256];
    int multiProcessorCount;
};

struct hipPointerAttribute_t {
    hipMemoryType type;
    int device;
};

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceGetPCIBusId(char* buf, int len, int d);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);

enum hipDeviceAttribute_t { hipDeviceAttributeIsLargeBar = 1 };
enum { hipDeviceMallocFinegrained = 1 };
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t attr, int device);
hipError_t hipExtMallocWithFlags(void** p, size_t bytes, unsigned flags);
hipError_t hipMalloc(void** p, size_t bytes);
template <class T>
static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
template <class T>
static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags) { return hipHostMalloc(reinterpret_cast<void**>(p), bytes, flags); }
hipError_t hipHostFree(void* p);
enum { hipHostRegisterDefault = 0 };
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
hipError_t hipHostGetDevicePointer(void** dp, void* hp, unsigned flags);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);

hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemset(void* dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t s);

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamGetDevice(hipStream_t s, hipDevice_t* d);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipThreadExchangeStreamCaptureMode(hipStreamCaptureMode* mode);
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);

#endif

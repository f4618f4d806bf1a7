// A CPU stand-in for the few HIP runtime calls the HOST code of the library makes (host_path.hip's staging ring and copy pool,
// blmm_multi.hip's per-device worker threads, blmm_internal.h's types), so that those translation units compile with g++ and
// run under AddressSanitizer / UBSan / ThreadSanitizer on the CPU (tests/sanitize/Makefile; GPU sanitizers are not available
// on the pool).  "Device" memory is host memory, streams and events complete at once, every copy is a memcpy.  TEST
// INFRASTRUCTURE: nothing under bulklmm.jl_amd/ includes this file.
#pragma once
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cstddef>

// function qualifiers of the inline helpers in blmm_internal.h
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
typedef struct stub_stream* hipStream_t;
typedef struct stub_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostRegisterDefault = 0, hipStreamNonBlocking = 1 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };

extern thread_local int stub_current_device;
extern int stub_device_count;

inline const char* hipGetErrorString(hipError_t) { return "stub error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = stub_device_count; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= stub_device_count) return hipErrorInvalidValue; stub_current_device = d; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = stub_current_device; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t b) { *p = std::malloc(b ? b : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t b, unsigned) { *p = std::malloc(b ? b : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t b, hipMemcpyKind) { std::memcpy(d, s, b); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t b, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, b); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t b, hipStream_t) { std::memcpy(d, s, b); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(std::malloc(1)); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }

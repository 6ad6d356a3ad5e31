// flagstat_ctx.h -- internal: the process-global HIP context of flagstat_capi.hip as seen by
// the other host translation units of libflagstats_hip.so.
#ifndef FLAGSTAT_CTX_H_
#define FLAGSTAT_CTX_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>

namespace fsint {
std::recursive_mutex& mutex();
int bind_ctx();                                  // lazy init + hipSetDevice; 0 on success
int fail_text(const char* msg);                  // records + prints, returns -1
int fail_hip(const char* what, hipError_t e);    // records + prints, returns non-zero
int stage_reserve(uint64_t flags);               // two device staging buffers of >= flags uint16
uint16_t* stage_buf(int slot);
hipStream_t stream(int slot);
uint64_t* dev_out(int slot);                     // device uint64[32] per slot
uint64_t* host_out();                            // pinned uint64[2][32]
int count_async(const uint16_t* d, uint64_t n, int slot);   // K1+K2 on stream(slot): dev_out(slot) += counters
int count_async_to(const uint16_t* d, uint64_t n, uint64_t* d_out, int slot);  // same, caller-owned device counters
int count_host_array(const uint16_t* h, uint64_t n, uint64_t* out);  // the FLAGSTATS_u16_x64 body
uint64_t chunk_bytes();                            // host streaming chunk (knob chunk_flags) in bytes
int pinned_reserve(uint64_t bytes, void* bufs[3]);   // persistent pinned chunk buffers
}  // namespace fsint

#endif

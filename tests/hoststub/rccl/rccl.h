// TEST-ONLY stand-in for <rccl/rccl.h>: the types flagstat_multi.hip names (RCCL itself is bound by dlopen at run time
// and is not exercised by the host-stub build).
#ifndef FLAGSTATS_TEST_RCCL_STUB_H_
#define FLAGSTATS_TEST_RCCL_STUB_H_
#include <hip/hip_runtime.h>
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
#endif

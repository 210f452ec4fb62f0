// pg_rccl.h -- RCCL, loaded on first use.
// The library is not LINKED against librccl: a process that never creates a communicator (one GPU: the Python / Julia host
// of a single-GPU run, every test but the RCCL ones) never maps it.  That is not only thrift.  A shared object that brings
// librccl into a process BEFORE PyTorch is imported makes the interpreter abort at exit (`double free or corruption`) --
// any hipcc-built library linked with -lrccl does, an empty one included, with one copy of every runtime library mapped
// (scripts/which_hip_runtime.py; round 2 met it in tests/test_gpu_rccl.py and blamed a second HIP runtime).  Loaded when
// the first communicator is created, RCCL arrives after whatever the host has imported, or is torch's own copy already.
#pragma once
#include <rccl/rccl.h>   // types and enums only

namespace pg {
namespace rccl {

struct Api {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*) = nullptr;   // may be absent in an old RCCL
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// dlopen("librccl.so.1") on first use (the copy the process has mapped already, if any: same SONAME); throws pg::Error
// when RCCL cannot be loaded or lacks a required entry point
const Api& api();
bool loaded();

}  // namespace rccl
}  // namespace pg

// RCCL behind the C ABI (me_comm_* / me_pooled_moments_allreduce* in include/metropolis_engine.h): the one collective of
// the engine -- a sum all-reduce of the fp64 pooled-moment vector (1.5 KB at 16 parameters, 18 KB at 64) -- enqueued
// on a HIP stream of the engine, with no PyTorch in the loop.  The reference has no counterpart (one chain, one process).
//
// librccl is loaded on first use (dlopen), not linked: engines that never pool across GPUs do not pay for mapping a
// 0.3-0.5 GB library, and libmetropolis_hip.so loads on hosts without it.  The types and prototypes come from
// <rccl/rccl.h> at compile time.  Search order: $METROPOLIS_RCCL_LIB, the soname (which resolves to a copy the process
// has loaded already, e.g. PyTorch's), /opt/rocm/lib.
#pragma once

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <string>

#include <rccl/rccl.h>

namespace me {

struct RcclApi {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclGetVersion) get_version = nullptr;
  std::string path;   // what dlopen was given
};

// nullptr (and *err) when librccl cannot be loaded or lacks a symbol
inline const RcclApi *rccl_api(std::string *err) {
  static std::mutex mutex;
  static RcclApi api;
  static bool tried = false, ok = false;
  static std::string failure;
  std::lock_guard<std::mutex> lock(mutex);
  if (!tried) {
    tried = true;
    const char *candidates[] = {std::getenv("METROPOLIS_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *handle = nullptr;
    for (const char *name : candidates) {
      if (!name || !*name) continue;
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) {
        api.path = name;
        break;
      }
      failure += std::string(failure.empty() ? "" : "; ") + dlerror();
    }
    if (handle) {
      auto sym = [&](const char *name) -> void * {
        void *p = dlsym(handle, name);
        if (!p) failure = std::string("librccl lacks ") + name;
        return p;
      };
      api.get_unique_id = (decltype(api.get_unique_id))sym("ncclGetUniqueId");
      api.comm_init_rank = (decltype(api.comm_init_rank))sym("ncclCommInitRank");
      api.comm_destroy = (decltype(api.comm_destroy))sym("ncclCommDestroy");
      api.all_reduce = (decltype(api.all_reduce))sym("ncclAllReduce");
      api.error_string = (decltype(api.error_string))sym("ncclGetErrorString");
      api.get_version = (decltype(api.get_version))sym("ncclGetVersion");
      ok = api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_reduce && api.error_string && api.get_version;
    }
  }
  if (!ok) {
    if (err) *err = "RCCL is not available: " + failure;
    return nullptr;
  }
  return &api;
}

}  // namespace me

// RCCL binding of the user-sharded train step (the reference is single-device, so there is no
// reference line to cite: this is the exchange step the partition over users needs).
//
// librccl is resolved at RUN time (dlopen / dlsym), not linked: a single-GPU caller never loads it, and a process that
// already holds an RCCL - a PyTorch process does: libtorch_hip.so depends on the wheel's own librccl.so.1 - gets THAT
// instance (the loader matches by SONAME), so a communicator made here and one made by the caller's own RCCL code come
// from the same library.  SDRM_RCCL_LIB overrides the name.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <mutex>
#include <string>

namespace sdrm {

struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;   // why loading failed (empty when loaded)
};

// The process-wide RCCL entry points (loaded once).  Returns nullptr and fills `why` when librccl cannot be loaded.
inline const RcclApi* rccl_api(std::string* why) {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {std::getenv("SDRM_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
      api.err = dlerror();
    }
    if (!api.lib) {
      api.err = "cannot load librccl: " + api.err;
      return;
    }
    bool ok = true;
    auto sym = [&](const char* s) {
      void* p = dlsym(api.lib, s);
      if (!p) { ok = false; api.err = std::string("librccl lacks ") + s; }
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (ok) api.err.clear();
    else { dlclose(api.lib); api.lib = nullptr; }
  });
  if (!api.lib) {
    if (why) *why = api.err;
    return nullptr;
  }
  return &api;
}

// What an engine holds once sdrm_comm_init_rank / sdrm_allreduce_init has run.
struct Exchange {
  ncclComm_t comm = nullptr;
  bool comm_owned = false;        // created by sdrm_comm_init_rank (destroyed with the engine) vs adopted from the caller
  int nranks = 0, rank = -1;
  hipStream_t aux = nullptr;      // the first gradient bucket's all-reduce runs here, beside the upper layers' weight gradients
  bool aux_owned = false;
  hipEvent_t ev_bucket = nullptr, ev_done = nullptr;
};

}  // namespace sdrm

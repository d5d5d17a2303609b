// RCCL entry points, resolved at run time.  The library only talks to RCCL when
// a caller asks for an exchange between handles (mcmcgpu_comm_*), so librccl is
// opened with dlopen on first use instead of being a load-time dependency: a
// single-GPU user never maps it.  Types and prototypes come from RCCL's own
// header; the function pointers below have exactly those prototypes.
#ifndef SLAM_AMD_RCCL_LOADER_HPP
#define SLAM_AMD_RCCL_LOADER_HPP

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <type_traits>

namespace mg {

struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;  // why loading failed, empty when loaded
};

// The process-wide table; error is set (and every pointer null) when librccl cannot be opened.
inline const RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) {
      const char* why = dlerror();
      api.error = std::string("cannot open librccl: ") + (why ? why : "unknown reason");
      return;
    }
    bool ok = true;
    auto get = [&](auto& fn, const char* symbol) {
      fn = reinterpret_cast<std::decay_t<decltype(fn)>>(dlsym(lib, symbol));
      if (!fn) {
        ok = false;
        api.error = std::string("librccl lacks ") + symbol;
      }
    };
    get(api.GetUniqueId, "ncclGetUniqueId");
    get(api.CommInitRank, "ncclCommInitRank");
    get(api.CommInitAll, "ncclCommInitAll");
    get(api.CommDestroy, "ncclCommDestroy");
    get(api.AllGather, "ncclAllGather");
    get(api.Broadcast, "ncclBroadcast");
    get(api.AllReduce, "ncclAllReduce");
    get(api.GroupStart, "ncclGroupStart");
    get(api.GroupEnd, "ncclGroupEnd");
    get(api.GetErrorString, "ncclGetErrorString");
    if (!ok) {
      const std::string why = api.error;
      api = RcclApi{};
      api.error = why;
    }
  });
  return api;
}

}  // namespace mg

#endif  // SLAM_AMD_RCCL_LOADER_HPP

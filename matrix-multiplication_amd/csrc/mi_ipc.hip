// Peer mappings of an output buffer for the multi-GPU exchange (include/mi_spmm.h, "Peer mappings"):
// hipIpcGetMemHandle / hipIpcOpenMemHandle / hipIpcCloseMemHandle behind three entry points with EXPLICIT
// lifetimes.  New relative to the reference, which is single-device (`cudaSetDevice(0)`,
// src/sparse_mm.cu:295; no collective or peer access anywhere).
//
// Why not torch's CUDA-IPC tensors: they keep a reference-counter FILE per shared allocation and unlink it from
// a destructor; two consumers that let go of one producer's buffer at the same moment made that unlink throw
// inside a destructor and abort the process (round-5 rehearsal).  Here a mapping lives from mi_ipc_open to the
// matching mi_ipc_close and nothing else: no file, no destructor, no garbage collector.
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "mi_common.h"

namespace {
static_assert(sizeof(hipIpcMemHandle_t) == MI_IPC_HANDLE_BYTES, "MI_IPC_HANDLE_BYTES must be HIP's handle size");

struct Mapping {
  void* base;
  int opens;
};
std::mutex g_mu;
// one process maps an exported allocation ONCE however many buffers of it are opened: HIP refuses (or aliases) a
// second hipIpcOpenMemHandle of the same handle in one process, so the opens are counted here
std::map<std::string, Mapping> g_open;  // handle bytes -> mapping
}  // namespace

extern "C" {

int mi_ipc_export(const void* dev_ptr, void* handle_out, int64_t* offset_out, int64_t* alloc_bytes_out) {
  if (!dev_ptr || !handle_out || !offset_out) return MI_EINVAL;
  void* base = nullptr;
  size_t size = 0;
  // the handle names the ALLOCATION (a caching allocator hands out pieces of larger ones): export its base and say
  // where in it the buffer starts
  MI_HIP_TRY(hipMemGetAddressRange(reinterpret_cast<hipDeviceptr_t*>(&base), &size,
                                   reinterpret_cast<hipDeviceptr_t>(const_cast<void*>(dev_ptr))));
  hipIpcMemHandle_t h;
  MI_HIP_TRY(hipIpcGetMemHandle(&h, base));
  std::memcpy(handle_out, &h, MI_IPC_HANDLE_BYTES);
  *offset_out = static_cast<const char*>(dev_ptr) - static_cast<const char*>(base);
  if (alloc_bytes_out) *alloc_bytes_out = static_cast<int64_t>(size);
  return MI_OK;
}

int mi_ipc_open(const void* handle, void** base_out) {
  if (!handle || !base_out) return MI_EINVAL;
  const std::string key(static_cast<const char*>(handle), MI_IPC_HANDLE_BYTES);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_open.find(key);
  if (it != g_open.end()) {
    it->second.opens++;
    *base_out = it->second.base;
    return MI_OK;
  }
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle, MI_IPC_HANDLE_BYTES);
  void* base = nullptr;
  MI_HIP_TRY(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
  g_open[key] = Mapping{base, 1};
  *base_out = base;
  return MI_OK;
}

int mi_ipc_close(const void* handle) {
  if (!handle) return MI_EINVAL;
  const std::string key(static_cast<const char*>(handle), MI_IPC_HANDLE_BYTES);
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_open.find(key);
  if (it == g_open.end()) return MI_EINVAL;  // never opened here (or closed already)
  if (--it->second.opens > 0) return MI_OK;
  void* base = it->second.base;
  g_open.erase(it);
  MI_HIP_TRY(hipIpcCloseMemHandle(base));
  return MI_OK;
}

int mi_ipc_open_count(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  int n = 0;
  for (const auto& kv : g_open) n += kv.second.opens;
  return n;
}

}  // extern "C"

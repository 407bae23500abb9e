// Guard-page device allocator for PyTorch (torch.cuda.memory.CUDAPluggableAllocator): every tensor gets its own
// mapping made with the HIP virtual-memory API, with UNMAPPED address space on both sides, and sits flush against one
// of the two edges:
//   GUARD_MODE=end   (default)  the tensor's last byte (rounded up to 16) is the mapping's last byte: a read or write past
//                               the end of ANY tensor is a GPU memory access fault instead of a silent hit in a neighbour;
//   GUARD_MODE=start            the tensor starts at the mapping's first byte: so is an access at a negative offset.
// GUARD_POISON=0xEF fills every fresh tensor with that byte (reads of memory nobody wrote), GUARD_SYNC=stream: see guard_free.
// GPU AddressSanitizer is not available on this pool; this is the memory-safety check of the HIP kernels
// (tools/guard_run.py runs train steps under it with HIP_LAUNCH_BLOCKING=1, so that the faulting launch is the one the
// host is in).  Frees synchronise the device first (a stream-ordered caching allocator would keep the block alive) and never
// recycle addresses: a stale pointer into a freed tensor faults as well.
//   hipcc -shared -fPIC -O1 -o tools/libguard_alloc.so tools/guard_alloc.cpp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Rec {
  hipDeviceptr_t va;
  size_t va_size;
  hipMemGenericAllocationHandle_t handle;
  hipDeviceptr_t map_at;
  size_t map_size;
  size_t user_size;
  std::vector<hipStream_t> streams;  // the stream it was allocated on + every stream guard_record() named
};
std::mutex g_mu;
std::unordered_map<void *, Rec> g_live;
size_t g_gran = 0;
size_t g_count = 0, g_bytes = 0;

void die(const char *what, hipError_t e) {
  std::fprintf(stderr, "guard_alloc: %s: %s\n", what, hipGetErrorString(e));
  std::abort();
}
#define GA(x)                      \
  do {                             \
    hipError_t e_ = (x);           \
    if (e_ != hipSuccess) die(#x, e_); \
  } while (0)
}  // namespace

extern "C" void *guard_malloc(ssize_t size, int device, hipStream_t stream) {
  if (size <= 0) return nullptr;
  static const bool at_start = [] {
    const char *m = std::getenv("GUARD_MODE");
    return m != nullptr && std::strcmp(m, "start") == 0;
  }();
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_gran == 0) {
    GA(hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum));
    std::fprintf(stderr, "guard_alloc: granularity %zu bytes, mode %s\n", g_gran, at_start ? "start" : "end");
  }
  const size_t user = ((size_t)size + 15) / 16 * 16;
  Rec r;
  r.map_size = (user + g_gran - 1) / g_gran * g_gran;
  r.va_size = r.map_size + 2 * g_gran;
  GA(hipMemAddressReserve(&r.va, r.va_size, g_gran, nullptr, 0));
  GA(hipMemCreate(&r.handle, r.map_size, &prop, 0));
  r.map_at = (hipDeviceptr_t)((char *)r.va + g_gran);
  GA(hipMemMap(r.map_at, r.map_size, 0, r.handle, 0));
  hipMemAccessDesc desc = {};
  desc.location = prop.location;
  desc.flags = hipMemAccessFlagsProtReadWrite;
  GA(hipMemSetAccess(r.map_at, r.map_size, &desc, 1));
  void *p = at_start ? (void *)r.map_at : (void *)((char *)r.map_at + r.map_size - user);
  // GUARD_POISON=<byte>: fresh tensors are filled with that byte (torch.empty memory is otherwise zero here, stale data of
  // an earlier tensor under a caching allocator): a kernel that reads what nobody wrote sees garbage indices / NaN-like floats
  static const int poison = [] {
    const char *m = std::getenv("GUARD_POISON");
    return m != nullptr ? (int)std::strtol(m, nullptr, 0) & 255 : -1;
  }();
  if (poison >= 0) {
    GA(hipMemset(p, poison, user));
    GA(hipDeviceSynchronize());
  }
  r.user_size = (size_t)size;
  r.streams.push_back(stream);
  g_live[p] = r;
  static const bool log = std::getenv("GUARD_LOG") != nullptr;  // one line per allocation: locate a fault address
  if (log) std::fprintf(stderr, "guard_alloc: + %p .. %p (%zd bytes; mapped %p .. %p)\n", p, (void *)((char *)p + size), size,
                        (void *)r.map_at, (void *)((char *)r.map_at + r.map_size));
  ++g_count;
  g_bytes += r.map_size;
  return p;
}

extern "C" void guard_free(void *p, ssize_t, int, hipStream_t) {
  if (p == nullptr) return;
  // GUARD_SYNC=stream: wait only for the streams a stream-ordered caching allocator would know about -- the one the tensor
  // was allocated on and those named by Tensor.record_stream (guard_record) -- so that a kernel on any OTHER stream that
  // still uses the tensor faults: a missing record_stream / missing event between streams.  Default: the whole device.
  static const bool by_stream = [] {
    const char *m = std::getenv("GUARD_SYNC");
    return m != nullptr && std::strcmp(m, "stream") == 0;
  }();
  if (!by_stream) GA(hipDeviceSynchronize());
  Rec r;
  if (by_stream) {
    std::vector<hipStream_t> streams;
    {
      std::lock_guard<std::mutex> lock(g_mu);
      auto it = g_live.find(p);
      if (it != g_live.end()) streams = it->second.streams;
    }
    for (hipStream_t s : streams) GA(hipStreamSynchronize(s));
  }
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) {
      std::fprintf(stderr, "guard_alloc: free of unknown pointer %p\n", p);
      std::abort();
    }
    r = it->second;
    g_live.erase(it);
    if (std::getenv("GUARD_LOG") != nullptr) std::fprintf(stderr, "guard_alloc: - %p\n", p);
  }
  GA(hipMemUnmap(r.map_at, r.map_size));
  GA(hipMemRelease(r.handle));
  // The address range is NOT given back (hipMemAddressFree): no later tensor ever gets these addresses, so a launch that
  // still holds a pointer into a freed tensor faults too (use after free).  (It is also what keeps this allocator usable:
  // re-mapping a just-released range hung the first kernel that touched it, ROCm 7.2.)  A run reserves a few hundred GiB of
  // the 128 TiB address space.
}

// Tensor.record_stream(stream) of the tensor whose storage holds address `p`
extern "C" int guard_record(void *p, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mu);
  for (auto &kv : g_live) {
    const char *b = (const char *)kv.first;
    if ((const char *)p >= b && (const char *)p < b + kv.second.user_size + 16) {
      for (hipStream_t s : kv.second.streams)
        if (s == stream) return 0;
      kv.second.streams.push_back(stream);
      return 0;
    }
  }
  return 1;
}

extern "C" void guard_stats(size_t *count, size_t *bytes, size_t *live) {
  std::lock_guard<std::mutex> lock(g_mu);
  *count = g_count;
  *bytes = g_bytes;
  *live = g_live.size();
}

// TEST INFRASTRUCTURE (tests/test_gpu_memsafety.py only; never loaded by the product path).
//
// Guard-page device allocations through the HIP virtual-memory API: every buffer is mapped into its own reserved
// address range with one UNMAPPED granule in front of it and one behind it, and is placed either flush against the end of
// its mapping (an overrun of one byte is a GPU memory access fault) or at its start (an underrun is).  The caching
// allocator of PyTorch never gives that: its segments are 2 MiB .. 1 GiB and an out-of-bounds read of a kernel lands in a
// neighbouring tensor unless the tensor happens to end its segment ("What's weak" 6: two such reads were found
// by accident only).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

namespace {
struct GuardAlloc {
  void* va;                              // reserved range [va, va + reserved)
  size_t reserved, mapped, gran;
  hipMemGenericAllocationHandle_t mem;
  int dev;
};
thread_local char g_err[256] = "";
int fail(const char* what, hipError_t e) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return 1;
}
hipMemAllocationProp prop_for(int dev) {
  hipMemAllocationProp p = {};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = dev;
  return p;
}
}  // namespace

extern "C" {

const char* gm_last_error() { return g_err; }

int gm_granularity(int dev, size_t* out) {
  hipMemAllocationProp p = prop_for(dev);
  hipError_t e = hipMemGetAllocationGranularity(out, &p, hipMemAllocationGranularityMinimum);
  return e == hipSuccess ? 0 : fail("hipMemGetAllocationGranularity", e);
}

// bytes > 0; align: power of two (the user pointer is a multiple of it).  flush_end != 0: the buffer ends at the last
// byte of the mapping (less than `align` bytes of slack when bytes is not a multiple of align); 0: it starts at the first.
int gm_alloc(int dev, size_t bytes, size_t align, int flush_end, void** user_ptr, void** handle) {
  size_t gran = 0;
  if (gm_granularity(dev, &gran)) return 1;
  if (bytes == 0 || align == 0 || (align & (align - 1)) || align > gran) {
    snprintf(g_err, sizeof g_err, "bad request: %zu bytes, alignment %zu, granularity %zu", bytes, align, gran);
    return 1;
  }
  hipError_t e = hipSetDevice(dev);
  if (e != hipSuccess) return fail("hipSetDevice", e);
  GuardAlloc* g = new GuardAlloc{};
  g->dev = dev;
  g->gran = gran;
  g->mapped = (bytes + gran - 1) / gran * gran;
  g->reserved = g->mapped + 2 * gran;
  if ((e = hipMemAddressReserve(&g->va, g->reserved, gran, nullptr, 0)) != hipSuccess) {
    delete g;
    return fail("hipMemAddressReserve", e);
  }
  hipMemAllocationProp p = prop_for(dev);
  if ((e = hipMemCreate(&g->mem, g->mapped, &p, 0)) != hipSuccess) {
    hipMemAddressFree(g->va, g->reserved);
    delete g;
    return fail("hipMemCreate", e);
  }
  char* base = static_cast<char*>(g->va) + gran;
  if ((e = hipMemMap(base, g->mapped, 0, g->mem, 0)) != hipSuccess) {
    hipMemRelease(g->mem);
    hipMemAddressFree(g->va, g->reserved);
    delete g;
    return fail("hipMemMap", e);
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = dev;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemSetAccess(base, g->mapped, &acc, 1)) != hipSuccess) {
    hipMemUnmap(base, g->mapped);
    hipMemRelease(g->mem);
    hipMemAddressFree(g->va, g->reserved);
    delete g;
    return fail("hipMemSetAccess", e);
  }
  uintptr_t u = reinterpret_cast<uintptr_t>(base);
  if (flush_end) u = (u + g->mapped - bytes) & ~static_cast<uintptr_t>(align - 1);
  *user_ptr = reinterpret_cast<void*>(u);
  *handle = g;
  return 0;
}

// the mapped range of an allocation (tests fill it with a pattern before use)
int gm_mapped_range(void* handle, void** base, size_t* bytes) {
  GuardAlloc* g = static_cast<GuardAlloc*>(handle);
  *base = static_cast<char*>(g->va) + g->gran;
  *bytes = g->mapped;
  return 0;
}

int gm_free(void* handle) {
  GuardAlloc* g = static_cast<GuardAlloc*>(handle);
  if (!g) return 0;
  char* base = static_cast<char*>(g->va) + g->gran;
  hipError_t e = hipMemUnmap(base, g->mapped);
  if (e == hipSuccess) e = hipMemRelease(g->mem);
  if (e == hipSuccess) e = hipMemAddressFree(g->va, g->reserved);
  delete g;
  return e == hipSuccess ? 0 : fail("gm_free", e);
}

}  // extern "C"

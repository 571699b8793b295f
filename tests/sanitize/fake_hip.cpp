// Stand-in for libamdhip64 / libhipfft in the HOST sanitizer build (tests/sanitize/Makefile): the library's translation units
// are compiled for the host only (hipcc --cuda-host-only -fsanitize=address,undefined -DQI_HOST_SANITIZE), every kernel launch
// lands in hipLaunchKernel below and does nothing, and "device memory" is host memory -- small allocations from malloc (so
// AddressSanitizer sees every table upload: hipMemcpy with a wrong size is a heap-buffer-overflow report), large ones
// (workspaces, panels) reserved with mmap and never touched.  What is exercised is the library's HOST arithmetic: band
// assignment, work-item lists, scratch carving, launch geometry, argument blocks.  Test infrastructure only: never linked
// into libqi_tfr.so.
#include <hip/hip_runtime_api.h>
#include <hipfft/hipfft.h>
#include <sys/mman.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace {
constexpr size_t kMapFrom = (size_t)32 << 20;  // allocations from 32 MiB are address-space reservations
std::mutex g_mu;
std::map<void*, size_t> g_mapped;   // mmap'ed "device" buffers
std::map<void*, size_t> g_malloced;
size_t g_launches = 0;
int g_fft_handles = 0;
}  // namespace

extern "C" size_t qi_fake_hip_launches() { return g_launches; }
// is [ptr, ptr + bytes) inside one live "device" allocation?  (the driver's bounds check for argument blocks)
extern "C" int qi_fake_hip_inside(const void* ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (const auto* m : {&g_mapped, &g_malloced}) {
    auto it = m->upper_bound(const_cast<void*>(ptr));
    if (it == m->begin()) continue;
    --it;
    const char* base = static_cast<const char*>(it->first);
    if ((const char*)ptr >= base && (const char*)ptr + bytes <= base + it->second) return 1;
  }
  return 0;
}

extern "C" {
hipError_t hipMalloc(void** p, size_t bytes) {
  if (!p) return hipErrorInvalidValue;
  if (bytes == 0) bytes = 1;
  std::lock_guard<std::mutex> lk(g_mu);
  if (bytes >= kMapFrom) {
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return hipErrorOutOfMemory;
    g_mapped[m] = bytes;
    *p = m;
  } else {
    void* m = malloc(bytes);
    if (!m) return hipErrorOutOfMemory;
    memset(m, 0, bytes);
    g_malloced[m] = bytes;
    *p = m;
  }
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_mapped.find(p);
  if (it != g_mapped.end()) {
    munmap(p, it->second);
    g_mapped.erase(it);
    return hipSuccess;
  }
  auto jt = g_malloced.find(p);
  if (jt == g_malloced.end()) {
    fprintf(stderr, "fake hip: hipFree of %p, which hipMalloc never returned\n", p);
    abort();
  }
  g_malloced.erase(jt);
  free(p);
  return hipSuccess;
}
static bool is_mapped_range(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_mapped.upper_bound(const_cast<void*>(p));
  if (it == g_mapped.begin()) return false;
  --it;
  return (const char*)p >= (const char*)it->first && (const char*)p + bytes <= (const char*)it->first + it->second;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  // (a copy that ends outside a reserved buffer would be a silent overrun of mmap'ed memory: check those by hand; malloc'ed
  // buffers are AddressSanitizer's)
  for (const void* q : {(const void*)dst, src}) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_mapped.upper_bound(const_cast<void*>(q));
    if (it != g_mapped.begin()) {
      --it;
      const char* base = (const char*)it->first;
      if ((const char*)q >= base && (const char*)q < base + it->second && (const char*)q + bytes > base + it->second) {
        fprintf(stderr, "fake hip: hipMemcpy of %zu bytes runs past the end of a %zu-byte buffer\n", bytes, it->second);
        abort();
      }
    }
  }
  if (bytes > ((size_t)256 << 20) && (is_mapped_range(dst, bytes) || is_mapped_range(src, bytes))) return hipSuccess;  // (panels: not moved)
  memmove(dst, src, bytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* d, const void* s, size_t b, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, b, k); }
hipError_t hipMemset(void* p, int v, size_t bytes) {
  if (bytes > ((size_t)256 << 20) && is_mapped_range(p, bytes)) return hipSuccess;
  if (!is_mapped_range(p, bytes) || bytes <= ((size_t)256 << 20)) memset(p, v, bytes);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* p, int v, size_t b, hipStream_t) { return hipMemset(p, v, b); }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake hip: no error"; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeMaxSharedMemoryPerBlock ? 160 * 1024 : (a == hipDeviceAttributeMultiprocessorCount ? 256 : 0);
  return hipSuccess;
}
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "fake gfx950");
  strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
  p->multiProcessorCount = 256;
  p->totalGlobalMem = (size_t)288 << 30;
  p->sharedMemPerBlock = 160 * 1024;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  p->warpSize = 64;
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3 grid, dim3 block, void**, size_t shmem, hipStream_t) {
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 ||
      grid.y > 65535 || grid.z > 65535 || shmem > 160 * 1024) {
    fprintf(stderr, "fake hip: launch geometry grid (%u, %u, %u) block (%u, %u, %u) shared %zu\n", grid.x, grid.y, grid.z, block.x,
            block.y, block.z, shmem);
    abort();
  }
  ++g_launches;
  return hipSuccess;
}
// clang's host stubs: push / pop of the <<<>>> configuration
static thread_local struct { dim3 g, b; size_t sh; hipStream_t st; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) { t_cfg = {g, b, sh, st}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) {
  *g = t_cfg.g; *b = t_cfg.b; *sh = t_cfg.sh; *st = t_cfg.st;
  return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}

// ---- hipFFT: plans are counted, transforms are not performed
hipfftResult hipfftCreate(hipfftHandle* h) { *h = reinterpret_cast<hipfftHandle>(malloc(16)); ++g_fft_handles; return HIPFFT_SUCCESS; }
hipfftResult hipfftDestroy(hipfftHandle h) { free(h); --g_fft_handles; return HIPFFT_SUCCESS; }
hipfftResult hipfftSetAutoAllocation(hipfftHandle, int) { return HIPFFT_SUCCESS; }
hipfftResult hipfftSetStream(hipfftHandle, hipStream_t) { return HIPFFT_SUCCESS; }
hipfftResult hipfftSetWorkArea(hipfftHandle, void*) { return HIPFFT_SUCCESS; }
hipfftResult hipfftMakePlanMany(hipfftHandle, int rank, int* n, int*, int, int, int*, int, int, hipfftType, int batch, size_t* ws) {
  if (rank != 1 || n[0] < 1 || batch < 1) return HIPFFT_INVALID_VALUE;
  *ws = (size_t)n[0] * 16;  // (some work area, so that the library's sharing of one area across plans runs)
  return HIPFFT_SUCCESS;
}
hipfftResult hipfftExecC2C(hipfftHandle, hipfftComplex*, hipfftComplex*, int) { return HIPFFT_SUCCESS; }
hipfftResult hipfftExecZ2Z(hipfftHandle, hipfftDoubleComplex*, hipfftDoubleComplex*, int) { return HIPFFT_SUCCESS; }
hipfftResult hipfftExecR2C(hipfftHandle, hipfftReal*, hipfftComplex*) { return HIPFFT_SUCCESS; }
hipfftResult hipfftExecD2Z(hipfftHandle, hipfftDoubleReal*, hipfftDoubleComplex*) { return HIPFFT_SUCCESS; }
hipfftResult hipfftExecC2R(hipfftHandle, hipfftComplex*, hipfftReal*) { return HIPFFT_SUCCESS; }
hipfftResult hipfftExecZ2D(hipfftHandle, hipfftDoubleComplex*, hipfftDoubleReal*) { return HIPFFT_SUCCESS; }
}

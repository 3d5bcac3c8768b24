// Stand-in for libamdhip64 used ONLY by `make asan` (host-side AddressSanitizer / UBSan build of the engine's planning,
// packing and argument code; SURVEY.md section 5 "ASan on the host C++ shim").  "Device" memory is host heap, copies are memcpy,
// kernel launches do nothing: what runs under the sanitizers is every line of HOST code in lavie_amd/csrc (parameter inventory,
// weight-arena carving, workspace dry run and bump allocation, split-K / tile planning, C-ABI argument checks), with every
// hipMemcpy* bounds-checked by ASan against the exact-size heap blocks behind it.  Never linked into liblavie_hip.so.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" {

hipError_t hipMalloc(void** p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void* d, const void*, size_t n, size_t, hipMemcpyKind) { memset(d, 0, n); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip stub"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// kernel launch plumbing of clang's host stubs: configuration push / pop and a launch that does nothing
static thread_local struct { dim3 grid, block; size_t shmem; hipStream_t stream; } g_cfg;
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    g_cfg.grid = grid; g_cfg.block = block; g_cfg.shmem = shmem; g_cfg.stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = g_cfg.grid; *block = g_cfg.block; *shmem = g_cfg.shmem; *stream = g_cfg.stream;
    return hipSuccess;
}
static long g_launches = 0;
hipError_t hipLaunchKernel(const void* f, dim3 grid, dim3 block, void**, size_t shmem, hipStream_t) {
    // the launch geometry itself is host logic worth checking
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 ||
        shmem > 160 * 1024) abort();
    ++g_launches;
    if (getenv("LAVIE_HOSTCHECK_TRACE")) fprintf(stderr, "launch %p\n", f);
    return hipSuccess;
}
hipError_t hipExtLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t s, hipEvent_t, hipEvent_t, int) {
    return hipLaunchKernel(f, grid, block, args, shmem, s);
}
long lavie_hostcheck_launches() { return g_launches; }
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void*, void*, void*, const char*, size_t, unsigned) {}
}

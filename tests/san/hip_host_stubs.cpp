// Host-only link of a HIP translation unit for the sanitizer harness (tests/san/Makefile): the device half of hn_pack2.hip is
// not part of this program, so the few HIP runtime entry points it refers to resolve here.  Registering the (absent) device code
// at start-up is a no-op; anything else must never be reached by the layout planner and aborts loudly if it is.
#include <stdio.h>
#include <stdlib.h>

#define HN_UNREACHABLE(name)                                                                    \
    extern "C" int name(...) {                                                                  \
        fprintf(stderr, "host-only sanitizer build reached the HIP runtime (%s)\n", #name);     \
        abort();                                                                                \
    }
extern "C" void** __hipRegisterFatBinary(const void*) {
    static void* handle = nullptr;
    return &handle;
}
extern "C" void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
extern "C" void __hipUnregisterFatBinary(void**) {}
HN_UNREACHABLE(__hipPopCallConfiguration)
HN_UNREACHABLE(hipHostFree)
HN_UNREACHABLE(hipHostMalloc)
HN_UNREACHABLE(hipLaunchKernel)
HN_UNREACHABLE(hipMemcpyAsync)
HN_UNREACHABLE(hipStreamSynchronize)
extern "C" const char* hipGetErrorString(int) { return "host-only build"; }

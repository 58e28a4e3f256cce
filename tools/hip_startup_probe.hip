// tools/hip_startup_probe.hip -- what a process pays the HIP runtime before and after any work of ours (profiles/r04_cli_startup.txt).
// Build: hipcc --offload-arch=gfx950 -O2 tools/hip_startup_probe.hip -o tools/hip_startup_probe   (cross-compiles without a GPU)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
static double now() { using namespace std::chrono; return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int *p) { if (p) *p = 1; }
#define T(label, stmt) do { const double t0 = now(); stmt; printf("%-44s %8.2f ms\n", label, now() - t0); } while (0)
int main(int argc, char **argv)
{
    int nd = 0, *d = nullptr;
    void *big = nullptr, *pin = nullptr;
    hipStream_t s[4];
    hipEvent_t ev[9];
    T("hipInit(0)", (void)hipInit(0));
    T("hipGetDeviceCount", (void)hipGetDeviceCount(&nd));
    T("hipSetDevice(0)", (void)hipSetDevice(0));
    T("first hipStreamCreateWithFlags", (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking));
    T("three more streams", for (int i = 1; i < 4; i++) (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    T("nine events", for (auto &e : ev) (void)hipEventCreate(&e));
    T("hipMalloc 4 bytes (first)", (void)hipMalloc(&d, 4));
    T("hipMalloc 3 GB", (void)hipMalloc(&big, (size_t)3 << 30));
    T("hipHostMalloc 32 MB", (void)hipHostMalloc(&pin, (size_t)32 << 20, hipHostMallocPortable));
    T("first kernel launch + sync (code object load)", { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s[0], d); (void)hipStreamSynchronize(s[0]); });
    T("second kernel launch + sync", { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s[0], d); (void)hipStreamSynchronize(s[0]); });
    T("hipMemset 3 GB + sync (first touch)", { (void)hipMemsetAsync(big, 0, (size_t)3 << 30, s[0]); (void)hipStreamSynchronize(s[0]); });
    printf("devices: %d\n", nd);
    fflush(stdout);
    if (argc > 1) _exit(0);          /* any argument: leave without the runtime's atexit teardown */
    return 0;
}

// When does a flag that a kernel writes into pinned host memory become visible to a polling host, if ANOTHER
// kernel is queued behind it on the same stream?  (round 6: the criterion's answer of the drop-in's CG arrived
// only when the cg::step_1 enqueued behind it had ended - 40 us of idle device per iteration.)
// Variants: allocation flags (default / coherent / non-coherent), store kind (plain + __threadfence_system,
// system-scope atomic release store).   hipcc --offload-arch=gfx950 -O2 pinned_flag_lab.hip -o lab_bin/pinned_flag_lab
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void flag_kernel(volatile unsigned char* flags, int kind)
{
    if (threadIdx.x == 0) {
        if (kind == 0) {
            flags[0] = 1;
            flags[1] = 1;
            __threadfence_system();
        } else {
            __hip_atomic_store(const_cast<unsigned char*>(flags), (unsigned char)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(const_cast<unsigned char*>(flags) + 1, (unsigned char)1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void busy_kernel(long long cycles, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 1000) *sink = 1;
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const unsigned fl[3] = {hipHostMallocDefault, hipHostMallocCoherent, hipHostMallocNonCoherent};
    const char* fn[3] = {"default", "coherent", "non-coherent"};
    for (int a = 0; a < 3; ++a) {
        volatile unsigned char* p = nullptr;
        if (hipHostMalloc((void**)&p, 64, fl[a]) != hipSuccess) {
            std::printf("%s: allocation failed\n", fn[a]);
            continue;
        }
        for (int kind = 0; kind < 2; ++kind) {
            for (int behind = 0; behind < 2; ++behind) {
                double sum = 0;
                for (int rep = 0; rep < 20; ++rep) {
                    p[0] = p[1] = 0xFF;
                    hipStreamSynchronize(s);
                    const auto t0 = std::chrono::steady_clock::now();
                    flag_kernel<<<1, 64, 0, s>>>(p, kind);
                    if (behind) busy_kernel<<<1, 64, 0, s>>>(100 * 200, nullptr);   // 100 MHz clock: 200 us
                    while (p[0] == 0xFF || p[1] == 0xFF) {
                    }
                    sum += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    hipStreamSynchronize(s);
                }
                std::printf("%-13s %-28s %-22s flag seen after %7.1f us\n", fn[a],
                            kind == 0 ? "plain + __threadfence_system" : "system-scope release store",
                            behind ? "200 us kernel behind" : "nothing behind", sum / 20);
            }
        }
        hipHostFree((void*)p);
    }
    return 0;
}

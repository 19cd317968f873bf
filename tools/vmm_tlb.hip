// vmm_tlb (development tool): semantics of hipMemUnmap / hipMemMap on this system.
// (1) does the content of a physical handle survive unmap + map at another address?
// (2) after unmap + map of ANOTHER handle at the same address, do kernels reach the new memory?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); fflush(stdout); exit(1);} } while (0)
__global__ void fill(unsigned* p, size_t n, unsigned v) { size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; if (i < n) p[i] = v; }
__global__ void first_words(const unsigned* p, size_t n, unsigned* out) { out[threadIdx.x] = p[threadIdx.x * (n / 8)]; }
int main()
{
    const size_t GiB = size_t(1) << 30, n = GiB / 4;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned* out;
    CK(hipMalloc(&out, 64));
    void *V, *V2;
    CK(hipMemAddressReserve(&V, GiB, GiB, nullptr, 0));
    CK(hipMemAddressReserve(&V2, GiB, GiB, nullptr, 0));
    hipMemGenericAllocationHandle_t A, B;
    CK(hipMemCreate(&A, GiB, &prop, 0));
    CK(hipMemCreate(&B, GiB, &prop, 0));
    auto show = [&](const char* what, void* p) {
        first_words<<<1, 8>>>((const unsigned*)p, n, out);
        CK(hipGetLastError());
        unsigned h[8];
        CK(hipMemcpy(h, out, 32, hipMemcpyDeviceToHost));
        printf("%-60s %08x %08x %08x %08x %08x %08x %08x %08x\n", what, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        fflush(stdout);
    };
    auto map = [&](void* va, hipMemGenericAllocationHandle_t h) {
        CK(hipMemMap(va, GiB, 0, h, 0));
        CK(hipMemSetAccess(va, GiB, &acc, 1));
    };
    map(V, A);
    fill<<<unsigned(n / 256), 256>>>((unsigned*)V, n, 0xAAAA0001u);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    show("A mapped at V, filled with AAAA0001, read at V:", V);
    CK(hipMemUnmap(V, GiB));
    map(V2, A);
    show("A unmapped, mapped at V2, read at V2:", V2);
    map(V, B);
    fill<<<unsigned(n / 256), 256>>>((unsigned*)V, n, 0xBBBB0002u);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    show("B mapped at V (where A was), filled with BBBB0002, read at V:", V);
    show("   ... and A, still mapped at V2, now reads:", V2);
    CK(hipMemUnmap(V, GiB));
    CK(hipMemUnmap(V2, GiB));
    map(V, A);
    map(V2, B);
    show("swapped: A at V reads:", V);
    show("swapped: B at V2 reads:", V2);
    return 0;
}

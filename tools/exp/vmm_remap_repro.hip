// HIP-only reproducer: does a virtual range that is unmapped, freed, reserved again (same address) and given NEW
// physical memory lose part of the first kernel's writes?  (garlic_device_alloc's pool exists because it did:
// DESIGN.md section 4; this file pins the observation to the runtime, without the library or torch.)
//
//   for each round:  reserve(size) -> create+map chunks -> set access -> fill kernel (value = round) -> verify on the
//                    host -> unmap -> release -> address free
//   variants:  0 plain            (free the range, let the next reserve pick its address)
//              1 same address     (ask hipMemAddressReserve for the previous address)
//              2 as 1 + hipDeviceSynchronize and a dummy touch kernel after hipMemSetAccess, before the real fill
//              3 keep the range reserved, only unmap / map new memory
// Build: hipcc --offload-arch=gfx950 -O2 -o vmm_remap_repro vmm_remap_repro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void fill(double *p, size_t n, double v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (double)(i & 1023);
}
__global__ void verify(const double *p, size_t n, double v, unsigned long long *bad)
{
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += p[i] != v + (double)(i & 1023);
    if (b) atomicAdd(bad, b);
}
__global__ void touch(double *p, size_t n)
{
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 512; i < n; i += (size_t)gridDim.x * blockDim.x * 512) p[i] = -1.0;
}

int main(int argc, char **argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const size_t bytes = (size_t)(argc > 2 ? atof(argv[2]) : 1.0) * (1ull << 30);
    const int rounds = argc > 3 ? atoi(argv[3]) : 6;
    int rt = 0, drv = 0;
    CK(hipRuntimeGetVersion(&rt));
    CK(hipDriverGetVersion(&drv));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  HIP runtime %d driver %d  variant %d  %.2f GiB x %d rounds\n", prop.gcnArchName, rt, drv, variant, bytes / 1073741824.0, rounds);
    hipMemAllocationProp ap{};
    ap.type = hipMemAllocationTypePinned;
    ap.location.type = hipMemLocationTypeDevice;
    ap.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &ap, hipMemAllocationGranularityRecommended));
    const size_t size = (bytes + gran - 1) / gran * gran, chunk = ((256ull << 20) + gran - 1) / gran * gran;
    const size_t n = size / 8;
    std::vector<double> host(n);
    void *prev = nullptr, *range = nullptr;
    long total_bad = 0;
    for (int r = 0; r < rounds; r++) {
        if (!(variant == 3 && range)) CK(hipMemAddressReserve(&range, size, 0, (variant == 1 || variant == 2) ? prev : nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t off = 0; off < size; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            const size_t m = off + chunk <= size ? chunk : size - off;
            CK(hipMemCreate(&h, m, &ap, 0));
            CK(hipMemMap((char *)range + off, m, 0, h, 0));
            hs.push_back(h);
        }
        hipMemAccessDesc acc{};
        acc.location = ap.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(range, size, &acc, 1));
        if (variant == 2) {
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(touch, dim3(256), dim3(256), 0, 0, (double *)range, n);
            CK(hipDeviceSynchronize());
        }
        hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (double *)range, n, 1000.0 * (r + 1));
        CK(hipDeviceSynchronize());
        unsigned long long *dbad, hbad = 0;     // the same check through the shader path, before any copy engine touches the range
        CK(hipMalloc(&dbad, 8));
        CK(hipMemset(dbad, 0, 8));
        hipLaunchKernelGGL(verify, dim3(2048), dim3(256), 0, 0, (const double *)range, n, 1000.0 * (r + 1), dbad);
        CK(hipMemcpy(&hbad, dbad, 8, hipMemcpyDeviceToHost));
        CK(hipFree(dbad));
        CK(hipMemcpy(host.data(), range, size, hipMemcpyDeviceToHost));
        long bad = 0;
        size_t first = 0;
        for (size_t i = 0; i < n; i++)
            if (host[i] != 1000.0 * (r + 1) + (double)(i & 1023)) { if (!bad) first = i; bad++; }
        printf("round %d  range %p%s  wrong by kernel %llu, by hipMemcpy %ld%s", r, range, range == prev ? " (same address as before)" : "", hbad, bad, bad ? "" : "\n");
        if (bad) printf("  first at %zu = %g (expected %g)\n", first, host[first], 1000.0 * (r + 1) + (double)(first & 1023));
        total_bad += bad;
        CK(hipMemUnmap(range, size));
        for (auto h : hs) CK(hipMemRelease(h));
        prev = range;
        if (variant != 3) { CK(hipMemAddressFree(range, size)); range = nullptr; }
    }
    printf("variant %d: %ld wrong elements in all\n", variant, total_bad);
    return total_bad ? 1 : 0;
}

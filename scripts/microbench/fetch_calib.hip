// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the particle kernels
// (MI355X_MICROARCH.md calibrates FETCH_SIZE only for 16-byte coalesced streams: x2).  Every kernel moves a known
// number of bytes over arrays far larger than the 256 MiB Infinity Cache:
//   read16        16 B per lane, coalesced (the calibrated case)
//   read8          8 B per lane, coalesced (the gather's particle loads)
//   read8_pair     lane i loads a[2 i] and a[2 i + 1] with two 8-byte loads (the deposition's two particles per lane)
//   write8         8 B per lane, coalesced stores
//   atomic8        one global fp64 atomic add per lane on consecutive doubles (the deposition's J write-back)
//   atomic8_l2     the same on a 64 MiB array, 16 sweeps (J-sized: lives in the Infinity Cache)
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o fetch_calib fetch_calib.hip ; rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

__global__ void read16(const double2* __restrict__ a, double* __restrict__ out, long n) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 1.2345e-300) out[0] = s;
}
__global__ void read8(const double* __restrict__ a, double* __restrict__ out, long n) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += a[i];
    if (s == 1.2345e-300) out[0] = s;
}
__global__ void read8_pair(const double* __restrict__ a, double* __restrict__ out, long n, int one) {
    double s = 0.0;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; 2 * i + 1 < n; i += (long)gridDim.x * blockDim.x) {
        const double v0 = a[2 * i];
        const double v1 = a[2 * i + one];   // one = 1 at run time: two separate 8-byte loads, as in the kernel
        s += v0 + v1;
    }
    if (s == 1.2345e-300) out[0] = s;
}
__global__ void write8(double* __restrict__ a, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] = 1.0;
}
__global__ void atomic8(double* __restrict__ a, long n, int sweeps) {
    for (int s = 0; s < sweeps; ++s)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
            unsafeAtomicAdd(a + i, 1.0);
}

int main() {
    const long n = 1L << 28;   // 2 GiB of doubles
    double *a, *out;
    CHECK(hipMalloc(&a, n * sizeof(double)));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(a, 0, n * sizeof(double)));
    const dim3 grid(256 * 16), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms;
#define RUN(name, bytes, ...)                                                              \
    do {                                                                                   \
        CHECK(hipEventRecord(e0));                                                         \
        hipLaunchKernelGGL(name, grid, block, 0, 0, __VA_ARGS__);                          \
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));                         \
        CHECK(hipEventElapsedTime(&ms, e0, e1));                                           \
        printf("%-12s known %.1f KiB  %.3f ms  %.2f TB/s\n", #name, (bytes) / 1024.0, ms, (bytes) / ms * 1e-9); \
    } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        RUN(read16, (double)n * 8, (const double2*)a, out, n / 2);
        RUN(read8, (double)n * 8, a, out, n);
        RUN(read8_pair, (double)n * 8, a, out, n, 1);
        RUN(write8, (double)n * 8, a, n);
        RUN(atomic8, (double)n * 8, a, n, 1);
        RUN(atomic8, (double)(1L << 23) * 8 * 16, a, 1L << 23, 16);
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}

// LDS read cost per wave instruction on gfx950: ds_read_b64 vs ds_read_b128 (vs ds_read2_b64),
// 8 or 16 waves per CU, lanes on near-by addresses (the gather's pattern: ~8 lanes per address).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NPTS = 8192;
template <int MODE, int ODD>
__global__ void __launch_bounds__(512) bench(double* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) double lds[NPTS];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int a = tid; a < NPTS; a += blockDim.x) lds[a] = a * 0.5;
    __syncthreads();
    const int base = (((lane >> 3) * 2 + (tid >> 6) * 16) & 1023) + ODD;   // 8 lanes per address; ODD: 8-byte aligned only
    double acc0 = 0, acc1 = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const int b = base + (it & 7) * 2;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (MODE == 0) {
                double v; asm volatile("ds_read_b64 %0, %1 offset:%2\n" : "=v"(v) : "v"((b + j * 130) * 8), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc0 += v;
            } else if constexpr (MODE == 1) {
                double2 v;
                asm volatile("ds_read_b128 %0, %1\n" : "=v"(v) : "v"((b + j * 130) * 8));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc0 += v.x; acc1 += v.y;
            } else {
                double2 v;
                asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1\n" : "=v"(v) : "v"((b + j * 130) * 8));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc0 += v.x; acc1 += v.y;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + tid] = acc0 + acc1;
}
template <int MODE, int ODD>
static void run(const char* name, int threads) {
    const int blocks = 256, iters = 2000;
    double* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(double) * blocks * 512); (void)hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL((bench<MODE, ODD>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipLaunchKernelGGL((bench<MODE, ODD>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= blocks;
    std::vector<double> ho(512);
    (void)hipMemcpy(ho.data(), out, sizeof(double) * 512, hipMemcpyDeviceToHost);
    // expected sum for thread 0 (lane 0, wave 0): values lds[a] = a/2
    double expect = 0;
    for (int it = 0; it < iters; ++it) for (int j = 0; j < 16; ++j) {
        const int a = ODD + (it & 7) * 2 + j * 130;
        expect += a * 0.5 + (MODE == 0 ? 0.0 : (a + 1) * 0.5);
    }
    printf("%-14s %s threads %4d: %6.2f cycles per wave-instruction per CU   (check %s)\n", name, ODD ? "odd " : "even", threads,
           avg / ((double)iters * 16 * (threads / 64)), ho[0] == expect ? "ok" : "MISMATCH");
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int t : {512}) {
        run<0, 0>("ds_read_b64", t); run<1, 0>("ds_read_b128", t); run<2, 0>("ds_read2_b64", t);
        run<0, 1>("ds_read_b64", t); run<1, 1>("ds_read_b128", t); run<2, 1>("ds_read2_b64", t);
    }
    return 0;
}

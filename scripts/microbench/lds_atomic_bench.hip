// LDS fp64 atomic-add throughput on gfx950: cycles per wave-instruction, per CU, for several lane->address patterns.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_bench.hip -o lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NPTS = 3 * 15 * 15 * 15;

template <int MODE>
__global__ void __launch_bounds__(1024) bench(double* out, long long* cyc, int iters, int stride, int group, int wave_off) {
    __shared__ double lds[NPTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int a = tid; a < NPTS; a += blockDim.x) lds[a] = 0.0;
    __syncthreads();
    // lane -> base address: (lane / group) * stride doubles, waves offset from each other
    // group > 0: lanes / group share an address; group < 0: lane % (-group) (lanes -group apart share an address: the
    // deposition's chunk layout, lane = 16 r + c, has the four pairs of a cell on lanes c, c + 16, c + 32, c + 48)
    // wave_off: distance between the address ranges of consecutive waves (97: disjoint; 0: every wave on the same 64
    // addresses, 4 / 16: overlapping ranges -- the deposition's neighbouring chunks overlap in most of their stencil points)
    const int base = (group > 0 ? lane / group : lane % (-group)) * stride + wave * wave_off;
    double v = 1.0 + lane;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double* p = lds + base + j * 225 + (it & 3);
            if constexpr (MODE == 0) unsafeAtomicAdd(p, v);
            else if constexpr (MODE == 1) *(volatile double*)p = v;
            else if constexpr (MODE == 2) unsafeAtomicAdd((float*)p, (float)v);
            else if constexpr (MODE == 3) v += *(volatile double*)p;
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    double s = v;
    for (int a = tid; a < NPTS; a += blockDim.x) s += lds[a];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE>
static void run(const char* name, int threads, int stride, int group, int wave_off = 97) {
    const int blocks = 256, iters = 2000;
    double* out; long long* cyc;
    hipMalloc(&out, sizeof(double) * blocks * 1024);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10, stride, group, wave_off);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, stride, group, wave_off);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= blocks;
    const double winstr = (double)iters * 16 * (threads / 64);   // wave-instructions per CU
    printf("%-10s threads %4d stride %3d group %2d wave_off %2d : %7.2f clock64-ticks per wave-instr per CU, %7.3f ms, %6.2f ns per wave-instr per CU\n",
           name, threads, stride, group, wave_off, avg / winstr, ms, ms * 1e6 / winstr);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {64, 256, 512}) {
        run<0>("add_f64", threads, 1, 1);
        run<1>("write_b64", threads, 1, 1);
        run<3>("read_b64", threads, 1, 1);
        run<2>("add_f32", threads, 1, 1);
    }
    run<0>("add_f64", 512, 1, 2);    // pairs of lanes on the same address
    run<0>("add_f64", 512, 1, 4);
    run<0>("add_f64", 512, 1, 8);
    run<0>("add_f64", 512, 2, 1);    // 16-byte stride: 2-way bank conflict for 8-byte accesses?
    run<0>("add_f64", 512, 4, 1);
    run<0>("add_f64", 512, 15, 1);   // one tile row apart
    run<0>("add_f64", 512, 16, 1);
    run<0>("add_f64", 512, 32, 1);
    run<0>("add_f64", 512, 1, -32);  // lanes l and l + 32 on one address
    run<0>("add_f64", 512, 1, -16);  // lanes l, l + 16, l + 32, l + 48 on one address (the deposition's layout)
    run<0>("add_f64", 512, 1, -8);
    run<0>("add_f64", 768, 1, 1);
    for (int off : {0, 1, 4, 16, 32, 64}) run<0>("add_f64", 768, 1, 1, off);   // waves on overlapping address ranges
    run<0>("add_f64", 768, 1, -16);
    return 0;
}

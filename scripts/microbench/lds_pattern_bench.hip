// ds_add_f64 cost for explicit lane->address tables (in doubles): which row stride of the
// deposition tile avoids bank conflicts, and what same-address lanes cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int NPTS = 12000;
__global__ void __launch_bounds__(512) bench(const int* __restrict__ lane_base, double* out, long long* cyc, int iters) {
    __shared__ double lds[NPTS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int a = tid; a < NPTS; a += blockDim.x) lds[a] = 0.0;
    __syncthreads();
    const int base = lane_base[lane] + wave * 2;
    const double v = 1.0 + lane;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) unsafeAtomicAdd(lds + base + j * 257 + (it & 3), v);
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    double s = 0;
    for (int a = tid; a < NPTS; a += blockDim.x) s += lds[a];
    out[blockIdx.x * blockDim.x + tid] = s;
}
static double run(const std::vector<int>& tab, int threads) {
    const int blocks = 256, iters = 1000;
    int* d; double* out; long long* cyc;
    (void)hipMalloc(&d, 64 * sizeof(int)); (void)hipMalloc(&out, sizeof(double) * blocks * 512); (void)hipMalloc(&cyc, sizeof(long long) * blocks);
    (void)hipMemcpy(d, tab.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(bench, dim3(blocks), dim3(threads), 0, 0, d, out, cyc, 10);
    hipLaunchKernelGGL(bench, dim3(blocks), dim3(threads), 0, 0, d, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= blocks;
    (void)hipFree(d); (void)hipFree(out); (void)hipFree(cyc);
    return avg / ((double)iters * 16 * (threads / 64));
}
int main() {
    std::vector<int> tab(64);
    for (int N : {13, 14, 15, 16, 17, 18, 19, 20, 21, 23}) {
        for (int l = 0; l < 64; ++l) tab[l] = (l % 8) + N * (l / 8);
        printf("rows of 8, row stride %2d            : %6.2f cycles per wave-atomic per CU (512 thr)\n", N, run(tab, 512));
    }
    for (int N : {15, 16, 17}) {   // 4 x-cells apart rows of 4? two k-planes: stride N*N*? skip; rows of 8 with plane wrap after 4 rows
        for (int l = 0; l < 64; ++l) tab[l] = (l % 8) + N * ((l / 8) % 4) + N * 15 * (l / 32);
        printf("4 rows + next plane, row stride %2d   : %6.2f\n", N, run(tab, 512));
    }
    srand(7);
    for (int N : {15, 17}) {
        for (double dup : {0.0, 0.1, 0.25, 0.5}) {   // fraction of lanes that share the previous lane's cell
            int cell = 0;
            for (int l = 0; l < 64; ++l) {
                if (l > 0 && (rand() / (double)RAND_MAX) >= dup) cell += 1 + (rand() % 8 == 0);
                tab[l] = (cell % 8) + N * (cell / 8);
            }
            printf("irregular walk, stride %2d, dup %.2f    : %6.2f\n", N, dup, run(tab, 512));
        }
    }
    for (int l = 0; l < 64; ++l) tab[l] = l % 16;
    printf("4 quarters on the same 16 addresses : %6.2f\n", run(tab, 512));
    for (int l = 0; l < 64; ++l) tab[l] = l % 32;
    printf("2 halves on the same 32 addresses   : %6.2f\n", run(tab, 512));
    for (int l = 0; l < 64; ++l) tab[l] = (l % 16) + 32 * (l / 16);
    printf("quarters 32 doubles apart           : %6.2f\n", run(tab, 512));
    for (int l = 0; l < 64; ++l) tab[l] = ((l % 16) * 7) % 16 + 16 * ((l * 5) % 64 / 16) + 64 * (l / 16);
    printf("quarters with permuted banks        : %6.2f\n", run(tab, 512));
    for (int l = 0; l < 64; ++l) tab[l] = (l % 16) + 16 * ((l / 16 + l) % 4);
    printf("bank-distinct per quarter, scattered : %6.2f\n", run(tab, 512));
    for (int l = 0; l < 64; ++l) tab[l] = l;
    printf("linear                              : %6.2f\n", run(tab, 512));
    for (int l = 0; l < 64; ++l) tab[l] = (l * 37) % 1024;
    printf("pseudo-random distinct              : %6.2f\n", run(tab, 512));
    return 0;
}

// Which way v_mov_b32_dpp wave_shr:1 / wave_shl:1 move data on gfx950, and what lane 0 / lane 63 keep (bound_ctrl off,
// old = a second register): the row-neighbour exchange of the plane kernels in fields.hip relies on
//   wave_shr:1  lane l <- lane l - 1, lane 0 keeps old;   wave_shl:1  lane l <- lane l + 1, lane 63 keeps old.
// Also: cycles of a ds_read_b64 with 64 active lanes against 2 (the halo reads of lanes 0 and 63).
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/dpp_wave_shift.hip -o scripts/microbench/dpp_wave_shift
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void shifts(int* out) {
    const int t = threadIdx.x;
    out[t] = __builtin_amdgcn_update_dpp(1000 + t, t, 0x138, 0xF, 0xF, false);        // wave_shr:1
    out[64 + t] = __builtin_amdgcn_update_dpp(1000 + t, t, 0x130, 0xF, 0xF, false);   // wave_shl:1
}
template <int SPARSE>
__global__ void __launch_bounds__(256) lds_reads(double* out, long long* cyc, int reps) {
    __shared__ double s[4096];
    for (int a = threadIdx.x; a < 4096; a += 256) s[a] = a;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    const bool on = !SPARSE || lane == 0 || lane == 63;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (on) acc += s[(threadIdx.x + 66 * u + r) & 4095];
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    int* d; hipMalloc(&d, 128 * sizeof(int));
    shifts<<<1, 64>>>(d);
    int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shr:1  lane0 %d lane1 %d lane32 %d lane63 %d\n", h[0], h[1], h[32], h[63]);
    printf("wave_shl:1  lane0 %d lane31 %d lane62 %d lane63 %d\n", h[64], h[64 + 31], h[64 + 62], h[64 + 63]);
    double* o; long long* c; hipMalloc(&o, 1024 * 256 * 8); hipMalloc(&c, 1024 * 8);
    for (int sparse = 0; sparse < 2; ++sparse) {
        const int reps = 200;
        if (sparse) lds_reads<1><<<1024, 256>>>(o, c, reps); else lds_reads<0><<<1024, 256>>>(o, c, reps);
        hipDeviceSynchronize();
        long long hc[1024]; hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 1024; ++i) m += hc[i];
        printf("ds_read_b64, %s lanes: %.1f cycles (clock64) per read instruction of a wave, 4 waves per workgroup, 4 workgroups per CU\n",
               sparse ? "2 of 64" : "64 of 64", m / 1024 / (reps * 16));
    }
    return 0;
}

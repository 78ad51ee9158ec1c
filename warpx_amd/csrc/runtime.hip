// Library runtime: error string, workspace lifetime, blocking copies.
#include "workspace.hpp"

#include <stdarg.h>
#include <string.h>

namespace wxa {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace wxa

extern "C" {

const char* wxa_version(void) { return "warpx_amd 0.1.0 (gfx950, fp64)"; }
const char* wxa_last_error(void) { return wxa::g_err; }

wxa_status wxa_workspace_create(wxa_workspace** ws) {
    if (!ws) return WXA_ERR_INVALID_ARG;
    *ws = new wxa_workspace();
    return WXA_OK;
}

void wxa_workspace_destroy(wxa_workspace* ws) {
    if (!ws) return;
    ws->cell.release(); ws->rank.release(); ws->hist.release(); ws->offsets.release();
    ws->scan_tmp.release(); ws->tile_offsets.release(); ws->stragglers.release(); ws->counters.release(); ws->lens_tab.release(); ws->ext_pp.release();
    for (int b = 0; b < 2; ++b) { ws->ps.kr[b].release(); ws->ps.offs[b].release(); ws->ps.own[b].release(); }
    ws->ps.hist.release(); ws->heavy.release();
    delete ws;
}

wxa_status wxa_workspace_set_external_particle_fields(wxa_workspace* ws, const double E[3], const double B[3]) {
    WXA_REQUIRE(ws && E && B, "null argument");
    for (int d = 0; d < 3; ++d) { ws->ext_eb[d] = E[d]; ws->ext_eb[3 + d] = B[d]; }
    return WXA_OK;
}

wxa_status wxa_workspace_set_streaming_plasma(wxa_workspace* ws, int32_t on) {
    WXA_REQUIRE(ws, "null argument");
    ws->streaming_plasma = on ? 1 : 0;
    return WXA_OK;
}

wxa_status wxa_workspace_set_repeated_plasma_lens(wxa_workspace* ws, const wxa_repeated_plasma_lens* lens) {
    WXA_REQUIRE(ws && lens, "null argument");
    WXA_REQUIRE(lens->n_lenses >= 0, "negative number of lenses");
    const int n = lens->n_lenses;
    if (n > 0) {
        WXA_REQUIRE(lens->starts && lens->lengths && lens->strengths_E && lens->strengths_B, "null lens array");
        WXA_REQUIRE(lens->period > 0.0, "repeated_plasma_lens_period must be > 0");
        WXA_REQUIRE(lens->gamma_boost >= 1.0, "gamma_boost must be >= 1");
        std::vector<double> tab((size_t)4 * n);
        for (int i = 0; i < n; ++i) {
            tab[i] = lens->starts[i]; tab[n + i] = lens->lengths[i];
            tab[2 * n + i] = lens->strengths_E[i]; tab[3 * n + i] = lens->strengths_B[i];
        }
        wxa_status rc;
        if ((rc = ws->lens_tab.reserve(sizeof(double) * tab.size())) != WXA_OK) return rc;
        WXA_HIP_CHECK(hipMemcpy(ws->lens_tab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
    }
    ws->lens_n = n;
    ws->lens_period = lens->period;
    ws->lens_dt = lens->dt;
    ws->lens_gamma_boost = n > 0 ? lens->gamma_boost : 1.0;
    return WXA_OK;
}

wxa_status wxa_workspace_set_time(wxa_workspace* ws, double t) {
    WXA_REQUIRE(ws, "null argument");
    ws->ext_time = t;
    return WXA_OK;
}

wxa_status wxa_workspace_set_deposit_accumulator(wxa_workspace* ws, int32_t acc) {
    WXA_REQUIRE(ws, "null argument");
    WXA_REQUIRE(acc == WXA_ACC_FP64 || acc == WXA_ACC_FP32, "accumulator must be WXA_ACC_FP64 or WXA_ACC_FP32");
    ws->deposit_accumulator = acc;
    return WXA_OK;
}

wxa_status wxa_copy_to_host(void* dst_host, const void* src_dev, int64_t bytes) {
    WXA_REQUIRE(bytes >= 0 && (bytes == 0 || (dst_host && src_dev)), "bad copy arguments");
    if (bytes == 0) return WXA_OK;
    WXA_HIP_CHECK(hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
    return WXA_OK;
}

wxa_status wxa_copy_to_device(void* dst_dev, const void* src_host, int64_t bytes) {
    WXA_REQUIRE(bytes >= 0 && (bytes == 0 || (dst_dev && src_host)), "bad copy arguments");
    if (bytes == 0) return WXA_OK;
    WXA_HIP_CHECK(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
    return WXA_OK;
}

wxa_status wxa_device_synchronize(void) {
    WXA_HIP_CHECK(hipDeviceSynchronize());
    return WXA_OK;
}

}  // extern "C"

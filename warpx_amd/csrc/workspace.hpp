// Per-device scratch owned by the caller through wxa_workspace_create/destroy.
#ifndef WXA_WORKSPACE_HPP_
#define WXA_WORKSPACE_HPP_

#include "common.hpp"
#include "shapes.hpp"

#define WXA_TILE 8   // tile edge (cells) of the tile-major cell sort and of the LDS-tile kernels

namespace wxa {

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    wxa_status reserve(size_t bytes) {
        if (bytes <= cap) return WXA_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) {
            set_last_error("hipMalloc of %zu bytes failed", want);
            return WXA_ERR_NOMEM;
        }
        cap = want;
        return WXA_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace wxa

struct wxa_workspace {
    wxa::DevBuf cell, rank, hist, offsets, scan_tmp, tile_offsets, stragglers, counters;
    // description of the last cell sort (consumed by the tile-based deposition)
    bool sorted_valid = false;
    int64_t sorted_np = 0;              // particles covered by the tile offsets (live ones after wxa_sort_live_count)
    int64_t sorted_bins = 0;            // number of cell bins of that sort (the retired bin follows)
    const double* sorted_x = nullptr;   // identity of the sorted particle array
    int32_t sort_nc[3] = {0, 0, 0};
    int32_t sort_cell_lo[3] = {0, 0, 0};
    double sort_plo[3] = {0, 0, 0};
    double sort_dinv[3] = {0, 0, 0};
    // particles.E_external_particle / B_external_particle of the container that owns this workspace
    // (wxa_workspace_set_external_particle_fields); added to the gathered fields in PushPX / PushP
    double ext_eb[6] = {0, 0, 0, 0, 0, 0};
    // accumulator type of the LDS-tile Esirkepov deposition (wxa_workspace_set_deposit_accumulator)
    int32_t deposit_accumulator = WXA_ACC_FP64;
};

namespace wxa {
inline ExtEB ext_of(const wxa_workspace* ws) {
    if (!ws) return ExtEB{0, 0, 0, 0, 0, 0};
    return ExtEB{ws->ext_eb[0], ws->ext_eb[1], ws->ext_eb[2], ws->ext_eb[3], ws->ext_eb[4], ws->ext_eb[5]};
}
// LDS-tile deposition (deposit_tile.hip)
bool deposit_tile_available(const wxa_workspace* ws, const wxa_particle_view* p);
wxa_status deposit_current_tiled(const wxa_particle_view* p, const wxa_field_view J[3],
                                 const wxa_grid_geom* geom, double q, double dt, double relative_time,
                                 int order, int algo, wxa_workspace* ws, hipStream_t stream);
// LDS-tile gather + push (gather_tile.hip)
bool gather_tile_available(const wxa_workspace* ws, const wxa_particle_view* p);
wxa_status gather_push_tiled(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                             const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                             int pusher, bool move, wxa_workspace* ws, hipStream_t stream);
wxa_status gather_push_tiled_part(const wxa_particle_view* p, const wxa_field_view E[3], const wxa_field_view B[3],
                                  const wxa_grid_geom* geom, double q, double m, double dt, int order, int galerkin,
                                  int pusher, int part, wxa_workspace* ws, hipStream_t stream);
}  // namespace wxa
#endif

// LDS-tile current deposition (placeholder until the tile kernels land).
#include "workspace.hpp"

namespace wxa {

bool deposit_tile_available(const wxa_workspace*, const wxa_particle_view*) { return false; }

wxa_status deposit_current_tiled(const wxa_particle_view*, const wxa_field_view*, const wxa_grid_geom*, double,
                                 double, double, int, int, wxa_workspace*, hipStream_t) {
    set_last_error("tile deposition not built");
    return WXA_ERR_UNSUPPORTED;
}

}  // namespace wxa

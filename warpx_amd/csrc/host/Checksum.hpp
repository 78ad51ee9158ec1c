// The reference's regression checksum (Regression/Checksum/checksum.py:60-130) of the state held by a
// simulation: per field, the sum of |value| over the cell-centred covering grid (the plotfile writer
// averages staggered components to the cell centres, CellCenterFunctor.cpp:20-29 ->
// ablastr/coarsen/sample.H:47-99 with coarsening ratio 1); per species, the sums of |x|, |m u| and w over
// the particles (the plotfile holds momenta, Source/Particles/ParticleIO.H:41-82).  Output stage, not on
// the step path: fields and particles are copied to the host and reduced there.  Sums are over this brick.
#ifndef WXA_HOST_CHECKSUM_HPP_
#define WXA_HOST_CHECKSUM_HPP_

#include <cstdio>

#include "sim_capi.hpp"

namespace wxa::host {

inline double cell_centered_abs_sum(const Backend* be, const amrex::MultiFab& mf) {
    const wxa_field_view& v = mf.view();
    std::vector<double> a((size_t)v.kstride * (size_t)v.n[2]);
    if (be->memcpy_d2h(a.data(), v.p, sizeof(double) * a.size()) != 0) throw std::runtime_error("checksum: device copy failed");
    int np[3], ncell[3];
    for (int d = 0; d < 3; ++d) {
        np[d] = 1 + v.stag[d];                                  // sample::Interp: 1 + |sf - sc|, sc = 0
        ncell[d] = v.n[d] - 2 * v.ng[d] - v.stag[d];
    }
    const double wx = 1.0 / np[0], wy = 1.0 / np[1], wz = 1.0 / np[2];
    double sum = 0.0;
    for (int k = 0; k < ncell[2]; ++k)
        for (int j = 0; j < ncell[1]; ++j)
            for (int i = 0; i < ncell[0]; ++i) {
                double c = 0.0;
                for (int kr = 0; kr < np[2]; ++kr)
                    for (int jr = 0; jr < np[1]; ++jr)
                        for (int ir = 0; ir < np[0]; ++ir)
                            c += wx * wy * wz * a[(size_t)(i + ir + v.ng[0]) + (size_t)(j + jr + v.ng[1]) * v.jstride +
                                                 (size_t)(k + kr + v.ng[2]) * v.kstride];
                sum += std::fabs(c);
            }
    return sum;
}

inline std::string checksum_json(SimHandle& h, const std::vector<std::string>& species_names) {
    using warpx::fields::FieldType;
    using ablastr::fields::Direction;
    WarpX& wx = *h.warpx;
    const Backend* be = wx.context().be;
    be->stream_sync(wx.context().stream);
    char buf[128];
    std::string out = "{\n  \"lev=0\": {\n";
    const struct { const char* name; FieldType ft; } groups[3] = {
        {"B", FieldType::Bfield_fp}, {"E", FieldType::Efield_fp}, {"j", FieldType::current_fp}};
    for (const auto& g : groups)
        for (int d = 0; d < 3; ++d) {
            std::snprintf(buf, sizeof buf, "    \"%s%c\": %.17g,\n", g.name, "xyz"[d],
                          cell_centered_abs_sum(be, *wx.fields().get(g.ft, Direction{d}, 0)));
            out += buf;
        }
    // part_per_cell (PartPerCellFunctor): summed over the cells it is the number of macro-particles
    double nparts = 0.0;
    for (int s = 0; s < wx.GetPartContainer().nSpecies(); ++s)
        nparts += (double)wx.GetPartContainer().GetParticleContainer(s).tile().numParticles();
    std::snprintf(buf, sizeof buf, "    \"part_per_cell\": %.17g,\n", nparts);
    out += buf;
    std::snprintf(buf, sizeof buf, "    \"rho\": %.17g\n  }", cell_centered_abs_sum(be, wx.ComputeRho()));
    out += buf;
    for (int s = 0; s < wx.GetPartContainer().nSpecies(); ++s) {
        WarpXParticleContainer& pc = wx.GetPartContainer().GetParticleContainer(s);
        ParticleTile& t = pc.tile();
        const size_t n = (size_t)t.numParticles();
        std::vector<double> col(n);
        double sums[7];
        for (int c = 0; c < 7; ++c) {
            if (n && be->memcpy_d2h(col.data(), t.comp(c), sizeof(double) * n) != 0)
                throw std::runtime_error("checksum: device copy failed");
            double acc = 0.0;
            for (size_t i = 0; i < n; ++i) acc += std::fabs(col[i]);
            sums[c] = c >= 4 ? acc * pc.mass : acc;
        }
        const std::string name = s < (int)species_names.size() ? species_names[s] : "species" + std::to_string(s);
        out += ",\n  \"" + name + "\": {\n";
        const char* keys[7] = {"particle_position_x", "particle_position_y", "particle_position_z", "particle_weight",
                               "particle_momentum_x", "particle_momentum_y", "particle_momentum_z"};
        for (int c = 0; c < 7; ++c) {
            std::snprintf(buf, sizeof buf, "    \"%s\": %.17g%s\n", keys[c], sums[c], c == 6 ? "" : ",");
            out += buf;
        }
        out += "  }";
    }
    out += "\n}\n";
    return out;
}

}  // namespace wxa::host
#endif

// The text and binary layouts of a single-level AMReX plotfile with any number of grids, as the reference restates them
// in Source/Diagnostics/BTD_Plotfile_Header_Impl.cpp (AMReX itself is not on disk):
//   <dir>/Header                       BTDPlotfileHeaderImpl::WriteHeader (:113-176)
//   <dir>/Level_0/Cell_H               BTDMultiFabHeaderImpl::WriteMultiFabHeader (:263-311), VisMF version 1
//   <dir>/Level_0/Cell_D_<id>          "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))<box> <ncomp>\n" + the
//                                      components as native doubles, Fortran order, component slowest
//   <dir>/<species>/Header             BTDSpeciesHeaderImpl::WriteHeader (:416-453)
//   <dir>/<species>/Level_0/Particle_H BTDParticleDataHeaderImpl::WriteHeader (:520-534): the particle BoxArray
//   <dir>/<species>/Level_0/DATA_<id>  per particle x y z w px py pz as doubles
// Used by Plotfile.hpp (one grid: the brick) and by BTDiagnostics.hpp (one grid per flushed buffer: what
// BTDiagnostics::MergeBuffersForPlotfile, BTDiagnostics.cpp:1146-1314, leaves on disk).
#ifndef WXA_HOST_PLOTFILE_FORMAT_HPP_
#define WXA_HOST_PLOTFILE_FORMAT_HPP_

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace wxa::host {

inline void make_dir(const std::string& path) {
    if (::mkdir(path.c_str(), 0755) != 0 && errno != EEXIST) throw std::runtime_error("plotfile: cannot create " + path);
}
// every missing directory on the way to `path` ("diags/lab00003" from a deck's file_prefix)
inline void make_dirs(const std::string& path) {
    for (size_t at = 1; at <= path.size(); ++at)
        if (at == path.size() || path[at] == '/') make_dir(path.substr(0, at));
}

inline std::string box_string(const int lo[3], const int hi[3]) {
    std::ostringstream s;
    s << "((" << lo[0] << ',' << lo[1] << ',' << lo[2] << ") (" << hi[0] << ',' << hi[1] << ',' << hi[2] << ") (0,0,0))";
    return s.str();
}

inline std::string numbered(const std::string& stem, int64_t id, int digits) {
    char buf[32];
    std::snprintf(buf, sizeof(buf), "%0*lld", digits, (long long)id);
    return stem + buf;
}

// one grid of the plotfile: its index box, the fab file that holds it and the extrema VisMF keeps per component
struct PlotGrid {
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    std::string fab_file;                      // "Cell_D_00000"
    std::vector<double> vmin, vmax;
};

// <dir>/Level_0/<fab_file>: comps[c] is component c of the box lo..hi in Fortran order; fills the grid's extrema
inline void write_fab(const std::string& dir, PlotGrid& g, const std::vector<const double*>& comps) {
    const size_t npts = (size_t)(g.hi[0] - g.lo[0] + 1) * (size_t)(g.hi[1] - g.lo[1] + 1) * (size_t)(g.hi[2] - g.lo[2] + 1);
    std::ofstream f(dir + "/Level_0/" + g.fab_file, std::ios::binary | std::ios::trunc);
    if (!f.good()) throw std::runtime_error("plotfile: cannot open " + g.fab_file);
    f << "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))" << box_string(g.lo, g.hi) << ' ' << comps.size() << '\n';
    g.vmin.assign(comps.size(), 0.0);
    g.vmax.assign(comps.size(), 0.0);
    for (size_t c = 0; c < comps.size(); ++c) {
        f.write(reinterpret_cast<const char*>(comps[c]), (std::streamsize)(sizeof(double) * npts));
        double lo = npts ? comps[c][0] : 0.0, hi = lo;
        for (size_t q = 0; q < npts; ++q) { lo = std::min(lo, comps[c][q]); hi = std::max(hi, comps[c][q]); }
        g.vmin[c] = lo;
        g.vmax[c] = hi;
    }
}

// Header + Level_0/Cell_H for the grids written so far.  `dom_lo/hi`: the index box the grids lie in, `rlo/rhi` its
// physical extent; a grid's extent follows from its indices.
inline void write_cell_headers(const std::string& dir, const std::vector<std::string>& names, const std::vector<PlotGrid>& grids,
                               const int dom_lo[3], const int dom_hi[3], const double rlo[3], const double rhi[3],
                               const double dx[3], double time, int64_t step) {
    const size_t ncomp = names.size(), ng = grids.size();
    {
        std::ofstream f(dir + "/Level_0/Cell_H", std::ios::binary | std::ios::trunc);
        f.precision(17);
        f << 1 << '\n' << 1 << '\n' << ncomp << '\n' << 0 << '\n';      // version, how (one fab per file), ncomp, ngrow
        f << '(' << ng << " 0\n";                                       // BoxArray::writeOn
        for (const PlotGrid& g : grids) f << box_string(g.lo, g.hi) << '\n';
        f << ")\n" << ng << '\n';
        for (const PlotGrid& g : grids) f << "FabOnDisk: " << g.fab_file << " 0\n";
        f << '\n' << ng << ',' << ncomp << '\n';
        for (const PlotGrid& g : grids) { for (double v : g.vmin) f << v << ','; f << '\n'; }
        f << '\n' << ng << ',' << ncomp << '\n';
        for (const PlotGrid& g : grids) { for (double v : g.vmax) f << v << ','; f << '\n'; }
    }
    {
        std::ofstream f(dir + "/Header", std::ios::binary | std::ios::trunc);
        f.precision(17);
        f << "HyperCLaw-V1.1\n" << ncomp << '\n';
        for (const auto& n : names) f << n << '\n';
        f << 3 << '\n' << time << '\n' << 0 << '\n';
        for (int d = 0; d < 3; ++d) f << rlo[d] << ' ';
        f << '\n';
        for (int d = 0; d < 3; ++d) f << rhi[d] << ' ';
        f << '\n' << '\n';                                              // no refinement ratios on a single level
        f << box_string(dom_lo, dom_hi) << '\n' << step << '\n';
        for (int d = 0; d < 3; ++d) f << dx[d] << ' ';
        f << '\n' << 0 << '\n' << 0 << '\n';                            // Cartesian, bwidth
        f << 0 << ' ' << ng << ' ' << time << '\n' << step << '\n';
        for (const PlotGrid& g : grids)
            for (int d = 0; d < 3; ++d)
                f << rlo[d] + (g.lo[d] - dom_lo[d]) * dx[d] << ' ' << rlo[d] + (g.hi[d] + 1 - dom_lo[d]) * dx[d] << '\n';
        f << "Level_0/Cell\n";
    }
}

// the particles of one species on one grid: which DATA file holds them and how many there are
struct ParticleGrid {
    int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    int which = 0;
    int64_t count = 0;
};

// <dir>/<name>/Level_0/DATA_<which>: rec holds x y z w px py pz per particle
inline void write_particle_records(const std::string& dir, const std::string& name, int which, const std::vector<double>& rec,
                                   size_t n) {
    std::ofstream f(numbered(dir + "/" + name + "/Level_0/DATA_", which, 5), std::ios::binary | std::ios::trunc);
    if (!f.good()) throw std::runtime_error("plotfile: cannot open the particle data file of " + name);
    f.write(reinterpret_cast<const char*>(rec.data()), (std::streamsize)(sizeof(double) * 7 * n));
}

// <dir>/<name>/Header + Level_0/Particle_H for the grids written so far
inline void write_species_headers(const std::string& dir, const std::string& name, const std::vector<ParticleGrid>& grids) {
    int64_t total = 0;
    for (const ParticleGrid& g : grids) total += g.count;
    {
        std::ofstream f(dir + "/" + name + "/Level_0/Particle_H", std::ios::binary | std::ios::trunc);
        f << '(' << grids.size() << " 0\n";
        for (const ParticleGrid& g : grids) f << box_string(g.lo, g.hi) << '\n';
        f << ")\n";
    }
    {
        std::ofstream f(dir + "/" + name + "/Header", std::ios::binary | std::ios::trunc);
        f.precision(17);
        f << "Version_Two_Dot_One_double\n" << 3 << '\n' << 4 << '\n'
          << "weight\nmomentum_x\nmomentum_y\nmomentum_z\n" << 0 << '\n'      // no int components
          << 0 << '\n' << total << '\n' << (total + 1) << '\n' << 0 << '\n'   // is_checkpoint, count, next id, finest level
          << grids.size() << '\n';
        for (const ParticleGrid& g : grids) f << g.which << ' ' << g.count << ' ' << 0 << '\n';   // file, count, offset
    }
}

}  // namespace wxa::host
#endif

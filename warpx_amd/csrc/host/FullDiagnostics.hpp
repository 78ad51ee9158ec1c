// Full diagnostics of the host layer: `<diag>.diag_type = Full` with `format = plotfile` -- the plotfiles a WarpX run
// leaves under diags/ at the steps of `<diag>.intervals`, so that a deck run here gives the output tree the reference's
// analysis scripts and regression checksum expect.
//   Source/Diagnostics/Diagnostics.cpp:47-60        BaseReadParameters: file_prefix ("diags/<name>"), file_min_digits (6),
//                                                   format, dump_last_timestep (1), fields_to_plot (Ex .. jz)
//   Source/Diagnostics/FullDiagnostics.cpp:60-130   ReadParameters: intervals, write_species / species
//   .../FullDiagnostics.cpp:295-303, Diagnostics.cpp:611-625   DoDump / FilterComputePackFlush, m_already_done
//   Source/Diagnostics/MultiDiagnostics.cpp:83-115  the loop over diagnostics, the forced flush of the last time step
//   call sites: WarpXInitData.cpp:612-613 (before the first step), WarpXEvolve.cpp:306 (every step), :341-343 (istep ==
//   max_step); file name amrex::Concatenate(file_prefix, istep, file_min_digits) (FlushFormatPlotfile.cpp:69)
// The fields are those the plotfile writer knows (Ex Ey Ez Bx By Bz jx jy jz rho divE part_per_cell, cell-centred); other
// names of the reference's list (rho_<species>, F, G, ...) are left out with a warning.  openPMD output is not produced.
#ifndef WXA_HOST_FULL_DIAGNOSTICS_HPP_
#define WXA_HOST_FULL_DIAGNOSTICS_HPP_

#include <cstdio>

#include "Plotfile.hpp"

namespace wxa::host {

class FullDiagnostics {
public:
    std::string m_diag_name;
    std::string m_file_prefix;
    int m_file_min_digits = 6;
    bool m_dump_last_timestep = true;
    utils::parser::IntervalsParser m_intervals;
    std::vector<std::string> m_varnames_fields;        // what is written, in this order
    bool m_write_species = true;
    std::vector<std::string> m_output_species_names;   // empty: every species
    bool m_already_done = false;

    void NewIteration() { m_already_done = false; }                      // Diagnostics.H:110
    bool DoDump(int step, bool force_flush) {                            // FullDiagnostics.cpp:295-303
        if (m_already_done) return false;
        if (force_flush || m_intervals.contains(step + 1)) {
            m_already_done = true;
            return true;
        }
        return false;
    }
};

class MultiDiagnostics {
public:
    static const std::vector<std::string>& known_fields() {
        static const std::vector<std::string> k{"Ex", "Ey", "Ez", "Bx", "By", "Bz", "jx", "jy", "jz", "rho", "divE", "part_per_cell"};
        return k;
    }
    // fields: empty = the reference's default (Ex .. jz); names this writer does not know are dropped with a warning
    void Add(const std::string& name, const std::string& intervals, const std::string& file_prefix, int file_min_digits,
             const std::vector<std::string>& fields, bool write_species, const std::vector<std::string>& species,
             bool dump_last_timestep) {
        for (const auto& d : alldiags)
            if (d.m_diag_name == name) throw std::runtime_error("diagnostics: " + name + " is defined twice");
        FullDiagnostics d;
        d.m_diag_name = name;
        d.m_file_prefix = file_prefix.empty() ? "diags/" + name : file_prefix;
        d.m_file_min_digits = file_min_digits;
        d.m_intervals = utils::parser::IntervalsParser(intervals.empty() ? std::string("0") : intervals);
        d.m_dump_last_timestep = dump_last_timestep;
        d.m_write_species = write_species;
        d.m_output_species_names = species;
        const std::vector<std::string> want =
            fields.empty() ? std::vector<std::string>(known_fields().begin(), known_fields().begin() + 9) : fields;
        for (const std::string& f : want) {
            if (std::find(known_fields().begin(), known_fields().end(), f) != known_fields().end()) d.m_varnames_fields.push_back(f);
            else std::fprintf(stderr, "[warpx_amd] %s.fields_to_plot: %s is not written by this library (left out)\n", name.c_str(), f.c_str());
        }
        if (d.m_varnames_fields.empty())   // (the reference writes particle-only plotfiles for fields_to_plot = none)
            throw std::runtime_error("diagnostics: " + name + ".fields_to_plot names no field this library writes");
        alldiags.push_back(std::move(d));
    }
    int size() const { return (int)alldiags.size(); }
    void NewIteration() { for (auto& d : alldiags) d.NewIteration(); }

    // MultiDiagnostics::FilterComputePackFlush (step) / FilterComputePackFlushLastTimestep (step, forced)
    void FilterComputePackFlush(SimHandle& h, int step, bool last_timestep) {
        for (auto& d : alldiags) {
            if (last_timestep && !d.m_dump_last_timestep) continue;
            if (!d.DoDump(step, last_timestep)) continue;
            const std::string dir = numbered(d.m_file_prefix, h.warpx->getistep(), d.m_file_min_digits);
            const size_t slash = dir.find_last_of('/');
            if (slash != std::string::npos && slash > 0) make_dirs(dir.substr(0, slash));
            write_plotfile(h, dir, h.species_names, &d.m_varnames_fields, d.m_write_species ? &d.m_output_species_names : nullptr,
                           d.m_write_species);
        }
    }

private:
    std::vector<FullDiagnostics> alldiags;
};

// hooks class WarpX's step loop up to the diagnostics of this handle (WarpX.hpp cannot see the plotfile writer)
inline MultiDiagnostics& full_diagnostics(SimHandle& h) {
    if (!h.multi_diags) {
        h.multi_diags = std::make_shared<MultiDiagnostics>();
        SimHandle* hp = &h;
        h.warpx->diag_hook = [hp](int step, int what) {
            MultiDiagnostics& md = *std::static_pointer_cast<MultiDiagnostics>(hp->multi_diags);
            if (what == WarpX::kDiagNewIteration) md.NewIteration();
            else md.FilterComputePackFlush(*hp, step, what == WarpX::kDiagLastTimestep);
        };
    }
    return *std::static_pointer_cast<MultiDiagnostics>(h.multi_diags);
}

}  // namespace wxa::host
#endif

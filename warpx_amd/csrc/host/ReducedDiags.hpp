// Reduced diagnostics of the host layer: the quantities BASELINE.json's parity gate is stated in ("field energies and
// particle moments"), computed on the device and written in the reference's text format.
//   Source/Diagnostics/ReducedDiags/ReducedDiags.{H,cpp}        base class: path, extension, separator, precision,
//                                                               intervals, WriteToFile (:97-125)
//   .../MultiReducedDiags.cpp:36-144                            warpx.reduced_diags_names, <name>.type, the loop
//   .../FieldEnergy.cpp:36-157, ParticleEnergy.cpp:36-220, ParticleMomentum.cpp:36-253, ParticleNumber.cpp:36-139
//   Source/Utils/Parser/IntervalsParser.cpp:17-140              "start:stop:period" slices, comma separated
// Call sites as in the reference: once before the first step (WarpXInitData.cpp:612-619, step -1 -> row "0") and after
// every step (WarpXEvolve.cpp:299-305).  The other 15 types of the reference's dictionary are not on this path.
#ifndef WXA_HOST_REDUCED_DIAGS_HPP_
#define WXA_HOST_REDUCED_DIAGS_HPP_

#include <sys/stat.h>

#include <cctype>
#include <cerrno>
#include <climits>
#include <fstream>
#include <iomanip>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "BrickComm.hpp"

namespace wxa::host {

namespace utils::parser {

// Source/Utils/Parser/IntervalsParser.H:22-83, .cpp:17-64
class SliceParser {
public:
    explicit SliceParser(const std::string& instr) {
        std::vector<std::string> parts(1);
        for (char c : instr) {
            if (c == ':') parts.emplace_back();
            else if (!std::isspace((unsigned char)c)) parts.back().push_back(c);
        }
        auto to_int = [&](const std::string& s, const char* what) {
            try {
                size_t used = 0;
                const double v = std::stod(s, &used);   // parseStringtoInt evaluates an expression; decks write numbers here
                if (used != s.size()) throw std::invalid_argument(s);
                return (int)std::lround(v);
            } catch (const std::exception&) {
                throw std::runtime_error(std::string("intervals: cannot read the ") + what + " in '" + instr + "'");
            }
        };
        if (parts.size() == 1) {
            m_period = to_int(parts[0], "interval period");
        } else if (parts.size() == 2 || parts.size() == 3) {
            if (!parts[0].empty()) m_start = to_int(parts[0], "interval start");
            if (!parts[1].empty()) m_stop = to_int(parts[1], "interval stop");
            if (parts.size() == 3 && !parts[2].empty()) m_period = to_int(parts[2], "interval period");
        } else {
            throw std::runtime_error("intervals: '" + instr + "' is not a valid syntax for a slice.");
        }
    }
    bool contains(const int n) const {
        if (m_period <= 0) return false;
        return (n - m_start) % m_period == 0 && n >= m_start && n <= m_stop;
    }

private:
    int m_start = 0, m_stop = std::numeric_limits<int>::max(), m_period = 1;
};

// IntervalsParser.H:85-160, .cpp:83-107
class IntervalsParser {
public:
    IntervalsParser() = default;
    explicit IntervalsParser(const std::string& instr) {
        std::string one;
        for (size_t i = 0; i <= instr.size(); ++i) {
            if (i == instr.size() || instr[i] == ',') {
                m_slices.emplace_back(one);
                one.clear();
            } else {
                one.push_back(instr[i]);
            }
        }
    }
    bool contains(const int n) const {
        for (const SliceParser& s : m_slices)
            if (s.contains(n)) return true;
        return false;
    }

private:
    std::vector<SliceParser> m_slices;
};

}  // namespace utils::parser

// what the diagnostics read from the run (class WarpX, through a template parameter: this header precedes it)
class ReducedDiags {
public:
    std::string m_path = "./diags/reducedfiles/";   // ReducedDiags.H:28-46
    std::string m_extension = "txt";
    std::string m_rd_name;
    utils::parser::IntervalsParser m_intervals{"1"};
    bool m_write_header = true;
    bool m_write_file = true;     // false: <name>.path was given as "" -- the rows are only kept in m_data
    std::string m_sep = " ";
    int m_precision = 14;
    std::vector<double> m_data;

    ReducedDiags(std::string rd_name, const std::string& intervals, const char* path) : m_rd_name(std::move(rd_name)) {
        if (!intervals.empty()) m_intervals = utils::parser::IntervalsParser(intervals);
        if (!path || !*path) m_write_file = false;
        else m_path = path;
    }
    virtual ~ReducedDiags() = default;

    std::string file_name() const { return m_path + m_rd_name + "." + m_extension; }

    // the header row of the type (written by the constructors in the reference), without the leading '#'
    virtual std::vector<std::string> columns(const std::vector<std::string>& species_names) const = 0;

    // ReducedDiags.cpp:44-58 + the header block of every type's constructor: directory, truncated file, header row
    void InitFile(const std::vector<std::string>& species_names) {
        if (!m_write_file) return;
        make_directories(m_path);
        std::ofstream ofs{file_name(), std::ofstream::out | std::ofstream::trunc};
        if (!ofs) throw std::runtime_error("reduced diagnostics: cannot create " + file_name());
        int c = 0;
        ofs << "#";
        ofs << "[" << c++ << "]step()";
        ofs << m_sep;
        ofs << "[" << c++ << "]time(s)";
        for (const std::string& col : columns(species_names)) {
            ofs << m_sep;
            ofs << "[" << c++ << "]" << col;
        }
        ofs << "\n";
    }

    // ReducedDiags.cpp:97-125
    void WriteToFile(int step, double t_new) const {
        if (!m_write_file) return;
        std::ofstream ofs{file_name(), std::ofstream::out | std::ofstream::app};
        if (!ofs) throw std::runtime_error("reduced diagnostics: cannot append to " + file_name());   // a row is never dropped silently
        ofs << step + 1;
        ofs << m_sep;
        ofs << std::fixed << std::setprecision(m_precision) << std::scientific;
        ofs << t_new;
        for (const auto& item : m_data) ofs << m_sep << item;
        ofs << "\n";
    }

    static void make_directories(const std::string& path) {   // amrex::UtilCreateDirectory: every level of the path
        for (size_t i = 1; i <= path.size(); ++i)
            if (i == path.size() || path[i] == '/') {
                const std::string sub = path.substr(0, i);
                if (sub.empty() || sub == "." || sub == "..") continue;
                if (::mkdir(sub.c_str(), 0755) != 0 && errno != EEXIST)
                    throw std::runtime_error("reduced diagnostics: cannot create directory " + sub);
            }
    }
};

// (ReduceRealSum: BrickComm.hpp)

// FieldEnergy.cpp: [total, E, B] of level 0
class FieldEnergy : public ReducedDiags {
public:
    using ReducedDiags::ReducedDiags;
    std::vector<std::string> columns(const std::vector<std::string>&) const override {
        return {"total_lev0(J)", "E_lev0(J)", "B_lev0(J)"};   // :64-71
    }
    template <class WX>
    void ComputeDiags(WX& wx) {
        using warpx::fields::FieldType;
        using ablastr::fields::Direction;
        const Backend* be = wx.context().be;
        if (!be->reduce_field) throw std::runtime_error("FieldEnergy: not in this backend");
        std::vector<double> sums(2, 0.0);   // sum E^2, sum B^2 over the points this brick owns
        for (int which = 0; which < 2; ++which)
            for (int d = 0; d < 3; ++d) {
                const amrex::MultiFab* mf =
                    wx.fields().get(which == 0 ? FieldType::Efield_aux : FieldType::Bfield_aux, Direction{d}, 0);
                const wxa_field_view& f = mf->view();
                int32_t lo[3], hi[3];
                wx.owned_points(f, lo, hi);
                double s = 0.0;
                if (be->reduce_field(&f, lo, hi, &s, nullptr, wx.context().stream) != 0)
                    throw std::runtime_error("reduce_field failed");
                sums[(size_t)which] += s;   // tmpEx*tmpEx + tmpEy*tmpEy + tmpEz*tmpEz (:127-135)
            }
        ReduceRealSum(wx.comm(), be, sums, wx.context().stream);   // inside MultiFab::norm2
        const auto& dx = wx.context().dx;
        const double dV = dx[0] * dx[1] * dx[2];
        constexpr double ep0 = 8.8541878128e-12, mu0 = 1.25663706212e-06;   // ablastr/constant.H:45-46
        m_data.assign(3, 0.0);
        m_data[1] = 0.5 * sums[0] * ep0 * dV;      // :146-150
        m_data[2] = 0.5 * sums[1] / mu0 * dV;
        m_data[0] = m_data[1] + m_data[2];
    }
};

// the per-species sums the three particle diagnostics share: {w Ekin, w, w m ux, w m uy, w m uz, live count}
template <class WX>
inline std::vector<double> reduce_species(WX& wx) {
    const Backend* be = wx.context().be;
    if (!be->reduce_particles) throw std::runtime_error("particle diagnostics: not in this backend");
    auto& mypc = wx.GetPartContainer();
    const int ns = mypc.nSpecies();
    std::vector<double> all((size_t)ns * 6, 0.0);
    for (int i = 0; i < ns; ++i) {
        auto& pc = mypc.GetParticleContainer(i);
        const wxa_particle_view p = pc.tile().view();
        if (be->reduce_particles(&p, pc.mass, /*photon=*/0, all.data() + (size_t)i * 6, wx.context().stream) != 0)
            throw std::runtime_error("reduce_particles failed");
    }
    ReduceRealSum(wx.comm(), be, all, wx.context().stream);
    return all;
}

// ParticleEnergy.cpp: [total, species..., total_mean, species_mean...]
class ParticleEnergy : public ReducedDiags {
public:
    using ReducedDiags::ReducedDiags;
    std::vector<std::string> columns(const std::vector<std::string>& names) const override {   // :65-83
        std::vector<std::string> c{"total(J)"};
        for (const auto& n : names) c.push_back(n + "(J)");
        c.emplace_back("total_mean(J)");
        for (const auto& n : names) c.push_back(n + "_mean(J)");
        return c;
    }
    template <class WX>
    void ComputeDiags(WX& wx) {
        const std::vector<double> r = reduce_species(wx);
        const int nSpecies = (int)(r.size() / 6);
        m_data.assign((size_t)(2 * nSpecies + 2), 0.0);
        double Wtot = 0.0;
        for (int i_s = 0; i_s < nSpecies; ++i_s) {
            const double Etot = r[(size_t)i_s * 6 + 0], Ws = r[(size_t)i_s * 6 + 1];
            Wtot += Ws;
            m_data[(size_t)(1 + i_s)] = Etot;                                                          // :168-169
            m_data[(size_t)(1 + nSpecies + 1 + i_s)] = Ws > std::numeric_limits<double>::min() ? Etot / Ws : 0.0;   // :176-184
        }
        for (int i_s = 0; i_s < nSpecies; ++i_s) m_data[0] += m_data[(size_t)(1 + i_s)];             // :188-197
        m_data[(size_t)(1 + nSpecies)] = Wtot > std::numeric_limits<double>::min() ? m_data[0] / Wtot : 0.0;   // :202-210
    }
};

// ParticleMomentum.cpp: [total xyz, species xyz..., total_mean xyz, species_mean xyz...]
class ParticleMomentum : public ReducedDiags {
public:
    using ReducedDiags::ReducedDiags;
    std::vector<std::string> columns(const std::vector<std::string>& names) const override {   // :62-110
        std::vector<std::string> c;
        const char* xyz[3] = {"x", "y", "z"};
        for (const char* d : xyz) c.push_back(std::string("total_") + d + "(kg*m/s)");
        for (const auto& n : names)
            for (const char* d : xyz) c.push_back(n + "_" + d + "(kg*m/s)");
        for (const char* d : xyz) c.push_back(std::string("total_mean_") + d + "(kg*m/s)");
        for (const auto& n : names)
            for (const char* d : xyz) c.push_back(n + "_mean_" + d + "(kg*m/s)");
        return c;
    }
    template <class WX>
    void ComputeDiags(WX& wx) {
        const std::vector<double> r = reduce_species(wx);
        const int nSpecies = (int)(r.size() / 6);
        m_data.assign((size_t)(6 * nSpecies + 6), 0.0);
        double Wtot = 0.0;
        for (int i_s = 0; i_s < nSpecies; ++i_s) {
            const double* s = r.data() + (size_t)i_s * 6;
            const double Ws = s[1];
            Wtot += Ws;
            const size_t ot = (size_t)(3 + i_s * 3), om = (size_t)(3 + nSpecies * 3 + 3 + i_s * 3);     // :175-203
            for (int d = 0; d < 3; ++d) {
                m_data[ot + (size_t)d] = s[2 + d];
                m_data[om + (size_t)d] = Ws > std::numeric_limits<double>::min() ? s[2 + d] / Ws : 0.0;
            }
        }
        for (int i_s = 0; i_s < nSpecies; ++i_s)
            for (int d = 0; d < 3; ++d) m_data[(size_t)d] += m_data[(size_t)(3 + i_s * 3 + d)];       // :207-222
        const size_t oa = (size_t)(3 + nSpecies * 3);                                                  // :227-240
        for (int d = 0; d < 3; ++d)
            m_data[oa + (size_t)d] = Wtot > std::numeric_limits<double>::min() ? m_data[(size_t)d] / Wtot : 0.0;
    }
};

// ParticleNumber.cpp: [total macroparticles, species..., total weight, species...]
class ParticleNumber : public ReducedDiags {
public:
    using ReducedDiags::ReducedDiags;
    std::vector<std::string> columns(const std::vector<std::string>& names) const override {   // :66-89
        std::vector<std::string> c{"total_macroparticles()"};
        for (const auto& n : names) c.push_back(n + "_macroparticles()");
        c.emplace_back("total_weight()");
        for (const auto& n : names) c.push_back(n + "_weight()");
        return c;
    }
    template <class WX>
    void ComputeDiags(WX& wx) {
        const std::vector<double> r = reduce_species(wx);
        const int nSpecies = (int)(r.size() / 6);
        m_data.assign((size_t)(2 * nSpecies + 2), 0.0);
        const size_t iw = (size_t)(1 + nSpecies);
        for (int i_s = 0; i_s < nSpecies; ++i_s) {                                                    // :113-127
            m_data[(size_t)(1 + i_s)] = r[(size_t)i_s * 6 + 5];
            m_data[iw + 1 + (size_t)i_s] = r[(size_t)i_s * 6 + 1];
            m_data[0] += m_data[(size_t)(1 + i_s)];
            m_data[iw] += m_data[iw + 1 + (size_t)i_s];
        }
    }
};

// MultiReducedDiags.cpp:36-144
class MultiReducedDiags {
public:
    enum class Type { FieldEnergy, ParticleEnergy, ParticleMomentum, ParticleNumber };
    static bool known_type(const std::string& t) {
        return t == "FieldEnergy" || t == "ParticleEnergy" || t == "ParticleMomentum" || t == "ParticleNumber";
    }
    void Add(const std::string& name, const std::string& type, const std::string& intervals, const char* path) {
        for (const auto& e : m_multi_rd)
            if (e.rd->m_rd_name == name) throw std::runtime_error("reduced diagnostics: " + name + " is defined twice");
        Entry e;
        if (type == "FieldEnergy") { e.type = Type::FieldEnergy; e.rd = std::make_unique<FieldEnergy>(name, intervals, path); }
        else if (type == "ParticleEnergy") { e.type = Type::ParticleEnergy; e.rd = std::make_unique<ParticleEnergy>(name, intervals, path); }
        else if (type == "ParticleMomentum") { e.type = Type::ParticleMomentum; e.rd = std::make_unique<ParticleMomentum>(name, intervals, path); }
        else if (type == "ParticleNumber") { e.type = Type::ParticleNumber; e.rd = std::make_unique<ParticleNumber>(name, intervals, path); }
        else throw std::runtime_error(type + " is not a valid type for reduced diagnostic " + name + " on this path "
                                      "(FieldEnergy, ParticleEnergy, ParticleMomentum, ParticleNumber)");
        m_multi_rd.push_back(std::move(e));
    }
    int size() const { return (int)m_multi_rd.size(); }
    void SetSpeciesNames(const std::vector<std::string>& names) { m_species_names = names; }

    // ComputeDiags(step) + WriteToFile(step) of WarpXEvolve.cpp:299-305; step = -1 before the first step
    template <class WX>
    void ComputeAndWrite(WX& wx, int step) {
        if (m_multi_rd.empty()) return;
        const bool io = wx.comm().rank_of(wx.comm().coord()) == 0;   // ParallelDescriptor::IOProcessor()
        if (!m_initialized) {
            std::vector<std::string> names = m_species_names;
            const int ns = wx.GetPartContainer().nSpecies();
            for (int i = (int)names.size(); i < ns; ++i) names.push_back("species" + std::to_string(i));
            names.resize((size_t)ns);
            if (io)
                for (auto& e : m_multi_rd) e.rd->InitFile(names);
            m_initialized = true;
        }
        for (auto& e : m_multi_rd) {
            if (!e.rd->m_intervals.contains(step + 1)) continue;
            Compute(wx, e);
            if (io) e.rd->WriteToFile(step, wx.gett_new());
        }
    }
    // the last row of diagnostic `name`; compute_now: evaluate it at the current state first
    template <class WX>
    const std::vector<double>& Data(WX& wx, const std::string& name, bool compute_now) {
        for (auto& e : m_multi_rd)
            if (e.rd->m_rd_name == name) {
                if (compute_now) Compute(wx, e);
                return e.rd->m_data;
            }
        throw std::runtime_error("reduced diagnostics: no diagnostic named " + name);
    }

private:
    struct Entry {
        Type type;
        std::unique_ptr<ReducedDiags> rd;
    };
    template <class WX>
    static void Compute(WX& wx, Entry& e) {
        switch (e.type) {
            case Type::FieldEnergy: static_cast<FieldEnergy&>(*e.rd).ComputeDiags(wx); break;
            case Type::ParticleEnergy: static_cast<ParticleEnergy&>(*e.rd).ComputeDiags(wx); break;
            case Type::ParticleMomentum: static_cast<ParticleMomentum&>(*e.rd).ComputeDiags(wx); break;
            case Type::ParticleNumber: static_cast<ParticleNumber&>(*e.rd).ComputeDiags(wx); break;
        }
    }
    std::vector<Entry> m_multi_rd;
    std::vector<std::string> m_species_names;
    bool m_initialized = false;
};

}  // namespace wxa::host
#endif

// Input-deck front end: reads a WarpX inputs file (the `key = values` format of amrex::ParmParse with
// WarpX's FILE includes, my_constants and math expressions) and builds the simulation the deck describes,
// for the part of the parameter space this library covers.  Restates what WarpX::ReadParameters
// (Source/WarpX.cpp:480-1440), PlasmaInjector (Source/Initialization/PlasmaInjector.cpp), WarpX::InitData
// (Source/Initialization/WarpXInitData.cpp) and LaserParticleContainer's constructor
// (Source/Particles/LaserParticleContainer.cpp:70-250) read, with the reference's defaults.
// A key that is neither understood nor known to be harmless (diagnostics, AMReX box sizes, verbosity)
// is an error: a deck that asks for something outside this path must not run silently without it.
#ifndef WXA_HOST_WARPX_INPUTS_HPP_
#define WXA_HOST_WARPX_INPUTS_HPP_

#include <algorithm>
#include <fstream>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <sstream>

#include "Parser.hpp"
#include "sim_capi.hpp"

namespace wxa::host {

// ---- amrex::ParmParse (table of `name = value value ...` definitions; the last definition wins) ----------
class ParmParse {
public:
    void load_file(const std::string& path) {
        std::ifstream in(path);
        if (!in) throw std::runtime_error("inputs: cannot open " + path);
        std::stringstream ss;
        ss << in.rdbuf();
        const size_t slash = path.find_last_of('/');
        add_text(ss.str(), slash == std::string::npos ? std::string(".") : path.substr(0, slash));
    }
    // command-line style override: "name=value value"
    void add_override(const std::string& def) { add_text(def, "."); }

    bool contains(const std::string& key) const { return m_table.count(key) != 0; }
    bool queryarr(const std::string& key, std::vector<std::string>& out) {
        const auto it = m_table.find(key);
        if (it == m_table.end()) return false;
        m_used.insert(key);
        out = it->second;
        return true;
    }
    bool query(const std::string& key, std::string& out) {
        std::vector<std::string> v;
        if (!queryarr(key, v) || v.empty()) return false;
        out = v[0];
        return true;
    }
    // enum-like words are matched case-insensitively with '-' and '_' ignored (query_enum_sloppy)
    bool query_word(const std::string& key, std::string& out) {
        std::string v;
        if (!query(key, v)) return false;
        out.clear();
        for (char c : v)
            if (c != '-' && c != '_') out.push_back((char)std::tolower((unsigned char)c));
        return true;
    }
    // utils::parser::queryWithParser / queryArrWithParser (Source/Utils/Parser/ParserUtils.H): every value is an
    // expression of the user constants
    bool queryArrWithParser(const std::string& key, std::vector<double>& out) {
        std::vector<std::string> v;
        if (!queryarr(key, v)) return false;
        out.clear();
        for (const std::string& e : v) out.push_back(evaluate(e));
        return true;
    }
    bool queryWithParser(const std::string& key, double& out) {
        std::vector<double> v;
        if (!queryArrWithParser(key, v) || v.empty()) return false;
        out = v[0];
        return true;
    }
    bool queryWithParser(const std::string& key, int& out) {
        double v;
        if (!queryWithParser(key, v)) return false;
        out = safe_int(v, key);
        return true;
    }
    void getArrWithParser(const std::string& key, std::vector<double>& out, size_t n) {
        if (!queryArrWithParser(key, out)) throw std::runtime_error("inputs: " + key + " must be set");
        if (out.size() != n) throw std::runtime_error("inputs: " + key + " needs " + std::to_string(n) + " values");
    }
    double getWithParser(const std::string& key) {
        double v;
        if (!queryWithParser(key, v)) throw std::runtime_error("inputs: " + key + " must be set");
        return v;
    }
    static int safe_int(double v, const std::string& what) {   // safeCastToInt(std::round(x))
        const double r = std::round(v);
        if (!(std::fabs(r) < 2.0e9)) throw std::runtime_error("inputs: " + what + " does not fit an integer");
        return (int)r;
    }

    double evaluate(const std::string& expr) { return Parser(expr, {}, constants()).eval(); }
    Parser makeParser(const std::string& expr, const std::vector<std::string>& vars) {
        return Parser(expr, vars, constants());
    }

    // q_e, m_e, ... (Source/Utils/Parser/ParserUtils.cpp:120-135) + my_constants.*, which may use each other
    const std::map<std::string, double>& constants() {
        if (m_constants_ready) return m_constants;
        m_constants = {{"clight", 299'792'458.},        {"epsilon0", 8.8541878128e-12}, {"mu0", 1.25663706212e-06},
                       {"q_e", 1.602176634e-19},        {"m_e", 9.1093837015e-31},      {"m_p", 1.67262192369e-27},
                       {"m_u", 1.66053906660e-27},      {"kb", 1.380649e-23},           {"pi", 3.14159265358979323846}};
        std::map<std::string, std::string> todo;
        const std::string prefix = "my_constants.";
        for (const auto& kv : m_table)
            if (kv.first.compare(0, prefix.size(), prefix) == 0 && !kv.second.empty()) {
                todo[kv.first.substr(prefix.size())] = kv.second[0];
                m_used.insert(kv.first);
            }
        while (!todo.empty()) {
            bool progress = false;
            std::string last_error;
            for (auto it = todo.begin(); it != todo.end();) {
                try {
                    const double v = Parser(it->second, {}, m_constants).eval();
                    m_constants[it->first] = v;
                    it = todo.erase(it);
                    progress = true;
                } catch (const std::exception& e) {
                    last_error = e.what();
                    ++it;
                }
            }
            if (!progress) throw std::runtime_error("inputs: cannot resolve my_constants: " + last_error);
        }
        m_constants_ready = true;
        return m_constants;
    }

    void ignore(const std::string& key) { if (contains(key)) m_used.insert(key); }
    void ignore_prefix(const std::string& prefix) {
        for (const auto& kv : m_table)
            if (kv.first.compare(0, prefix.size(), prefix) == 0) m_used.insert(kv.first);
    }
    std::vector<std::string> unused() const {
        std::vector<std::string> out;
        for (const auto& kv : m_table)
            if (!m_used.count(kv.first)) out.push_back(kv.first);
        return out;
    }

private:
    // tokens: whitespace separated, "quoted strings" kept whole, '#' starts a comment, '=' stands alone
    void add_text(const std::string& text, const std::string& dir) {
        std::vector<std::string> tok;
        std::vector<bool> quoted;
        size_t i = 0;
        const size_t n = text.size();
        while (i < n) {
            const char c = text[i];
            if (std::isspace((unsigned char)c)) { ++i; continue; }
            if (c == '#') { while (i < n && text[i] != '\n') ++i; continue; }
            if (c == '=') { tok.emplace_back("="); quoted.push_back(false); ++i; continue; }
            if (c == '"') {
                const size_t e = text.find('"', i + 1);
                if (e == std::string::npos) throw std::runtime_error("inputs: unterminated string");
                tok.push_back(text.substr(i + 1, e - i - 1));
                quoted.push_back(true);
                i = e + 1;
                continue;
            }
            size_t e = i;
            int depth = 0;   // a name such as f(x,y,z) keeps its parentheses
            while (e < n && (depth > 0 || (!std::isspace((unsigned char)text[e]) && text[e] != '=' && text[e] != '#'))) {
                if (text[e] == '(') ++depth;
                if (text[e] == ')') --depth;
                if (text[e] == '"') break;
                ++e;
            }
            tok.push_back(text.substr(i, e - i));
            quoted.push_back(false);
            i = e;
        }
        for (size_t t = 0; t < tok.size();) {
            if (t + 1 >= tok.size() || tok[t + 1] != "=" || quoted[t + 1] || quoted[t])
                throw std::runtime_error("inputs: expected 'name = value' near '" + tok[t] + "'");
            const std::string name = tok[t];
            t += 2;
            std::vector<std::string> vals;
            while (t < tok.size() && !(t + 1 < tok.size() && tok[t + 1] == "=" && !quoted[t + 1])) vals.push_back(tok[t++]);
            if (name == "FILE") {   // include, relative to the including file
                for (const std::string& f : vals) load_file(f[0] == '/' ? f : dir + "/" + f);
                continue;
            }
            m_table[name] = vals;
            m_constants_ready = false;
        }
    }

    std::map<std::string, std::vector<std::string>> m_table;
    std::set<std::string> m_used;
    std::map<std::string, double> m_constants;
    bool m_constants_ready = false;
};

// ---- the deck -> simulation builder ----------------------------------------------------------------------
namespace inputs_detail {

// Six standard normal draws for particle `index` of a gaussian_beam: a counter-based stream (splitmix64 of stream + 8 index
// + draw), Box-Muller on pairs of 53-bit uniforms in (0, 1].  Host arithmetic only, the same for every backend.
inline void beam_normals(uint64_t stream, uint64_t index, double n[6]) {
    auto mix = [](uint64_t z) {
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto uniform = [&](uint64_t k) { return ((double)(mix(stream + 0xD1B54A32D192ED03ull * (8 * index + k)) >> 11) + 1.0) * (1.0 / 9007199254740992.0); };
    const double two_pi = 6.283185307179586476925286766559;
    for (int k = 0; k < 3; ++k) {
        const double r = std::sqrt(-2.0 * std::log(uniform(2 * k))), t = two_pi * uniform(2 * k + 1);
        n[2 * k] = r * std::cos(t);
        n[2 * k + 1] = r * std::sin(t);
    }
}

inline int axis_of(const std::string& w, const std::string& key) {
    if (w == "x") return 0;
    if (w == "y") return 1;
    if (w == "z") return 2;
    throw std::runtime_error("inputs: " + key + " must be x, y or z");
}

// WarpX::ComputeExternalFieldOnGridUsingParser (Source/Initialization/WarpXInitData.cpp:1062-1180): the whole
// array, guards included, at x = i dx + prob_lo + (1 - nodal) dx / 2
inline void fill_from_parser(const Backend* be, amrex::MultiFab& mf, const Parser& f, const WarpXContext& ctx) {
    const wxa_field_view& v = mf.view();
    std::vector<double> host((size_t)v.kstride * (size_t)v.n[2], 0.0);
    double fac[3];
    for (int d = 0; d < 3; ++d) fac[d] = (1.0 - v.stag[d]) * ctx.dx[d] * 0.5;
    for (int k = 0; k < v.n[2]; ++k)
        for (int j = 0; j < v.n[1]; ++j)
            for (int i = 0; i < v.n[0]; ++i) {
                const double xyzt[4] = {(i + v.lo[0]) * ctx.dx[0] + ctx.prob_lo[0] + fac[0],
                                        (j + v.lo[1]) * ctx.dx[1] + ctx.prob_lo[1] + fac[1],
                                        (k + v.lo[2]) * ctx.dx[2] + ctx.prob_lo[2] + fac[2], 0.0};
                host[(size_t)i + (size_t)j * v.jstride + (size_t)k * v.kstride] = f.eval(xyzt);
            }
    if (be->memcpy_h2d(v.p, host.data(), sizeof(double) * host.size()) != 0)
        throw std::runtime_error("inputs: copying an external field to the device failed");
}

// One brick per rank: the prime factors of nranks, largest first, go one by one to the splittable direction
// whose bricks are currently the thickest (and still divisible); rank = cx + nbx (cy + nby cz) as in
// BrickComm::rank_of.  Periodic directions only: a wall or the moving window must stay inside one brick.
inline void choose_bricks(int nranks, int rank, const int32_t n_cell[3], const bool splittable[3], int32_t nb[3],
                          int32_t coord[3]) {
    for (int d = 0; d < 3; ++d) nb[d] = 1;
    std::vector<int> factors;
    for (int n = nranks, f = 2; n > 1;) {
        if (n % f == 0) { factors.push_back(f); n /= f; } else { ++f; }
    }
    std::sort(factors.rbegin(), factors.rend());
    for (int f : factors) {
        int best = -1;
        for (int d = 2; d >= 0; --d) {   // ties go to z, then y (contiguous x rows stay long)
            if (!splittable[d] || (n_cell[d] / nb[d]) % f != 0) continue;
            if (best < 0 || n_cell[d] / nb[d] > n_cell[best] / nb[best]) best = d;
        }
        if (best < 0)
            throw std::runtime_error("inputs: cannot split the domain into " + std::to_string(nranks) +
                                     " bricks");
        nb[best] *= f;
    }
    coord[0] = rank % nb[0];
    coord[1] = (rank / nb[0]) % nb[1];
    coord[2] = rank / (nb[0] * nb[1]);
}

}  // namespace inputs_detail

struct DeckInfo {
    int max_step = -1;
    std::vector<std::string> species_names;
};

// "<diag>.intervals = 0:nsteps:nsteps/4, 5" -> "0:40:10,5": the words of an intervals entry concatenated (IntervalsParser.cpp:
// 86-87), every number of every slice evaluated with the deck's constants (parseStringtoInt, IntervalsParser.cpp:27-47)
inline std::string deck_intervals(ParmParse& pp, const std::vector<std::string>& words, const std::string& key) {
    std::string all, out, part;
    for (const std::string& w : words) all += w;
    auto flush = [&]() {
        if (!part.empty()) out += std::to_string(ParmParse::safe_int(pp.evaluate(part), key));
        part.clear();
    };
    for (char c : all) {
        if (c == ':' || c == ',') { flush(); out.push_back(c); }
        else if (!std::isspace((unsigned char)c)) part.push_back(c);
    }
    flush();
    return out;
}

// <diag>.diag_type = Full (FullDiagnostics.hpp, which sits above the plotfile writer and is included below)
inline void add_full_diagnostic(SimHandle& h, const std::string& name, const std::string& intervals, const std::string& file_prefix,
                                int file_min_digits, const std::vector<std::string>& fields, bool write_species,
                                const std::vector<std::string>& species, bool dump_last_timestep);

// nbricks / coord: this library's decomposition (one brick per GPU), not the deck's amr.max_grid_size
inline std::unique_ptr<SimHandle> sim_from_inputs(const Backend* be, const std::string& path,
                                                  const std::vector<std::string>& overrides, const wxa_comm* comm,
                                                  const int32_t nbricks[3], const int32_t coord[3], DeckInfo& info) {
    using namespace inputs_detail;
    ParmParse pp;
    pp.load_file(path);
    for (const std::string& o : overrides) pp.add_override(o);

    wxa_sim_config cfg{};
    // ---- geometry, amr (WarpX.cpp:480-560; AmrCore) ----
    int dims = 3, max_level = 0;
    pp.queryWithParser("geometry.dims", dims);
    pp.queryWithParser("amr.max_level", max_level);
    if (dims != 3) throw std::runtime_error("inputs: only geometry.dims = 3 is on this path");
    if (max_level != 0) throw std::runtime_error("inputs: mesh refinement (amr.max_level > 0) is not on this path");
    std::vector<double> v;
    pp.getArrWithParser("amr.n_cell", v, 3);
    for (int d = 0; d < 3; ++d) cfg.n_cell[d] = ParmParse::safe_int(v[d], "amr.n_cell");
    pp.getArrWithParser("geometry.prob_lo", v, 3);
    for (int d = 0; d < 3; ++d) cfg.prob_lo[d] = v[d];
    pp.getArrWithParser("geometry.prob_hi", v, 3);
    for (int d = 0; d < 3; ++d) cfg.prob_hi[d] = v[d];
    std::string w;
    // ---- boosted frame (ReadBoostedFrameParameters + ConvertLabParamsToBoost, WarpXUtil.cpp:114-141,180-262) ----
    double gamma_boost = 1.0, beta_boost = 0.0;
    if (pp.queryWithParser("warpx.gamma_boost", gamma_boost) && gamma_boost > 1.0) {
        beta_boost = std::sqrt(1.0 - 1.0 / std::pow(gamma_boost, 2.0));
        if (!pp.query_word("warpx.boost_direction", w)) throw std::runtime_error("inputs: warpx.boost_direction must be set");
        if (w != "z") throw std::runtime_error("inputs: The boost must be in the z direction.");
        int window = 0;
        pp.queryWithParser("warpx.do_moving_window", window);
        double beta_window = beta_boost;
        if (window) {
            std::string wd;
            if (pp.query_word("warpx.moving_window_dir", wd) && wd == "z") beta_window = pp.getWithParser("warpx.moving_window_v");
        }
        const double convert_factor = 1.0 / (gamma_boost * (1 - beta_boost * beta_window));
        cfg.prob_lo[2] *= convert_factor;
        cfg.prob_hi[2] *= convert_factor;
        cfg.gamma_boost = gamma_boost;
    } else {
        gamma_boost = 1.0;
        pp.ignore("warpx.boost_direction");
    }
    if (pp.query_word("geometry.coord_sys", w) && w != "0" && w != "cartesian")
        throw std::runtime_error("inputs: only cartesian geometry is on this path");

    // ---- boundaries (WarpX.cpp ReadBoundaryConditions, Source/Utils/WarpXAlgorithmSelection.cpp) ----
    std::vector<std::string> words;
    auto field_bc = [&](const char* key, int32_t out[3]) {
        if (!pp.queryarr(key, words)) throw std::runtime_error(std::string("inputs: ") + key + " must be set");
        if (words.size() != 3) throw std::runtime_error(std::string("inputs: ") + key + " needs 3 values");
        for (int d = 0; d < 3; ++d) {
            std::string b = words[d];
            std::transform(b.begin(), b.end(), b.begin(), [](unsigned char c) { return (char)std::tolower(c); });
            if (b == "periodic") out[d] = WXA_BOUNDARY_PERIODIC;
            else if (b == "pec") out[d] = WXA_BOUNDARY_PEC;
            else throw std::runtime_error("inputs: field boundary '" + words[d] + "' is not on this path (periodic, pec)");
        }
    };
    field_bc("boundary.field_lo", cfg.field_boundary_lo);
    field_bc("boundary.field_hi", cfg.field_boundary_hi);
    auto particle_bc = [&](const char* key, int32_t out[3]) {
        for (int d = 0; d < 3; ++d) out[d] = WXA_PBOUNDARY_DEFAULT;
        if (!pp.queryarr(key, words)) return;
        if (words.size() != 3) throw std::runtime_error(std::string("inputs: ") + key + " needs 3 values");
        for (int d = 0; d < 3; ++d) {
            std::string b = words[d];
            std::transform(b.begin(), b.end(), b.begin(), [](unsigned char c) { return (char)std::tolower(c); });
            if (b == "periodic") out[d] = WXA_PBOUNDARY_PERIODIC;
            else if (b == "absorbing") out[d] = WXA_PBOUNDARY_ABSORBING;
            else if (b == "reflecting") out[d] = WXA_PBOUNDARY_REFLECTING;
            else throw std::runtime_error("inputs: particle boundary '" + words[d] + "' is not on this path");
        }
    };
    particle_bc("boundary.particle_lo", cfg.particle_boundary_lo);
    particle_bc("boundary.particle_hi", cfg.particle_boundary_hi);

    // ---- algorithms (WarpX.cpp:1100-1330) with the reference's defaults ----
    cfg.cfl = 0.999;                                            // WarpX.H: cfl
    pp.queryWithParser("warpx.cfl", cfg.cfl);
    cfg.maxwell_solver = WXA_SOLVER_YEE;
    if (pp.query_word("algo.maxwell_solver", w)) {
        if (w == "ckc") cfg.maxwell_solver = WXA_SOLVER_CKC;
        else if (w != "yee")
            throw std::runtime_error("inputs: algo.maxwell_solver = " + w + " is not on this path (yee, ckc)");
    }
    cfg.grid_type = WXA_GRID_STAGGERED;
    if (pp.query_word("warpx.grid_type", w) && w != "staggered")
        throw std::runtime_error("inputs: warpx.grid_type = " + w + " is not on this path (staggered)");
    cfg.current_deposition = WXA_DEPOSIT_ESIRKEPOV;             // default with an FDTD solver
    if (pp.query_word("algo.current_deposition", w)) {
        if (w == "esirkepov") cfg.current_deposition = WXA_DEPOSIT_ESIRKEPOV;
        else if (w == "direct") cfg.current_deposition = WXA_DEPOSIT_DIRECT;
        else throw std::runtime_error("inputs: algo.current_deposition = " + w + " is not on this path");
    }
    if (pp.query_word("algo.charge_deposition", w) && w != "standard")
        throw std::runtime_error("inputs: algo.charge_deposition = " + w + " is not on this path");
    cfg.particle_pusher = WXA_PUSHER_BORIS;
    if (pp.query_word("algo.particle_pusher", w)) {
        if (w == "boris") cfg.particle_pusher = WXA_PUSHER_BORIS;
        else if (w == "vay") cfg.particle_pusher = WXA_PUSHER_VAY;
        else if (w == "higuera") cfg.particle_pusher = WXA_PUSHER_HC;
        else throw std::runtime_error("inputs: algo.particle_pusher = " + w + " is not on this path (boris, vay, higuera)");
    }
    // same shape factors in all directions with direct deposition and an EM solver (WarpX.cpp:1208-1214)
    cfg.galerkin = cfg.current_deposition == WXA_DEPOSIT_DIRECT ? 0 : 1;
    if (pp.query_word("algo.field_gathering", w) && w != "energyconserving")
        throw std::runtime_error("inputs: algo.field_gathering = " + w + " is not on this path (energy-conserving)");
    int galerkin_scheme = cfg.galerkin;
    if (pp.queryWithParser("interpolation.galerkin_scheme", galerkin_scheme)) cfg.galerkin = galerkin_scheme != 0;
    cfg.nox = 0;
    const bool has_shape = pp.queryWithParser("algo.particle_shape", cfg.nox);
    int use_filter = 1;                                         // WarpX.cpp:158
    pp.queryWithParser("warpx.use_filter", use_filter);
    cfg.use_filter = use_filter != 0;
    int use_fdtd_nci_corr = 0;                                  // MultiParticleContainer.cpp:327 (WarpX.cpp:153)
    pp.queryWithParser("particles.use_fdtd_nci_corr", use_fdtd_nci_corr);
    cfg.use_fdtd_nci_corr = use_fdtd_nci_corr != 0;
    if (pp.queryArrWithParser("warpx.filter_npass_each_dir", v))
        for (double np : v)
            if (np != 1.0) throw std::runtime_error("inputs: only one bilinear filter pass per direction is on this path");
    int flag = 0;
    if (pp.queryWithParser("warpx.use_filter_compensation", flag) && flag)
        throw std::runtime_error("inputs: warpx.use_filter_compensation is not on this path");
    cfg.sort_interval = 4;                                      // WarpX.cpp:168 (GPU default)
    if (pp.query("warpx.sort_intervals", w)) cfg.sort_interval = ParmParse::safe_int(pp.evaluate(w), "warpx.sort_intervals");
    for (const char* key : {"warpx.do_dive_cleaning", "warpx.do_divb_cleaning", "warpx.do_subcycling", "warpx.do_pml",
                            "warpx.do_electrostatic", "warpx.do_multi_J"})
        if (pp.queryWithParser(key, flag) && flag)
            throw std::runtime_error(std::string("inputs: ") + key + " = 1 is not on this path");

    info.max_step = -1;
    pp.queryWithParser("max_step", info.max_step);
    if (pp.contains("stop_time")) throw std::runtime_error("inputs: stop_time is not supported, use max_step");

    // ---- species and lasers present? (algo.particle_shape is mandatory then, WarpX.cpp:1283-1325) ----
    std::vector<std::string> species_names, laser_names;
    pp.queryarr("particles.species_names", species_names);
    pp.queryarr("lasers.names", laser_names);
    if (!species_names.empty() || !laser_names.empty()) {
        if (!has_shape) throw std::runtime_error("inputs: algo.particle_shape must be set");
    } else if (!has_shape) {
        cfg.nox = 1;   // no particles: the guard depths only need a valid order
    }

    // ---- decomposition: given, or chosen here for comm->nranks bricks ----
    int do_moving_window = 0, window_dir = -1;
    pp.queryWithParser("warpx.do_moving_window", do_moving_window);
    if (do_moving_window) {
        if (!pp.query_word("warpx.moving_window_dir", w)) throw std::runtime_error("inputs: warpx.moving_window_dir must be set");
        window_dir = axis_of(w, "warpx.moving_window_dir");
    }
    if (nbricks) {
        for (int d = 0; d < 3; ++d) { cfg.nbricks[d] = nbricks[d]; cfg.coord[d] = coord ? coord[d] : 0; }
    } else {
        // every direction can be cut (round 3: also across PEC walls and along the moving window), the longest first:
        // a 32 x 32 x 256 wakefield stage on 8 GPUs becomes 8 cubes of 32^3 instead of eight 16 x 8 x 256 slivers
        bool splittable[3] = {true, true, true};
        (void)window_dir;
        choose_bricks(comm ? comm->nranks : 1, comm ? comm->rank : 0, cfg.n_cell, splittable, cfg.nbricks, cfg.coord);
    }

    auto h = std::make_unique<SimHandle>();
    h->warpx = std::make_unique<WarpX>(be, cfg, comm);
    {   // Source/WarpX.cpp:614,625
        int safe = 0, single = 0;
        pp.queryWithParser("warpx.safe_guard_cells", safe);
        pp.queryWithParser("warpx.do_single_precision_comms", single);
        if (safe) h->warpx->SetSafeGuardCells(true);
        if (single) h->warpx->SetSinglePrecisionComms(true);
    }
    WarpX& wx = *h->warpx;
    const WarpXContext& ctx = wx.context();

    // ---- moving window (WarpX.cpp:620-660) ----
    if (do_moving_window) {
        wx.SetMovingWindow(window_dir, pp.getWithParser("warpx.moving_window_v"));
    } else {
        pp.ignore("warpx.moving_window_dir");
        pp.ignore("warpx.moving_window_v");
    }
    // warpx.zmax_plasma_to_compute_max_step (WarpX.cpp:636-640; WarpX::computeMaxStepBoostAccelerator,
    // WarpXInitData.cpp:820-855): max_step = the step at which the lower end of the (boosted-frame) domain passes the end
    // of the plasma, given in the lab frame; replaces the deck's max_step
    {
        double zmax_plasma = 0.0;
        if (pp.queryWithParser("warpx.zmax_plasma_to_compute_max_step", zmax_plasma)) {
            if (!do_moving_window || window_dir != 2)
                throw std::runtime_error("inputs: Can use zmax_plasma_to_compute_max_step only if moving window along z.");
            const double len_plasma_boost = zmax_plasma / gamma_boost;
            const double v_plasma_boost = -beta_boost * 299'792'458.;
            const double zmin_domain_boost_step_0 = cfg.prob_lo[2];                       // WarpXInitData.cpp:529-531
            const double interaction_time_boost = (len_plasma_boost - zmin_domain_boost_step_0) / (wx.moving_window_v - v_plasma_boost);
            info.max_step = static_cast<int>(interaction_time_boost / wx.getdt(0));
        }
        int from_btd = 0;
        if (pp.queryWithParser("warpx.compute_max_step_from_btd", from_btd) && from_btd)
            throw std::runtime_error("inputs: warpx.compute_max_step_from_btd is not on this path");
    }

    // ---- external fields on the grid (WarpXInitData.cpp:940-1060) ----
    using warpx::fields::FieldType;
    using ablastr::fields::Direction;
    for (const char* eb : {"E", "B"}) {
        const std::string style_key = std::string("warpx.") + eb + "_ext_grid_init_style";
        const FieldType ft = eb[0] == 'E' ? FieldType::Efield_fp : FieldType::Bfield_fp;
        if (!pp.query_word(style_key, w) || w == "default") continue;
        // WarpX::shiftMF fills the cells that enter a moving window with the external field
        // (WarpXMovingWindow.cpp:225-246); the window of this library fills them with zero, so the combination is
        // refused rather than run differently from the reference
        auto refuse_with_window = [&]() {
            if (do_moving_window)
                throw std::runtime_error("inputs: " + style_key + " = " + w +
                                         " with warpx.do_moving_window = 1 is not on this path (cells entering the "
                                         "window are filled with zero, not with the external field)");
        };
        if (w == "constant") {
            pp.getArrWithParser(std::string("warpx.") + eb + "_external_grid", v, 3);
            if (v[0] != 0.0 || v[1] != 0.0 || v[2] != 0.0) refuse_with_window();
            std::vector<std::string> exprs;
            pp.queryarr(std::string("warpx.") + eb + "_external_grid", exprs);
            for (int d = 0; d < 3; ++d) {
                amrex::MultiFab& mf = *wx.fields().get(ft, Direction{d}, 0);
                if (v[d] == 0.0) mf.setVal(0.0, ctx.stream);
                else fill_from_parser(be, mf, pp.makeParser(exprs[d], {"x", "y", "z", "t"}), ctx);   // the value at every point
            }
        } else if (w == std::string("parse") + (char)std::tolower(eb[0]) + "extgridfunction") {
            refuse_with_window();
            for (int d = 0; d < 3; ++d) {
                const std::string key = std::string("warpx.") + eb + "xyz"[d] + "_external_grid_function(x,y,z)";
                std::string expr;
                if (!pp.query(key, expr)) throw std::runtime_error("inputs: " + key + " must be set");
                fill_from_parser(be, *wx.fields().get(ft, Direction{d}, 0), pp.makeParser(expr, {"x", "y", "z", "t"}), ctx);
            }
        } else {
            throw std::runtime_error("inputs: " + style_key + " = " + w + " is not on this path");
        }
    }
    // particles.E/B_ext_particle_init_style = constant + particles.E/B_external_particle
    // (MultiParticleContainer::ReadParameters, MultiParticleContainer.cpp:120-160): the same for every species
    double ext_E[3] = {0, 0, 0}, ext_B[3] = {0, 0, 0};
    bool any_ext = false;
    bool lens_style[2] = {false, false};
    {
        const char* style[2] = {"particles.E_ext_particle_init_style", "particles.B_ext_particle_init_style"};
        const char* value[2] = {"particles.E_external_particle", "particles.B_external_particle"};
        double* dst[2] = {ext_E, ext_B};
        for (int f = 0; f < 2; ++f) {
            if (!pp.query_word(style[f], w) || w == "none" || w == "default") continue;
            if (w == "repeatedplasmalens") { lens_style[f] = true; continue; }
            if (w != "constant")
                throw std::runtime_error(std::string("inputs: ") + style[f] + " = " + w +
                                         " is not on this path (constant, repeated_plasma_lens)");
            std::vector<double> v;
            if (!pp.queryArrWithParser(value[f], v) || v.size() != 3)
                throw std::runtime_error(std::string("inputs: ") + value[f] + " needs three values");
            for (int d = 0; d < 3; ++d) dst[f][d] = v[d];
            any_ext = true;
        }
    }

    // particles.repeated_plasma_lens_* (MultiParticleContainer.cpp:210-260): the strengths are read only for the field
    // whose style asks for the lens, the other one stays zero
    wxa_repeated_plasma_lens lens{};
    std::vector<double> lens_starts, lens_lengths, lens_sE, lens_sB;
    if (lens_style[0] || lens_style[1]) {
        lens.period = pp.getWithParser("particles.repeated_plasma_lens_period");
        if (!(lens.period > 0.0)) throw std::runtime_error("inputs: particles.repeated_plasma_lens_period must be > 0");
        if (!pp.queryArrWithParser("particles.repeated_plasma_lens_starts", lens_starts) ||
            !pp.queryArrWithParser("particles.repeated_plasma_lens_lengths", lens_lengths))
            throw std::runtime_error("inputs: particles.repeated_plasma_lens_starts and _lengths must be set");
        const size_t n = lens_starts.size();
        if (lens_lengths.size() != n) throw std::runtime_error("inputs: particles.repeated_plasma_lens_* lengths differ");
        lens_sE.assign(n, 0.0);
        lens_sB.assign(n, 0.0);
        const char* keys[2] = {"particles.repeated_plasma_lens_strengths_E", "particles.repeated_plasma_lens_strengths_B"};
        std::vector<double>* dst[2] = {&lens_sE, &lens_sB};
        for (int f = 0; f < 2; ++f) {
            if (!lens_style[f]) { pp.ignore(keys[f]); continue; }
            if (!pp.queryArrWithParser(keys[f], *dst[f]) || dst[f]->size() != n)
                throw std::runtime_error(std::string("inputs: ") + keys[f] + " needs one value per lens");
        }
        lens.n_lenses = (int32_t)n;
        lens.starts = lens_starts.data(); lens.lengths = lens_lengths.data();
        lens.strengths_E = lens_sE.data(); lens.strengths_B = lens_sB.data();
    }
    const double c = 299'792'458.;
    // MapParticletoBoostedFrame (PhysicalParticleContainer.cpp:456-500) with t_lab = 0, at simulation time t0 = 0,
    // boost_adjust_transverse_positions off, forward propagation: single / multiple particles are given in the lab frame
    auto map_to_boosted_frame = [&](double pos[3], double u[3]) {
        const double uz_boost = gamma_boost * beta_boost * c;
        const double t_lab = 0.0, t0 = wx.gett_new();
        const double tpr = gamma_boost * t_lab - uz_boost * pos[2] / (c * c);
        const double zpr = gamma_boost * pos[2] - uz_boost * t_lab;
        const double gamma_lab = std::sqrt(1.0 + (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) / (c * c));
        u[2] = gamma_boost * u[2] - uz_boost * gamma_lab;
        const double gammapr = std::sqrt(1.0 + (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) / (c * c));
        const double vzpr = u[2] / gammapr;
        pos[2] = zpr - (tpr - t0) * vzpr;
    };
    // ---- species (PlasmaInjector.cpp, PhysicalParticleContainer::AddParticles) ----
    for (const std::string& name : species_names) {
        double charge = 0.0, mass = 0.0;
        bool have_q = false, have_m = false;
        if (pp.query_word(name + ".species_type", w)) {   // Source/Particles/SpeciesPhysicalProperties.H
            const auto& k = pp.constants();
            if (w == "electron") { charge = -k.at("q_e"); mass = k.at("m_e"); }
            else if (w == "positron") { charge = k.at("q_e"); mass = k.at("m_e"); }
            else if (w == "proton") { charge = k.at("q_e"); mass = k.at("m_p"); }
            else throw std::runtime_error("inputs: " + name + ".species_type = " + w + " is not on this path");
            have_q = have_m = true;
        }
        have_q = pp.queryWithParser(name + ".charge", charge) || have_q;
        have_m = pp.queryWithParser(name + ".mass", mass) || have_m;
        if (!have_q || !have_m) throw std::runtime_error("inputs: " + name + ".charge and .mass (or .species_type) must be set");
        if (charge != 0.0 && wx.any_reflecting_wall())
            throw std::runtime_error("inputs: charged species with a reflecting particle boundary are not on this path");
        for (const char* off : {".do_not_push", ".do_not_deposit", ".do_not_gather", ".do_field_ionization", ".do_qed_quantum_sync",
                                ".do_qed_breit_wheeler", ".do_backward_propagation",
                                ".rigid_advance", ".initialize_self_fields", ".do_resampling"})
            if (pp.queryWithParser(name + off, flag) && flag)
                throw std::runtime_error("inputs: " + name + off + " is not on this path");
        pp.ignore(name + ".addIntegerAttributes");   // extra per-particle attributes are diagnostics only
        pp.ignore(name + ".addRealAttributes");
        pp.ignore_prefix(name + ".attribute.");

        const int sid = wx.GetPartContainer().AddSpecies(charge, mass);
        if (any_ext) wx.GetPartContainer().GetParticleContainer(sid).SetExternalParticleFields(ext_E, ext_B);
        if (lens.n_lenses > 0) wx.GetPartContainer().GetParticleContainer(sid).SetRepeatedPlasmaLens(lens, wx.getdt(0));
        double crr = 0;
        if (pp.queryWithParser(name + ".do_classical_radiation_reaction", crr) && crr != 0)
            wx.GetPartContainer().GetParticleContainer(sid).SetRadiationReaction(true);
        auto* pc = dynamic_cast<PhysicalParticleContainer*>(&wx.GetPartContainer().GetParticleContainer(sid));
        if (!pp.query_word(name + ".injection_style", w)) throw std::runtime_error("inputs: " + name + ".injection_style must be set");
        std::vector<double> cols[7];
        auto add_if_mine = [&](const double pos_in[3], const double u_in[3], double weight) {
            double pos[3] = {pos_in[0], pos_in[1], pos_in[2]};
            double u[3] = {u_in[0] * c, u_in[1] * c, u_in[2] * c};          // setupSingleParticle: u *= c
            if (gamma_boost > 1.0) map_to_boosted_frame(pos, u);           // AddParticles, :862-893
            // AddNParticles keeps the particles of this rank's boxes: here [brick_plo, brick_phi)
            for (int d = 0; d < 3; ++d)
                if (!(pos[d] >= ctx.brick_plo[d] && pos[d] < ctx.brick_phi[d])) return;
            for (int d = 0; d < 3; ++d) cols[d].push_back(pos[d]);
            cols[3].push_back(weight);
            for (int d = 0; d < 3; ++d) cols[4 + d].push_back(u[d]);
        };
        if (w == "nuniformpercell") {
            wxa_plasma_injector inj{};
            pp.getArrWithParser(name + ".num_particles_per_cell_each_dim", v, 3);
            for (int d = 0; d < 3; ++d) inj.ppc[d] = ParmParse::safe_int(v[d], name + ".num_particles_per_cell_each_dim");
            const char* lo_keys[3] = {".xmin", ".ymin", ".zmin"};
            const char* hi_keys[3] = {".xmax", ".ymax", ".zmax"};
            for (int d = 0; d < 3; ++d) {
                inj.lo[d] = -std::numeric_limits<double>::max();   // PlasmaInjector.cpp:70-80
                inj.hi[d] = std::numeric_limits<double>::max();
                pp.queryWithParser(name + lo_keys[d], inj.lo[d]);
                pp.queryWithParser(name + hi_keys[d], inj.hi[d]);
            }
            if (!pp.query_word(name + ".profile", w) || w != "constant")
                throw std::runtime_error("inputs: " + name + ".profile must be constant on this path");
            inj.density = pp.getWithParser(name + ".density");
            std::string mom = "atrest";
            pp.query_word(name + ".momentum_distribution_type", mom);
            if (mom == "constant") {
                double u[3] = {0.0, 0.0, 0.0};
                pp.queryWithParser(name + ".ux", u[0]);
                pp.queryWithParser(name + ".uy", u[1]);
                pp.queryWithParser(name + ".uz", u[2]);
                const double none[3] = {0.0, 0.0, 0.0};
                pc->SetGaussianMomentum(u, none, 0);
            } else if (mom == "parsemomentumfunction") {
                Parser f[3];
                for (int d = 0; d < 3; ++d) {
                    const std::string key = name + ".momentum_function_u" + "xyz"[d] + "(x,y,z)";
                    std::string expr;
                    if (!pp.query(key, expr)) throw std::runtime_error("inputs: " + key + " must be set");
                    f[d] = pp.makeParser(expr, {"x", "y", "z"});
                }
                pc->SetMomentumFunction([f0 = f[0], f1 = f[1], f2 = f[2]](double x, double y, double z, double* out) {
                    const double xyz[3] = {x, y, z};
                    out[0] = f0.eval(xyz); out[1] = f1.eval(xyz); out[2] = f2.eval(xyz);
                });
            } else if (mom == "gaussian") {
                // InjectorMomentumGaussian: u = u_m + u_th N(0,1) per component.  The reference draws from
                // AMReX's generator, which no other program reproduces; here a counter-based stream keyed by
                // warpx.random_seed (default 1) and the species index gives every lattice point its draws, so a
                // run is repeatable, the same on any brick layout, and statistically the same plasma.
                double um[3] = {0.0, 0.0, 0.0}, uth[3] = {0.0, 0.0, 0.0};
                const char* mk[3] = {".ux_m", ".uy_m", ".uz_m"};
                const char* tk[3] = {".ux_th", ".uy_th", ".uz_th"};
                for (int d = 0; d < 3; ++d) { pp.queryWithParser(name + mk[d], um[d]); pp.queryWithParser(name + tk[d], uth[d]); }
                std::string seed_word = "default";
                pp.query("warpx.random_seed", seed_word);
                if (seed_word == "random" && comm && comm->nranks > 1)   // each brick would seed its own stream
                    throw std::runtime_error("inputs: warpx.random_seed = random with several bricks: the bricks of a run must share "
                                             "the stream (a plasma that is the same on any brick layout): give a number");
                const uint64_t seed = seed_word == "default" ? 1u : (seed_word == "random" ? (uint64_t)std::random_device{}()
                                                                                            : (uint64_t)pp.evaluate(seed_word));
                pc->SetGaussianMomentum(um, uth, seed * 0x9E3779B97F4A7C15ull + 1000003ull * (uint64_t)sid);
            } else if (mom != "atrest") {
                throw std::runtime_error("inputs: " + name + ".momentum_distribution_type = " + mom +
                                         " is not on this path (at_rest, constant, gaussian, parse_momentum_function)");
            }
            int continuous = 0;
            pp.queryWithParser(name + ".do_continuous_injection", continuous);
            pc->SetPlasmaInjector(inj, continuous != 0);
            pc->AddPlasma(ctx.prob_lo.data(), ctx.prob_hi.data());
        } else if (w == "singleparticle") {
            std::vector<double> pos, u;
            pp.getArrWithParser(name + ".single_particle_pos", pos, 3);
            pp.getArrWithParser(name + ".single_particle_u", u, 3);
            add_if_mine(pos.data(), u.data(), pp.getWithParser(name + ".single_particle_weight"));
            pc->AppendFromHost(cols);
        } else if (w == "multipleparticles") {
            std::vector<double> q[7];
            const char* keys[7] = {".multiple_particles_pos_x", ".multiple_particles_pos_y", ".multiple_particles_pos_z",
                                   ".multiple_particles_weight", ".multiple_particles_ux", ".multiple_particles_uy",
                                   ".multiple_particles_uz"};
            for (int a = 0; a < 7; ++a)
                if (!pp.queryArrWithParser(name + keys[a], q[a])) throw std::runtime_error("inputs: " + name + keys[a] + " must be set");
            for (int a = 1; a < 7; ++a)
                if (q[a].size() != q[0].size()) throw std::runtime_error("inputs: " + name + ".multiple_particles_* lengths differ");
            for (size_t i = 0; i < q[0].size(); ++i) {
                const double pos[3] = {q[0][i], q[1][i], q[2][i]}, u[3] = {q[4][i], q[5][i], q[6][i]};
                add_if_mine(pos, u, q[3][i]);
            }
            pc->AppendFromHost(cols);
        } else if (w == "gaussianbeam") {   // query_word: lower case, quotes and underscores dropped
            // PlasmaInjector::setupGaussianBeam (PlasmaInjector.cpp:221-262) + PhysicalParticleContainer::AddGaussianBeam
            // (PhysicalParticleContainer.cpp:503-677), 3-D: npart draws of N(x_m, x_rms) x N(y_m, y_rms) x N(z_m, z_rms),
            // kept inside the species' bounds and the cuts, weight q_tot / (npart charge), momentum from the species'
            // momentum distribution, optional 4- or 8-fold symmetrisation; CheckAndAddParticle maps to the boosted frame.
            // The reference draws from AMReX's generator on the I/O rank; here every particle has its own counter-based
            // stream (warpx.random_seed, species index, particle index), so the beam is the same on any brick layout and
            // for the CPU and the HIP backend -- and statistically, not bit for bit, the reference's.
            const double x_m = pp.getWithParser(name + ".x_m"), y_m = pp.getWithParser(name + ".y_m"), z_m = pp.getWithParser(name + ".z_m");
            const double x_rms = pp.getWithParser(name + ".x_rms"), y_rms = pp.getWithParser(name + ".y_rms"), z_rms = pp.getWithParser(name + ".z_rms");
            double cut[3] = {std::numeric_limits<double>::max(), std::numeric_limits<double>::max(), std::numeric_limits<double>::max()};
            pp.queryWithParser(name + ".x_cut", cut[0]); pp.queryWithParser(name + ".y_cut", cut[1]); pp.queryWithParser(name + ".z_cut", cut[2]);
            const double q_tot = pp.getWithParser(name + ".q_tot");
            long npart = (long)pp.getWithParser(name + ".npart");
            int do_symmetrize = 0, symmetrization_order = 4;
            pp.queryWithParser(name + ".do_symmetrize", do_symmetrize);
            pp.queryWithParser(name + ".symmetrization_order", symmetrization_order);
            if (symmetrization_order != 4 && symmetrization_order != 8)
                throw std::runtime_error("inputs: Symmetrization only supported to orders 4 or 8");
            if (pp.contains(name + ".focal_distance")) throw std::runtime_error("inputs: " + name + ".focal_distance is not on this path");
            if (charge == 0.0) throw std::runtime_error("inputs: " + name + ": a gaussian_beam needs a charge");
            double blo[3], bhi[3];
            const char* lo_keys[3] = {".xmin", ".ymin", ".zmin"};
            const char* hi_keys[3] = {".xmax", ".ymax", ".zmax"};
            for (int d = 0; d < 3; ++d) {
                blo[d] = -std::numeric_limits<double>::max(); bhi[d] = std::numeric_limits<double>::max();   // PlasmaInjector.cpp:70-80
                pp.queryWithParser(name + lo_keys[d], blo[d]); pp.queryWithParser(name + hi_keys[d], bhi[d]);
            }
            std::string mom = "atrest";
            pp.query_word(name + ".momentum_distribution_type", mom);
            double um[3] = {0.0, 0.0, 0.0}, uth[3] = {0.0, 0.0, 0.0};
            if (mom == "gaussian" || mom == "constant") {
                const char* mk[3] = {".ux_m", ".uy_m", ".uz_m"};
                const char* ck[3] = {".ux", ".uy", ".uz"};
                const char* tk[3] = {".ux_th", ".uy_th", ".uz_th"};
                for (int d = 0; d < 3; ++d) {
                    pp.queryWithParser(name + (mom == "gaussian" ? mk[d] : ck[d]), um[d]);
                    if (mom == "gaussian") pp.queryWithParser(name + tk[d], uth[d]);
                }
            } else if (mom != "atrest") {
                throw std::runtime_error("inputs: " + name + ".momentum_distribution_type = " + mom + " is not on this path for a gaussian_beam (at_rest, constant, gaussian)");
            }
            std::string seed_word = "default";
            pp.query("warpx.random_seed", seed_word);
            // every brick draws the whole beam and keeps its share: all bricks need the same stream
            if (seed_word == "random" && comm && comm->nranks > 1)
                throw std::runtime_error("inputs: warpx.random_seed = random with several bricks would give every brick another beam "
                                         "(each draws all of it and keeps its share): give a number");
            const uint64_t seed = seed_word == "default" ? 1u : (seed_word == "random" ? (uint64_t)std::random_device{}() : (uint64_t)pp.evaluate(seed_word));
            const uint64_t stream = seed * 0x9E3779B97F4A7C15ull + 1000003ull * (uint64_t)sid + 0x6A09E667F3BCC909ull;
            if (do_symmetrize) npart /= symmetrization_order;
            const double weight = q_tot / ((double)npart * charge);
            for (long i = 0; i < npart; ++i) {
                double n6[6];
                beam_normals(stream, (uint64_t)i, n6);
                const double x = x_m + x_rms * n6[0], y = y_m + y_rms * n6[1], z = z_m + z_rms * n6[2];
                if (!(x < bhi[0] && x >= blo[0] && y < bhi[1] && y >= blo[1] && z < bhi[2] && z >= blo[2])) continue;
                if (!(std::abs(x - x_m) <= cut[0] * x_rms && std::abs(y - y_m) <= cut[1] * y_rms && std::abs(z - z_m) <= cut[2] * z_rms)) continue;
                const double u[3] = {um[0] + uth[0] * n6[3], um[1] + uth[1] * n6[4], um[2] + uth[2] * n6[5]};   // gamma beta
                auto add = [&](double px, double py, double ux, double uy, double wgt) {
                    const double pos[3] = {px, py, z}, uu[3] = {ux, uy, u[2]};
                    add_if_mine(pos, uu, wgt);
                };
                if (do_symmetrize) {
                    const double wn = weight / symmetrization_order;
                    add(x, y, u[0], u[1], wn); add(x, -y, u[0], -u[1], wn); add(-x, y, -u[0], u[1], wn); add(-x, -y, -u[0], -u[1], wn);
                    if (symmetrization_order == 8) {
                        add(y, x, u[1], u[0], wn); add(-y, x, -u[1], u[0], wn); add(y, -x, u[1], -u[0], wn); add(-y, -x, -u[1], -u[0], wn);
                    }
                } else {
                    add(x, y, u[0], u[1], weight);
                }
            }
            pc->AppendFromHost(cols);
        } else {
            throw std::runtime_error("inputs: " + name + ".injection_style = " + w +
                                     " is not on this path (NUniformPerCell, SingleParticle, MultipleParticles, gaussian_beam)");
        }
        if (wx.sort_intervals > 0 && pc->TotalNumberOfParticles() > 0) pc->SortParticlesByBin(amrex::IntVect(1));
        info.species_names.push_back(name);
    }

    // ---- laser antennas (LaserParticleContainer.cpp:70-250) ----
    for (const std::string& name : laser_names) {
        if (!pp.query_word(name + ".profile", w) || w != "gaussian")
            throw std::runtime_error("inputs: " + name + ".profile must be Gaussian on this path");
        wxa_laser_antenna la{};
        pp.getArrWithParser(name + ".position", v, 3);
        for (int d = 0; d < 3; ++d) la.position[d] = v[d];
        pp.getArrWithParser(name + ".direction", v, 3);
        for (int d = 0; d < 3; ++d) la.direction[d] = v[d];
        pp.getArrWithParser(name + ".polarization", v, 3);
        for (int d = 0; d < 3; ++d) la.polarization[d] = v[d];
        la.e_max = pp.getWithParser(name + ".e_max");
        la.wavelength = pp.getWithParser(name + ".wavelength");
        la.waist = pp.getWithParser(name + ".profile_waist");
        la.duration = pp.getWithParser(name + ".profile_duration");
        la.t_peak = pp.getWithParser(name + ".profile_t_peak");
        la.focal_distance = pp.getWithParser(name + ".profile_focal_distance");
        for (const char* zero : {".zeta", ".beta", ".phi2", ".phi0", ".stc_direction"})
            if (pp.contains(name + zero)) throw std::runtime_error("inputs: " + name + zero + " is not on this path");
        wx.GetPartContainer().AddLaser(la);
    }

    // ---- harmless: output, AMReX box sizes, verbosity ----
    std::vector<std::string> diag_names, rdiag_names;
    pp.queryarr("diagnostics.diags_names", diag_names);
    pp.queryarr("warpx.reduced_diags_names", rdiag_names);
    // <diag>.diag_type = BackTransformed (BTDiagnostics::ReadParameters, BTDiagnostics.cpp:206-292): the field snapshots are
    // assembled in memory (wxa_sim_btd_info / _data); formats, species output and the other diagnostics are not produced
    // warpx_amd.write_diagnostics = 0 (this library's own key; default 1 = what the reference does): nothing is written
    // to disk -- no Full or reduced diagnostics, and a BackTransformed diagnostic keeps its snapshots in memory
    int write_diagnostics = 1;
    pp.queryWithParser("warpx_amd.write_diagnostics", write_diagnostics);
    for (const std::string& d : diag_names) {
        std::string type;
        if (!pp.query_word(d + ".diag_type", type) || type != "backtransformed") continue;
        int fields_on = 1, nsnap = 0, buffer = 256;
        pp.queryWithParser(d + ".do_back_transformed_fields", fields_on);
        if (!fields_on) continue;
        bool in_memory_only = false;
        {   // formats this library does not write (openpmd ...): the snapshots are still assembled and can be read through
            // wxa_sim_btd_info / _data / _particles, only the files are not written -- said on stderr, not silently.  A
            // second BackTransformed diagnostic (one only here) or one given by `intervals` is left out, also by name.
            std::string format = "plotfile";
            pp.query_word(d + ".format", format);
            if (format != "plotfile") {
                if (wx.btd() || pp.contains(d + ".intervals")) {
                    std::fprintf(stderr, "[warpx_amd] diagnostic %s (BackTransformed, format = %s) is left out: this library writes "
                                 "plotfiles and keeps one BackTransformed diagnostic\n", d.c_str(), format.c_str());
                    pp.ignore_prefix(d + ".");
                    continue;
                }
                //  In memory only on request: assembling the snapshots costs a device-to-host pass over nine
                // fields and a charge deposition every step, and whole lab-frame snapshots accumulate in host memory for the
                // length of the run -- a production deck written for openPMD output must not pay that for files nobody gets.
                int keep = 0;
                pp.queryWithParser("warpx_amd.btd_in_memory", keep);
                if (!keep) {
                    std::fprintf(stderr, "[warpx_amd] diagnostic %s (BackTransformed, format = %s) is left out: this library writes "
                                 "plotfiles (warpx_amd.btd_in_memory = 1 assembles the snapshots in host memory for "
                                 "wxa_sim_btd_info / _data / _particles instead)\n", d.c_str(), format.c_str());
                    pp.ignore_prefix(d + ".");
                    continue;
                }
                std::fprintf(stderr, "[warpx_amd] diagnostic %s: format = %s is not written by this library; its snapshots stay in "
                             "memory (wxa_sim_btd_info / _data / _particles)\n", d.c_str(), format.c_str());
                in_memory_only = true;
            }
        }
        if (pp.contains(d + ".intervals")) throw std::runtime_error("inputs: " + d + ".intervals is not on this path (num_snapshots_lab)");
        if (!pp.queryWithParser(d + ".num_snapshots_lab", nsnap)) throw std::runtime_error("inputs: " + d + ".num_snapshots_lab must be set");
        double dt_snap = 0.0, dz_snap = 0.0;
        const bool have_dt = pp.queryWithParser(d + ".dt_snapshots_lab", dt_snap);
        if (pp.queryWithParser(d + ".dz_snapshots_lab", dz_snap)) dt_snap = dz_snap / 299792458.0;   // :258-261
        else if (!have_dt) throw std::runtime_error("inputs: " + d + ".dt_snapshots_lab or dz_snapshots_lab must be set");
        pp.queryWithParser(d + ".buffer_size", buffer);
        if (wx.btd()) throw std::runtime_error("inputs: one BackTransformed diagnostic only");
        int write_species = 1;                                     // BTDiagnostics.cpp:113-114
        pp.queryWithParser(d + ".write_species", write_species);
        wx.AddBTDiagnostics(nsnap, dt_snap, buffer, write_species != 0);
        wx.btd()->SetSpeciesNames(info.species_names);
        // <diag>.file_prefix (BTDiagnostics.cpp:98-99 reads it through Diagnostics::BaseReadParameters): given, the snapshots
        // are flushed to <file_prefix><i>/ buffer by buffer as plotfiles; absent, they stay in memory.  On several bricks
        // brick 0 writes, one plotfile per snapshot for the whole run (BTDiagnostics::flush_bricks).
        std::string prefix;
        if (pp.query(d + ".file_prefix", prefix) && write_diagnostics && !in_memory_only) {
            int digits = 6;                                        // Diagnostics.H: m_file_min_digits
            pp.queryWithParser(d + ".file_min_digits", digits);
            wx.btd()->SetFlush(prefix, digits);
        }
    }
    // <diag>.diag_type = Full with format = plotfile (Diagnostics::BaseReadParameters, Diagnostics.cpp:47-60;
    // FullDiagnostics::ReadParameters, FullDiagnostics.cpp:109-125): plotfiles <file_prefix><step> at the steps of
    // <diag>.intervals and after the last step.  Other formats (openpmd, checkpoint ...) and the other diagnostic types
    // (TimeAveraged, BoundaryScraping) are output this library does not write.
    h->species_names = info.species_names;
    wx.max_step = info.max_step;
    for (const std::string& d : diag_names) {
        if (!write_diagnostics) break;
        std::string type, format = "plotfile";
        if (!pp.query_word(d + ".diag_type", type)) throw std::runtime_error("inputs: " + d + ".diag_type must be set");
        pp.query_word(d + ".format", format);
        if (type != "full" || format != "plotfile") continue;
        std::vector<std::string> iv;
        if (!pp.queryarr(d + ".intervals", iv)) throw std::runtime_error("inputs: " + d + ".intervals must be set");
        const std::string intervals = deck_intervals(pp, iv, d + ".intervals");
        std::string prefix = "diags/" + d;
        pp.query(d + ".file_prefix", prefix);
        int digits = 6, write_species = 1, dump_last = 1;
        pp.queryWithParser(d + ".file_min_digits", digits);
        pp.queryWithParser(d + ".write_species", write_species);
        pp.queryWithParser(d + ".dump_last_timestep", dump_last);
        std::vector<std::string> fields, species;
        pp.queryarr(d + ".fields_to_plot", fields);
        if (fields.size() == 1 && fields[0] == "none") fields = {"none"};
        pp.queryarr(d + ".species", species);
        for (const std::string& sp : species)
            if (std::find(info.species_names.begin(), info.species_names.end(), sp) == info.species_names.end())
                throw std::runtime_error("inputs: " + d + ".species names the unknown species " + sp);
        if (comm && comm->nranks > 1 && !prefix.empty() && prefix[0] != '/') {
            // the bricks of a run write into one directory: a relative prefix is relative to every process's own
            // working directory, which a launcher normally makes the same
        }
        add_full_diagnostic(*h, d, intervals, prefix, digits, fields, write_species != 0, species, dump_last != 0);
    }
    // warpx.reduced_diags_names (MultiReducedDiags.cpp:36-83, ReducedDiags.cpp:26-72): the four types of ReducedDiags.hpp
    // are produced; the others (probes, histograms, load-balance costs ...) are output this library does not write
    for (const std::string& d : rdiag_names) {
        std::string type;
        if (!pp.query(d + ".type", type)) throw std::runtime_error("inputs: " + d + ".type must be set");
        if (!MultiReducedDiags::known_type(type) || !write_diagnostics) continue;
        if (pp.contains(d + ".frequency"))   // ReducedDiags::BackwardCompatibility
            throw std::runtime_error("inputs: " + d + ".frequency is no longer a valid option. Please use the renamed option " +
                                     d + ".intervals instead.");
        std::vector<std::string> iv{"1"};
        pp.queryarr(d + ".intervals", iv);   // getarr in the reference: the default is never used there
        const std::string intervals = deck_intervals(pp, iv, d + ".intervals");
        std::string path = "./diags/reducedfiles/";
        pp.query(d + ".path", path);
        for (const char* fixed : {".extension", ".separator", ".precision"})
            if (pp.contains(d + fixed)) throw std::runtime_error("inputs: " + d + fixed + " is not on this path (txt, ' ', 14)");
        wx.reduced_diags.Add(d, type, intervals, path.c_str());
    }
    wx.reduced_diags.SetSpeciesNames(info.species_names);
    for (const std::string& d : diag_names) pp.ignore_prefix(d + ".");
    for (const std::string& d : rdiag_names) pp.ignore_prefix(d + ".");
    pp.ignore_prefix("diagnostics.");
    pp.ignore_prefix("amrex.");
    for (const char* key : {"amr.max_grid_size", "amr.max_grid_size_x", "amr.max_grid_size_y", "amr.max_grid_size_z",
                            "amr.blocking_factor", "amr.blocking_factor_x", "amr.blocking_factor_y", "amr.blocking_factor_z",
                            "warpx.verbose", "warpx.serialize_initial_conditions", "warpx.do_dynamic_scheduling",
                            "warpx.numprocs", "warpx.random_seed", "algo.load_balance_intervals", "warpx.always_warn_immediately",
                            "warpx.abort_on_warning_threshold"})
        pp.ignore(key);
    const std::vector<std::string> left = pp.unused();
    if (!left.empty()) {
        std::string msg = "inputs: parameters that this library does not understand:";
        for (const std::string& k : left) msg += " " + k;
        throw std::runtime_error(msg);
    }
    be->stream_sync(ctx.stream);
    return h;
}

}  // namespace wxa::host

#include "Checksum.hpp"
#include "FullDiagnostics.hpp"

namespace wxa::host {
inline void add_full_diagnostic(SimHandle& h, const std::string& name, const std::string& intervals, const std::string& file_prefix,
                                int file_min_digits, const std::vector<std::string>& fields, bool write_species,
                                const std::vector<std::string>& species, bool dump_last_timestep) {
    full_diagnostics(h).Add(name, intervals, file_prefix, file_min_digits, fields, write_species, species, dump_last_timestep);
}
}  // namespace wxa::host

// C entry points of the deck front end (include/warpx_amd.h), instantiated next to WXA_SIM_CAPI
#define WXA_INPUTS_CAPI(PFX, RET, SIMTYPE, BACKEND_GETTER, SET_ERROR)                                   \
    extern "C" {                                                                                       \
    RET PFX##sim_create_from_inputs(const char* path, int32_t n_overrides, const char* const* overrides, \
                                    const wxa_comm* comm, const int32_t* nbricks, const int32_t* coord, \
                                    SIMTYPE** out) {                                                   \
        if (!path || !out || n_overrides < 0 || (n_overrides > 0 && !overrides)) return (RET)WXA_ERR_INVALID_ARG; \
        try {                                                                                          \
            std::vector<std::string> ov;                                                               \
            for (int32_t i = 0; i < n_overrides; ++i) ov.emplace_back(overrides[i]);                   \
            wxa::host::DeckInfo info;                                                                  \
            auto h = wxa::host::sim_from_inputs(BACKEND_GETTER(), path, ov, comm, nbricks, coord, info); \
            h->max_step = info.max_step;                                                               \
            h->species_names = info.species_names;                                                     \
            *out = reinterpret_cast<SIMTYPE*>(h.release());                                            \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    int32_t PFX##sim_max_step(const SIMTYPE* s) {                                                      \
        return s ? reinterpret_cast<const wxa::host::SimHandle*>(s)->max_step : -1;                    \
    }                                                                                                  \
    int32_t PFX##sim_num_species(const SIMTYPE* s) {                                                   \
        if (!s) return -1;                                                                             \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(const_cast<SIMTYPE*>(s));                    \
        return h->warpx->GetPartContainer().nSpecies();                                                \
    }                                                                                                  \
    const char* PFX##sim_species_name(const SIMTYPE* s, int32_t id) {                                  \
        if (!s) return nullptr;                                                                        \
        const auto* h = reinterpret_cast<const wxa::host::SimHandle*>(s);                              \
        return (id >= 0 && id < (int32_t)h->species_names.size()) ? h->species_names[id].c_str() : nullptr; \
    }                                                                                                  \
    /* writes the JSON text (NUL-terminated) into buf; returns the length needed (without the NUL) or < 0 */ \
    int64_t PFX##sim_checksum_json(SIMTYPE* s, char* buf, int64_t capacity) {                          \
        if (!s) return -1;                                                                             \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        try {                                                                                          \
            const std::string js = wxa::host::checksum_json(*h, h->species_names);                     \
            if (buf && capacity > 0) {                                                                 \
                const size_t n = std::min((size_t)capacity - 1, js.size());                            \
                std::memcpy(buf, js.data(), n);                                                        \
                buf[n] = 0;                                                                            \
            }                                                                                          \
            return (int64_t)js.size();                                                                 \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return -2;                                                                                 \
        }                                                                                              \
    }                                                                                                  \
    /* FlushFormatPlotfile::WriteToFile for this brick (host/Plotfile.hpp) */                          \
    RET PFX##sim_write_plotfile(SIMTYPE* s, const char* dir) {                                         \
        if (!s || !dir) return (RET)WXA_ERR_INVALID_ARG;                                               \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        try {                                                                                          \
            wxa::host::write_plotfile(*h, dir, h->species_names);                                      \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    /* <diag>.diag_type = Full, format = plotfile (host/FullDiagnostics.hpp); fields: space-separated names or NULL */ \
    RET PFX##sim_add_full_diag(SIMTYPE* s, const char* name, const char* intervals, const char* file_prefix, \
                               int32_t file_min_digits, const char* fields, int32_t write_species,    \
                               int32_t dump_last_timestep) {                                           \
        if (!s || !name || !*name) return (RET)WXA_ERR_INVALID_ARG;                                    \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        try {                                                                                          \
            std::vector<std::string> f;                                                                \
            std::istringstream words(fields ? fields : "");                                            \
            for (std::string w; words >> w;) f.push_back(w);                                           \
            wxa::host::add_full_diagnostic(*h, name, intervals ? intervals : "", file_prefix ? file_prefix : "", \
                                           file_min_digits > 0 ? file_min_digits : 6, f, write_species != 0, {}, \
                                           dump_last_timestep != 0);                                   \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    /* MultiDiagnostics::FilterComputePackFlushLastTimestep for a run the caller ends itself (no max_step known) */ \
    RET PFX##sim_flush_diags_last_timestep(SIMTYPE* s) {                                               \
        if (!s) return (RET)WXA_ERR_INVALID_ARG;                                                       \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        try {                                                                                          \
            if (h->warpx->btd()) h->warpx->btd()->SetSpeciesNames(h->species_names);                       \
            h->warpx->FlushDiagsLastTimestep();                                                            \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_HIP;                                                                   \
        }                                                                                              \
    }                                                                                                  \
    /* lab-frame snapshot i of the back-transformed diagnostics as a plotfile (host/Plotfile.hpp, write_btd_plotfile) */ \
    RET PFX##sim_btd_write_plotfile(SIMTYPE* s, int32_t i, const char* dir) {                          \
        if (!s || !dir) return (RET)WXA_ERR_INVALID_ARG;                                               \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        try {                                                                                          \
            wxa::host::write_btd_plotfile(*h, i, dir, h->species_names);                                      \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    /* snapshots to disk: <file_prefix><i, file_min_digits>/ receives every full buffer as one more grid of a plotfile */ \
    /* (BTDiagnostics::Flush + MergeBuffersForPlotfile, BTDiagnostics.cpp:1027-1314); after wxa_sim_add_btd, before stepping */ \
    RET PFX##sim_btd_set_flush(SIMTYPE* s, const char* file_prefix, int32_t file_min_digits) {        \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        if (!s || !file_prefix || !h->warpx->btd()) return (RET)WXA_ERR_INVALID_ARG;                   \
        try {                                                                                          \
            h->warpx->btd()->SetSpeciesNames(h->species_names);                                        \
            h->warpx->btd()->SetFlush(file_prefix, file_min_digits);                                   \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    /* the forced flush after the last step (FilterComputePackFlushLastTimestep): partly filled buffers go to disk */ \
    RET PFX##sim_btd_flush(SIMTYPE* s) {                                                               \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        if (!s || !h->warpx->btd()) return (RET)WXA_ERR_INVALID_ARG;                                   \
        try {                                                                                          \
            h->warpx->btd()->SetSpeciesNames(h->species_names);                                        \
            h->warpx->btd()->FlushLast(*h->warpx);                                                     \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    /* index box of this brick's share of snapshot i in the snapshot's (x, y, k_lab) index space (m_snapshot_box, */ \
    /* BTDiagnostics.cpp:489-506, cut to the brick in x and y), inclusive bounds */                  \
    RET PFX##sim_btd_box(SIMTYPE* s, int32_t i, int32_t lo[3], int32_t hi[3]) {                       \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        if (!s || !lo || !hi || !h->warpx->btd() || i < 0 || i >= h->warpx->btd()->num_snapshots())   \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        const auto& sn = h->warpx->btd()->snapshot(i);                                                 \
        lo[0] = sn.ilo[0]; lo[1] = sn.ilo[1]; lo[2] = sn.ksmall;                                       \
        hi[0] = sn.ilo[0] + sn.n[0] - 1; hi[1] = sn.ilo[1] + sn.n[1] - 1; hi[2] = sn.kbig;             \
        return (RET)WXA_OK;                                                                            \
    }                                                                                                  \
    /* Evolve without / with the velocity synchronisation at the end of the call, and the synchronisation alone */ \
    RET PFX##sim_set_synchronize_at_end(SIMTYPE* s, int32_t on) {                                      \
        if (!s) return (RET)WXA_ERR_INVALID_ARG;                                                       \
        reinterpret_cast<wxa::host::SimHandle*>(s)->warpx->synchronize_at_end = on != 0;               \
        return (RET)WXA_OK;                                                                            \
    }                                                                                                  \
    RET PFX##sim_set_safe_guard_cells(SIMTYPE* s, int32_t on) {                                        \
        if (!s) return (RET)WXA_ERR_INVALID_ARG;                                                       \
        try {                                                                                          \
            reinterpret_cast<wxa::host::SimHandle*>(s)->warpx->SetSafeGuardCells(on != 0);             \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    RET PFX##sim_set_single_precision_comms(SIMTYPE* s, int32_t on) {                                  \
        if (!s) return (RET)WXA_ERR_INVALID_ARG;                                                       \
        try {                                                                                          \
            reinterpret_cast<wxa::host::SimHandle*>(s)->warpx->SetSinglePrecisionComms(on != 0);       \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    RET PFX##sim_synchronize(SIMTYPE* s) {                                                             \
        if (!s) return (RET)WXA_ERR_INVALID_ARG;                                                       \
        auto* h = reinterpret_cast<wxa::host::SimHandle*>(s);                                          \
        try {                                                                                          \
            h->warpx->SynchronizeNow();                                                                \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_HIP;                                                                   \
        }                                                                                              \
    }                                                                                                  \
    /* the expression evaluator of the decks, for tests and tools */                                   \
    RET PFX##parser_eval(const char* expr, int32_t nvars, const char* const* names, const double* values, \
                         double* out) {                                                                \
        if (!expr || !out || nvars < 0) return (RET)WXA_ERR_INVALID_ARG;                               \
        try {                                                                                          \
            std::vector<std::string> vars;                                                             \
            for (int32_t i = 0; i < nvars; ++i) vars.emplace_back(names[i]);                           \
            wxa::host::ParmParse pp;                                                                   \
            *out = wxa::host::Parser(expr, vars, pp.constants()).eval(values);                         \
            return (RET)WXA_OK;                                                                        \
        } catch (const std::exception& e) {                                                            \
            SET_ERROR(e.what());                                                                       \
            return (RET)WXA_ERR_INVALID_ARG;                                                           \
        }                                                                                              \
    }                                                                                                  \
    }

#endif

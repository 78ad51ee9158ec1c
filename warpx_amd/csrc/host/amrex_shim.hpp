// Source-compatible shim types for exactly the AMReX / ablastr symbols that the
// explicit-FDTD branch of Source/Evolve/WarpXEvolve.cpp touches (hard part 4).
// One process owns ONE brick on ONE GPU, so a MultiFab here is a single box
// (amrex::MultiFab with one FArrayBox) and there is no MFIter.
#ifndef WXA_HOST_AMREX_SHIM_HPP_
#define WXA_HOST_AMREX_SHIM_HPP_

#include <array>
#include <cmath>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "backend.hpp"

namespace amrex {

using Real = double;          // WarpX_PRECISION = DOUBLE (CMakeLists.txt:109-118)
using ParticleReal = double;  // WarpX_PARTICLE_PRECISION = DOUBLE
using Long = int64_t;

struct IntVect {
    std::array<int, 3> v{0, 0, 0};
    IntVect() = default;
    explicit IntVect(int a) : v{a, a, a} {}
    IntVect(int a, int b, int c) : v{a, b, c} {}
    int& operator[](int d) { return v[d]; }
    int operator[](int d) const { return v[d]; }
    const int* data() const { return v.data(); }
    IntVect operator+(const IntVect& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    IntVect operator-(const IntVect& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    bool operator==(const IntVect& o) const { return v == o.v; }
    bool allLE(const IntVect& o) const { return v[0] <= o.v[0] && v[1] <= o.v[1] && v[2] <= o.v[2]; }
    static IntVect TheZeroVector() { return IntVect(0); }
    static IntVect TheUnitVector() { return IntVect(1); }
};

inline IntVect min(const IntVect& a, const IntVect& b) {
    return {std::min(a[0], b[0]), std::min(a[1], b[1]), std::min(a[2], b[2])};
}

// Cell-centred index box [lo, hi] inclusive, like amrex::Box
struct Box {
    IntVect lo, hi;
    Box() = default;
    Box(const IntVect& l, const IntVect& h) : lo(l), hi(h) {}
    Box& grow(const IntVect& n) { lo = lo - n; hi = hi + n; return *this; }
    int length(int d) const { return hi[d] - lo[d] + 1; }
    Long numPts() const { return (Long)length(0) * length(1) * length(2); }
};

struct Periodicity {
    IntVect period;  // domain length in cells along periodic directions, 0 otherwise
    bool isPeriodic(int d) const { return period[d] > 0; }
};

// One staggered component with guard cells on the device (amrex::MultiFab, one box).
class MultiFab {
public:
    MultiFab(const wxa::host::Backend* be, const Box& cell_box, const IntVect& ixtype, const IntVect& ngrow,
             bool pad_rows = true)
        : m_be(be), m_box(cell_box), m_type(ixtype), m_ng(ngrow) {
        for (int d = 0; d < 3; ++d) {
            m_view.lo[d] = cell_box.lo[d] - ngrow[d];
            m_view.n[d] = cell_box.length(d) + ixtype[d] + 2 * ngrow[d];
            m_view.ng[d] = ngrow[d];
            m_view.stag[d] = ixtype[d];
        }
        // rows padded to 128 B and the first valid point of each row 128-B aligned, so that
        // the stencil kernels issue full-line coalesced loads on the staggered (n / n+1) rows
        const int64_t js = pad_rows ? (m_view.n[0] + 15) / 16 * 16 : m_view.n[0];
        const int64_t front = pad_rows ? (16 - ngrow[0] % 16) % 16 : 0;
        m_view.jstride = js;
        m_view.kstride = js * m_view.n[1];
        m_bytes = sizeof(double) * (size_t)(front + m_view.kstride * m_view.n[2]);
        m_alloc = be->dmalloc(m_bytes);
        if (!m_alloc) throw std::runtime_error("MultiFab: device allocation failed");
        m_view.p = static_cast<double*>(m_alloc) + front;
        be->memset_async(m_alloc, 0, m_bytes, nullptr);
        be->stream_sync(nullptr);
    }
    ~MultiFab() { if (m_alloc) m_be->dfree(m_alloc); }
    MultiFab(const MultiFab&) = delete;
    MultiFab& operator=(const MultiFab&) = delete;

    const wxa_field_view& view() const { return m_view; }
    IntVect nGrowVect() const { return m_ng; }
    IntVect ixType() const { return m_type; }
    const Box& box() const { return m_box; }
    void setVal(Real val, void* stream = nullptr) {
        if (val != 0.0) throw std::runtime_error("MultiFab::setVal: only 0 is supported on this path");
        m_be->field_set_zero(&m_view, stream);
    }
    size_t bytes() const { return m_bytes; }
    // exchange the storage of two identically shaped MultiFabs (used instead of a copy-back)
    void swap_storage(MultiFab& o) {
        for (int d = 0; d < 3; ++d)
            if (m_view.n[d] != o.m_view.n[d] || m_view.lo[d] != o.m_view.lo[d])
                throw std::runtime_error("MultiFab::swap_storage: shapes differ");
        if (m_view.jstride != o.m_view.jstride || m_view.kstride != o.m_view.kstride || m_bytes != o.m_bytes)
            throw std::runtime_error("MultiFab::swap_storage: layouts differ");
        std::swap(m_alloc, o.m_alloc);
        std::swap(m_view.p, o.m_view.p);
    }

private:
    const wxa::host::Backend* m_be;
    Box m_box;
    IntVect m_type, m_ng;
    wxa_field_view m_view{};
    void* m_alloc = nullptr;
    size_t m_bytes = 0;
};

}  // namespace amrex

// Source/ablastr/utils/Enums.H:19-33, Source/Evolve/WarpXDtType.H:10-15, WarpXPushType.H:11-16
enum struct PatchType : int { fine = 0, coarse = 1 };
enum struct DtType : int { Full = 0, FirstHalf = 1, SecondHalf = 2 };
enum struct PushType : int { Explicit = 0, Implicit = 1 };
// Source/Utils/WarpXAlgorithmSelection.H:72-84
enum struct ParticlePusherAlgo : int { Boris = 0, Vay = 1 };
enum struct CurrentDepositionAlgo : int { Esirkepov = 0, Direct = 1 };

namespace warpx::fields {
// the subset of Source/Fields.H FieldType used on this path
enum struct FieldType : int { Efield_fp, Bfield_fp, current_fp, Efield_aux, Bfield_aux, current_buf };
}

namespace ablastr::fields {

struct Direction {
    int dir;
    operator int() const { return dir; }
};

using VectorField = std::array<amrex::MultiFab*, 3>;

// Named-field registry (Source/ablastr/fields/MultiFabRegister.H:161-660): owns every
// MultiFab, hands out non-owning pointers; aliases share storage (alias_init :273-320).
class MultiFabRegister {
public:
    explicit MultiFabRegister(const wxa::host::Backend* be) : m_be(be) {}

    amrex::MultiFab* alloc_init(warpx::fields::FieldType name, Direction dir, int level, const amrex::Box& box,
                                const amrex::IntVect& ixtype, const amrex::IntVect& ngrow,
                                std::optional<amrex::Real> initial_value = 0.0) {
        (void)initial_value;
        const Key k{(int)name, dir.dir, level};
        if (m_owned.count(k) || m_alias.count(k)) throw std::runtime_error("MultiFabRegister: field already registered");
        m_owned[k] = std::make_unique<amrex::MultiFab>(m_be, box, ixtype, ngrow);
        return m_owned[k].get();
    }
    amrex::MultiFab* alias_init(warpx::fields::FieldType new_name, warpx::fields::FieldType alias_name,
                                Direction dir, int level) {
        const Key kn{(int)new_name, dir.dir, level}, ka{(int)alias_name, dir.dir, level};
        if (!m_owned.count(ka)) throw std::runtime_error("MultiFabRegister::alias_init: unknown source field");
        m_alias[kn] = m_owned[ka].get();
        return m_alias[kn];
    }
    bool has(warpx::fields::FieldType name, Direction dir, int level) const {
        const Key k{(int)name, dir.dir, level};
        return m_owned.count(k) || m_alias.count(k);
    }
    amrex::MultiFab* get(warpx::fields::FieldType name, Direction dir, int level) const {
        const Key k{(int)name, dir.dir, level};
        auto it = m_owned.find(k);
        if (it != m_owned.end()) return it->second.get();
        auto ia = m_alias.find(k);
        if (ia != m_alias.end()) return ia->second;
        throw std::runtime_error("MultiFabRegister::get: field not registered");
    }
    VectorField get_alldirs(warpx::fields::FieldType name, int level) const {
        return {get(name, Direction{0}, level), get(name, Direction{1}, level), get(name, Direction{2}, level)};
    }

private:
    using Key = std::array<int, 3>;
    const wxa::host::Backend* m_be;
    std::map<Key, std::unique_ptr<amrex::MultiFab>> m_owned;
    std::map<Key, amrex::MultiFab*> m_alias;
};

}  // namespace ablastr::fields
#endif

// Guard-cell exchange and particle migration between bricks (one brick per GPU).
//
// Replaces what the reference gets from amrex FabArray::FillBoundary /
// FillBoundaryAndSync / SumBoundary and ParticleContainer::Redistribute over MPI
// (Source/ablastr/utils/Communication.cpp:71-175,
// Source/Parallelization/WarpXSumGuardCells.cpp:17-37).  Design for xGMI: the domain is
// periodic and split into nbricks[0] x nbricks[1] x nbricks[2] bricks; exchanges run
// direction by direction (x, then y, then z), each direction moving two face slabs that
// already contain the guards filled by the previous directions, so edges and corners need
// no extra messages: 6 large point-to-point transfers per exchange instead of 26 small
// ones.  A direction with a single brick is self-periodic and handled on the device with
// no communication at all.  The bytes themselves are moved by the host program's
// `wxa_comm::exchange` callback (torch.distributed P2P over RCCL in bench.py).
#ifndef WXA_HOST_BRICKCOMM_HPP_
#define WXA_HOST_BRICKCOMM_HPP_

#include "amrex_shim.hpp"

namespace wxa::host {

struct DeviceBuffer {
    const Backend* be = nullptr;
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t bytes) {
        if (bytes <= cap) return;
        if (p) be->dfree(p);
        cap = bytes + bytes / 4 + 1024;
        p = be->dmalloc(cap);
        if (!p) throw std::runtime_error("DeviceBuffer: allocation failed");
    }
    ~DeviceBuffer() { if (p) be->dfree(p); }
};

class BrickComm {
public:
    BrickComm(const Backend* be, const wxa_comm* comm, const int nbricks[3], const int coord[3]) : m_be(be) {
        if (comm) { m_comm = *comm; m_has_comm = true; }
        for (int d = 0; d < 3; ++d) { m_nb[d] = nbricks[d]; m_coord[d] = coord[d]; }
        for (auto& b : m_send) b.be = be;
        for (auto& b : m_recv) b.be = be;
        const int total = m_nb[0] * m_nb[1] * m_nb[2];
        if (total > 1 && !m_has_comm) throw std::runtime_error("BrickComm: more than one brick needs a wxa_comm");
        if (m_has_comm && m_comm.nranks != total) throw std::runtime_error("BrickComm: nranks != number of bricks");
    }

    bool self_periodic(int d) const { return m_nb[d] == 1; }
    int rank_of(const int c[3]) const { return c[0] + m_nb[0] * (c[1] + m_nb[1] * c[2]); }
    int neighbor(int d, int side) const {  // side 0 = minus, 1 = plus
        int c[3] = {m_coord[0], m_coord[1], m_coord[2]};
        c[d] = (c[d] + (side ? 1 : -1) + m_nb[d]) % m_nb[d];
        return rank_of(c);
    }
    const int* nbricks() const { return m_nb; }
    const int* coord() const { return m_coord; }

    // FabArray::FillBoundary(ng, period) / FillBoundaryAndSync when nodal_sync.
    void FillBoundary(amrex::MultiFab& mf, const amrex::IntVect& ng, bool nodal_sync, void* stream) {
        const wxa_field_view& f = mf.view();
        int lo[3], hi[3];
        for (int d = 0; d < 3; ++d) {
            if (ng[d] > f.ng[d]) throw std::runtime_error("FillBoundary: ng exceeds allocated guard cells");
            lo[d] = f.lo[d] + f.ng[d];
            hi[d] = f.lo[d] + f.n[d] - f.ng[d];
        }
        for (int d = 0; d < 3; ++d) {
            if (ng[d] <= 0 && !(nodal_sync && f.stag[d])) continue;
            if (self_periodic(d)) {
                if (nodal_sync && f.stag[d]) self_op(f, d, 0, lo, hi, /*sync=*/true, stream);
                self_op(f, d, ng[d], lo, hi, /*sync=*/false, stream);
            } else {
                const int v0 = f.lo[d] + f.ng[d], v1 = f.lo[d] + f.n[d] - f.ng[d];
                const int sync = (nodal_sync && f.stag[d]) ? 1 : 0;
                // to minus: my low slab (fills its high guards; with sync also its high-edge node)
                int32_t smlo[3], smhi[3], splo[3], sphi[3], rplo[3], rphi[3], rmlo[3], rmhi[3];
                for (int e = 0; e < 3; ++e) {
                    smlo[e] = splo[e] = rplo[e] = rmlo[e] = lo[e];
                    smhi[e] = sphi[e] = rphi[e] = rmhi[e] = hi[e];
                }
                smlo[d] = v0 + (sync ? 0 : f.stag[d]); smhi[d] = v0 + f.stag[d] + ng[d];
                splo[d] = v1 - f.stag[d] - ng[d];      sphi[d] = v1 - f.stag[d];
                rplo[d] = v1 - (sync ? f.stag[d] : 0); rphi[d] = v1 + ng[d];   // from plus neighbour
                rmlo[d] = v0 - ng[d];                  rmhi[d] = v0;            // from minus neighbour
                exchange_slabs(f, d, smlo, smhi, splo, sphi, rplo, rphi, rmlo, rmhi, /*mode=*/0, stream);
            }
            lo[d] = f.lo[d] + f.ng[d] - ng[d];
            hi[d] = f.lo[d] + f.n[d] - f.ng[d] + ng[d];
        }
    }

    // FabArray::SumBoundary(src_ng, dst_ng): valid points receive every image's deposit;
    // guards are refreshed afterwards only where that is free (self-periodic directions)
    // or when refresh_guards is set -- nothing on the step path reads J guards after this.
    void SumBoundary(amrex::MultiFab& mf, const amrex::IntVect& src_ng, bool refresh_guards, void* stream) {
        const wxa_field_view& f = mf.view();
        for (int d = 0; d < 3; ++d) {
            if (self_periodic(d)) {
                int per[3] = {0, 0, 0}, g[3] = {0, 0, 0};
                per[d] = 1; g[d] = src_ng[d];
                check(m_be->sum_boundary_periodic(&f, g, per, stream));
            } else {
                const int v0 = f.lo[d] + f.ng[d], v1 = f.lo[d] + f.n[d] - f.ng[d];
                const int sng = src_ng[d], st = f.stag[d];
                int32_t smlo[3], smhi[3], splo[3], sphi[3], rplo[3], rphi[3], rmlo[3], rmhi[3];
                for (int e = 0; e < 3; ++e) {  // transverse: the whole allocation
                    smlo[e] = splo[e] = rplo[e] = rmlo[e] = f.lo[e];
                    smhi[e] = sphi[e] = rphi[e] = rmhi[e] = f.lo[e] + f.n[e];
                }
                smlo[d] = v0 - sng;      smhi[d] = v0 + st;        // low guards (+ shared node) -> minus
                splo[d] = v1 - st;       sphi[d] = v1 + sng;       // (shared node +) high guards -> plus
                rplo[d] = v1 - st - sng; rphi[d] = v1;             // added from plus neighbour
                rmlo[d] = v0;            rmhi[d] = v0 + st + sng;  // added from minus neighbour
                exchange_slabs(f, d, smlo, smhi, splo, sphi, rplo, rphi, rmlo, rmhi, /*mode=*/1, stream);
            }
        }
        if (refresh_guards) {
            bool any = false;
            for (int d = 0; d < 3; ++d) any = any || !self_periodic(d);
            if (any) FillBoundary(mf, mf.nGrowVect(), false, stream);
        }
    }

    // post `n` (<= 2) sends/recvs of raw device buffers with the +/- neighbours in direction d
    void exchange_raw(int d, void* send_minus, int64_t sm_bytes, void* send_plus, int64_t sp_bytes,
                      void* recv_plus, int64_t rp_bytes, void* recv_minus, int64_t rm_bytes, void* stream) {
        int32_t speer[2] = {neighbor(d, 0), neighbor(d, 1)};
        int32_t rpeer[2] = {neighbor(d, 1), neighbor(d, 0)};
        void* sb[2] = {send_minus, send_plus};
        void* rb[2] = {recv_plus, recv_minus};
        int64_t sbytes[2] = {sm_bytes, sp_bytes};
        int64_t rbytes[2] = {rp_bytes, rm_bytes};
        if (m_comm.exchange(m_comm.ctx, 2, speer, sb, sbytes, rpeer, rb, rbytes, stream) != 0)
            throw std::runtime_error("BrickComm: exchange callback failed");
    }
    void exchange_counts(int d, int64_t to_minus, int64_t to_plus, int64_t& from_plus, int64_t& from_minus) {
        int32_t speer[2] = {neighbor(d, 0), neighbor(d, 1)};
        int32_t rpeer[2] = {neighbor(d, 1), neighbor(d, 0)};
        int64_t sv[2] = {to_minus, to_plus};
        int64_t rv[2] = {0, 0};
        if (m_comm.exchange_counts(m_comm.ctx, 2, speer, sv, rpeer, rv) != 0)
            throw std::runtime_error("BrickComm: exchange_counts callback failed");
        from_plus = rv[0];
        from_minus = rv[1];
    }

    static void check(int rc) {
        if (rc != 0) throw std::runtime_error("BrickComm: kernel returned status " + std::to_string(rc));
    }

private:
    // self-periodic fill (or nodal sync) along d over the transverse box [lo,hi) of the other
    // directions: restrict the view transversally so the device routine touches exactly that box
    void self_op(const wxa_field_view& f, int d, int ng, const int lo[3], const int hi[3], bool sync,
                 void* stream) {
        if (!sync && ng <= 0) return;
        wxa_field_view v = f;
        for (int e = 0; e < 3; ++e) {
            if (e == d) continue;
            const int64_t shift = lo[e] - f.lo[e];
            v.p += (e == 0 ? shift : e == 1 ? shift * f.jstride : shift * f.kstride);
            v.lo[e] = lo[e];
            v.n[e] = hi[e] - lo[e];
            v.ng[e] = 0;
            v.stag[e] = 0;
        }
        int per[3] = {0, 0, 0}, g[3] = {0, 0, 0};
        per[d] = 1; g[d] = ng;
        if (sync) {
            v.stag[d] = f.stag[d];
            check(m_be->sync_nodal_periodic(&v, per, stream));
        } else {
            check(m_be->fill_boundary_periodic(&v, g, per, stream));
        }
    }

    static int64_t box_pts(const int32_t lo[3], const int32_t hi[3]) {
        int64_t n = 1;
        for (int e = 0; e < 3; ++e) n *= std::max(0, hi[e] - lo[e]);
        return n;
    }

    void exchange_slabs(const wxa_field_view& f, int d, const int32_t smlo[3], const int32_t smhi[3],
                        const int32_t splo[3], const int32_t sphi[3], const int32_t rplo[3],
                        const int32_t rphi[3], const int32_t rmlo[3], const int32_t rmhi[3], int mode,
                        void* stream) {
        const int64_t nsm = box_pts(smlo, smhi), nsp = box_pts(splo, sphi);
        const int64_t nrp = box_pts(rplo, rphi), nrm = box_pts(rmlo, rmhi);
        m_send[0].reserve(8 * nsm); m_send[1].reserve(8 * nsp);
        m_recv[0].reserve(8 * nrp); m_recv[1].reserve(8 * nrm);
        check(m_be->pack_box(&f, smlo, smhi, (double*)m_send[0].p, stream));
        check(m_be->pack_box(&f, splo, sphi, (double*)m_send[1].p, stream));
        exchange_raw(d, m_send[0].p, 8 * nsm, m_send[1].p, 8 * nsp, m_recv[0].p, 8 * nrp, m_recv[1].p, 8 * nrm,
                     stream);
        check(m_be->unpack_box(&f, rplo, rphi, (const double*)m_recv[0].p, mode, stream));
        check(m_be->unpack_box(&f, rmlo, rmhi, (const double*)m_recv[1].p, mode, stream));
    }

    const Backend* m_be;
    wxa_comm m_comm{};
    bool m_has_comm = false;
    int m_nb[3], m_coord[3];
    DeviceBuffer m_send[2], m_recv[2];
};

}  // namespace wxa::host
#endif

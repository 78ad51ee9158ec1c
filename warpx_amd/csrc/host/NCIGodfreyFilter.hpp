// The NCI corrector's filter coefficients: NCIGodfreyFilter (Source/Filter/NCIGodfreyFilter.H:18-52,
// NCIGodfreyFilter.cpp:27-154).  The filter is applied along z only (stencil lengths 1, 1, 5) to E and B before the
// gather of a species' Evolve (PhysicalParticleContainer.cpp:1900-1911, applyNCIFilter :2097-2172); the five coefficients
// come from the fitted tables of Source/Utils/NCIGodfreyTables.H (nci_godfrey_tables.hpp), interpolated in c dt / dz.
#ifndef WXA_HOST_NCIGODFREYFILTER_HPP_
#define WXA_HOST_NCIGODFREYFILTER_HPP_

#include "nci_godfrey_tables.hpp"

#include <algorithm>

namespace wxa::host {

enum struct godfrey_coeff_set : int { Ex_Ey_Bz = 0, Bx_By_Ez = 1 };   // NCIGodfreyFilter.H:18

class NCIGodfreyFilter {
public:
    static constexpr int m_stencil_width = 4;   // NCIGodfreyFilter.H:38 (guard cells the filter needs along z)

    NCIGodfreyFilter(godfrey_coeff_set coeff_set, double cdtodz, bool nodal_gather)
        : m_coeff_set(coeff_set), m_cdtodz(cdtodz), m_nodal_gather(nodal_gather) {}

    // NCIGodfreyFilter::ComputeStencils (:45-154)
    void ComputeStencils() {
        using namespace nci_godfrey;
        // :57-61 interpolate the coefficients from the table.  (weight_right is taken from the reference as it stands:
        // cdtodz minus the row's position index / tab_length.)
        int index = static_cast<int>(tab_length * m_cdtodz);
        index = std::min(index, tab_length - 2);
        index = std::max(index, 0);
        const double weight_right = m_cdtodz - double(index) / double(tab_length);
        // :66-98 Galerkin tables for a gather from the staggered grid, momentum-conserving ones for a nodal gather
        const int set = (m_nodal_gather ? 2 : 0) + (m_coeff_set == godfrey_coeff_set::Ex_Ey_Bz ? 0 : 1);
        const double* tab = tables + (size_t)set * tab_length * tab_width;
        double prestencil[4];
        for (int i = 0; i < tab_width; i++)
            prestencil[i] = (1. - weight_right) * tab[index * tab_width + i] + weight_right * tab[(index + 1) * tab_width + i];
        // :100-105
        stencil_z[0] = (256 + 128 * prestencil[0] + 96 * prestencil[1] + 80 * prestencil[2] + 70 * prestencil[3]) / 256;
        stencil_z[1] = -(64 * prestencil[0] + 64 * prestencil[1] + 60 * prestencil[2] + 56 * prestencil[3]) / 256;
        stencil_z[2] = (16 * prestencil[1] + 24 * prestencil[2] + 28 * prestencil[3]) / 256;
        stencil_z[3] = -(4 * prestencil[2] + 8 * prestencil[3]) / 256;
        stencil_z[4] = (1 * prestencil[3]) / 256;
        // :118-126 no filter along x and y; Filter::DoFilter visits coefficient 0 twice
        stencil_x[0] = 1. / 2.;
        stencil_y[0] = 1. / 2.;
        stencil_z[0] /= 2.;
    }

    double stencil_x[1] = {0.5}, stencil_y[1] = {0.5}, stencil_z[5] = {0.5, 0., 0., 0., 0.};

private:
    godfrey_coeff_set m_coeff_set;
    double m_cdtodz;
    bool m_nodal_gather;
};

}  // namespace wxa::host
#endif

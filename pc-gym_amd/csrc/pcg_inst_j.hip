// pcg_inst_j.hip -- kernel instantiations for: multistage_extraction_reactive with eq_exponent == 2
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_me_reactive_sq() { return make_kernels<PCG_KID_ME_REACTIVE_SQ>(); }
}  // namespace pcg

// pcg_inst_i.hip -- kernel instantiations for: multistage_extraction with eq_exponent == 2 (multiply-only RHS)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_me_sq() { return make_kernels<PCG_KID_ME_SQ>(); }
}  // namespace pcg

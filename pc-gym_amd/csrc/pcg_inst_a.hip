// pcg_inst_a.hip -- kernel instantiations for: cstr, four_tank, affine  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_cstr() { return make_kernels<PCG_MODEL_CSTR>(); }
Kernels kernels_four_tank() { return make_kernels<PCG_MODEL_FOUR_TANK>(); }
Kernels kernels_affine() { return make_kernels<PCG_MODEL_AFFINE>(); }
}  // namespace pcg

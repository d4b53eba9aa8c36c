// pcg_inst_h.hip -- kernel instantiations for: heat_ex  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_heat_ex() { return make_kernels<PCG_MODEL_HEAT_EX>(); }
}  // namespace pcg

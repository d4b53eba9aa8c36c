// pcg_inst_c.hip -- kernel instantiations for: me_reactive  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_me_reactive() { return make_kernels<PCG_MODEL_ME_REACTIVE>(); }
}  // namespace pcg

// pcg_inst_b.hip -- kernel instantiations for: me  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_me() { return make_kernels<PCG_MODEL_ME>(); }
}  // namespace pcg

// pcg_inst_g.hip -- kernel instantiations for: biofilm  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_biofilm() { return make_kernels<PCG_MODEL_BIOFILM>(); }
}  // namespace pcg

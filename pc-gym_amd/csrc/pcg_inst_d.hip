// pcg_inst_d.hip -- kernel instantiations for: cryst, disease, inv_batch  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_cryst() { return make_kernels<PCG_MODEL_CRYST>(); }
Kernels kernels_disease() { return make_kernels<PCG_MODEL_DISEASE>(); }
Kernels kernels_inv_batch() { return make_kernels<PCG_MODEL_INV_BATCH>(); }
}  // namespace pcg

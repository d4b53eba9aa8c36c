// pcg_inst_e.hip -- kernel instantiations for: complex_cstr, batch, photo, polymer  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_complex_cstr() { return make_kernels<PCG_MODEL_COMPLEX_CSTR>(); }
Kernels kernels_batch() { return make_kernels<PCG_MODEL_BATCH>(); }
Kernels kernels_photo() { return make_kernels<PCG_MODEL_PHOTO>(); }
Kernels kernels_polymer() { return make_kernels<PCG_MODEL_POLYMER>(); }
}  // namespace pcg

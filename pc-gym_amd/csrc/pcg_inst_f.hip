// pcg_inst_f.hip -- kernel instantiations for: cstr_series, distillation, oscillators  (see pcg_kernels.hpp)
#include "pcg_kernels.hpp"

namespace pcg {
Kernels kernels_cstr_series() { return make_kernels<PCG_MODEL_CSTR_SERIES>(); }
Kernels kernels_distillation() { return make_kernels<PCG_MODEL_DISTILLATION>(); }
Kernels kernels_oscillators() { return make_kernels<PCG_MODEL_OSCILLATORS>(); }
}  // namespace pcg

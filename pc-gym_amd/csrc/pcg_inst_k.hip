// pcg_inst_k.hip -- feature-masked pipelined step kernels (pcg_step_feat.hpp) of the small HBM-bound models
#include "pcg_kernels.hpp"
#include "pcg_step_feat.hpp"

namespace pcg {
int feat_fill_cstr(FeatEntry* out, int cap) { return feat_table<Model<PCG_MODEL_CSTR>>(out, cap); }
int feat_fill_four_tank(FeatEntry* out, int cap) { return feat_table<Model<PCG_MODEL_FOUR_TANK>>(out, cap); }
}  // namespace pcg

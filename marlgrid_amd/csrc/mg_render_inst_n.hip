// mg_render_inst_n.hip — instantiations of mg::render_kernel, group N (mg_render_kernel.h: MG_RENDER_GROUP_N): the fused step
// that also writes MultiGrid.encode of its batch (mg_step_render_encode)
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif
namespace mg {
#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_N(MG_RENDER_INSTANTIATE)
#endif
}  // namespace mg

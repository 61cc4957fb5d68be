// mg_render_inst_b.hip — instantiations of mg::render_kernel, group B (mg_render_kernel.h: MG_RENDER_GROUP_B)
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif
namespace mg {
#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_B(MG_RENDER_INSTANTIATE)
#endif
}  // namespace mg

// mg_render_inst_g.hip — instantiations of mg::render_kernel, group G (mg_render_kernel.h: MG_RENDER_GROUP_G)
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif
namespace mg {
#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_G(MG_RENDER_INSTANTIATE)
MG_RENDER_GROUP_X(MG_RENDER_INSTANTIATE)
#endif
}  // namespace mg

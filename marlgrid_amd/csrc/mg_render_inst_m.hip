// mg_render_inst_m.hip — instantiations of mg::render_kernel, group M (mg_render_kernel.h: MG_RENDER_GROUP_M)
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif
namespace mg {
#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_M(MG_RENDER_INSTANTIATE)
#endif
}  // namespace mg

// mg_render_inst_k.hip — instantiations of mg::render_kernel, group K (mg_render_kernel.h: MG_RENDER_GROUP_K)
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif
namespace mg {
#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_K(MG_RENDER_INSTANTIATE)
#endif
}  // namespace mg

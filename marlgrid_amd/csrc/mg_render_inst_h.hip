// mg_render_inst_h.hip — instantiations of mg::render_kernel, group H (mg_render_kernel.h: MG_RENDER_GROUP_H)
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif
namespace mg {
#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_H(MG_RENDER_INSTANTIATE)
#endif
}  // namespace mg

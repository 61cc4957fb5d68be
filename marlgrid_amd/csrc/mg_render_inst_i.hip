// mg_render_inst_i.hip — instantiations of mg::render_kernel, group I (mg_render_kernel.h: MG_RENDER_GROUP_I)
#include "mg_render_kernel.h"
#if defined(MG_AB_VARIANTS)
#include <stdlib.h>
#endif
namespace mg {
#if !defined(MG_DEV_ONLY)
MG_RENDER_GROUP_I(MG_RENDER_INSTANTIATE)
#endif
}  // namespace mg

/*
 * rows/everything.hip - the all-features kernel, the ray census (COUNT == 1) and the volume camera's instantiations
 * (one object per row of renderImpl's table: see renderer.h).  gfx950 only.
 */
#include "../renderer_kernel.h"

namespace solrrows
{
RendererFn everything(int count, int features, bool volume)
{
    if ((features & ~F_DEEP) != F_ALL)
        return nullptr;
    if (count == 1)
        return volume ? k_standardRenderer<1, F_ALL, true> : k_standardRenderer<1, F_ALL>;
    if (count == 0)
        return volume ? k_standardRenderer<0, F_ALL | F_DEEP, true> : k_standardRenderer<0, F_ALL | F_DEEP>;
    return nullptr;
}
} // namespace solrrows

/*
 * rows/special_cameras.hip - the special cameras, global illumination and the box-debug view over the usual untextured primitives; every primitive type with textures
 * (one object per row of renderImpl's table: see renderer.h).  gfx950 only.
 */
#include "../renderer_kernel.h"

namespace solrrows
{
RendererFn specialCameras(int count, int features)
{
    if (count != 0)
        return nullptr;
    switch (features & ~F_DEEP)
    {
    case F_SPHERE | F_PLANE | F_TRI | F_CYL | F_FULL: return k_standardRenderer<0, F_SPHERE | F_PLANE | F_TRI | F_CYL | F_FULL | F_DEEP>;
    case F_ALL & ~F_FULL: return k_standardRenderer<0, (F_ALL & ~F_FULL) | F_DEEP>;
    default: return nullptr;
    }
}
} // namespace solrrows

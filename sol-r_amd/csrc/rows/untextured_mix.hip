/*
 * rows/untextured_mix.hip - spheres, planes, triangles and cylinders without textures (three-bank node loop only)
 * (one object per row of renderImpl's table: see renderer.h).  gfx950 only.
 */
#include "../renderer_kernel.h"

namespace solrrows
{
RendererFn untexturedMix(int count, int features)
{
    if ((features & ~(F_DEEP | F_STACK | F_STREAM)) != (F_SPHERE | F_PLANE | F_TRI | F_CYL))
        return nullptr;
    if (count == 0 && (features & F_STREAM)) /* a frame whose image leaves in bands while it renders (renderer.h) */
        return (features & F_STACK) ? k_standardRenderer<0, (F_SPHERE | F_PLANE | F_TRI | F_CYL) | F_DEEP | F_STACK | F_STREAM> : k_standardRenderer<0, (F_SPHERE | F_PLANE | F_TRI | F_CYL) | F_DEEP | F_STREAM>;
    if (count == 0 && (features & F_STACK)) /* a frame that may bounce deeper than the LDS stack holds */
        return k_standardRenderer<0, (F_SPHERE | F_PLANE | F_TRI | F_CYL) | F_DEEP | F_STACK>;
    if (features & (F_STACK | F_STREAM))
        return nullptr;
    if (count == 0)
        return k_standardRenderer<0, (F_SPHERE | F_PLANE | F_TRI | F_CYL) | F_DEEP>;
    if (count == 2)
        return k_standardRenderer<2, (F_SPHERE | F_PLANE | F_TRI | F_CYL) | F_DEEP>;
    return nullptr;
}

/* (each lean row file answers for its own row; the others return null) */
WalkBoundFn walkBoundRow3(int features)
{
    const int row = 3;
    if (row == 3)
        return k_walkBound<(F_SPHERE | F_PLANE | F_TRI | F_CYL) | F_DEEP>;
    return nullptr;
}
} // namespace solrrows

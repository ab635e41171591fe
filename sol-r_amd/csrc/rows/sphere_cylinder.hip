/*
 * rows/sphere_cylinder.hip - the lean kernel of spheres + cylinders (molecules)
 * (one object per row of renderImpl's table: see renderer.h).  gfx950 only.
 */
#include "../renderer_kernel.h"

namespace solrrows
{
RendererFn sphereCylinder(int count, int features)
{
    if ((features & ~(F_DEEP | F_STACK | F_STREAM)) != (F_SPHERE | F_CYL))
        return nullptr;
    const bool deep = (features & F_DEEP) != 0;
    if (count == 0 && (features & F_STREAM)) /* a frame whose image leaves in bands while it renders (renderer.h) */
        return (features & F_STACK) ? (deep ? k_standardRenderer<0, (F_SPHERE | F_CYL) | F_DEEP | F_STACK | F_STREAM> : k_standardRenderer<0, (F_SPHERE | F_CYL) | F_STACK | F_STREAM>)
                                    : (deep ? k_standardRenderer<0, (F_SPHERE | F_CYL) | F_DEEP | F_STREAM> : k_standardRenderer<0, (F_SPHERE | F_CYL) | F_STREAM>);
    if (count == 0 && (features & F_STACK)) /* a frame that may bounce deeper than the LDS stack holds */
        return deep ? k_standardRenderer<0, (F_SPHERE | F_CYL) | F_DEEP | F_STACK> : k_standardRenderer<0, (F_SPHERE | F_CYL) | F_STACK>;
    if (features & (F_STACK | F_STREAM))
        return nullptr;
    if (count == 0)
        return deep ? k_standardRenderer<0, (F_SPHERE | F_CYL) | F_DEEP> : k_standardRenderer<0, (F_SPHERE | F_CYL)>;
    if (count == 2)
        return deep ? k_standardRenderer<2, (F_SPHERE | F_CYL) | F_DEEP> : k_standardRenderer<2, (F_SPHERE | F_CYL)>;
    return nullptr;
}

/* (each lean row file answers for its own row; the others return null) */
WalkBoundFn walkBoundRow2(int features)
{
    const int row = 2;
    if (row == 2)
        return (features & F_DEEP) ? k_walkBound<(F_SPHERE | F_CYL) | F_DEEP> : k_walkBound<(F_SPHERE | F_CYL)>;
    return nullptr;
}
} // namespace solrrows

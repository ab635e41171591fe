/*
 * rows/sphere_plane.hip - the lean kernel of spheres + axis planes (the Cornell box): frames, recorded frames, the walk replay
 * (one object per row of renderImpl's table: see renderer.h).  gfx950 only.
 */
#include "../renderer_kernel.h"

namespace solrrows
{
RendererFn spherePlane(int count, int features)
{
    if ((features & ~(F_DEEP | F_STACK | F_STREAM)) != (F_SPHERE | F_PLANE))
        return nullptr;
    const bool deep = (features & F_DEEP) != 0;
    if (count == 0 && (features & F_STREAM)) /* a frame whose image leaves in bands while it renders (renderer.h) */
        return (features & F_STACK) ? (deep ? k_standardRenderer<0, (F_SPHERE | F_PLANE) | F_DEEP | F_STACK | F_STREAM> : k_standardRenderer<0, (F_SPHERE | F_PLANE) | F_STACK | F_STREAM>)
                                    : (deep ? k_standardRenderer<0, (F_SPHERE | F_PLANE) | F_DEEP | F_STREAM> : k_standardRenderer<0, (F_SPHERE | F_PLANE) | F_STREAM>);
    if (count == 0 && (features & F_STACK)) /* a frame that may bounce deeper than the LDS stack holds */
        return deep ? k_standardRenderer<0, (F_SPHERE | F_PLANE) | F_DEEP | F_STACK> : k_standardRenderer<0, (F_SPHERE | F_PLANE) | F_STACK>;
    if (features & (F_STACK | F_STREAM))
        return nullptr;
    if (count == 0)
        return deep ? k_standardRenderer<0, (F_SPHERE | F_PLANE) | F_DEEP> : k_standardRenderer<0, (F_SPHERE | F_PLANE)>;
    if (count == 2)
        return deep ? k_standardRenderer<2, (F_SPHERE | F_PLANE) | F_DEEP> : k_standardRenderer<2, (F_SPHERE | F_PLANE)>;
    return nullptr;
}

/* (each lean row file answers for its own row; the others return null) */
WalkBoundFn walkBoundRow0(int features)
{
    const int row = 0;
    if (row == 0)
        return (features & F_DEEP) ? k_walkBound<(F_SPHERE | F_PLANE) | F_DEEP> : k_walkBound<(F_SPHERE | F_PLANE)>;
    return nullptr;
}
} // namespace solrrows

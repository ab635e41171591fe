/*
 * rows/textured.hip - the texture tier over the usual primitives: textured sphere + triangle scenes (OBJ meshes with MTL images), the textured mix
 * (one object per row of renderImpl's table: see renderer.h).  gfx950 only.
 */
#include "../renderer_kernel.h"

namespace solrrows
{
RendererFn textured(int count, int features)
{
    if (count != 0)
        return nullptr;
    switch (features)
    {
    case F_SPHERE | F_TRI | F_TEX: return k_standardRenderer<0, F_SPHERE | F_TRI | F_TEX>;
    case F_SPHERE | F_TRI | F_TEX | F_DEEP: return k_standardRenderer<0, F_SPHERE | F_TRI | F_TEX | F_DEEP>;
    case F_SPHERE | F_PLANE | F_TRI | F_CYL | F_TEX:
    case F_SPHERE | F_PLANE | F_TRI | F_CYL | F_TEX | F_DEEP: return k_standardRenderer<0, F_SPHERE | F_PLANE | F_TRI | F_CYL | F_TEX | F_DEEP>;
    default: return nullptr;
    }
}
} // namespace solrrows

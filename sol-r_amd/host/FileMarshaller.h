/*
 * FileMarshaller.h - the .irt scene file (reference: solr/io/FileMarshaller.{h,cpp}).
 *
 * An .irt file is what the reference's FileMarshaller::saveToFile writes and the viewer's "load scene"
 * reads back: a raw dump, in the reference's 64-bit struct layouts, of
 *
 *     size_t version            2 for the CUDA engine's structs (1: the OpenCL engine's; refused, as there)
 *     SceneInfo                 112 bytes, read and ignored by the loader
 *     size_t nbPrimitives
 *     nbPrimitives x 160 bytes  the reference's CPUPrimitive (GPUKernel.h:46-65 there): two bools, then from
 *                               byte 4 p0 p1 p2 n0 n1 n2 size (7 x float3), type, materialId, vt0 vt1 vt2
 *                               (float2, from byte 96), speed0..2 (float3), 4 bytes of padding
 *     size_t nbTextures, then per texture: size_t id, TextureInfo (32 bytes, pointer field meaningless),
 *                               size.x * size.y * size.z bytes of pixels
 *     size_t nbMaterials, then per material: size_t id, Material (176 bytes)
 *
 * loadFromFile appends the primitives to the kernel (not movable, belonging to the model), the textures
 * after the kernel's current ones (material texture ids are shifted accordingly), overwrites the materials
 * with the ids in the file, and finally rescales EVERY primitive of the kernel so that the model's height
 * becomes `scale` - all as FileMarshaller.cpp:54-192 does.  The reference's sample medias/irt/test.irt
 * (20 958 triangles, 16 materials) is the format's golden vector, tests/test_scene_files.py.
 */
#pragma once

#include <string>

#include "GPUKernel.h"

namespace solr
{
class FileMarshaller
{
public:
    /* returns the extent of the loaded primitives (x, y, z), zeros if the file was refused */
    vec4f loadFromFile(GPUKernel &kernel, const std::string &filename, const vec4f &center, const float scale);
    void saveToFile(GPUKernel &kernel, const std::string &filename);
};
}

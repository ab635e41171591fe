/* FileMarshaller.cpp - see FileMarshaller.h */
#include "FileMarshaller.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <vector>

namespace solr
{
namespace
{
const uint64_t FORMAT_VERSION = 2; /* FileMarshaller.cpp:47: the CUDA engine's structs */

#pragma pack(push, 1)
struct IrtPrimitive /* the reference's CPUPrimitive as its 64-bit builds lay it out */
{
    uint8_t belongsToModel, movable, pad0[2];
    float p0[3], p1[3], p2[3], n0[3], n1[3], n2[3], size[3];
    int32_t type, materialId;
    float vt0[2], vt1[2], vt2[2];
    float speed0[3], speed1[3], speed2[3];
    uint8_t pad1[4];
};
struct IrtTexture /* TextureInfo, types.h:301-308 there */
{
    uint64_t buffer;
    int32_t offset;
    int32_t size[3];
    int32_t type;
    uint8_t pad[4];
};
#pragma pack(pop)
static_assert(sizeof(IrtPrimitive) == 160, "CPUPrimitive on disk");
static_assert(sizeof(IrtTexture) == 32, "TextureInfo on disk");
static_assert(sizeof(SceneInfo) == 112 && sizeof(Material) == 176, "SceneInfo / Material on disk");

vec3f v3(const float *f) { return make_vec3f(f[0], f[1], f[2]); }
vec2f v2(const float *f) { return make_vec2f(f[0], f[1]); }
void put3(float *f, const vec3f &v) { f[0] = v.x, f[1] = v.y, f[2] = v.z; }
void put2(float *f, const vec2f &v) { f[0] = v.x, f[1] = v.y; }

void shiftTextureIds(Material &m, int by)
{
    int *ids[] = {&m.textureIds.x,         &m.textureIds.y,         &m.textureIds.z,         &m.textureIds.w,
                  &m.advancedTextureIds.x, &m.advancedTextureIds.y, &m.advancedTextureIds.z, &m.advancedTextureIds.w};
    for (int *id : ids)
        if (*id != TEXTURE_NONE)
            *id += by;
}
}

vec4f FileMarshaller::loadFromFile(GPUKernel &kernel, const std::string &filename, const vec4f &center,
                                   const float scale)
{
    vec4f result = make_vec4f();
    const float vd = kernel.getSceneInfo().viewDistance;
    float mn[3] = {vd, vd, vd}, mx[3] = {-vd, -vd, -vd};

    std::ifstream file(filename.c_str(), std::ifstream::binary);
    if (file.is_open())
    {
        uint64_t version = 0;
        file.read((char *)&version, sizeof(version));
        if (!file || version != FORMAT_VERSION)
        {
            std::cerr << "FileMarshaller: " << filename << " is not compatible with this engine (version " << version
                      << ", expected " << FORMAT_VERSION << ")" << std::endl;
            return result;
        }
        SceneInfo ignored;
        file.read((char *)&ignored, sizeof(ignored));

        uint64_t nbPrimitives = 0;
        file.read((char *)&nbPrimitives, sizeof(nbPrimitives));
        for (uint64_t i = 0; i < nbPrimitives && file; ++i)
        {
            IrtPrimitive p;
            file.read((char *)&p, sizeof(p));
            if (!file)
            {
                std::cerr << "FileMarshaller: " << filename << " ends inside primitive " << i << std::endl;
                break;
            }
            const int n = kernel.addPrimitive(static_cast<PrimitiveType>(p.type));
            kernel.setPrimitive(n, center.x + p.p0[0], center.y + p.p0[1], center.z + p.p0[2], center.x + p.p1[0],
                                center.y + p.p1[1], center.z + p.p1[2], center.x + p.p2[0], center.y + p.p2[1],
                                center.z + p.p2[2], p.size[0], p.size[1], p.size[2], p.materialId);
            kernel.setPrimitiveBellongsToModel(n, true);
            kernel.setPrimitiveIsMovable(n, false);
            kernel.setPrimitiveNormals(n, v3(p.n0), v3(p.n1), v3(p.n2));
            kernel.setPrimitiveTextureCoordinates(n, v2(p.vt0), v2(p.vt1), v2(p.vt2));
            /* the extent is taken over all three points whatever the type, before the offset */
            for (int k = 0; k < 3; ++k)
            {
                const float lo = std::min(std::min(p.p0[k], p.p1[k]), p.p2[k]);
                const float hi = std::max(std::max(p.p0[k], p.p1[k]), p.p2[k]);
                mn[k] = (lo < mn[k]) ? lo : mn[k];
                mx[k] = (hi > mx[k]) ? hi : mx[k];
            }
        }

        uint64_t nbTextures = 0;
        file.read((char *)&nbTextures, sizeof(nbTextures));
        const int firstSlot = (int)kernel.getNbActiveTextures();
        for (uint64_t i = 0; i < nbTextures && file; ++i)
        {
            uint64_t id = 0;
            IrtTexture t;
            file.read((char *)&id, sizeof(id));
            file.read((char *)&t, sizeof(t));
            if (!file || t.size[0] < 0 || t.size[1] < 0 || t.size[2] < 0)
                break;
            const size_t bytes = (size_t)t.size[0] * t.size[1] * t.size[2];
            std::vector<BitmapBuffer> pixels(bytes);
            file.read((char *)pixels.data(), (std::streamsize)bytes);
            if (!file)
            {
                std::cerr << "FileMarshaller: " << filename << " ends inside texture " << i << std::endl;
                break;
            }
            TextureInfo info;
            memset(&info, 0, sizeof(info));
            info.buffer = pixels.data(); /* setTexture copies */
            info.offset = t.offset;
            info.size.x = t.size[0];
            info.size.y = t.size[1];
            info.size.z = t.size[2];
            info.type = t.type;
            kernel.setTexture(firstSlot + (int)i, info);
        }

        uint64_t nbMaterials = 0;
        file.read((char *)&nbMaterials, sizeof(nbMaterials));
        for (uint64_t i = 0; i < nbMaterials && file; ++i)
        {
            uint64_t id = 0;
            Material material;
            file.read((char *)&id, sizeof(id));
            file.read((char *)&material, sizeof(material));
            if (!file)
            {
                std::cerr << "FileMarshaller: " << filename << " ends inside material " << i << std::endl;
                break;
            }
            shiftTextureIds(material, firstSlot);
            kernel.setMaterial(static_cast<unsigned int>(id), material);
        }
    }
    /* FileMarshaller.cpp:180-189: also when the file could not be opened (the extent is then 2 x
     * viewDistance) - every primitive the kernel holds is rescaled */
    result.x = fabsf(mx[0] - mn[0]);
    result.y = fabsf(mx[1] - mn[1]);
    result.z = fabsf(mx[2] - mn[2]);
    const float ratio = scale / result.y;
    kernel.scalePrimitives(ratio, 0, NB_MAX_BOXES);
    return result;
}

void FileMarshaller::saveToFile(GPUKernel &kernel, const std::string &filename)
{
    std::ofstream file(filename.c_str(), std::ofstream::binary);
    if (!file.is_open())
    {
        std::cerr << "FileMarshaller: cannot write " << filename << std::endl;
        return;
    }
    const uint64_t version = FORMAT_VERSION;
    file.write((const char *)&version, sizeof(version));
    file.write((const char *)&kernel.getSceneInfo(), sizeof(SceneInfo));

    /* FileMarshaller.cpp:213-234: the COUNT is of the primitives that belong to the model, the records
     * written are the first `count` primitives by id */
    const uint64_t nbTotal = kernel.getNbActivePrimitives();
    uint64_t nbPrimitives = 0;
    for (uint64_t i = 0; i < nbTotal; ++i)
    {
        CPUPrimitive *p = kernel.getPrimitive((unsigned int)i);
        if (p && p->belongsToModel)
            ++nbPrimitives;
    }
    file.write((const char *)&nbPrimitives, sizeof(nbPrimitives));
    std::map<uint64_t, Material *> materials;
    for (uint64_t i = 0; i < nbPrimitives; ++i)
    {
        IrtPrimitive out;
        memset(&out, 0, sizeof(out));
        if (CPUPrimitive *p = kernel.getPrimitive((unsigned int)i))
        {
            out.belongsToModel = p->belongsToModel;
            out.movable = p->movable;
            put3(out.p0, p->p0), put3(out.p1, p->p1), put3(out.p2, p->p2);
            put3(out.n0, p->n0), put3(out.n1, p->n1), put3(out.n2, p->n2);
            put3(out.size, p->size);
            out.type = p->type;
            out.materialId = p->materialId;
            put2(out.vt0, p->vt0), put2(out.vt1, p->vt1), put2(out.vt2, p->vt2);
            put3(out.speed0, p->speed0), put3(out.speed1, p->speed1), put3(out.speed2, p->speed2);
            if (Material *m = kernel.getMaterial(p->materialId))
                materials[(uint64_t)p->materialId] = m;
        }
        file.write((const char *)&out, sizeof(out));
    }

    /* textures in use, renumbered 0.. in ascending id order */
    std::map<uint64_t, TextureInfo> textures;
    for (auto &entry : materials)
    {
        const Material &m = *entry.second;
        const int ids[] = {m.textureIds.x,         m.textureIds.y,         m.textureIds.z,         m.textureIds.w,
                           m.advancedTextureIds.x, m.advancedTextureIds.y, m.advancedTextureIds.z, m.advancedTextureIds.w};
        for (int id : ids)
            if (id != TEXTURE_NONE && id >= 0 && id < NB_MAX_TEXTURES)
                textures[(uint64_t)id] = kernel.getTextureInformation(id);
    }
    const uint64_t nbTextures = textures.size();
    file.write((const char *)&nbTextures, sizeof(nbTextures));
    std::map<int, int> renumbered;
    renumbered[TEXTURE_NONE] = TEXTURE_NONE;
    uint64_t index = 0;
    for (auto &entry : textures)
    {
        const TextureInfo &info = entry.second;
        IrtTexture t;
        memset(&t, 0, sizeof(t));
        t.size[0] = info.size.x, t.size[1] = info.size.y, t.size[2] = info.size.z;
        t.type = info.type;
        renumbered[(int)entry.first] = (int)index;
        file.write((const char *)&index, sizeof(index));
        file.write((const char *)&t, sizeof(t));
        if (info.buffer)
            file.write((const char *)info.buffer, (std::streamsize)((size_t)info.size.x * info.size.y * info.size.z));
        ++index;
    }

    /* FileMarshaller.cpp:291-303 renumbers the texture ids inside the kernel's own materials while
     * writing them; the file gets the same bytes from a copy, the kernel keeps rendering what it rendered */
    const uint64_t nbMaterials = materials.size();
    file.write((const char *)&nbMaterials, sizeof(nbMaterials));
    for (auto &entry : materials)
    {
        Material m = *entry.second;
        int *ids[] = {&m.textureIds.x,         &m.textureIds.y,         &m.textureIds.z,         &m.textureIds.w,
                      &m.advancedTextureIds.x, &m.advancedTextureIds.y, &m.advancedTextureIds.z, &m.advancedTextureIds.w};
        for (int *id : ids)
        {
            std::map<int, int>::const_iterator it = renumbered.find(*id);
            *id = (it == renumbered.end()) ? 0 : it->second; /* std::map::operator[] of an unknown id gives 0 */
        }
        file.write((const char *)&entry.first, sizeof(uint64_t));
        file.write((const char *)&m, sizeof(m));
    }
}
}

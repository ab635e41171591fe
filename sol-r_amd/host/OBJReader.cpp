/* OBJReader.cpp - see OBJReader.h */
#include "OBJReader.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

namespace solr
{
namespace
{

/* the blank-separated words of a line */
std::vector<std::string> words(const std::string &line)
{
    std::vector<std::string> out;
    std::string word;
    for (char c : line)
    {
        if (c == ' ')
        {
            if (!word.empty())
                out.push_back(word);
            word.clear();
        }
        else
            word += c;
    }
    if (!word.empty())
        out.push_back(word);
    return out;
}

float number(const std::vector<std::string> &w, size_t i)
{
    return i < w.size() ? static_cast<float>(atof(w[i].c_str())) : 0.f;
}

/* up to three numbers after a keyword (Kd, Ks) */
vec4f triple(const std::string &text)
{
    const std::vector<std::string> w = words(text);
    return make_vec4f(number(w, 0), number(w, 1), number(w, 2));
}

/* "v/t/n": vertex, texture coordinate and normal number; what is absent reads 0 */
vec4i corner(const std::string &text)
{
    int part[3] = {0, 0, 0};
    size_t from = 0;
    for (int k = 0; k < 3 && from <= text.size(); ++k)
    {
        const size_t slash = text.find('/', from);
        const std::string field = text.substr(from, slash == std::string::npos ? std::string::npos : slash - from);
        if (!field.empty())
            part[k] = atoi(field.c_str());
        if (slash == std::string::npos)
            break;
        from = slash + 1;
    }
    return make_vec4i(part[0], part[1], part[2]);
}

std::string readLine(std::ifstream &file)
{
    std::string line;
    std::getline(file, line);
    line.erase(std::remove(line.begin(), line.end(), '\r'), line.end());
    return line;
}

void commit(GPUKernel &kernel, const MaterialMTL &m)
{
    const float innerDiffusion = 1000.f, diffusionRatio = 4.f; /* OBJReader.cpp:164-165 */
    kernel.setMaterial(m.index, m.Kd.x, m.Kd.y, m.Kd.z, m.noise, m.reflection, m.refraction, false, false, 0,
                       m.transparency, m.opacity, m.diffuseTextureId, m.normalTextureId, m.bumpTextureId,
                       m.specularTextureId, m.reflectionTextureId, m.transparencyTextureId, m.ambientOcclusionTextureId,
                       m.Ks.x, 100.f * m.Ks.y, m.Ks.z, m.illumination, innerDiffusion, innerDiffusion * diffusionRatio,
                       false);
}

template <class T>
const T &lookup(std::map<int, T> &table, int key)
{
    return table[key]; /* an undefined number reads as zeros, and stays defined */
}
}

unsigned int OBJReader::loadMaterialsFromFile(const std::string &filename, std::map<std::string, MaterialMTL> &materials,
                                              GPUKernel &kernel, int materialId)
{
    std::ifstream file(filename.c_str());
    if (!file.is_open())
        return 0;
    std::string id;
    while (file.good())
    {
        std::string line = readLine(file);
        while (!line.empty() && line[0] < 32) /* leading tabs; blanks stay */
            line = line.substr(1);

        if (line.find("newmtl") == 0)
        {
            if (!id.empty())
                commit(kernel, materials[id]);
            id = line.length() > 7 ? line.substr(7) : std::string();
            MaterialMTL m;
            memset(&m, 0, sizeof(m));
            m.index = static_cast<unsigned int>(materials.size() + materialId);
            m.diffuseTextureId = m.normalTextureId = m.bumpTextureId = m.specularTextureId = MATERIAL_NONE;
            m.reflectionTextureId = m.transparencyTextureId = m.ambientOcclusionTextureId = MATERIAL_NONE;
            m.Ks.x = 1.f;
            m.Ks.y = 500.f;
            materials[id] = m;
        }
        if (line.find("Kd") == 0 && line.length() > 3)
        {
            MaterialMTL &m = materials[id];
            m.Kd = triple(line.substr(3));
            if (m.isSketchupLightMaterial)
                m.illumination = (m.Kd.x + m.Kd.y + m.Kd.z) / 3.f;
        }
        if (line.find("Ks") == 0 && line.length() > 3)
            materials[id].Ks = triple(line.substr(3));
        if (line.find("map_Kd") == 0 || line.find("map_bump") == 0 || line.find("map_norm") == 0 ||
            line.find("map_spec") == 0)
            std::cerr << "OBJReader: " << filename << ": texture map not loaded (no image codecs in this engine): "
                      << line << std::endl;
        if (line.find("Tr") == 0)
        {
            const float d = static_cast<float>(atof(line.substr(2).c_str()));
            MaterialMTL &m = materials[id];
            m.reflection = 1.f;
            m.transparency = 0.5f + (d / 50.f);
            m.refraction = 1.1f;
            m.noise = 0.f;
        }
        if (line.find("SoL_R_Light") != std::string::npos)
            materials[id].isSketchupLightMaterial = true;
        if (line.find("illum") == 0 && line.length() > 6)
        {
            MaterialMTL &m = materials[id];
            switch (atoi(line.substr(6).c_str()))
            {
            case 3:
            case 5:
            case 8:
                m.transparency = 0.f; /* reflection only */
                break;
            case 6:
            case 7:
            case 9:
                break; /* transparency */
            default:
                m.reflection = 0.f;
                m.transparency = 0.f;
                m.refraction = 0.f;
            }
        }
    }
    if (!id.empty())
        commit(kernel, materials[id]);
    return 0;
}

void OBJReader::addLightComponent(GPUKernel &kernel, std::vector<vec4f> &faceCenters, const vec4f &center,
                                  const vec4f &objectCenter, const vec4f &objectScale, const int material,
                                  CPUBoundingBox &aabb)
{
    if (!faceCenters.empty())
    {
        /* the caller's bounds are overwritten with the component's, as in the reference */
        aabb.parameters[0] = make_vec3f(1000000.f, 1000000.f, 1000000.f);
        aabb.parameters[1] = make_vec3f(-1000000.f, -1000000.f, -1000000.f);
        for (const vec4f &c : faceCenters)
        {
            aabb.parameters[0].x = std::min(c.x, aabb.parameters[0].x);
            aabb.parameters[1].x = std::max(c.x, aabb.parameters[1].x);
            aabb.parameters[0].y = std::min(c.y, aabb.parameters[0].y);
            aabb.parameters[1].y = std::max(c.y, aabb.parameters[1].y);
            aabb.parameters[0].z = std::min(c.z, aabb.parameters[0].z);
            aabb.parameters[1].z = std::max(c.z, aabb.parameters[1].z);
        }
        const float lx = (aabb.parameters[1].x + aabb.parameters[0].x) / 2.f;
        const float ly = (aabb.parameters[1].y + aabb.parameters[0].y) / 2.f;
        const float lz = (aabb.parameters[1].z + aabb.parameters[0].z) / 2.f;
        const float dx = aabb.parameters[1].x - aabb.parameters[0].x, dy = aabb.parameters[1].y - aabb.parameters[0].y,
                    dz = aabb.parameters[1].z - aabb.parameters[0].z;
        const float radius = sqrtf(dx * dx + dy * dy + dz * dz) / 2.f;
        const int n = kernel.addPrimitive(ptSphere);
        kernel.setPrimitive(n, center.x + objectScale.x * (-objectCenter.x + lx),
                            center.y + objectScale.y * (-objectCenter.y + ly),
                            center.z + objectScale.z * (-objectCenter.z + lz), radius, 0.f, 0.f, material);
        kernel.setPrimitiveBellongsToModel(n, true);
    }
    faceCenters.clear();
}

vec4f OBJReader::loadModelFromFile(const std::string &filename, GPUKernel &kernel, const vec4f &objectPosition,
                                   const bool autoScale, const vec4f &scale, bool loadMaterials, int materialId,
                                   bool allSpheres, bool autoCenter, CPUBoundingBox &aabb, const bool &checkInAABB,
                                   const CPUBoundingBox &inAABB)
{
    std::map<int, vec3f> vertices, normals;
    std::map<int, vec2f> textureCoordinates;
    std::map<std::string, MaterialMTL> materials;
    int nbVertices = 1, nbNormals = 1, nbTextureCoordinates = 1;

    std::string stem(filename);
    const size_t ext = stem.find(".obj");
    if (ext != std::string::npos)
        stem = filename.substr(0, ext);
    std::replace(stem.begin(), stem.end(), '\\', '/');
    const std::string modelFilename = stem + ".obj";

    vec4f objectSize = make_vec4f();
    aabb.parameters[0] = make_vec3f(100000.f, 100000.f, 100000.f);
    aabb.parameters[1] = make_vec3f(-100000.f, -100000.f, -100000.f);

    /* pass 1: points */
    std::ifstream file(modelFilename.c_str());
    if (file.is_open())
    {
        while (file.good())
        {
            const std::string line = readLine(file);
            if (line.length() <= 1)
                continue;
            if (loadMaterials && line.find("mtllib") != std::string::npos && line.length() > 7)
            {
                const std::string folder = stem.substr(0, stem.rfind('/'));
                loadMaterialsFromFile(folder + '/' + line.substr(7), materials, kernel, materialId);
            }
            if (line[0] != 'v')
                continue;
            const std::vector<std::string> w = words(line);
            vec3f v = make_vec3f(number(w, 1), number(w, 2), number(w, 3));
            if (line[1] == 'n')
            {
                v.z = -v.z;
                normals[nbNormals++] = v;
            }
            else if (line[1] == 't')
            {
                vec2f t = make_vec2f(v.x, v.y);
                if (t.x < 0.f)
                    t.x = fabsf(t.x) - static_cast<int>(fabsf(t.x));
                if (t.y < 0.f)
                    t.y = fabsf(t.y) - static_cast<int>(fabsf(t.y));
                textureCoordinates[nbTextureCoordinates++] = t;
            }
            else if (line[1] == ' ')
            {
                v.z = -v.z;
                vertices[nbVertices++] = v;
                aabb.parameters[0].x = (v.x < aabb.parameters[0].x) ? v.x : aabb.parameters[0].x;
                aabb.parameters[0].y = (v.y < aabb.parameters[0].y) ? v.y : aabb.parameters[0].y;
                aabb.parameters[0].z = (v.z < aabb.parameters[0].z) ? v.z : aabb.parameters[0].z;
                aabb.parameters[1].x = (v.x > aabb.parameters[1].x) ? v.x : aabb.parameters[1].x;
                aabb.parameters[1].y = (v.y > aabb.parameters[1].y) ? v.y : aabb.parameters[1].y;
                aabb.parameters[1].z = (v.z > aabb.parameters[1].z) ? v.z : aabb.parameters[1].z;
            }
        }
        file.close();
    }

    if (checkInAABB)
        for (int k = 0; k < 3; ++k)
            if ((&aabb.parameters[0].x)[k] < (&inAABB.parameters[0].x)[k] ||
                (&aabb.parameters[1].x)[k] > (&inAABB.parameters[1].x)[k])
                return objectSize;

    vec4f objectCenter = objectPosition;
    vec4f objectScale = scale;
    if (autoScale)
    {
        const float os = std::max(aabb.parameters[1].x - aabb.parameters[0].x,
                                  std::max(aabb.parameters[1].y - aabb.parameters[0].y,
                                           aabb.parameters[1].z - aabb.parameters[0].z));
        objectScale.x = scale.x / os;
        objectScale.y = scale.y / os;
        objectScale.z = scale.z / os;
        if (autoCenter)
        {
            objectCenter.x = (aabb.parameters[0].x + aabb.parameters[1].x) / 2.f;
            objectCenter.y = (aabb.parameters[0].y + aabb.parameters[1].y) / 2.f;
            objectCenter.z = (aabb.parameters[0].z + aabb.parameters[1].z) / 2.f;
        }
    }
    auto place = [&](float x, float y, float z) {
        return make_vec3f(objectPosition.x + objectScale.x * (-objectCenter.x + x),
                          objectPosition.y + objectScale.y * (-objectCenter.y + y),
                          objectPosition.z + objectScale.z * (-objectCenter.z + z));
    };

    /* pass 2: faces */
    file.open(modelFilename.c_str());
    if (file.is_open())
    {
        int material = materialId;
        int lightMaterial = MATERIAL_NONE;
        bool lightComponent = false;
        std::vector<vec4f> faceCenters;
        std::string component;
        while (file.good())
        {
            const std::string line = readLine(file);
            if (line.empty())
                continue;
            if (line.find("g") == 0)
            {
                lightComponent = (line.find("SoL_R") != std::string::npos);
                if (lightComponent)
                {
                    if (line != component)
                        addLightComponent(kernel, faceCenters, objectPosition, objectCenter, objectScale, lightMaterial,
                                          aabb);
                    component = line;
                }
            }
            if (line.find("usemtl") == 0 && line.length() > 7)
            {
                std::map<std::string, MaterialMTL>::const_iterator it = materials.find(line.substr(7));
                if (it != materials.end())
                {
                    material = (int)it->second.index;
                    if (lightComponent)
                        lightMaterial = material;
                }
                else
                    std::cerr << "OBJReader: unknown material " << line.substr(7) << std::endl;
            }
            if (line[0] != 'f')
                continue;
            std::vector<vec4i> face;
            {
                const std::vector<std::string> w = words(line.substr(1));
                for (const std::string &c : w)
                    face.push_back(corner(c));
            }
            if (face.size() < 3)
                continue; /* the reference reads past the end of its vector here */

            int n = 0; /* the primitive the texture coordinates and normals below go to */
            const vec3f a = lookup(vertices, face[0].x), b = lookup(vertices, face[1].x), c = lookup(vertices, face[2].x);
            if (allSpheres || lightComponent)
            {
                vec4f middle = make_vec4f((a.x + b.x + c.x) / 3.f, (a.y + b.y + c.y) / 3.f, (a.z + b.z + c.z) / 3.f);
                if (lightComponent)
                    faceCenters.push_back(middle);
                else
                {
                    const float sx = std::max(middle.x - a.x, std::max(middle.x - b.x, middle.x - c.x));
                    const float sy = std::max(middle.y - a.y, std::max(middle.y - b.y, middle.y - c.y));
                    const float sz = std::max(middle.z - a.z, std::max(middle.z - b.z, middle.z - c.z));
                    const vec3f p = place(middle.x, middle.y, middle.z);
                    n = kernel.addPrimitive(ptEllipsoid);
                    kernel.setPrimitive(n, p.x, p.y, p.z, objectScale.x * sx, objectScale.y * sy, objectScale.z * sz,
                                        material);
                    kernel.setPrimitiveBellongsToModel(n, true);
                }
            }
            else
            {
                const vec3f p0 = place(a.x, a.y, a.z), p1 = place(b.x, b.y, b.z), p2 = place(c.x, c.y, c.z);
                n = kernel.addPrimitive(ptTriangle);
                kernel.setPrimitive(n, p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, 0.f, 0.f, 0.f, material);
                kernel.setPrimitiveBellongsToModel(n, true);
            }
            kernel.setPrimitiveTextureCoordinates(n, lookup(textureCoordinates, face[0].y),
                                                  lookup(textureCoordinates, face[1].y),
                                                  lookup(textureCoordinates, face[2].y));
            kernel.setPrimitiveNormals(n, lookup(normals, face[0].z), lookup(normals, face[1].z),
                                       lookup(normals, face[2].z));

            if (face.size() == 4)
            {
                const vec3f d = lookup(vertices, face[3].x);
                if (allSpheres)
                {
                    const vec3f p = place((d.x + c.x + a.x) / 3.f, (d.y + c.y + a.y) / 3.f, (d.z + c.z + a.z) / 3.f);
                    n = kernel.addPrimitive(ptSphere);
                    kernel.setPrimitive(n, p.x, p.y, p.z, 100.f, 0.f, 0.f, material); /* OBJReader.cpp:729 */
                    kernel.setPrimitiveBellongsToModel(n, true);
                }
                else
                {
                    const vec3f p0 = place(d.x, d.y, d.z), p1 = place(c.x, c.y, c.z), p2 = place(a.x, a.y, a.z);
                    n = kernel.addPrimitive(ptTriangle);
                    kernel.setPrimitive(n, p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, 0.f, 0.f, 0.f,
                                        material);
                    kernel.setPrimitiveBellongsToModel(n, true);
                }
                kernel.setPrimitiveTextureCoordinates(n, lookup(textureCoordinates, face[3].y),
                                                      lookup(textureCoordinates, face[2].y),
                                                      lookup(textureCoordinates, face[0].y));
                kernel.setPrimitiveNormals(n, lookup(normals, face[3].z), lookup(normals, face[2].z),
                                           lookup(normals, face[0].z));
            }
        }
        file.close();
        if (!faceCenters.empty())
            addLightComponent(kernel, faceCenters, objectPosition, objectCenter, objectScale, lightMaterial, aabb);
    }

    objectSize.x = objectScale.x * (aabb.parameters[1].x - aabb.parameters[0].x);
    objectSize.y = objectScale.y * (aabb.parameters[1].y - aabb.parameters[0].y);
    objectSize.z = objectScale.z * (aabb.parameters[1].z - aabb.parameters[0].z);
    return objectSize;
}
}

/* SWCReader.cpp - see SWCReader.h */
#include "SWCReader.h"

#include <algorithm>
#include <cstdlib>
#include <fstream>

namespace solr
{
CPUBoundingBox SWCReader::loadMorphologyFromFile(const std::string &filename, GPUKernel &kernel, const vec4f &position,
                                                 const vec4f &scale, const int materialId)
{
    CPUBoundingBox bounds; /* the reference starts from an uninitialised box; zeros here */
    bounds.parameters[0] = bounds.parameters[1] = bounds.center = make_vec3f();
    bounds.indexForNextBox = 0;

    std::ifstream file(filename.c_str());
    if (file.is_open())
    {
        while (file.good())
        {
            std::string dropped, word[7];
            std::getline(file, dropped);
            for (std::string &w : word)
                file >> w;
            if (word[0] == "#")
                continue;
            Morphology m;
            const int id = atoi(word[0].c_str());
            m.branch = atoi(word[1].c_str());
            m.x = static_cast<float>(scale.x * (position.x + atof(word[2].c_str())));
            m.y = static_cast<float>(scale.y * (position.y + atof(word[3].c_str())));
            m.z = static_cast<float>(scale.z * (position.z + atof(word[4].c_str())));
            m.radius = static_cast<float>(scale.w * atof(word[5].c_str()));
            m.parent = atoi(word[6].c_str());
            m_morphologies[id] = m;
            bounds.parameters[0].x = std::min(bounds.parameters[0].x, m.x);
            bounds.parameters[0].y = std::min(bounds.parameters[0].y, m.y);
            bounds.parameters[0].z = std::min(bounds.parameters[0].z, m.z);
            bounds.parameters[1].x = std::max(bounds.parameters[1].x, m.x);
            bounds.parameters[1].y = std::max(bounds.parameters[1].y, m.y);
            bounds.parameters[1].z = std::max(bounds.parameters[1].z, m.z);
        }
        file.close();
    }

    const vec2f zero = make_vec2f(0.f, 0.f);
    for (Morphologies::iterator it = m_morphologies.begin(); it != m_morphologies.end(); ++it)
    {
        Morphology &a = it->second;
        if (a.parent == -1)
        {
            a.primitiveId = kernel.addPrimitive(ptSphere, true);
            kernel.setPrimitive(a.primitiveId, a.x, a.y, a.z, a.radius * 1.5f, 0.f, 0.f, materialId);
            kernel.setPrimitiveTextureCoordinates(a.primitiveId, zero, make_vec2f(2.f, 2.f), zero);
            continue;
        }
        Morphology &b = m_morphologies[a.parent]; /* an unknown parent comes into being, all zeros */
        if (b.parent == -1)
            continue;
        const vec2f one = make_vec2f(1.f, 1.f);
        b.primitiveId = kernel.addPrimitive(ptCylinder, true);
        kernel.setPrimitive(b.primitiveId, a.x, a.y, a.z, b.x, b.y, b.z, a.radius, 0.f, 0.f, materialId);
        kernel.setPrimitiveTextureCoordinates(b.primitiveId, zero, one, zero);
        const int p = kernel.addPrimitive(ptSphere, true);
        kernel.setPrimitive(p, b.x, b.y, b.z, b.radius, 0.f, 0.f, materialId);
        kernel.setPrimitiveTextureCoordinates(p, zero, one, zero);
    }
    return bounds;
}
}

/* PDBReader.cpp - see PDBReader.h */
#include "PDBReader.h"

#include <dlfcn.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <vector>

namespace solr
{
namespace
{
const size_t NB_ELEMENTS = 119;             /* PDBReader.cpp:57: both tables are that long, zero-filled at the end */
const float DEFAULT_ATOM_DISTANCE = 30.f;   /* :250 */
const float DEFAULT_STICK_DISTANCE = 1.7f;  /* :251 */

struct Colour
{
    std::string symbol;
    int r = 0, g = 0, b = 0;
};
struct Radius
{
    std::string name;
    float pm = 0.f;
};
struct Tables
{
    std::vector<Colour> colours;
    std::vector<Radius> radii;
    bool loaded = false;
};

std::string tablePath()
{
    if (const char *e = getenv("SOLR_PDB_ELEMENTS"))
        return e;
    Dl_info info;
    if (dladdr((const void *)&tablePath, &info) && info.dli_fname)
    {
        std::string lib(info.dli_fname);
        const size_t slash = lib.rfind('/');
        return (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/pdb_elements.txt";
    }
    return "pdb_elements.txt";
}

const Tables &tables()
{
    static Tables t;
    if (t.loaded)
        return t;
    std::ifstream file(tablePath().c_str());
    std::string line;
    while (std::getline(file, line))
    {
        std::istringstream words(line);
        std::string kind;
        words >> kind;
        if (kind == "colour")
        {
            Colour c;
            words >> c.symbol >> c.r >> c.g >> c.b;
            /* the reference upper-cases its symbols before comparing (PDBReader.cpp:394) */
            std::transform(c.symbol.begin(), c.symbol.end(), c.symbol.begin(), ::toupper);
            t.colours.push_back(c);
        }
        else if (kind == "radius")
        {
            Radius r;
            words >> r.name >> r.pm;
            t.radii.push_back(r);
        }
    }
    t.loaded = !t.colours.empty() && !t.radii.empty();
    if (t.loaded)
    {
        t.colours.resize(NB_ELEMENTS);
        t.radii.resize(NB_ELEMENTS);
    }
    return t;
}

struct Atom
{
    int id = 0, index = 0;
    float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
    int materialId = 0, chainId = 0, residue = 0;
    bool isBackbone = false;
};

/* the non-blank characters of line[from, to], provided the scanner reaches position `closing` */
bool field(const std::string &line, size_t from, size_t to, size_t closing, std::string &out)
{
    out.clear();
    if (closing >= line.length())
        return false;
    for (size_t i = from; i <= to; ++i)
        if (line[i] != ' ')
            out += line[i];
    return true;
}
}

vec4f PDBReader::loadAtomsFromFile(const std::string &filename, GPUKernel &kernel, GeometryType geometryType,
                                   const float defaultAtomSize, const float defaultStickSize, const int materialType,
                                   const vec4f scale, const bool useModels)
{
    const Tables &t = tables();
    if (!t.loaded)
    {
        std::cerr << "PDBReader: element tables not found at " << tablePath() << std::endl;
        return make_vec4f(0.f, 0.f, 0.f, -1.f);
    }
    for (size_t i = 0; i < NB_ELEMENTS; ++i)
        kernel.setMaterial((unsigned int)i, static_cast<float>(t.colours[i].r) / 255.f,
                           static_cast<float>(t.colours[i].g) / 255.f, static_cast<float>(t.colours[i].b) / 255.f, 0.f,
                           0.f, 0.f, false, false, 0, 0.f, 0.f, TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE,
                           TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE, 1.f, 100.f, 0.f, 0.f,
                           kernel.getSceneInfo().viewDistance, 0.f, false);
    kernel.resetBoxes(true);

    const float distanceRatio = 2.f;
    std::map<int, Atom> atoms;
    float mn[3] = {100000.f, 100000.f, 100000.f}, mx[3] = {-100000.f, -100000.f, -100000.f};
    int index = 0;
    std::ifstream file(filename.c_str());
    if (file.is_open())
    {
        while (file.good())
        {
            std::string line, value, atomName, atomCode;
            std::getline(file, line);
            if (line.find("ATOM") != 0)
                continue;
            Atom atom;
            atom.index = index++;
            if (field(line, 7, 10, 11, value))
                atom.id = atoi(value.c_str());
            field(line, 13, 16, 17, atomCode);
            if (line.length() > 21)
                atom.chainId = (int)line[21] - 64;
            if (field(line, 23, 25, 26, value))
                atom.residue = atoi(value.c_str());
            if (field(line, 31, 36, 37, value))
                atom.x = static_cast<float>(atof(value.c_str()));
            if (field(line, 39, 44, 45, value))
                atom.y = static_cast<float>(atof(value.c_str()));
            if (field(line, 47, 52, 53, value))
                atom.z = -static_cast<float>(atof(value.c_str()));
            field(line, 77, 78, 79, atomName);

            atom.isBackbone = (geometryType == gtBackbone || geometryType == gtIsoSurface || atomCode.length() == 1);
            bool found = false;
            for (size_t i = 0; !found && i < NB_ELEMENTS; ++i)
                if (atomName == t.colours[i].symbol)
                {
                    found = true;
                    switch (materialType)
                    {
                    case 1:
                        atom.materialId = (atom.chainId % 2 == 0) ? static_cast<int>(i) : 1000;
                        break;
                    case 2:
                        atom.materialId = atom.residue % 10;
                        break;
                    default:
                        atom.materialId = static_cast<int>(i);
                        break;
                    }
                    atom.w = (geometryType == gtFixedSizeAtoms) ? defaultAtomSize : 0.5f * defaultAtomSize;
                }
            if (!found)
                std::cerr << "PDBReader: no colour for element '" << atomName << "' (atom " << atomCode << ")" << std::endl;
            if (geometryType == gtFixedSizeAtoms)
                atom.w = defaultAtomSize;
            else
            {
                found = false;
                for (size_t i = 0; !found && i < NB_ELEMENTS; ++i)
                    if (atomName == t.radii[i].name)
                    {
                        atom.w = t.radii[i].pm;
                        found = true;
                    }
                if (!found)
                    std::cerr << "PDBReader: no radius for element '" << atomName << "'" << std::endl;
            }
            if (geometryType != gtBackbone || atom.isBackbone)
            {
                const float p[3] = {atom.x, atom.y, atom.z};
                for (int k = 0; k < 3; ++k)
                {
                    mn[k] = (p[k] < mn[k]) ? p[k] : mn[k];
                    mx[k] = (p[k] > mx[k]) ? p[k] : mx[k];
                }
                /* keyed by serial number for the stick types, else by position in the file (one-based:
                 * the counter has moved on) */
                if (geometryType == gtSticks || (geometryType == gtAtomsAndSticks && atom.residue % 2 == 0))
                    atoms[atom.id] = atom;
                else
                    atoms[index] = atom;
            }
        }
        file.close();
    }

    vec4f objectSize = make_vec4f(mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]);
    const float cx = (mn[0] + mx[0]) / 2.f, cy = (mn[1] + mx[1]) / 2.f, cz = (mn[2] + mx[2]) / 2.f;
    const float sx = scale.x / (mx[0] - mn[0]), sy = scale.y / (mx[1] - mn[1]), sz = scale.z / (mx[2] - mn[2]);
    const float atomDistance = DEFAULT_ATOM_DISTANCE;
    const vec2f vt0 = make_vec2f(0.f, 0.f), vt1 = make_vec2f(1.f, 1.f), vt2 = make_vec2f(0.f, 0.f);
    auto place = [&](float x, float y, float z) {
        return make_vec3f(sx * distanceRatio * atomDistance * (x - cx), sy * distanceRatio * atomDistance * (y - cy),
                          sz * distanceRatio * atomDistance * (z - cz));
    };

    for (std::map<int, Atom>::iterator it = atoms.begin(); it != atoms.end(); ++it)
    {
        const Atom &atom = it->second;
        float radius = atom.w, stickRadius = atom.w;
        switch (geometryType)
        {
        case gtFixedSizeAtoms:
            radius = defaultAtomSize;
            break;
        case gtSticks:
        case gtBackbone:
            radius = defaultStickSize;
            stickRadius = defaultStickSize;
            break;
        case gtAtomsAndSticks:
            radius = atom.w / 2.f;
            stickRadius = defaultStickSize / 2.f;
            break;
        default:
            break;
        }
        if (geometryType == gtSticks || geometryType == gtAtomsAndSticks || geometryType == gtBackbone)
            for (std::map<int, Atom>::iterator it2 = atoms.begin(); it2 != atoms.end(); ++it2)
            {
                const Atom &other = it2->second;
                if (it2 == it || atom.isBackbone != other.isBackbone)
                    continue;
                const float ax = atom.x - other.x, ay = atom.y - other.y, az = atom.z - other.z;
                const float distance = sqrtf(ax * ax + ay * ay + az * az);
                const float reach = (geometryType == gtBackbone && other.isBackbone) ? DEFAULT_STICK_DISTANCE * 2.f
                                                                                     : DEFAULT_STICK_DISTANCE;
                if (distance < reach)
                {
                    const vec3f a = place(atom.x, atom.y, atom.z);
                    const vec3f h = place((atom.x + other.x) / 2.f, (atom.y + other.y) / 2.f, (atom.z + other.z) / 2.f);
                    const int nb = kernel.addPrimitive(ptCylinder, true);
                    kernel.setPrimitive(nb, a.x, a.y, a.z, h.x, h.y, h.z, sx * stickRadius, 0.f, 0.f,
                                        (geometryType == gtSticks) ? atom.materialId : 1010);
                    kernel.setPrimitiveTextureCoordinates(nb, vt0, vt1, vt2);
                    ++m_nbPrimitives;
                }
            }
        int m = atom.materialId;
        if (!useModels)
        {
            radius = stickRadius; /* PDBReader.cpp:664: the atom takes the stick's radius */
            if (geometryType == gtAtomsAndSticks)
                m = 11;
        }
        const vec3f p = place(atom.x, atom.y, atom.z);
        if (geometryType == gtIsoSurface && atom.isBackbone && atom.chainId % 2 == 0)
        {
            const int nb = kernel.addPrimitive(ptSphere, true);
            kernel.setPrimitive(nb, p.x, p.y, p.z, sx * radius * 2.f, 0.f, 0.f, 10);
            kernel.setPrimitiveTextureCoordinates(nb, vt0, vt1, vt2);
            ++m_nbPrimitives;
        }
        const int nb = kernel.addPrimitive(ptSphere, true);
        kernel.setPrimitive(nb, p.x, p.y, p.z, sx * radius, 0.f, 0.f, m);
        kernel.setPrimitiveTextureCoordinates(nb, vt0, vt1, vt2);
        ++m_nbPrimitives;
    }
    objectSize.x *= sx * distanceRatio * atomDistance;
    objectSize.y *= sy * distanceRatio * atomDistance;
    objectSize.z *= sz * distanceRatio * atomDistance;
    return objectSize;
}
}

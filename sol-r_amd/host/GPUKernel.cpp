/*
 * GPUKernel.cpp - host scene store and box-tree builder.
 *
 * Behavioural restatement of the parts of the reference's
 * solr/engines/GPUKernel.cpp that produce the inputs of the rendering path
 * (SURVEY.md section 8 row a13).  Each method cites the reference lines whose
 * observable behaviour it reproduces; storage and structure are this
 * project's own.
 */
#include "GPUKernel.h"

#include <chrono>
#include <cstdio>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <iostream>

namespace
{
/* reference: GPUKernel.cpp:69 */
const unsigned int AABB_MAGIC_NUMBER = 6400;

inline vec3f min2(const vec3f &a, const vec3f &b)
{
    return solr::make_vec3f(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z));
}
inline vec3f max2(const vec3f &a, const vec3f &b)
{
    return solr::make_vec3f(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z));
}
inline vec3f min3(const vec3f &a, const vec3f &b, const vec3f &c) { return min2(min2(a, b), c); }
inline vec3f max3(const vec3f &a, const vec3f &b, const vec3f &c) { return max2(max2(a, b), c); }

/* signed 32-bit multiply/add with wrap-around: the reference computes grid
 * keys in int (GPUKernel.cpp:1014) and overflows for large grids; x86 wraps */
inline int wrapMulAdd3(int X, int Y, int Z, int n)
{
    unsigned int u = (unsigned int)X * (unsigned int)n * (unsigned int)n + (unsigned int)Y * (unsigned int)n +
                     (unsigned int)Z;
    return (int)u;
}
}

namespace solr
{
GPUKernel *SingletonKernel::m_kernel = nullptr;

float GPUKernel::vectorLength(const vec3f &v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }

/* reference: GPUKernel.cpp:142-151 (division by the length, not rsqrt) */
void GPUKernel::normalizeVector(vec3f &v)
{
    float l = vectorLength(v);
    if (l != 0.f)
    {
        v.x /= l;
        v.y /= l;
        v.z /= l;
    }
}

vec3f GPUKernel::crossProduct(const vec3f &b, const vec3f &c)
{
    vec3f a;
    a.x = b.y * c.z - b.z * c.y;
    a.y = b.z * c.x - b.x * c.z;
    a.z = b.x * c.y - b.y * c.x;
    return a;
}

GPUKernel::GPUKernel()
    : m_nbActiveMaterials(-1)
    , m_nbActiveTextures(0)
    , m_lightInformationSize(0)
    , m_maxPrimitivesPerBox(0)
    , m_doneWithAdding(false)
    , m_addingIndex(0)
    , m_frame(0)
    , m_nbFrames(0)
    , m_treeDepth(2) /* reference: GPUKernel.cpp:198 */
    , m_primitivesTransfered(false)
    , m_materialsTransfered(false)
    , m_texturesTransfered(false)
    , m_randomsTransfered(false)
    , m_refresh(true)
    , m_gpuDescription("host")
    , m_buffersInitialized(false)
    , m_deterministicSeed(-1)
{
    memset(&m_sceneInfo, 0, sizeof(m_sceneInfo));
    memset(&m_postProcessingInfo, 0, sizeof(m_postProcessingInfo));
    memset(m_hTextures, 0, sizeof(m_hTextures));
    m_viewPos = make_vec3f();
    m_viewDir = make_vec3f();
    m_angles = make_vec4f();
    m_occupancyParameters = make_vec2i(1, 1);
}

GPUKernel::~GPUKernel()
{
    for (int i = 0; i < NB_MAX_TEXTURES; ++i)
        delete[] m_hTextures[i].buffer;
}

/* reference: GPUKernel.cpp:319-370.  Buffers are sized to the frame instead
 * of MAX_BITMAP_SIZE, except the random buffer whose length the device layer
 * fixes at MAX_BITMAP_SIZE (h2d_randoms). */
void GPUKernel::initBuffers()
{
    m_lightInformation.assign(NB_MAX_LIGHTINFORMATIONS, LightInformation());
    m_hMaterials.assign(NB_MAX_MATERIALS + 1, Material());
    memset(m_hMaterials.data(), 0, m_hMaterials.size() * sizeof(Material));
    m_hBoundingBoxes.clear();
    m_hPrimitives.clear();
    m_hMovable.clear();
    m_hLamps.clear();
    for (int i = 0; i < NB_MAX_TEXTURES; ++i)
        delete[] m_hTextures[i].buffer;
    memset(m_hTextures, 0, sizeof(m_hTextures));
    m_hRandoms.assign(randomsNeeded(), 0.f);
    m_randomsFilled = false;
    size_t pixels = std::max<size_t>((size_t)m_sceneInfo.size.x * (size_t)m_sceneInfo.size.y, 1);
    m_hPrimitivesXYIds.assign(pixels, make_vec4i());
    m_bitmap.assign(pixels * SOLR_COLOR_DEPTH, 0);
    m_buffersInitialized = true;
}

/* reference: GPUKernel.cpp:372-471 */
void GPUKernel::cleanup()
{
    m_frames.clear();
    for (unsigned int f = 0; f < 1; ++f)
    {
        Frame &fr = m_frames[f];
        fr.minPos = make_vec3f(-m_sceneInfo.viewDistance, -m_sceneInfo.viewDistance, -m_sceneInfo.viewDistance);
        fr.maxPos = make_vec3f(m_sceneInfo.viewDistance, m_sceneInfo.viewDistance, m_sceneInfo.viewDistance);
    }
    for (int i = 0; i < NB_MAX_TEXTURES; ++i)
        delete[] m_hTextures[i].buffer;
    memset(m_hTextures, 0, sizeof(m_hTextures));
    m_hRandoms.clear();
    m_bitmap.clear();
    m_hBoundingBoxes.clear();
    m_hPrimitives.clear();
    m_hLamps.clear();
    m_hMaterials.clear();
    m_hPrimitivesXYIds.clear();
    m_lightInformation.clear();
    m_nbActiveMaterials = -1;
    m_nbActiveTextures = 0;
    m_materialsTransfered = false;
    m_primitivesTransfered = false;
    m_texturesTransfered = false;
    m_randomsTransfered = false;
    m_buffersInitialized = false;
}

void GPUKernel::reshape() {}

/* reference: GPUKernel.cpp:484-493 */
void GPUKernel::setCamera(const vec3f &eye, const vec3f &dir, const vec4f &angles)
{
    m_viewPos = eye;
    m_viewDir = dir;
    m_angles = angles;
    m_refresh = true;
}

/* reference: GPUKernel.cpp:495-516 */
int GPUKernel::addPrimitive(PrimitiveType type, bool belongsToModel)
{
    if (m_doneWithAdding)
        return m_addingIndex++;
    CPUPrimitive primitive;
    memset(&primitive, 0, sizeof(CPUPrimitive));
    primitive.belongsToModel = belongsToModel;
    primitive.type = type;
    int index = static_cast<int>(frame().primitives.size());
    frame().primitives[index] = primitive;
    return index;
}

CPUPrimitive *GPUKernel::getPrimitive(const unsigned int index)
{
    PrimitiveContainer &prims = frame().primitives;
    PrimitiveContainer::iterator it = prims.find(index);
    return it == prims.end() ? nullptr : &it->second;
}

void GPUKernel::setPrimitive(const int &index, float x0, float y0, float z0, float w, float h, float d, int materialId)
{
    setPrimitive(index, x0, y0, z0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, w, h, d, materialId);
}

void GPUKernel::setPrimitive(const int &index, float x0, float y0, float z0, float x1, float y1, float z1, float w,
                             float h, float d, int materialId)
{
    setPrimitive(index, x0, y0, z0, x1, y1, z1, 0.f, 0.f, 0.f, w, h, d, materialId);
}

/* reference: GPUKernel.cpp:539-684: stores the points and derives what the
 * intersection code expects per type (cylinder axis n1 and centre p2, plane
 * normals, triangle face normal); grows the scene extent by p0 only. */
void GPUKernel::setPrimitive(const int &index, float x0, float y0, float z0, float x1, float y1, float z1, float x2,
                             float y2, float z2, float w, float h, float d, int materialId)
{
    m_primitivesTransfered = false;
    CPUPrimitive *pp = (index >= 0) ? getPrimitive(index) : nullptr;
    if (!pp)
    {
        std::cerr << "GPUKernel::setPrimitive: Out of bounds (" << index << "/" << NB_MAX_PRIMITIVES << ")"
                  << std::endl;
        return;
    }
    CPUPrimitive &p = *pp;
    p.movable = true;
    p.p0 = make_vec3f(x0, y0, z0);
    p.p1 = make_vec3f(x1, y1, z1);
    p.p2 = make_vec3f(x2, y2, z2);
    p.size = make_vec3f(w, h, d);
    p.n0 = p.n1 = p.n2 = make_vec3f();
    p.vt0 = p.vt1 = p.vt2 = make_vec2f();
    p.materialId = materialId;

    switch (p.type)
    {
    case ptSphere:
        p.size = make_vec3f(w, w, w);
        break;
    case ptEllipsoid:
        p.size = make_vec3f(w, h, d);
        break;
    case ptCylinder:
    case ptCone:
    {
        vec3f axis = make_vec3f(x1 - x0, y1 - y0, z1 - z0);
        float len = sqrtf(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
        if (len != 0.f)
        {
            axis.x /= len;
            axis.y /= len;
            axis.z /= len;
        }
        p.n1 = axis;
        p.p2 = make_vec3f((x0 + x1) / 2.f, (y0 + y1) / 2.f, (z0 + z1) / 2.f);
        p.size = make_vec3f(w, w, w);
        break;
    }
    case ptXYPlane:
        p.n0 = make_vec3f(0.f, 0.f, 1.f);
        p.n1 = p.n2 = p.n0;
        break;
    case ptYZPlane:
        p.n0 = make_vec3f(1.f, 0.f, 0.f);
        p.n1 = p.n2 = p.n0;
        break;
    case ptXZPlane:
    case ptCheckboard:
        p.n0 = make_vec3f(0.f, 1.f, 0.f);
        p.n1 = p.n2 = p.n0;
        break;
    case ptTriangle:
    {
        vec3f v0 = make_vec3f(p.p1.x - p.p0.x, p.p1.y - p.p0.y, p.p1.z - p.p0.z);
        normalizeVector(v0);
        vec3f v1 = make_vec3f(p.p2.x - p.p0.x, p.p2.y - p.p0.y, p.p2.z - p.p0.z);
        normalizeVector(v1);
        p.n0 = crossProduct(v0, v1);
        normalizeVector(p.n0);
        p.n1 = p.n2 = p.n0;
        break;
    }
    default:
        break;
    }
    Frame &f = frame();
    f.minPos = make_vec3f(std::min(x0, f.minPos.x), std::min(y0, f.minPos.y), std::min(z0, f.minPos.z));
    f.maxPos = make_vec3f(std::max(x0, f.maxPos.x), std::max(y0, f.maxPos.y), std::max(z0, f.maxPos.z));
}

void GPUKernel::setPrimitiveIsMovable(const int &index, bool movable)
{
    if (CPUPrimitive *p = (index >= 0) ? getPrimitive(index) : nullptr)
        p->movable = movable;
}

void GPUKernel::setPrimitiveBellongsToModel(const int &index, bool bellongsToModel)
{
    if (CPUPrimitive *p = (index >= 0) ? getPrimitive(index) : nullptr)
        p->belongsToModel = bellongsToModel;
}

void GPUKernel::setPrimitiveTextureCoordinates(const unsigned int index, const vec2f &vt0, const vec2f &vt1,
                                               const vec2f &vt2)
{
    if (CPUPrimitive *p = getPrimitive(index))
    {
        p->vt0 = vt0;
        p->vt1 = vt1;
        p->vt2 = vt2;
    }
}

/* reference: GPUKernel.cpp:715-727 */
void GPUKernel::setPrimitiveNormals(unsigned int index, vec3f n0, vec3f n1, vec3f n2)
{
    if (CPUPrimitive *p = getPrimitive(index))
    {
        normalizeVector(n0);
        p->n0 = n0;
        normalizeVector(n1);
        p->n1 = n1;
        normalizeVector(n2);
        p->n2 = n2;
    }
}

/* reference: GPUKernel.cpp:729-739 */
unsigned int GPUKernel::getPrimitiveAt(int x, int y)
{
    unsigned int returnValue = -1;
    fetchPrimitiveIds();
    unsigned int index = y * m_sceneInfo.size.x + x;
    if (index < static_cast<unsigned int>(m_sceneInfo.size.x * m_sceneInfo.size.y) &&
        index < m_hPrimitivesXYIds.size())
        returnValue = m_hPrimitivesXYIds[index].x;
    return returnValue;
}

void GPUKernel::setPrimitiveMaterial(unsigned int index, int materialId)
{
    if (CPUPrimitive *p = getPrimitive(index))
        p->materialId = materialId;
}

int GPUKernel::getPrimitiveMaterial(unsigned int index)
{
    const CPUPrimitive *p = peekPrimitive(index);
    return p ? p->materialId : -1;
}

vec4f GPUKernel::getPrimitiveCenter(unsigned int index)
{
    vec4f center = make_vec4f();
    if (const CPUPrimitive *p = peekPrimitive(index))
    {
        center.x = p->p0.x;
        center.y = p->p0.y;
        center.z = p->p0.z;
    }
    return center;
}

void GPUKernel::setPrimitiveCenter(unsigned int index, const vec3f &center)
{
    m_primitivesTransfered = false;
    if (CPUPrimitive *p = getPrimitive(index))
        p->p0 = center;
}

/* reference: GPUKernel.cpp:2482-2495 */
int GPUKernel::getLight(int index)
{
    syncHost();
    if (index >= 0 && index < frameAsIs().nbActiveLamps && index < (int)m_hLamps.size())
        return m_hLamps[index];
    return -1;
}

/* ---------------------------------------------------------------------- */
/* Box tree                                                                */
/* ---------------------------------------------------------------------- */

/* reference: GPUKernel.cpp:741-839.  Returns whether the LAST primitive of
 * the box is emissive (the reference overwrites the flag per primitive). */
bool GPUKernel::updateBoundingBox(CPUBoundingBox &box)
{
    bool result = false;
    box.parameters[0] = make_vec3f(1000000.f, 1000000.f, 1000000.f);
    box.parameters[1] = make_vec3f(-1000000.f, -1000000.f, -1000000.f);
    PrimitiveContainer &prims = frame().primitives;
    for (long id : box.primitives)
    {
        CPUPrimitive &primitive = prims[id];
        result = (m_hMaterials[primitive.materialId].innerIllumination.x != 0.f);
        vec3f corner0, corner1;
        switch (primitive.type)
        {
        case ptTriangle:
            corner0 = min3(primitive.p0, primitive.p1, primitive.p2);
            corner1 = max3(primitive.p0, primitive.p1, primitive.p2);
            break;
        case ptCylinder:
            corner0 = min2(primitive.p0, primitive.p1);
            corner1 = max2(primitive.p0, primitive.p1);
            break;
        default:
            corner0 = primitive.p0;
            corner1 = primitive.p0;
            break;
        }
        vec3f p0 = min2(corner0, corner1);
        vec3f p1 = make_vec3f((corner0.x > corner1.x) ? corner0.x : corner1.x,
                              (corner0.y > corner1.y) ? corner0.y : corner1.y,
                              (corner0.z > corner1.z) ? corner0.z : corner1.z);
        switch (primitive.type)
        {
        case ptCylinder:
        case ptSphere:
        case ptCone:
            p0.x -= primitive.size.x;
            p0.y -= primitive.size.x;
            p0.z -= primitive.size.x;
            p1.x += primitive.size.x;
            p1.y += primitive.size.x;
            p1.z += primitive.size.x;
            break;
        default:
            p0.x -= primitive.size.x;
            p0.y -= primitive.size.y;
            p0.z -= primitive.size.z;
            p1.x += primitive.size.x;
            p1.y += primitive.size.y;
            p1.z += primitive.size.z;
            break;
        }
        if (p0.x < box.parameters[0].x) box.parameters[0].x = p0.x;
        if (p0.y < box.parameters[0].y) box.parameters[0].y = p0.y;
        if (p0.z < box.parameters[0].z) box.parameters[0].z = p0.z;
        if (p1.x > box.parameters[1].x) box.parameters[1].x = p1.x;
        if (p1.y > box.parameters[1].y) box.parameters[1].y = p1.y;
        if (p1.z > box.parameters[1].z) box.parameters[1].z = p1.z;
    }
    box.center.x = (box.parameters[0].x + box.parameters[1].x) / 2.f;
    box.center.y = (box.parameters[0].y + box.parameters[1].y) / 2.f;
    box.center.z = (box.parameters[0].z + box.parameters[1].z) / 2.f;
    return result;
}

/* reference: GPUKernel.cpp:841-890 */
bool GPUKernel::updateOutterBoundingBox(CPUBoundingBox &outterBox, const int depth)
{
    const float vd = m_sceneInfo.viewDistance;
    outterBox.parameters[0] = make_vec3f(vd, vd, vd);
    outterBox.parameters[1] = make_vec3f(-vd, -vd, -vd);
    BoxContainer &level = frame().boundingBoxes[depth];
    const bool direct = outterBox.childAt.size() == outterBox.primitives.size();
    for (size_t c = 0; c < outterBox.primitives.size(); ++c)
    {
        const unsigned int key = (unsigned int)outterBox.primitives[c];
        CPUBoundingBox *known = direct ? level.atIfKey(outterBox.childAt[c], key) : nullptr;
        CPUBoundingBox &box = known ? *known : level[key];
        if (outterBox.parameters[0].x > box.parameters[0].x) outterBox.parameters[0].x = box.parameters[0].x;
        if (outterBox.parameters[0].y > box.parameters[0].y) outterBox.parameters[0].y = box.parameters[0].y;
        if (outterBox.parameters[0].z > box.parameters[0].z) outterBox.parameters[0].z = box.parameters[0].z;
        if (outterBox.parameters[1].x < box.parameters[1].x) outterBox.parameters[1].x = box.parameters[1].x;
        if (outterBox.parameters[1].y < box.parameters[1].y) outterBox.parameters[1].y = box.parameters[1].y;
        if (outterBox.parameters[1].z < box.parameters[1].z) outterBox.parameters[1].z = box.parameters[1].z;
    }
    outterBox.center.x = (outterBox.parameters[0].x + outterBox.parameters[1].x) / 2.f;
    outterBox.center.y = (outterBox.parameters[0].y + outterBox.parameters[1].y) / 2.f;
    outterBox.center.z = (outterBox.parameters[0].z + outterBox.parameters[1].z) / 2.f;
    return false;
}

/* reference: GPUKernel.cpp:892-915 */
void GPUKernel::resetBoxes(bool resetPrimitives)
{
    ensureLevels();
    BoxContainer &level0 = frame().boundingBoxes[0];
    if (resetPrimitives)
        for (unsigned int i = 0; i < level0.size(); ++i)
            resetBox(level0[i], resetPrimitives);
    else
        level0.clear();
}

void GPUKernel::resetBox(CPUBoundingBox &box, bool resetPrimitives)
{
    if (resetPrimitives)
    {
        box.primitives.clear();
        box.childAt.clear();
        box.indexForNextBox = 1;
    }
    const float vd = m_sceneInfo.viewDistance;
    box.parameters[0] = make_vec3f(vd, vd, vd);
    box.parameters[1] = make_vec3f(-vd, -vd, -vd);
}

/* reference: GPUKernel.cpp:917-992: hash every primitive into a 6400^3 grid
 * over the p0 extent; emissive primitives go to box 0 of the top level. */
int GPUKernel::processBoxes(const int boxSize, bool simulate)
{
    Frame &f = frame();
    vec3f boxSteps;
    boxSteps.x = (f.maxPos.x - f.minPos.x) / boxSize;
    boxSteps.y = (f.maxPos.y - f.minPos.y) / boxSize;
    boxSteps.z = (f.maxPos.z - f.minPos.z) / boxSize;
    boxSteps.x = (boxSteps.x == 0.f) ? 1 : boxSteps.x;
    boxSteps.y = (boxSteps.y == 0.f) ? 1 : boxSteps.y;
    boxSteps.z = (boxSteps.z == 0.f) ? 1 : boxSteps.z;

    if (simulate)
        return 0;

    BoxContainer &level0 = f.boundingBoxes[0];
    level0.reserve(f.primitives.size());
    const float vd = m_sceneInfo.viewDistance;
    size_t maxPrimitivesPerBox = 0;
    unsigned int p = 0;
    for (auto &entry : f.primitives)
    {
        const CPUPrimitive &primitive = entry.second;
        const vec3f &center = primitive.p0;
        unsigned int X = static_cast<int>((center.x - f.minPos.x) / boxSteps.x);
        unsigned int Y = static_cast<int>((center.y - f.minPos.y) / boxSteps.y);
        unsigned int Z = static_cast<int>((center.z - f.minPos.z) / boxSteps.z);
        unsigned int B = 1 + 1000 * (X * boxSize * boxSize + Y * boxSize + Z);

        /* the cell is created for every primitive, lights included */
        if (!level0.contains(B))
        {
            CPUBoundingBox box;
            box.parameters[0] = make_vec3f(vd, vd, vd);
            box.parameters[1] = make_vec3f(-vd, -vd, -vd);
            box.center = make_vec3f();
            box.indexForNextBox = 1;
            level0.insert(std::make_pair(B, box));
        }
        if (m_hMaterials[primitive.materialId].innerIllumination.x != 0.f)
            f.boundingBoxes[m_treeDepth][0].primitives.push_back(p);
        else
        {
            CPUBoundingBox &cell = level0[B];
            cell.primitives.push_back(p);
            maxPrimitivesPerBox = std::max(maxPrimitivesPerBox, cell.primitives.size());
        }
        ++p;
    }
    for (auto &box : level0)
        updateBoundingBox(box.second);
    return static_cast<int>(maxPrimitivesPerBox);
}

/* reference: GPUKernel.cpp:994-1039: re-hash the boxes of the level below by
 * their centre on a boxSize^3 grid; key 0 is reserved for the lights. */
int GPUKernel::processOutterBoxes(const int boxSize, const int boundingBoxesDepth)
{
    Frame &f = frame();
    vec3f boxSteps;
    boxSteps.x = (f.maxPos.x - f.minPos.x) / boxSize;
    boxSteps.y = (f.maxPos.y - f.minPos.y) / boxSize;
    boxSteps.z = (f.maxPos.z - f.minPos.z) / boxSize;
    boxSteps.x = (boxSteps.x == 0.f) ? 1 : boxSteps.x;
    boxSteps.y = (boxSteps.y == 0.f) ? 1 : boxSteps.y;
    boxSteps.z = (boxSteps.z == 0.f) ? 1 : boxSteps.z;

    const float vd = m_sceneInfo.viewDistance;
    BoxContainer &level = f.boundingBoxes[boundingBoxesDepth];
    level.reserve(f.boundingBoxes[boundingBoxesDepth - 1].size() + 1);
    size_t maxPrimitivesPerBox = 0;
    BoxContainer &below = f.boundingBoxes[boundingBoxesDepth - 1];
    for (BoxContainer::iterator it = below.begin(); it != below.end(); ++it)
    {
        const vec3f &center = it->second.center;
        int X = static_cast<int>((center.x - f.minPos.x) / boxSteps.x);
        int Y = static_cast<int>((center.y - f.minPos.y) / boxSteps.y);
        int Z = static_cast<int>((center.z - f.minPos.z) / boxSteps.z);
        int B = wrapMulAdd3(X, Y, Z, boxSize);
        B++;
        CPUBoundingBox &outer = level[(unsigned int)B];
        outer.parameters[0] = make_vec3f(vd, vd, vd);
        outer.parameters[1] = make_vec3f(-vd, -vd, -vd);
        outer.primitives.push_back(it->first);
        outer.childAt.push_back(it.position());
        maxPrimitivesPerBox = std::max(maxPrimitivesPerBox, outer.primitives.size());
    }
    for (auto &box : level)
        updateOutterBoundingBox(box.second, boundingBoxesDepth - 1);
    return static_cast<int>(maxPrimitivesPerBox);
}

/* reference: GPUKernel.cpp:1041-1083 */
int GPUKernel::compactBoxes(bool reconstructBoxes)
{
    /* rotations applied on the device: the flattened scene over there is already what the lines below
     * would produce and upload */
    if (!reconstructBoxes && (!m_pendingRotations.empty() || m_unrecordedRotations))
        return frameAsIs().nbActiveBoxes;
    m_primitivesTransfered = false;
    if (reconstructBoxes)
    {
        /* the engine builds the same tree on the device (0.2 s of maps and hashing for 100 k primitives on the
         * host, 1.4-1.6 s in the reference; a few milliseconds there): the flattened arrays come back, the
         * per-level maps are left for whoever needs them (ensureLevels) */
        if (buildTreeOnDevice())
            return frameAsIs().nbActiveBoxes; /* frame() would make the level maps at once */
        buildLevelsOnHost();
    }
    else
        ensureLevels();
    streamDataToGPU();
    return frame().nbActiveBoxes;
}

/* reference: GPUKernel.cpp:1047-1076, the levels without the flattening */
void GPUKernel::buildLevelsOnHost()
{
    m_frames[m_frame].levelsBuilt = true; /* first: frame() below asks */
    const bool outer = !m_buildingLevels;
    m_buildingLevels = true;
    Frame &f = frame();
    /* The reference resets only the lights box here
     * (GPUKernel.cpp:1049) and relies on resetFrame() having emptied the
     * levels; a second compactBoxes(true) on the same frame would hash
     * every primitive into its cell twice.  Rebuild from empty levels. */
    for (int level = 0; level < BOUNDING_BOXES_TREE_DEPTH; ++level)
        f.boundingBoxes[level].clear();
    const int gridGranularity = 2;
    const int gridDivider = 4;

    m_treeDepth = 0;
    int nbBoxes = static_cast<int>(f.primitives.size());
    while (nbBoxes > gridGranularity)
    {
        ++m_treeDepth;
        nbBoxes /= gridDivider;
    }
    processBoxes(AABB_MAGIC_NUMBER, false);

    m_treeDepth = 0;
    nbBoxes = static_cast<int>(f.primitives.size());
    do
    {
        ++m_treeDepth;
        processOutterBoxes(nbBoxes, m_treeDepth);
        nbBoxes /= gridDivider;
    } while (nbBoxes > gridGranularity);
    m_frames[m_frame].levelsBuilt = true;
    if (outer)
        m_buildingLevels = false;
}

void GPUKernel::ensureLevels()
{
    if (!m_frames[m_frame].levelsBuilt)
        buildLevelsOnHost();
}

namespace
{
struct HostPhase
{
    const bool on = getenv("SOLR_HIP_DEBUG_TIMING") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on)
            return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "solr host: %-30s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
        last = now;
    }
};
} // namespace

/* compactBoxes(true) through the engine: true when the flattened arrays, lamps and light list are in place */
bool GPUKernel::buildTreeOnDevice()
{
    HostPhase phase;
    if (m_hostBuildOnly || getenv("SOLR_HOST_BUILD"))
        return false;
    m_frames[m_frame].levelsBuilt = true; /* whatever was left unbuilt is about to be replaced */
    Frame &f = frame();
    const int n = static_cast<int>(f.primitives.size());
    if (n < 1 || n >= NB_MAX_PRIMITIVES)
        return false;
    /* the builder numbers the primitives by their rank in the map (GPUKernel.cpp:932-990: `p`) and looks them
     * up by that number when it streams them: the two agree when the ids are 0..n-1 */
    /* the records are complete (normals and texture coordinates too, which the builder does not read): the
     * flattened array is these records in the order the builder returns - a gather from one array instead of a
     * second visit of the primitive store in a random order, which took as long as the build */
    std::vector<Primitive> prims(n);
    std::vector<unsigned char> emissive(n), movable(n);
    std::vector<CPUPrimitive *> byId(n);
    int rank = 0;
    for (auto &entry : f.primitives)
    {
        if ((int)entry.first != rank)
            return false;
        CPUPrimitive &p = entry.second;
        byId[rank] = &p;
        Primitive &out = prims[rank];
        memset(&out, 0, sizeof(out));
        out.index = rank;
        out.type = p.type;
        out.p0 = p.p0;
        out.p1 = p.p1;
        out.p2 = p.p2;
        out.n0 = p.n0;
        out.n1 = p.n1;
        out.n2 = p.n2;
        out.size = p.size;
        out.materialId = p.materialId;
        out.vt0 = p.vt0;
        out.vt1 = p.vt1;
        out.vt2 = p.vt2;
        if (p.materialId < 0 || p.materialId > NB_MAX_MATERIALS)
            return false;
        emissive[rank] = m_hMaterials[p.materialId].innerIllumination.x != 0.f;
        movable[rank] = p.movable && p.type != ptCamera;
        ++rank;
    }
    std::vector<BoundingBox> boxes;
    std::vector<int> order;
    int nbLamps = 0;
    phase.mark("compactBoxes: primitive records");
    const int depth = deviceBuildTree(prims, emissive, f.minPos, f.maxPos, m_sceneInfo.viewDistance, boxes, order, nbLamps);
    phase.mark("compactBoxes: device build");
    if (depth < 1 || (int)order.size() != n || boxes.empty() || boxes.size() >= (size_t)NB_MAX_BOXES)
        return false;

    /* what streamDataToGPU leaves behind (GPUKernel.cpp:1151-1281), from the device's node list and order */
    for (int level = 0; level < BOUNDING_BOXES_TREE_DEPTH; ++level)
        f.boundingBoxes[level].clear();
    m_treeDepth = depth;
    m_primitivesTransfered = false;
    f.nbActiveBoxes = 0;
    f.nbActivePrimitives = 0;
    f.nbActiveLamps = 0;
    m_maxPrimitivesPerBox = 0;
    m_hBoundingBoxes.swap(boxes);
    f.nbActiveBoxes = (int)m_hBoundingBoxes.size();
    m_hPrimitives.clear();
    m_hMovable.clear();
    m_hLamps.clear();
    m_hPrimitives.reserve(n);
    m_hMovable.reserve(n);
    if (m_lightInformation.size() < NB_MAX_LIGHTINFORMATIONS)
        m_lightInformation.assign(NB_MAX_LIGHTINFORMATIONS, LightInformation());
    m_lightInformationSize = 0;
    for (int k = 0; k < n; ++k)
    {
        const long id = order[k];
        if (id < 0 || id >= n)
            return false;
        m_hPrimitives.push_back(prims[id]); /* appendPrimitive's record */
        m_hMovable.push_back(k >= nbLamps && movable[id]);
        if (k < nbLamps)
        {
            CPUPrimitive &primitive = *byId[id];
            Material &material = m_hMaterials[primitive.materialId];
            LightInformation li;
            memset(&li, 0, sizeof(li));
            li.primitiveId = (int)id;
            li.materialId = primitive.materialId;
            li.location = primitive.p0;
            li.color.x = material.color.x;
            li.color.y = material.color.y;
            li.color.z = material.color.z;
            li.color.w = material.innerIllumination.x;
            if (m_lightInformationSize < NB_MAX_LIGHTINFORMATIONS)
                m_lightInformation[m_lightInformationSize] = li;
            m_hLamps.push_back((Lamp)id);
            ++f.nbActiveLamps;
            ++m_lightInformationSize;
        }
    }
    f.nbActivePrimitives = n;
    for (const BoundingBox &b : m_hBoundingBoxes)
        m_maxPrimitivesPerBox = std::max(m_maxPrimitivesPerBox, (size_t)std::max(b.nbPrimitives, 0));
    m_frames[m_frame].levelsBuilt = false; /* the maps follow when somebody needs them (frame()) */
    phase.mark("compactBoxes: flattened arrays");
    return true;
}

void GPUKernel::appendPrimitive(long id, bool inLevel0Box)
{
    CPUPrimitive &primitive = frame().primitives[(unsigned int)id];
    Primitive out;
    memset(&out, 0, sizeof(out));
    out.index = (int)id;
    out.type = primitive.type;
    out.p0 = primitive.p0;
    out.p1 = primitive.p1;
    out.p2 = primitive.p2;
    out.n0 = primitive.n0;
    out.n1 = primitive.n1;
    out.n2 = primitive.n2;
    out.size = primitive.size;
    out.materialId = primitive.materialId;
    out.vt0 = primitive.vt0;
    out.vt1 = primitive.vt1;
    out.vt2 = primitive.vt2;
    m_hPrimitives.push_back(out);
    m_hMovable.push_back(inLevel0Box && primitive.movable && primitive.type != ptCamera);
    ++frame().nbActivePrimitives;
}

/* reference: GPUKernel.cpp:1085-1149: depth-first emission; a node's skip
 * pointer is the number of nodes emitted for its subtree. */
void GPUKernel::recursiveDataStreamToGPU(const int depth, CPUBoundingBox &parent)
{
    Frame &f = frameAsIs();
    BoxContainer &level = f.boundingBoxes[depth];
    const bool direct = parent.childAt.size() == parent.primitives.size();
    for (size_t c = 0; c < parent.primitives.size(); ++c)
    {
        /* operator[] semantics: a missing key yields an empty box (skipped).  It may also grow the level
         * and move its boxes: `parent` lives one level up and stays put */
        const unsigned int key = (unsigned int)parent.primitives[c];
        CPUBoundingBox *known = direct ? level.atIfKey(parent.childAt[c], key) : nullptr;
        CPUBoundingBox &box = known ? *known : level[key];
        if (box.primitives.size() != 0 && f.nbActiveBoxes < NB_MAX_BOXES)
        {
            int boxIndex = f.nbActiveBoxes;
            BoundingBox out;
            memset(&out, 0, sizeof(out));
            out.parameters[0] = box.parameters[0];
            out.parameters[1] = box.parameters[1];
            out.nbPrimitives = (depth == 0) ? static_cast<int>(box.primitives.size()) : 0;
            out.startIndex = (depth == 0) ? f.nbActivePrimitives : depth;
            m_hBoundingBoxes.push_back(out);
            ++f.nbActiveBoxes;
            if (depth == 0)
            {
                m_maxPrimitivesPerBox = std::max(m_maxPrimitivesPerBox, box.primitives.size());
                for (long id : box.primitives)
                    if (id < NB_MAX_PRIMITIVES)
                        appendPrimitive(id, true);
            }
            else
                recursiveDataStreamToGPU(depth - 1, box);
            m_hBoundingBoxes[boxIndex].indexForNextBox.x = (depth == 0) ? 1 : f.nbActiveBoxes - boxIndex;
        }
    }
}

/* reference: GPUKernel.cpp:1151-1281 */
void GPUKernel::streamDataToGPU()
{
    Frame &f = frame();
    m_primitivesTransfered = false;
    f.nbActiveBoxes = 0;
    f.nbActivePrimitives = 0;
    f.nbActiveLamps = 0;
    m_maxPrimitivesPerBox = 0;
    m_hBoundingBoxes.clear();
    m_hPrimitives.clear();
    m_hMovable.clear();
    m_hLamps.clear();
    {
        size_t nodes = 0;
        for (int level = 0; level < BOUNDING_BOXES_TREE_DEPTH; ++level)
            nodes += f.boundingBoxes[level].size();
        m_hBoundingBoxes.reserve(nodes);
        m_hPrimitives.reserve(f.primitives.size());
        m_hMovable.reserve(f.primitives.size());
    }
    if (m_lightInformation.size() < NB_MAX_LIGHTINFORMATIONS)
        m_lightInformation.assign(NB_MAX_LIGHTINFORMATIONS, LightInformation());

    const int maxDepth = m_treeDepth;
    const float vd = m_sceneInfo.viewDistance;
    BoxContainer &top = f.boundingBoxes[maxDepth];
    for (BoxContainer::iterator itob = top.begin(); itob != top.end(); ++itob)
    {
        CPUBoundingBox &box = itob->second;
        int boxIndex = f.nbActiveBoxes;
        BoundingBox out;
        memset(&out, 0, sizeof(out));
        out.parameters[0] = box.parameters[0];
        out.parameters[1] = box.parameters[1];
        out.nbPrimitives = 0;
        out.startIndex = maxDepth;
        if (itob == top.begin())
        {
            /* first master box: the lights, with scene-wide bounds */
            m_lightInformationSize = 0;
            out.parameters[0] = make_vec3f(-vd, -vd, -vd);
            out.parameters[1] = make_vec3f(vd, vd, vd);
            out.nbPrimitives = static_cast<int>(box.primitives.size());
            out.startIndex = 0;
            for (long id : box.primitives)
            {
                appendPrimitive(id, false);
                CPUPrimitive &primitive = f.primitives[(unsigned int)id];
                Material &material = m_hMaterials[primitive.materialId];
                LightInformation li;
                memset(&li, 0, sizeof(li));
                li.primitiveId = (int)id;
                li.materialId = primitive.materialId;
                li.location = primitive.p0;
                li.color.x = material.color.x;
                li.color.y = material.color.y;
                li.color.z = material.color.z;
                li.color.w = material.innerIllumination.x;
                if (m_lightInformationSize < NB_MAX_LIGHTINFORMATIONS)
                    m_lightInformation[m_lightInformationSize] = li;
                m_hLamps.push_back((Lamp)id);
                ++f.nbActiveLamps;
                ++m_lightInformationSize;
            }
        }
        m_hBoundingBoxes.push_back(out);
        ++f.nbActiveBoxes;
        if (maxDepth > 0)
            recursiveDataStreamToGPU(maxDepth - 1, box);
        m_hBoundingBoxes[boxIndex].indexForNextBox.x = f.nbActiveBoxes - boxIndex;
    }
    if (f.nbActivePrimitives != (int)f.primitives.size())
        std::cerr << "Lost primitives on the way for frame " << m_frame << "... " << f.nbActivePrimitives
                  << "!=" << f.primitives.size() << std::endl;
}

/* reference: GPUKernel.cpp:2764-2775 */
void GPUKernel::nextFrame()
{
    syncHost();
    ++m_frame;
    if (m_frame >= m_nbFrames)
        m_frame = m_nbFrames ? m_nbFrames - 1 : 0;
}

void GPUKernel::previousFrame()
{
    syncHost();
    if (m_frame > 0)
        --m_frame;
}

/* reference: GPUKernel.cpp:1686-1690 */
void GPUKernel::getPrimitiveOtherCenter(unsigned int index, vec3f &center)
{
    if (const CPUPrimitive *p = peekPrimitive(index))
        center = p->p1;
}

/* reference: GPUKernel.cpp:1513-1572: the frames between the first and the last key frame are rebuilt as
 * blends of the two, primitive by primitive in id order, with weight frame / nbFrames; each gets its own
 * box tree */
void GPUKernel::morphPrimitives()
{
    if (m_nbFrames < 3)
        return;
    syncHost();
    const float nbFrames = static_cast<float>(m_nbFrames);
    for (unsigned int frame = 1; frame < m_nbFrames - 1; ++frame)
    {
        setFrame((int)frame);
        resetFrame();
        PrimitiveContainer &first = m_frames[0].primitives;
        PrimitiveContainer &last = m_frames[m_nbFrames - 1].primitives;
        const float r = static_cast<float>(m_frame) / nbFrames;
        auto mix = [r](const vec3f &a, const vec3f &b) {
            return make_vec3f(a.x + r * (b.x - a.x), a.y + r * (b.y - a.y), a.z + r * (b.z - a.z));
        };
        PrimitiveContainer::iterator it2 = last.begin();
        for (PrimitiveContainer::iterator it1 = first.begin(); it1 != first.end() && it2 != last.end(); ++it1, ++it2)
        {
            const CPUPrimitive a = it1->second, b = it2->second; /* copies: adding below may move the store */
            const vec3f p0 = mix(a.p0, b.p0), p1 = mix(a.p1, b.p1), p2 = mix(a.p2, b.p2);
            const vec3f size = mix(a.size, b.size);
            const int i = addPrimitive(PrimitiveType(a.type));
            setPrimitive(i, p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, size.x, size.y, size.z, a.materialId);
            setPrimitiveNormals(i, mix(a.n0, b.n0), mix(a.n1, b.n1), mix(a.n2, b.n2));
            setPrimitiveTextureCoordinates(i, a.vt0, a.vt1, a.vt2);
            setPrimitiveIsMovable(i, a.movable);
        }
        compactBoxes(true);
    }
}

/* reference: GPUKernel.cpp:1283-1305 */
void GPUKernel::resetFrame()
{
    m_frames[m_frame].levelsBuilt = true; /* nothing to make them for any more */
    Frame &f = frame();
    vec3f mn = f.minPos, mx = f.maxPos; /* the reference keeps the extent across resets */
    f = Frame();
    f.minPos = mn;
    f.maxPos = mx;
}

/* reference: GPUKernel.cpp:1307-1349 */
void GPUKernel::resetAll()
{
    unsigned int oldFrame = m_frame;
    for (auto &entry : m_frames)
    {
        m_frame = entry.first;
        resetFrame();
    }
    m_frame = oldFrame;
    m_primitivesTransfered = false;
    m_nbActiveMaterials = -1;
    m_materialsTransfered = false;
    for (int i = 0; i < NB_MAX_TEXTURES; ++i)
        delete[] m_hTextures[i].buffer;
    memset(m_hTextures, 0, sizeof(m_hTextures));
    m_nbActiveTextures = 0;
    m_texturesTransfered = false;
}

/* reference: GPUKernel.cpp:1602-1630 */
void GPUKernel::rotateVector(vec3f &v, const vec3f &c, const vec3f &cosA, const vec3f &sinA)
{
    float vx = v.x - c.x, vy = v.y - c.y, vz = v.z - c.z;
    float ry = vy * cosA.x - vz * sinA.x;
    float rz = vy * sinA.x + vz * cosA.x;
    vy = ry;
    vz = rz;
    rz = vz * cosA.y - vx * sinA.y;
    float rx = vz * sinA.y + vx * cosA.y;
    vz = rz;
    vx = rx;
    rx = vx * cosA.z - vy * sinA.z;
    ry = vx * sinA.z + vy * cosA.z;
    v = make_vec3f(rx + c.x, ry + c.y, rz + c.z);
}

/* reference: GPUKernel.cpp:1378-1460, 1639-1672: rotates the primitives held
 * by level-0 boxes (lights live in the top-level box and stay put) and
 * refits every level without re-hashing. */
void GPUKernel::rotatePrimitives(const vec3f &rotationCenter, const vec4f &angles)
{
    vec3f cosA = make_vec3f(cosf(angles.x), cosf(angles.y), cosf(angles.z));
    vec3f sinA = make_vec3f(sinf(angles.x), sinf(angles.y), sinf(angles.z));
    /* the scene the device holds is the one the host holds (plus the rotations already pending): the
     * engine may turn it in place, and the host copy follows when somebody looks (syncHost) */
    if (m_primitivesTransfered && !m_hostTouched && deviceRotatePrimitives(rotationCenter, cosA, sinA))
    {
        if (m_pendingRotations.size() >= MAX_RECORDED_ROTATIONS)
        {
            ++m_unrecordedRotations;
            return;
        }
        PendingRotation r;
        r.center = rotationCenter;
        r.cosA = cosA;
        r.sinA = sinA;
        m_pendingRotations.push_back(r);
        return;
    }
    Frame &f = frame();
    m_primitivesTransfered = false;
    rotatePrimitivesOnly(f, rotationCenter, cosA, sinA);
    refitBoxes(f);
}

void GPUKernel::rotatePrimitivesOnly(Frame &f, const vec3f &rotationCenter, const vec3f &cosA, const vec3f &sinA)
{
    ensureLevels();
    const vec3f zero = make_vec3f();
    for (auto &entry : f.boundingBoxes[0])
    {
        CPUBoundingBox &box = entry.second;
        for (long id : box.primitives)
        {
            CPUPrimitive &p = f.primitives[(unsigned int)id];
            if (p.movable && p.type != ptCamera)
            {
                rotateVector(p.p0, rotationCenter, cosA, sinA);
                if (p.type == ptCylinder || p.type == ptTriangle)
                {
                    rotateVector(p.p1, rotationCenter, cosA, sinA);
                    rotateVector(p.p2, rotationCenter, cosA, sinA);
                    rotateVector(p.n0, zero, cosA, sinA);
                    rotateVector(p.n1, zero, cosA, sinA);
                    rotateVector(p.n2, zero, cosA, sinA);
                    if (p.type == ptCylinder)
                    {
                        vec3f axis = make_vec3f(p.p1.x - p.p0.x, p.p1.y - p.p0.y, p.p1.z - p.p0.z);
                        float len = sqrtf(axis.x * axis.x + axis.y * axis.y + axis.z * axis.z);
                        if (len != 0)
                        {
                            axis.x /= len;
                            axis.y /= len;
                            axis.z /= len;
                        }
                        p.n1 = axis;
                    }
                }
            }
        }
    }
}

/* the box updates of the reference's rotatePrimitives: they read the primitives only, so one pass after
 * several rotations leaves what one pass after each would */
void GPUKernel::refitBoxes(Frame &f)
{
    ensureLevels();
    for (auto &entry : f.boundingBoxes[0])
    {
        resetBox(entry.second, false);
        updateBoundingBox(entry.second);
    }
    for (int b = 1; b < BOUNDING_BOXES_TREE_DEPTH; ++b)
        for (auto &entry : f.boundingBoxes[b])
            updateOutterBoundingBox(entry.second, b - 1);
}

void GPUKernel::syncHost()
{
    if (m_pendingRotations.empty() && !m_unrecordedRotations)
        return;
    const bool transfered = m_primitivesTransfered, touched = m_hostTouched;
    ensureLevels(); /* from the primitives as they were built with, before the rotations reach the store */
    std::vector<PendingRotation> pending;
    pending.swap(m_pendingRotations);
    const size_t unrecorded = m_unrecordedRotations;
    m_unrecordedRotations = 0;
    Frame &f = m_frames[m_frame];
    /* a handful of rotations is replayed (4 ms each for 100 k primitives); a long animation is fetched */
    if (pending.size() <= 8 || !primitivesFromDevice(f))
    {
        if (unrecorded)
            std::cerr << "GPUKernel::syncHost: the engine did not hand back its primitives and " << unrecorded
                      << " of the rotations it applied were not recorded: the scene store lags behind" << std::endl;
        for (const PendingRotation &r : pending)
            rotatePrimitivesOnly(f, r.center, r.cosA, r.sinA);
    }
    refitBoxes(f);
    /* the flattened arrays follow.  What they now hold is what the device holds (the same arithmetic
     * ran there), so neither an upload is due nor has the scene been "touched" by this */
    streamDataToGPU();
    m_primitivesTransfered = transfered;
    m_hostTouched = touched;
}

/* reference: GPUKernel.cpp:1462-1511 */
void GPUKernel::translatePrimitives(const vec3f &t)
{
    m_primitivesTransfered = false;
    ensureLevels();
    Frame &f = frame();
    for (auto &entry : f.boundingBoxes[0])
    {
        CPUBoundingBox &box = entry.second;
        resetBox(box, false);
        for (long id : box.primitives)
        {
            CPUPrimitive &p = f.primitives[(unsigned int)id];
            if (p.movable && p.type != ptCamera)
            {
                p.p0 = make_vec3f(p.p0.x + t.x, p.p0.y + t.y, p.p0.z + t.z);
                p.p1 = make_vec3f(p.p1.x + t.x, p.p1.y + t.y, p.p1.z + t.z);
                p.p2 = make_vec3f(p.p2.x + t.x, p.p2.y + t.y, p.p2.z + t.z);
            }
        }
        updateBoundingBox(box);
    }
    for (int b = 1; b < BOUNDING_BOXES_TREE_DEPTH; ++b)
        for (auto &entry : f.boundingBoxes[b])
            updateOutterBoundingBox(entry.second, b - 1);
}

/* reference: GPUKernel.cpp:1574-1600 (from/to are ignored there too) */
void GPUKernel::scalePrimitives(float scale, unsigned int, unsigned int)
{
    m_primitivesTransfered = false;
    for (auto &entry : frame().primitives)
    {
        CPUPrimitive &p = entry.second;
        p.p0 = make_vec3f(p.p0.x * scale, p.p0.y * scale, p.p0.z * scale);
        p.p1 = make_vec3f(p.p1.x * scale, p.p1.y * scale, p.p1.z * scale);
        p.p2 = make_vec3f(p.p2.x * scale, p.p2.y * scale, p.p2.z * scale);
        p.size = make_vec3f(p.size.x * scale, p.size.y * scale, p.size.z * scale);
    }
}

/* reference: GPUKernel.cpp:1705-1739 */
int GPUKernel::addCube(float x, float y, float z, float radius, int materialId)
{
    return addRectangle(x, y, z, radius, radius, radius, materialId);
}

int GPUKernel::addRectangle(float x, float y, float z, float w, float h, float d, int materialId)
{
    int id = addPrimitive(ptXYPlane);
    setPrimitive(id, x, y, z + d, w, h, d, materialId);
    id = addPrimitive(ptXYPlane);
    setPrimitive(id, x, y, z - d, w, h, d, materialId);
    id = addPrimitive(ptYZPlane);
    setPrimitive(id, x - w, y, z, w, h, d, materialId);
    id = addPrimitive(ptYZPlane);
    setPrimitive(id, x + w, y, z, w, h, d, materialId);
    id = addPrimitive(ptXZPlane);
    setPrimitive(id, x, y + h, z, w, h, d, materialId);
    id = addPrimitive(ptXZPlane);
    setPrimitive(id, x, y - h, z, w, h, d, materialId);
    return id;
}

/* ---------------------------------------------------------------------- */
/* Materials and textures                                                  */
/* ---------------------------------------------------------------------- */

int GPUKernel::addMaterial() { return ++m_nbActiveMaterials; }

void GPUKernel::setMaterial(unsigned int index, const Material &material)
{
    if (index < NB_MAX_MATERIALS && index < m_hMaterials.size())
    {
        m_hMaterials[index] = material;
        m_materialsTransfered = false;
    }
}

/* reference: GPUKernel.cpp:1780-1909 */
void GPUKernel::setMaterial(unsigned int index, float r, float g, float b, float noise, float reflection,
                            float refraction, bool procedural, bool wireframe, int wireframeWidth, float transparency,
                            float opacity, int diffuseTextureId, int normalTextureId, int bumpTextureId,
                            int specularTextureId, int reflectionTextureId, int transparentTextureId,
                            int ambientOcclusionTextureId, float specValue, float specPower, float specCoef,
                            float innerIllumination, float illuminationDiffusion, float illuminationPropagation,
                            bool fastTransparency)
{
    if (index >= NB_MAX_MATERIALS || index >= m_hMaterials.size())
    {
        std::cerr << "GPUKernel::setMaterial: Out of bounds(" << index << "/" << NB_MAX_MATERIALS << ")" << std::endl;
        return;
    }
    Material &m = m_hMaterials[index];
    m.color = make_vec4f(r, g, b, 0.f);
    m.specular = make_vec4f(specValue, specPower, 0.f, specCoef);
    m.innerIllumination = make_vec4f(innerIllumination, illuminationDiffusion, illuminationPropagation, noise);
    m.reflection = reflection;
    m.refraction = refraction;
    m.transparency = transparency;
    m.opacity = opacity;
    m.attributes = make_vec4i(fastTransparency ? 1 : 0, procedural ? 1 : 0,
                              wireframe ? ((wireframeWidth == 0) ? 1 : 2) : 0, wireframeWidth);
    m.textureIds = make_vec4i(diffuseTextureId, normalTextureId, bumpTextureId, specularTextureId);
    m.advancedTextureIds =
        make_vec4i(reflectionTextureId, transparentTextureId, ambientOcclusionTextureId, TEXTURE_NONE);
    m.advancedTextureOffset = make_vec4i();
    m.mappingOffset = make_vec2f(1.f, 1.f);
    auto off = [this](int id) { return (id < 0 || id >= NB_MAX_TEXTURES) ? 0 : m_hTextures[id].offset; };
    if (diffuseTextureId >= 0 && diffuseTextureId < m_nbActiveTextures)
    {
        const TextureInfo &t = m_hTextures[diffuseTextureId];
        m.textureMapping = make_vec4i(t.size.x, t.size.y, TEXTURE_NONE, t.size.z);
        m.textureOffset = make_vec4i(t.offset, off(normalTextureId), off(bumpTextureId), off(specularTextureId));
        m.advancedTextureOffset =
            make_vec4i(off(reflectionTextureId), off(transparentTextureId), off(ambientOcclusionTextureId), 0);
    }
    else
    {
        /* computed textures (Mandelbrot, Julia) or none */
        m.textureMapping = make_vec4i(40000, 40000, TEXTURE_NONE, 3);
        m.textureIds = make_vec4i(diffuseTextureId, TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE);
        m.textureOffset = make_vec4i();
    }
    m_materialsTransfered = false;
}

/* reference: GPUKernel.cpp:1911-1930 */
void GPUKernel::setMaterialColor(unsigned int index, float r, float g, float b)
{
    if (index < NB_MAX_MATERIALS && index < m_hMaterials.size())
    {
        m_hMaterials[index].color.x = r;
        m_hMaterials[index].color.y = g;
        m_hMaterials[index].color.z = b;
        m_materialsTransfered = false;
    }
}

Material *GPUKernel::getMaterial(const int index)
{
    if (index >= 0 && index <= m_nbActiveMaterials && index < (int)m_hMaterials.size())
        return &m_hMaterials[index];
    return nullptr;
}

/* reference: GPUKernel.cpp:1932-1953: the current material becomes a plain textured one */
void GPUKernel::setMaterialTextureId(unsigned int textureId)
{
    if (m_currentMaterial < 0 || (size_t)m_currentMaterial >= m_hMaterials.size() || textureId >= NB_MAX_TEXTURES)
        return;
    Material &m = m_hMaterials[m_currentMaterial];
    if ((int)textureId != m.textureIds.x)
    {
        m.reflection = 0.3f;
        m.refraction = 0.f;
        m.transparency = 0.f;
        m.opacity = 0.f;
        m.textureMapping = make_vec4i(m_hTextures[textureId].size.x, m_hTextures[textureId].size.y, TEXTURE_NONE,
                                      m_hTextures[textureId].size.z);
        m.textureIds = make_vec4i((int)textureId, TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE);
        m_materialsTransfered = false;
    }
}

/* reference: GPUKernel.cpp:2017-2033 */
void GPUKernel::setTexture(const int index, const TextureInfo &textureInfo)
{
    if (index < 0 || index >= NB_MAX_TEXTURES)
        return;
    if (index >= m_nbActiveTextures)
        ++m_nbActiveTextures;
    delete[] m_hTextures[index].buffer;
    int size = textureInfo.size.x * textureInfo.size.y * textureInfo.size.z;
    m_hTextures[index].buffer = new BitmapBuffer[size];
    m_hTextures[index].offset = 0;
    m_hTextures[index].size = textureInfo.size;
    m_hTextures[index].type = textureInfo.type;
    memcpy(m_hTextures[index].buffer, textureInfo.buffer, size);
    realignTexturesAndMaterials();
    m_texturesTransfered = false;
}

void GPUKernel::getTexture(const int index, TextureInfo &textureInfo)
{
    if (index >= 0 && index < NB_MAX_TEXTURES && index <= m_nbActiveTextures)
        textureInfo = m_hTextures[index];
}

TextureInfo &GPUKernel::getTextureInformation(const int index) { return m_hTextures[index]; }

/* reference: GPUKernel.cpp:2691-2705 */
void GPUKernel::processTextureOffsets()
{
    int totalSize = 0;
    for (int i = 0; i < NB_MAX_TEXTURES; ++i)
    {
        if (m_hTextures[i].buffer != 0)
        {
            m_hTextures[i].offset = totalSize;
            totalSize += m_hTextures[i].size.x * m_hTextures[i].size.y * m_hTextures[i].size.z;
        }
        else
            m_hTextures[i].offset = 0;
    }
}

const std::vector<BitmapBuffer> &GPUKernel::hostTextureAtlas()
{
    processTextureOffsets();
    size_t total = 0;
    for (int i = 0; i < NB_MAX_TEXTURES; ++i)
        if (m_hTextures[i].buffer)
            total += (size_t)m_hTextures[i].size.x * m_hTextures[i].size.y * m_hTextures[i].size.z;
    m_textureAtlas.assign(total, 0);
    for (int i = 0; i < NB_MAX_TEXTURES; ++i)
        if (m_hTextures[i].buffer)
            memcpy(m_textureAtlas.data() + m_hTextures[i].offset, m_hTextures[i].buffer,
                   (size_t)m_hTextures[i].size.x * m_hTextures[i].size.y * m_hTextures[i].size.z);
    return m_textureAtlas;
}

/* reference: GPUKernel.cpp:2238-2348 (loop bound m_nbActiveMaterials, exclusive, as there) */
void GPUKernel::realignTexturesAndMaterials()
{
    processTextureOffsets();
    auto off = [this](int id) { return (id < 0 || id >= NB_MAX_TEXTURES) ? 0 : m_hTextures[id].offset; };
    for (int i = 0; i < m_nbActiveMaterials && i < (int)m_hMaterials.size(); ++i)
    {
        Material &m = m_hMaterials[i];
        const int diffuseTextureId = m.textureIds.x;
        const int normalTextureId = m.textureIds.y;
        const int bumpTextureId = m.textureIds.z;
        const int specularTextureId = m.textureIds.w;
        const int reflectionTextureId = m.advancedTextureIds.x;
        const int transparencyTextureId = m.advancedTextureIds.y;
        switch (diffuseTextureId)
        {
        case TEXTURE_MANDELBROT:
        case TEXTURE_JULIA:
            m.textureMapping = make_vec4i(40000, 40000, TEXTURE_NONE, 3);
            m.textureIds = make_vec4i(diffuseTextureId, TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE);
            m.textureOffset = make_vec4i();
            m.advancedTextureIds = make_vec4i(TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE, TEXTURE_NONE);
            m.advancedTextureOffset = make_vec4i();
            break;
        default:
            /* the reference indexes m_hTextures[-1] for untextured materials
             * (diffuseTextureId == TEXTURE_NONE < m_nbActiveTextures); it then
             * overwrites textureMapping with whatever precedes the array.  Not
             * reproduced: untextured materials keep their mapping. */
            if (diffuseTextureId >= 0 && diffuseTextureId < m_nbActiveTextures)
            {
                const TextureInfo &t = m_hTextures[diffuseTextureId];
                m.textureMapping = make_vec4i(t.size.x, t.size.y, TEXTURE_NONE, t.size.z);
                m.textureOffset =
                    make_vec4i(t.offset, off(normalTextureId), off(bumpTextureId), off(specularTextureId));
                m.advancedTextureIds.x = reflectionTextureId;
                m.advancedTextureIds.y = transparencyTextureId;
                m.advancedTextureOffset.x = off(reflectionTextureId);
                m.advancedTextureOffset.y = off(transparencyTextureId);
                m.mappingOffset = make_vec2f(1.f, 0.f);
            }
            else if (diffuseTextureId >= m_nbActiveTextures)
            {
                m.textureMapping = make_vec4i(1, 1, TEXTURE_NONE, 1);
                m.textureOffset = make_vec4i();
                m.advancedTextureIds = make_vec4i(reflectionTextureId, transparencyTextureId, TEXTURE_NONE, TEXTURE_NONE);
                m.advancedTextureOffset = make_vec4i();
                m.mappingOffset = make_vec2f(1.f, 1.f);
            }
        }
    }
}

/* ---------------------------------------------------------------------- */
/* Scene                                                                   */
/* ---------------------------------------------------------------------- */

/* reference: GPUKernel.cpp:2044-2069 (zeroes geometryEpsilon/rayEpsilon/
 * extendedGeometry: SURVEY.md appendix A.12) */
void GPUKernel::setSceneInfo(int width, int height, float transparentColor, int graphicsLevel, float viewDistance,
                             float shadowIntensity, int nbRayIterations, vec4f backgroundColor, int cameraType,
                             float eyeSeparation, bool renderBoxes, int pathTracingIteration,
                             int maxPathTracingIterations, FrameBufferType frameBufferType, int timestamp,
                             int atmosphericEffect, int skyboxSize, int skyboxMaterialId)
{
    (void)cameraType; /* the reference ignores it as well */
    memset(&m_sceneInfo, 0, sizeof(SceneInfo));
    m_sceneInfo.size.x = width;
    m_sceneInfo.size.y = height;
    m_sceneInfo.transparentColor = transparentColor;
    m_sceneInfo.graphicsLevel = graphicsLevel;
    m_sceneInfo.viewDistance = viewDistance;
    m_sceneInfo.shadowIntensity = shadowIntensity;
    m_sceneInfo.nbRayIterations = nbRayIterations;
    m_sceneInfo.backgroundColor = backgroundColor;
    m_sceneInfo.eyeSeparation = eyeSeparation;
    m_sceneInfo.renderBoxes = renderBoxes;
    m_sceneInfo.pathTracingIteration = pathTracingIteration;
    m_sceneInfo.maxPathTracingIterations = maxPathTracingIterations;
    m_sceneInfo.frameBufferType = frameBufferType;
    m_sceneInfo.timestamp = timestamp;
    m_sceneInfo.atmosphericEffect = atmosphericEffect;
    m_sceneInfo.skyboxRadius = skyboxSize;
    m_sceneInfo.skyboxMaterialId = skyboxMaterialId;
}

void GPUKernel::setSceneInfo(const SceneInfo &sceneInfo) { m_sceneInfo = sceneInfo; }
SceneInfo &GPUKernel::getSceneInfo() { return m_sceneInfo; }

void GPUKernel::setPostProcessingInfo(PostProcessingType type, float param1, float param2, int param3)
{
    m_postProcessingInfo.type = type;
    m_postProcessingInfo.param1 = param1;
    m_postProcessingInfo.param2 = param2;
    m_postProcessingInfo.param3 = param3;
}

void GPUKernel::setPostProcessingInfo(const PostProcessingInfo &postProcessingInfo)
{
    m_postProcessingInfo = postProcessingInfo;
}

unsigned int GPUKernel::getNbActiveBoxes() { return frameAsIs().nbActiveBoxes; }
unsigned int GPUKernel::getNbActivePrimitives() { return frameAsIs().nbActivePrimitives; }
unsigned int GPUKernel::getNbActiveLamps() { return frameAsIs().nbActiveLamps; }
unsigned int GPUKernel::getNbActiveMaterials() { return m_nbActiveMaterials; }
unsigned int GPUKernel::getNbActiveTextures() { return m_nbActiveTextures; }

void GPUKernel::setDeterministic(long seed)
{
    m_deterministicSeed = seed;
    m_randomsTransfered = false;
    m_randomsFilled = false;
}

/* The reference keeps MAX_BITMAP_SIZE values (GPUKernel.cpp:350) and its 1920 x 1080 limit; its natural depth
 * of field reads `pixel index + timestamp % (MAX_BITMAP_SIZE - 2)` (CudaRayTracer.cu:475), so a larger frame
 * needs width * height + 10 000 of them (W * H + 10002).  The first MAX_BITMAP_SIZE
 * values of a seed are the same whatever the length. */
size_t GPUKernel::randomsNeeded() const
{
    const size_t pixels = (size_t)std::max(m_sceneInfo.size.x, 0) * (size_t)std::max(m_sceneInfo.size.y, 0);
    return pixels > (size_t)MAX_BITMAP_SIZE ? pixels + 10002 : (size_t)MAX_BITMAP_SIZE;
}

/* what render_begin would upload with the next frame, for callers that look at the buffer before a frame was
 * rendered (the test harness hands it to the oracle) */
const std::vector<RandomBuffer> &GPUKernel::hostRandoms()
{
    if (!m_randomsFilled || m_hRandoms.size() != randomsNeeded())
    {
        m_randomsTransfered = false;
        fillRandoms();
    }
    return m_hRandoms;
}

void GPUKernel::fillRandoms()
{
    m_randomsFilled = true;
    if (m_hRandoms.size() != randomsNeeded())
        m_hRandoms.assign(randomsNeeded(), 0.f);
    if (m_deterministicSeed >= 0)
    {
        unsigned int state = (unsigned int)m_deterministicSeed;
        for (size_t i = 0; i < m_hRandoms.size(); ++i)
        {
            state = state * 1664525u + 1013904223u;
            int k = (int)((state >> 8) % 2000u);
            m_hRandoms[i] = 0.000005f * (k - 1000);
        }
    }
    else
    {
        srand(static_cast<int>(time(0)));
        const size_t size = (size_t)m_sceneInfo.size.x * m_sceneInfo.size.y;
        for (size_t i = 0; i < size && i < m_hRandoms.size(); ++i)
            m_hRandoms[i] = 0.000005f * (rand() % 2000 - 1000);
    }
}

/* reference: GPUKernel.cpp:2712-2727 */
void GPUKernel::render_end(BitmapBuffer *image)
{
    render_end();
    if (lastError() != 0)
        return; /* (nothing was rendered: the caller's array is left alone, SolR_RunKernel returns -1) */
    const BitmapBuffer *bitmap = getBitmap();
    if (image && bitmap)
        memcpy(image, bitmap, (size_t)m_sceneInfo.size.x * (size_t)m_sceneInfo.size.y * (size_t)SOLR_COLOR_DEPTH);
}

void GPUKernel::render_begin(const float)
{
    if (m_deterministicSeed < 0)
        m_sceneInfo.timestamp = rand() % 10000;
    const bool periodic = (m_deterministicSeed < 0) && (m_sceneInfo.pathTracingIteration % 50 == 1);
    if (!m_randomsTransfered || periodic || m_hRandoms.size() != randomsNeeded() /* the frame changed size */)
    {
        m_randomsTransfered = false;
        fillRandoms();
    }
    size_t pixels = (size_t)m_sceneInfo.size.x * m_sceneInfo.size.y;
    if (m_hPrimitivesXYIds.size() < pixels)
        m_hPrimitivesXYIds.resize(pixels, make_vec4i());
    if (m_bitmap.size() < pixels * SOLR_COLOR_DEPTH)
        m_bitmap.resize(pixels * SOLR_COLOR_DEPTH, 0);
}
}

/*
 * solr_scene.hip - the resident scene of the MI355X rendering engine: what h2d_scene / h2d_materials / h2d_textures /
 * h2d_randoms / h2d_lightInformation (CudaRayTracer.cu:1536-1625) leave on the device, and the lists the walks take.
 *   - the arena's device kernels: leaf records, thin copies of plain-plane leaves, the enclosure check, rotation and refit
 *     of animated scenes;
 *   - the host side of the uploads: row conversion (scene_layout.h), material tags, texture tables, the walk-order list
 *     (chains collapsed, siblings grouped, inner nodes that hardly cull pruned) and the order-free lists, built on the
 *     device (solr_lists.hip) when they are due;
 *   - makeScene(): the SceneArgs a launch is handed.
 * Part of the engine's host side (engine.h); the boundary that calls into it is solr_hip.hip.  gfx950 only.
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <functional>
#include <chrono>
#include <vector>

#include "../../include/solr_hip.h"
#include "rt_device.h"
#include "device_pool.h"
#include "lists_device.h"

using namespace solrdev;

#include "renderer.h"
#include "engine.h"

using namespace solreng;

namespace solreng
{
/* ---- animated scenes: rotate + refit on the device ---------------------------------------------------
 * The reference animates a scene by GPUKernel::rotatePrimitives + compactBoxes(false) on the host and a
 * full upload, every frame (MoleculeScene.cpp:75-81; GPUKernel.cpp:1378-1460 rotates the primitives of
 * the level-0 boxes and refits every level, :1151-1281 flattens again).  The flattened tree keeps its
 * shape under that - only primitive coordinates and node bounds change - so the same arithmetic runs
 * here on the resident arena instead: the primitive rows in place, then the nodes bottom-up.  Every
 * expression below is the host builder's (sol-r_amd/host/GPUKernel.cpp rotateVector, updateBoundingBox,
 * updateOutterBoundingBox), in its order and with its comparisons, so that the arena afterwards holds
 * bit for bit what a host rotation followed by a fresh upload would have put there. */
struct RotationArgs
{
    float cx, cy, cz;
    float cosx, cosy, cosz;
    float sinx, siny, sinz;
};

__device__ inline void rotateRow(float4 &v, float cx, float cy, float cz, const RotationArgs &R)
{
    float vx = v.x - cx, vy = v.y - cy, vz = v.z - cz;
    float ry = vy * R.cosx - vz * R.sinx;
    float rz = vy * R.sinx + vz * R.cosx;
    vy = ry;
    vz = rz;
    rz = vz * R.cosy - vx * R.siny;
    float rx = vz * R.siny + vx * R.cosy;
    vz = rz;
    vx = rx;
    rx = vx * R.cosz - vy * R.sinz;
    ry = vx * R.sinz + vy * R.cosz;
    v.x = rx + cx;
    v.y = ry + cy;
    v.z = rz + cz;
}

/* Leaf records (scene_layout.h): for every leaf of a node list, the first primitive's test data and index in
 * one 64-byte line.  A function of the primitive records and the list's start indices alone: run after every
 * upload of the arena and after every device-side rotation of the primitives. */
__global__ __launch_bounds__(256) void k_buildLeafRecords(float4 *__restrict__ arena, unsigned offNodes, unsigned offStart,
                                                         unsigned offPrims, unsigned offLeaf, int nbNodes)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    const int nb = __float_as_int(arena[offNodes + 2u * (unsigned)i + 1u].z);
    if (nb > 0)
    {
        const int start = ((const int *)arena)[offStart + (unsigned)i];
        const float4 *prim = arena + offPrims + 8u * (unsigned)start;
        r0 = prim[ROW_P0_TYPE];
        r1 = prim[ROW_SIZE_MAT];
        r2 = prim[ROW_P1_INDEX];
        r3 = prim[ROW_P2];
        if (planeClass(__float_as_int(r0.w) & PRIM_TYPE_MASK))
        {
            const float4 n0 = prim[ROW_N0];
            r2 = make_float4(n0.x, n0.y, n0.z, r2.w);
            r3 = make_float4(r3.w, 0.f, 0.f, 0.f);
        }
        r3.w = __int_as_float(start);
    }
    float4 *out = arena + offLeaf + 4u * (unsigned)i;
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
    out[3] = r3;
}

/* The thin copy of a node list (rt_device.h tightRay; scene_layout.h SceneArgs::tightLists): leaf by leaf.  A leaf
 * whose primitives are all plain axis planes becomes the union of their rectangles, `margin` thick and `margin` wider,
 * cut with the reference's box (never larger than it: a ray the thin box lets in, the reference's let in as well);
 * every other node is copied.  k_tightenInner then makes the inner nodes the unions of the leaves below them. */
__global__ __launch_bounds__(256) void k_tightenLeaves(float4 *__restrict__ arena, unsigned offNodes, unsigned offTight,
                                                      unsigned offStart, unsigned offPrims, int nbNodes, float margin)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    float4 row0 = arena[offNodes + 2u * (unsigned)i], row1 = arena[offNodes + 2u * (unsigned)i + 1u];
    const int nb = __float_as_int(row1.z);
    if (nb > 0)
    {
        const int start = ((const int *)arena)[offStart + (unsigned)i];
        float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
        bool plain = true;
        for (int k = 0; k < nb && plain; ++k)
        {
            const float4 *prim = arena + offPrims + 8u * (unsigned)(start + k);
            const float4 p = prim[ROW_P0_TYPE], s = prim[ROW_SIZE_MAT];
            const int kind = (__float_as_int(p.w) >> PRIM_KIND_SHIFT) & 15;
            plain = kind == KIND_PLANE_XY || kind == KIND_PLANE_YZ || kind == KIND_PLANE_XZ;
            /* (a size is compared with a distance: its sign cannot make the rectangle larger than |size|) */
            const float ex = kind == KIND_PLANE_YZ ? margin : fabsf(s.x) + margin;
            const float ey = kind == KIND_PLANE_XZ ? margin : fabsf(s.y) + margin;
            const float ez = kind == KIND_PLANE_XY ? margin : fabsf(s.z) + margin;
            lx = fminf(lx, p.x - ex), hx = fmaxf(hx, p.x + ex);
            ly = fminf(ly, p.y - ey), hy = fmaxf(hy, p.y + ey);
            lz = fminf(lz, p.z - ez), hz = fmaxf(hz, p.z + ez);
        }
        /* (finite, ordered bounds only: anything else keeps the reference's box) */
        plain = plain && lx <= hx && ly <= hy && lz <= hz && fabsf(lx) < 3.0e38f && fabsf(hx) < 3.0e38f && fabsf(ly) < 3.0e38f &&
                fabsf(hy) < 3.0e38f && fabsf(lz) < 3.0e38f && fabsf(hz) < 3.0e38f;
        if (plain)
        {
            const float nlx = fmaxf(row0.x, lx), nly = fmaxf(row0.y, ly), nlz = fmaxf(row0.z, lz);
            const float nhx = fminf(row1.x, hx), nhy = fminf(row1.y, hy), nhz = fminf(row0.w, hz);
            if (nlx <= nhx && nly <= nhy && nlz <= nhz)
            {
                row0 = make_float4(nlx, nly, nlz, nhz);
                row1 = make_float4(nhx, nhy, row1.z, row1.w);
            }
        }
    }
    arena[offTight + 2u * (unsigned)i] = row0;
    arena[offTight + 2u * (unsigned)i + 1u] = row1;
}

/* inner node i of the thin copy: the union of the leaves of its subtree (nodes i + 1 ... i + skip - 1: skip pointers
 * are nested intervals), cut with its own box.  A group still passes whenever one of its members does. */
__global__ __launch_bounds__(256) void k_tightenInner(float4 *__restrict__ arena, unsigned offTight, int nbNodes, int listLength)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    const float4 row0 = arena[offTight + 2u * (unsigned)i], row1 = arena[offTight + 2u * (unsigned)i + 1u];
    const int nb = __float_as_int(row1.z), skip = __float_as_int(row1.w);
    if (nb > 0 || skip <= 1)
        return;
    const int listEnd = (i / listLength + 1) * listLength; /* (several lists one behind the other: stay in this one) */
    const int end = min(i + skip, listEnd);
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
    for (int j = i + 1; j < end; ++j)
    {
        const float4 b = arena[offTight + 2u * (unsigned)j + 1u];
        if (__float_as_int(b.z) <= 0)
            continue;
        const float4 a = arena[offTight + 2u * (unsigned)j];
        lx = fminf(lx, a.x), ly = fminf(ly, a.y), lz = fminf(lz, a.z);
        hx = fmaxf(hx, b.x), hy = fmaxf(hy, b.y), hz = fmaxf(hz, a.w);
    }
    const float nlx = fmaxf(row0.x, lx), nly = fmaxf(row0.y, ly), nlz = fmaxf(row0.z, lz);
    const float nhx = fminf(row1.x, hx), nhy = fminf(row1.y, hy), nhz = fminf(row0.w, hz);
    if (!(nlx <= nhx && nly <= nhy && nlz <= nhz))
        return; /* no leaf below it, or bounds that are not numbers: the reference's box stays */
    arena[offTight + 2u * (unsigned)i] = make_float4(nlx, nly, nlz, nhz);
    arena[offTight + 2u * (unsigned)i + 1u] = make_float4(nhx, nhy, row1.z, row1.w);
}

/* maybeBuildOrderFreeLists' precondition, for the exact list as the arena holds it: every inner node holds its
 * direct children, every leaf its primitives (the same float arithmetic as the host loop there, which stays as the
 * route for an arena that is not laid out).  *bad is raised for a node that does not. */
__global__ __launch_bounds__(256) void k_listEncloses(const float4 *__restrict__ arena, unsigned offNodes, unsigned offStart,
                                                      unsigned offPrims, int nbNodes, int nbPrims, int *bad)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbNodes)
        return;
    const float4 a = arena[offNodes + 2u * (unsigned)i], b = arena[offNodes + 2u * (unsigned)i + 1u];
    const int count = __float_as_int(b.z);
    const int end = min(i + max(__float_as_int(b.w), 1), nbNodes);
    bool encloses = true;
    if (count <= 0)
    {
        for (int j = i + 1; j < end && encloses;)
        {
            const float4 ca = arena[offNodes + 2u * (unsigned)j], cb = arena[offNodes + 2u * (unsigned)j + 1u];
            encloses = ca.x >= a.x && ca.y >= a.y && ca.z >= a.z && cb.x <= b.x && cb.y <= b.y && ca.w <= a.w;
            j += max(__float_as_int(cb.w), 1);
        }
    }
    else
    {
        const int start = ((const int *)arena)[offStart + (unsigned)i];
        for (int k = 0; k < count && encloses; ++k)
        {
            const long long pi = (long long)start + k;
            if (start < 0 || pi >= nbPrims)
            {
                encloses = false;
                break;
            }
            const float4 *r = arena + offPrims + (size_t)PRIM_ROWS * (size_t)pi;
            const float4 p0 = r[ROW_P0_TYPE], size = r[ROW_SIZE_MAT];
            const int type = __float_as_int(p0.w) & PRIM_TYPE_MASK;
            float lo[3] = {p0.x, p0.y, p0.z}, hi[3] = {p0.x, p0.y, p0.z};
            auto add = [&](const float4 &v) {
                lo[0] = v.x < lo[0] ? v.x : lo[0], lo[1] = v.y < lo[1] ? v.y : lo[1], lo[2] = v.z < lo[2] ? v.z : lo[2];
                hi[0] = hi[0] < v.x ? v.x : hi[0], hi[1] = hi[1] < v.y ? v.y : hi[1], hi[2] = hi[2] < v.z ? v.z : hi[2];
            };
            float grow[3] = {size.x, size.y, size.z};
            if (type == ptTriangle)
            {
                add(r[ROW_P1_INDEX]);
                add(r[ROW_P2]);
                grow[0] = grow[1] = grow[2] = 0.f;
            }
            else if (type == ptCylinder)
            {
                add(r[ROW_P1_INDEX]);
                grow[1] = grow[2] = grow[0];
            }
            else if (type == ptSphere)
                grow[1] = grow[2] = grow[0];
            auto larger = [](float x, float y) { return x < y ? y : x; }; /* std::max */
            auto slack = [&](int k) { return 4.f * 1.1920929e-7f * larger(larger(fabsf(lo[k]), fabsf(hi[k])), fabsf(grow[k])); };
            const float ex = slack(0), ey = slack(1), ez = slack(2);
            encloses = a.x <= lo[0] - fabsf(grow[0]) + ex && a.y <= lo[1] - fabsf(grow[1]) + ey && a.z <= lo[2] - fabsf(grow[2]) + ez &&
                       b.x >= hi[0] + fabsf(grow[0]) - ex && b.y >= hi[1] + fabsf(grow[1]) - ey && a.w >= hi[2] + fabsf(grow[2]) - ez;
        }
    }
    if (!encloses)
        *bad = 1;
}

__global__ __launch_bounds__(256) void k_rotatePrimitives(float4 *__restrict__ arena, unsigned offPrims, int nbPrimitives,
                                                          const unsigned char *__restrict__ movable,
                                                          const RotationArgs R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbPrimitives || !movable[i])
        return;
    float4 *r = arena + offPrims + (size_t)PRIM_ROWS * i;
    float4 p0 = r[ROW_P0_TYPE];
    const int type = __float_as_int(p0.w) & PRIM_TYPE_MASK;
    rotateRow(p0, R.cx, R.cy, R.cz, R);
    r[ROW_P0_TYPE] = p0;
    if (type == ptCylinder || type == ptTriangle)
    {
        float4 p1 = r[ROW_P1_INDEX], p2 = r[ROW_P2], n0 = r[ROW_N0], n1 = r[ROW_N1], n2 = r[ROW_N2];
        rotateRow(p1, R.cx, R.cy, R.cz, R);
        rotateRow(p2, R.cx, R.cy, R.cz, R);
        rotateRow(n0, 0.f, 0.f, 0.f, R);
        rotateRow(n1, 0.f, 0.f, 0.f, R);
        rotateRow(n2, 0.f, 0.f, 0.f, R);
        if (type == ptCylinder)
        {
            float ax = p1.x - p0.x, ay = p1.y - p0.y, az = p1.z - p0.z;
            const float len = __builtin_sqrtf(ax * ax + ay * ay + az * az);
            if (len != 0)
            {
                ax /= len;
                ay /= len;
                az /= len;
            }
            n1.x = ax;
            n1.y = ay;
            n1.z = az;
        }
        r[ROW_P1_INDEX] = p1;
        r[ROW_P2] = p2;
        r[ROW_N0] = n0;
        r[ROW_N1] = n1;
        r[ROW_N2] = n2;
    }
}

/* One node per thread, the nodes of one height of the tree per launch (children first).  A node with
 * primitives is a level-0 box: updateBoundingBox; one without is the union of its children:
 * updateOutterBoundingBox, seeded like it (+-viewDistance; +-infinity for our own grouping nodes, which
 * the list marks with the sign bit). */
__global__ __launch_bounds__(256) void k_refitNodes(float4 *__restrict__ arena, unsigned offNodes, unsigned offStart,
                                                    unsigned offPrims, const int *__restrict__ list, int count,
                                                    float seed)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count)
        return;
    const int entry = list[t];
    const int node = entry & 0x7fffffff;
    if (entry < 0)
        seed = INFINITY; /* one of our own grouping nodes: the plain union */
    float4 *rows = arena + offNodes;
    const float4 row1 = rows[2 * node + 1];
    const int nb = __float_as_int(row1.z);
    const int skip = __float_as_int(row1.w);
    float lx, ly, lz, hx, hy, hz;
    if (nb > 0)
    {
        const int first = ((const int *)arena)[offStart + node];
        lx = ly = lz = 1000000.f;
        hx = hy = hz = -1000000.f;
        for (int k = 0; k < nb; ++k)
        {
            const float4 *r = arena + offPrims + (size_t)PRIM_ROWS * (first + k);
            const float4 p0 = r[ROW_P0_TYPE];
            const float4 size = r[ROW_SIZE_MAT];
            const int type = __float_as_int(p0.w) & PRIM_TYPE_MASK;
            /* std::min(a, b) is (b < a) ? b : a and std::max(a, b) is (a < b) ? b : a: kept as such, the
             * sign of a zero that ties depends on it */
            float c0x = p0.x, c0y = p0.y, c0z = p0.z, c1x = p0.x, c1y = p0.y, c1z = p0.z;
            if (type == ptTriangle || type == ptCylinder)
            {
                const float4 p1 = r[ROW_P1_INDEX];
                c0x = (p1.x < p0.x) ? p1.x : p0.x;
                c0y = (p1.y < p0.y) ? p1.y : p0.y;
                c0z = (p1.z < p0.z) ? p1.z : p0.z;
                c1x = (p0.x < p1.x) ? p1.x : p0.x;
                c1y = (p0.y < p1.y) ? p1.y : p0.y;
                c1z = (p0.z < p1.z) ? p1.z : p0.z;
                if (type == ptTriangle)
                {
                    const float4 p2 = r[ROW_P2];
                    c0x = (p2.x < c0x) ? p2.x : c0x;
                    c0y = (p2.y < c0y) ? p2.y : c0y;
                    c0z = (p2.z < c0z) ? p2.z : c0z;
                    c1x = (c1x < p2.x) ? p2.x : c1x;
                    c1y = (c1y < p2.y) ? p2.y : c1y;
                    c1z = (c1z < p2.z) ? p2.z : c1z;
                }
            }
            float ax = (c1x < c0x) ? c1x : c0x, ay = (c1y < c0y) ? c1y : c0y, az = (c1z < c0z) ? c1z : c0z;
            float bx = (c0x > c1x) ? c0x : c1x, by = (c0y > c1y) ? c0y : c1y, bz = (c0z > c1z) ? c0z : c1z;
            const bool round = type == ptCylinder || type == ptSphere || type == ptCone;
            const float sy = round ? size.x : size.y, sz = round ? size.x : size.z;
            ax -= size.x;
            ay -= sy;
            az -= sz;
            bx += size.x;
            by += sy;
            bz += sz;
            if (ax < lx) lx = ax;
            if (ay < ly) ly = ay;
            if (az < lz) lz = az;
            if (bx > hx) hx = bx;
            if (by > hy) hy = by;
            if (bz > hz) hz = bz;
        }
    }
    else
    {
        lx = ly = lz = seed;
        hx = hy = hz = -seed;
        for (int c = node + 1; c < node + skip;)
        {
            const float4 a = rows[2 * c], b = rows[2 * c + 1];
            if (lx > a.x) lx = a.x;
            if (ly > a.y) ly = a.y;
            if (lz > a.z) lz = a.z;
            if (hx < b.x) hx = b.x;
            if (hy < b.y) hy = b.y;
            if (hz < a.w) hz = a.w;
            const int s = __float_as_int(b.w);
            c += (s > 1) ? s : 1;
        }
    }
    rows[2 * node] = make_float4(lx, ly, lz, hz);
    rows[2 * node + 1] = make_float4(hx, hy, row1.z, row1.w);
}

/* skip pointers must describe nested intervals for the ballot-only walk */
int validateNesting(const BoundingBox *boxes, int n)
{
    std::vector<int> ends;
    for (int i = 0; i < n; ++i)
    {
        const int skip = boxes[i].indexForNextBox.x;
        if (skip < 1 || (long)i + skip > n)
            return 0;
        while (!ends.empty() && ends.back() <= i)
            ends.pop_back();
        const int end = i + skip;
        if (!ends.empty() && end > ends.back())
            return 0;
        ends.push_back(end);
    }
    return 1;
}

/* join the material facts the walks need into every primitive's tag (scene_layout.h) */
int materialTag(const Material &m)
{
    int tag = 0;
    if (m.attributes.x == 0)
        tag |= PRIM_FAST0;
    if (m.attributes.x == 1)
        tag |= PRIM_FAST1;
    if (m.attributes.y != 0)
        tag |= PRIM_PROCEDURAL;
    if (m.transparency != 0.f)
        tag |= PRIM_TRANSPARENT;
    if (m.attributes.z == 1)
        tag |= PRIM_WIRE1;
    if (m.attributes.z == 2)
        tag |= PRIM_WIRE2;
    if (m.innerIllumination.x != 0.f)
        tag |= PRIM_EMISSIVE;
    if (m.textureIds.x != TEXTURE_NONE)
        tag |= PRIM_TEXTURED;
    int w = m.attributes.w;
    w = w < -1 ? -1 : (w > 100 ? 100 : w); /* wireFrameMapping compares X % 100 <= width */
    tag |= (w + 1) << PRIM_WIDTH_SHIFT;
    return tag;
}

/* What solr_hip_rotate_primitives refits and in which order: the nodes of a list by height, children
 * before parents.  A frame walks the walk-order list, so that is the one refitted with every rotation; the
 * reference's own list (box-debug view, census, variant 3, read-back) follows when somebody needs it
 * (refreshExactList) - node bounds are a function of the primitives alone, so late is as good as at once.
 * Both give a node of the reference's tree the same bounds: min / max over the level-0 boxes below it,
 * clamped once or several times by the same +-viewDistance seed, first occurrence winning a tie in either
 * nesting.  Node 0, the light cell, keeps its +-viewDistance (GPUKernel.cpp:1189). */
static void buildRefitPlan(const std::vector<float4> &exact, const std::vector<float4> &walk, const std::vector<int> &origin,
                           const std::vector<float4> &free, const std::vector<int> &freeOrigin)
{
    g.refitReady = false;
    g.exactStale = false;
    g.refitLevels.clear();
    g.refitWalkLevels.clear();
    g.refitFreeLevels.clear();
    if (!g.nested)
        return;
    auto heights = [](const std::vector<float4> &rows, std::vector<int> &height) {
        const int n = (int)(rows.size() / 2);
        height.assign(n, 0);
        /* nested skip pointers: a node's subtree is the nodes after it up to its skip; going backwards
         * every child is finished before its parent reads it */
        std::vector<int> parent(n, -1), stack;
        for (int i = 0; i < n; ++i)
        {
            while (!stack.empty() && i >= stack.back() + std::max(bitsi(rows[2 * stack.back() + 1].w), 1))
                stack.pop_back();
            parent[i] = stack.empty() ? -1 : stack.back();
            stack.push_back(i);
        }
        int top = 0;
        for (int i = n - 1; i >= 0; --i)
        {
            if (parent[i] >= 0)
                height[parent[i]] = std::max(height[parent[i]], height[i] + 1);
            top = std::max(top, height[i]);
        }
        return n ? top + 1 : 0;
    };
    std::vector<int> plan;
    auto byHeight = [&](const std::vector<int> &height, int nbHeights, std::vector<int> &levels, auto entry) {
        std::vector<std::vector<int>> bucket((size_t)nbHeights);
        for (int i = 0; i < (int)height.size(); ++i)
        {
            const long e = entry(i);
            if (e != -1)
                bucket[(size_t)height[i]].push_back((int)e);
        }
        for (const std::vector<int> &b : bucket)
            if (!b.empty())
            {
                levels.push_back((int)plan.size());
                levels.push_back((int)b.size());
                plan.insert(plan.end(), b.begin(), b.end());
            }
    };
    std::vector<int> height;
    int nbHeights = heights(exact, height);
    byHeight(height, nbHeights, g.refitLevels, [](int i) { return i != 0 ? (long)i : -1L; });
    nbHeights = heights(walk, height);
    byHeight(height, nbHeights, g.refitWalkLevels, [&](int j) {
        if (origin[j] == 0)
            return -1L;                                   /* the light cell */
        return origin[j] < 0 ? (long)(j | (int)0x80000000) : (long)j; /* sign bit: a grouping node */
    });
    /* the eight order-free lists, one behind the other: a forest with the same kinds of node (leaves of the
     * reference's tree, unions above them) */
    if (!free.empty())
    {
        nbHeights = heights(free, height);
        byHeight(height, nbHeights, g.refitFreeLevels, [&](int j) {
            if (freeOrigin[j] == 0)
                return -1L;
            return freeOrigin[j] < 0 ? (long)(j | (int)0x80000000) : (long)j;
        });
    }
    if (plan.empty())
        plan.push_back(0);
    upload(g.refitPlan, plan);
    g.refitReady = ok();
}

static void refitList(const std::vector<int> &levels, unsigned offNodes, unsigned offStart, float viewDistance)
{
    float4 *arena = (float4 *)g.geometry.ptr;
    const int *plan = (const int *)g.refitPlan.ptr;
    for (size_t l = 0; l + 1 < levels.size(); l += 2)
        hipLaunchKernelGGL(k_refitNodes, dim3((unsigned)((levels[l + 1] + 255) / 256)), dim3(256), 0, g.stream, arena,
                           offNodes, offStart, g.offPrims, plan + levels[l], levels[l + 1], viewDistance);
}

/* the reference's node list is wanted: refit it from the primitives as they are now */
void refreshExactList()
{
    if (!g.exactStale || !g.geometry.ptr)
        return;
    quiesce();
    refitList(g.refitLevels, g.offBoxes, g.offBoxStart, g.exactStaleViewDistance);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(g.stream));
    g.exactStale = false;
}

/* the buffers the device builder left its lists in: rows and start indices until they are in the arena, the origins
 * (only the refit plan reads them) until the host has them or the lists go */
void dropFreeStage(bool originToo)
{
    if (g.freeStage.rows)
        (void)hipFree(g.freeStage.rows);
    if (g.freeStage.start)
        (void)hipFree(g.freeStage.start);
    g.freeStage.rows = nullptr;
    g.freeStage.start = nullptr;
    if (originToo && g.freeStage.origin)
    {
        (void)hipFree(g.freeStage.origin);
        g.freeStage.origin = nullptr;
    }
}

/* host images of order-free lists that were built on the device: from where they are now */
static void ensureHostFreeLists()
{
    if (g.freeHostValid || !ok())
        return;
    quiesce();
    g.hostBoxesFree.resize(g.freeRows);
    g.hostBoxStartFree.resize(g.freeRows / 2);
    g.hostOriginFree.resize(g.freeRows / 2);
    const bool staged = g.freeStage.rows != nullptr;
    const char *arena = (const char *)g.geometry.ptr;
    if (!staged && !arena)
    {
        setError(-1, "order-free lists neither staged nor in the arena", __FILE__, __LINE__);
        return;
    }
    HIPCHECK(hipMemcpy(g.hostBoxesFree.data(), staged ? (const void *)g.freeStage.rows : arena + (size_t)g.offBoxesFree * 16, g.freeRows * 16,
                       hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(g.hostBoxStartFree.data(), staged ? (const void *)g.freeStage.start : arena + (size_t)g.offBoxStartFree * 4,
                       g.freeRows / 2 * 4, hipMemcpyDeviceToHost));
    if (g.freeStage.origin)
        HIPCHECK(hipMemcpy(g.hostOriginFree.data(), g.freeStage.origin, g.freeRows / 2 * 4, hipMemcpyDeviceToHost));
    if (ok())
    {
        g.freeHostValid = true;
        dropFreeStage(true); /* the next layout takes them from the host images */
    }
}

/* the arena moved on (device-side rotations): bring the host images up to date before anything reads them */
static void pullGeometry()
{
    if (!g.deviceAhead || !g.geometry.ptr)
        return;
    refreshExactList();
    quiesce();
    auto get = [&](unsigned at, void *dst, size_t bytes) {
        if (bytes)
            HIPCHECK(hipMemcpy(dst, (const char *)g.geometry.ptr + (size_t)at * 16, bytes, hipMemcpyDeviceToHost));
    };
    get(g.offBoxes, g.hostBoxes.data(), g.hostBoxes.size() * 16);
    get(g.offBoxesCompact, g.hostBoxesCompact.data(), g.hostBoxesCompact.size() * 16);
    if (g.freeHostValid)
        get(g.offBoxesFree, g.hostBoxesFree.data(), g.hostBoxesFree.size() * 16);
    get(g.offPrims, g.hostPrims.data(), g.hostPrims.size() * 16);
    g.deviceAhead = false;
}

void retagPrimitives()
{
    pullGeometry();
    const size_t n = g.hostPrims.size() / PRIM_ROWS;
    const bool noKinds = getenv("SOLR_HIP_NO_KINDS") != nullptr; /* tests: every primitive through the general tests */
    int features = 0;
    bool contained = true, opaque = true, planes = false;
    float extent = 1.f;
    for (size_t i = 0; i < n; ++i)
    {
        float4 *r = &g.hostPrims[PRIM_ROWS * i];
        for (int row : {(int)ROW_P0_TYPE, (int)ROW_P1_INDEX, (int)ROW_P2})
            for (float c : {r[row].x, r[row].y, r[row].z})
                if (fabsf(c) < 3.0e38f) /* (a comparison with NaN is false: the extent stays a number) */
                    extent = std::max(extent, fabsf(c));
        int tag, mat;
        memcpy(&tag, &r[ROW_P0_TYPE].w, 4);
        memcpy(&mat, &r[ROW_SIZE_MAT].w, 4);
        const int type = tag & PRIM_TYPE_MASK;
        /* a material that was never uploaded reads as all zeros on the device */
        const int facts = (mat >= 0 && (size_t)mat < g.materialTags.size()) ? g.materialTags[mat] : (PRIM_FAST0 | (1 << PRIM_WIDTH_SHIFT));
        int kind = KIND_GENERAL;
        if (noKinds || !(facts & PRIM_FAST0))
            kind = KIND_GENERAL; /* the closest-hit walk lets every lane of the leaf test a primitive with a kind */
        else if (type == ptSphere && !(facts & PRIM_PROCEDURAL))
            kind = KIND_SPHERE;
        else if ((type == ptXYPlane || type == ptYZPlane || type == ptXZPlane) && !(facts & (PRIM_TEXTURED | PRIM_WIRE2)) &&
                 !(type == ptYZPlane && (facts & PRIM_EMISSIVE)))
            kind = type == ptXYPlane ? KIND_PLANE_XY : (type == ptYZPlane ? KIND_PLANE_YZ : KIND_PLANE_XZ);
        else if (type == ptTriangle)
            kind = KIND_TRIANGLE;
        else if (type == ptCylinder || type == ptCone)
            kind = KIND_CYLINDER;
        r[ROW_P0_TYPE].w = bitsf(type | facts | (kind << PRIM_KIND_SHIFT));
        planes = planes || kind == KIND_PLANE_XY || kind == KIND_PLANE_YZ || kind == KIND_PLANE_XZ;
        /* inside the box the reference's builder gives its leaf (GPUKernel.cpp:762-830: the vertices of a triangle,
         * p0 +- radius of a sphere, min / max (p0, p1) +- radius of a cylinder, p0 +- size of a plane; a cone's box
         * is built around p0 alone, a procedural sphere's surface is displaced, the others are not bounded by
         * their size) */
        /* every occluder saturates a shadow (GI:880: intensity 1 x sceneInfo.shadowIntensity) unless it is transparent
         * (GI:881-892 scales and tints) or a textured plane (its texel's alpha is the intensity, GI:553-558) */
        opaque = opaque && !(facts & PRIM_TRANSPARENT) && !(facts & PRIM_TEXTURED) && type != ptCamera;
        contained = contained && (type == ptTriangle || type == ptCylinder || (type == ptSphere && !(facts & PRIM_PROCEDURAL)) ||
                                  type == ptXYPlane || type == ptYZPlane || type == ptXZPlane);
        r[ROW_P2].w = (mat >= 0 && (size_t)mat < g.materialAverage.size()) ? g.materialAverage[mat] : 0.f;
        switch (type)
        {
        case ptSphere:
        case ptEnvironment:
            features |= (facts & PRIM_PROCEDURAL) ? F_PROC : F_SPHERE;
            break;
        case ptCylinder:
        case ptCone:
            features |= F_CYL;
            break;
        case ptEllipsoid:
            features |= F_ELL;
            break;
        case ptTriangle:
            features |= F_TRI;
            break;
        case ptCamera:
            features |= F_PLANE | F_TEX;
            break;
        default:
            features |= F_PLANE;
            break;
        }
        if (facts & PRIM_TEXTURED)
            features |= F_TEX;
    }
    /* |p0| + |size| of the largest primitive, at least: the scale the thin leaves' margin is a 2^-10 of */
    float reach = 0.f;
    for (size_t i = 0; i < n; ++i)
        for (float c : {g.hostPrims[PRIM_ROWS * i + ROW_SIZE_MAT].x, g.hostPrims[PRIM_ROWS * i + ROW_SIZE_MAT].y,
                        g.hostPrims[PRIM_ROWS * i + ROW_SIZE_MAT].z})
            if (fabsf(c) < 3.0e38f)
                reach = std::max(reach, fabsf(c));
    g.sceneExtent = extent + reach;
    g.plainPlanes = planes;
    g.sceneFeatures = features;
    g.primsContained = contained && n > 0;
    g.opaqueShadows = opaque && n > 0;
    g.geometryDirty = true;
}

/* The thin copy of a node list behind it (rows offNodes + 2 n + 2 ...; rt_device.h tightRay): made where the scene has
 * plain axis planes at all and the list is short enough for an inner node's thread to read its whole subtree (the
 * room of a 100 k-triangle model keeps the reference's boxes).  false: there is no copy to walk. */
bool tightenList(unsigned offNodes, unsigned offStart, int nbNodes, int listLength)
{
    static const bool off = getenv("SOLR_HIP_NO_TIGHT_LEAVES") != nullptr;
    if (off || !g.plainPlanes || nbNodes <= 0 || listLength <= 0 || listLength > 65536 || !ok())
        return false;
    float4 *arena = (float4 *)g.geometry.ptr;
    const unsigned offTight = offNodes + 2u * (unsigned)nbNodes + 2u;
    const float margin = g.sceneExtent * (1.f / 1024.f);
    const dim3 grid((unsigned)((nbNodes + 255) / 256));
    hipLaunchKernelGGL(k_tightenLeaves, grid, dim3(256), 0, g.stream, arena, offNodes, offTight, offStart, g.offPrims, nbNodes, margin);
    hipLaunchKernelGGL(k_tightenInner, grid, dim3(256), 0, g.stream, arena, offTight, nbNodes, listLength);
    HIPCHECK(hipGetLastError());
    return ok();
}

/* The eight order-free lists once more, behind their thin copies: every node's two rows with its bounds as (near, far) per
 * axis for the octant the list was flattened for (bit 0: x, 1: y, 2: z negative) - {n.x, n.y, n.z, f.z} {f.x, f.y, count,
 * 32 x skip} (scene_layout.h sortedLists; rt_device.h SOLR_ORDER_SORTED, SOLR_NEXT_BY_BYTES).  Made wherever the lists'
 * bounds change. */
__global__ __launch_bounds__(256) void k_sortNodeBounds(float4 *__restrict__ arena, unsigned offBoxesFree, int nb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned offSorted = offBoxesFree + 32u * (unsigned)nb + 4u;
    if (i > 8 * nb)
        return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a; /* (i == 8 nb: the pad record behind the last list) */
    if (i < 8 * nb)
    {
        const int octant = i / nb;
        a = arena[offBoxesFree + 2u * (unsigned)i];
        b = arena[offBoxesFree + 2u * (unsigned)i + 1u];
        if (octant & 1)
        {
            const float t = a.x;
            a.x = b.x;
            b.x = t;
        }
        if (octant & 2)
        {
            const float t = a.y;
            a.y = b.y;
            b.y = t;
        }
        if (octant & 4)
        {
            const float t = a.z;
            a.z = a.w;
            a.w = t;
        }
        /* the skip word in BYTES: the loop that walks this copy keeps its cursors in bytes (rt_device.h SOLR_NEXT_BY_BYTES) */
        b.w = __int_as_float(__float_as_int(b.w) << 5);
    }
    arena[offSorted + 2u * (unsigned)i] = a;
    arena[offSorted + 2u * (unsigned)i + 1u] = b;
}

static bool sortFreeLists(int nbRows)
{
    static const bool off = getenv("SOLR_HIP_NO_SORTED_LISTS") != nullptr;
    const int nb = nbRows / 16; /* nodes per list: eight lists of two rows a node */
    if (off || nb <= 0 || !ok())
        return false;
    hipLaunchKernelGGL(k_sortNodeBounds, dim3((unsigned)((8 * nb + 1 + 255) / 256)), dim3(256), 0, g.stream,
                       (float4 *)g.geometry.ptr, g.offBoxesFree, nb);
    HIPCHECK(hipGetLastError());
    return ok();
}

/* the leaf records of both node lists from the primitive records as the arena holds them now */
void buildLeafRecords()
{
    if (!ok() || !g.geometry.ptr)
        return;
    float4 *arena = (float4 *)g.geometry.ptr;
    const int n = (int)(g.hostBoxes.size() / 2), nc = (int)(g.hostBoxesCompact.size() / 2);
    if (n > 0)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, arena, g.offBoxes,
                           g.offBoxStart, g.offPrims, g.offLeaf, n);
    if (nc > 0)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, g.stream, arena,
                           g.offBoxesCompact, g.offBoxStartCompact, g.offPrims, g.offLeafCompact, nc);
    const int nf = (int)(g.freeRows / 2);
    if (nf > 0 && !g.freeStale)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, g.stream, arena,
                           g.offBoxesFree, g.offBoxStartFree, g.offPrims, g.offLeafFree, nf);
    HIPCHECK(hipGetLastError());
    /* the thin copies follow the bounds and the primitives they were made from (an upload, a rotation on the device) */
    g.tightCompact = tightenList(g.offBoxesCompact, g.offBoxStartCompact, nc, nc);
    g.tightFree = nf > 0 && !g.freeStale && tightenList(g.offBoxesFree, g.offBoxStartFree, nf, nf / 8);
    g.sortedFree = nf > 0 && !g.freeStale && sortFreeLists((int)g.freeRows);
    HIPCHECK(hipStreamSynchronize(g.stream));
}

/* where the order-free lists go: behind everything else, so that they can be added to an arena that is laid out */
static unsigned layoutFreeLists(unsigned row)
{
    g.offBoxesFree = row;
    /* (one pad record, as behind every node list; then the thin copy and the copy with sorted bounds, padded alike) */
    row += 3u * ((unsigned)g.freeRows + 2u);
    g.offBoxStartFree = row * 4;
    row += (unsigned)((g.freeRows / 2 + 3) / 4);
    row = (row + 3u) & ~3u; /* leaf records: one 64-byte line per node */
    g.offLeafFree = row;
    row += 2u * (unsigned)g.freeRows + 4u;
    return row;
}

/* the lists the device builder has just left (g.freeStage) into an arena that holds everything else already: what
 * is there stays where it is (moved to a larger allocation when this one is too small), nothing is uploaded again */
static void appendFreeLists()
{
    const unsigned end = layoutFreeLists(g.rowsFixed);
    const size_t bytes = (size_t)end * 16, fixedBytes = (size_t)g.rowsFixed * 16;
    PhaseTimer phase;
    if (g.geometry.bytes < bytes)
    {
        DeviceBuffer larger;
        reserve(larger, bytes);
        if (!ok())
            return;
        HIPCHECK(hipMemcpyAsync(larger.ptr, g.geometry.ptr, fixedBytes, hipMemcpyDeviceToDevice, g.stream));
        HIPCHECK(hipStreamSynchronize(g.stream));
        release(g.geometry);
        g.geometry = larger;
    }
    char *arena = (char *)g.geometry.ptr;
    HIPCHECK(hipMemsetAsync(arena + fixedBytes, 0, bytes - fixedBytes, g.stream));
    HIPCHECK(hipMemcpyAsync(arena + (size_t)g.offBoxesFree * 16, g.freeStage.rows, g.freeRows * 16, hipMemcpyDeviceToDevice, g.stream));
    HIPCHECK(hipMemcpyAsync(arena + (size_t)g.offBoxStartFree * 4, g.freeStage.start, g.freeRows / 2 * 4, hipMemcpyDeviceToDevice, g.stream));
    const int nf = (int)(g.freeRows / 2);
    if (ok() && nf > 0)
        hipLaunchKernelGGL(k_buildLeafRecords, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, g.stream, (float4 *)g.geometry.ptr,
                           g.offBoxesFree, g.offBoxStartFree, g.offPrims, g.offLeafFree, nf);
    HIPCHECK(hipGetLastError());
    g.tightFree = nf > 0 && tightenList(g.offBoxesFree, g.offBoxStartFree, nf, nf / 8);
    g.sortedFree = nf > 0 && sortFreeLists((int)g.freeRows);
    HIPCHECK(hipStreamSynchronize(g.stream));
    phase.mark("geometry: lists appended");
    if (ok())
    {
        dropFreeStage(false);
        g.freeDirty = false;
    }
}

/* assemble and upload the geometry arena from its host images (scene_layout.h) */
void flushGeometry()
{
    if (!g.geometryDirty)
    {
        if (g.freeDirty && g.freeStage.rows && g.geometry.ptr && g.rowsFixed > 0)
            appendFreeLists();
        else if (g.freeDirty)
            g.geometryDirty = true; /* (not the case this shortcut is for: everything again) */
        if (!g.geometryDirty)
            return;
    }
    pullGeometry();
    if (!g.freeStage.rows)
        ensureHostFreeLists(); /* laid out again from the host images: the lists too, then */
    auto rowsOfInts = [](size_t n) { return (unsigned)((n + 3) / 4); };
    unsigned row = 0;
    /* each node list is followed by one pad record: the walk requests the record after the node it tests
     * (rt_device.h advanceTidy), after the last node too */
    g.offBoxes = row;
    row += (unsigned)g.hostBoxes.size() + 2u;
    g.offBoxesCompact = row;
    row += 2u * ((unsigned)g.hostBoxesCompact.size() + 2u); /* ... and its thin copy (tightenList), padded alike */
    row = (row + 3u) & ~3u; /* primitive records start on a 64-byte line */
    g.offPrims = row;
    row += (unsigned)g.hostPrims.size();
    g.offLights = row;
    row += (unsigned)g.hostLights.size();
    const unsigned startRow = row;
    row += rowsOfInts(g.hostBoxStart.size());
    const unsigned startRowCompact = row;
    row += rowsOfInts(g.hostBoxStartCompact.size());
    g.offBoxStart = startRow * 4;
    g.offBoxStartCompact = startRowCompact * 4;
    row = (row + 3u) & ~3u; /* leaf records: one 64-byte line per node */
    g.offLeaf = row;
    row += 2u * (unsigned)g.hostBoxes.size() + 4u;
    g.offLeafCompact = row;
    row += 2u * (unsigned)g.hostBoxesCompact.size() + 4u;
    g.rowsFixed = row;
    row = layoutFreeLists(row);
    PhaseTimer phase;
    /* the pieces go straight to their rows of the arena (a staged host copy of the whole arena, zero-filled first,
     * took 10-14 ms for 100 k primitives); pad records and the leaf-record area start as zeros */
    reserve(g.geometry, (size_t)std::max(row, 1u) * 16);
    if (!ok())
        return;
    HIPCHECK(hipMemsetAsync(g.geometry.ptr, 0, (size_t)std::max(row, 1u) * 16, g.stream));
    auto put = [&](unsigned at, const void *src, size_t bytes) {
        if (bytes && ok())
            HIPCHECK(hipMemcpyAsync((char *)g.geometry.ptr + (size_t)at * 16, src, bytes, hipMemcpyHostToDevice, g.stream));
    };
    put(g.offBoxes, g.hostBoxes.data(), g.hostBoxes.size() * 16);
    put(g.offBoxesCompact, g.hostBoxesCompact.data(), g.hostBoxesCompact.size() * 16);
    put(g.offPrims, g.hostPrims.data(), g.hostPrims.size() * 16);
    put(g.offLights, g.hostLights.data(), g.hostLights.size() * 16);
    put(startRow, g.hostBoxStart.data(), g.hostBoxStart.size() * 4);
    put(startRowCompact, g.hostBoxStartCompact.data(), g.hostBoxStartCompact.size() * 4);
    if (g.freeStage.rows && ok())
    {
        HIPCHECK(hipMemcpyAsync((char *)g.geometry.ptr + (size_t)g.offBoxesFree * 16, g.freeStage.rows, g.freeRows * 16,
                                hipMemcpyDeviceToDevice, g.stream));
        HIPCHECK(hipMemcpyAsync((char *)g.geometry.ptr + (size_t)g.offBoxStartFree * 4, g.freeStage.start, g.freeRows / 2 * 4,
                                hipMemcpyDeviceToDevice, g.stream));
    }
    else
    {
        put(g.offBoxesFree, g.hostBoxesFree.data(), g.hostBoxesFree.size() * 16);
        put(g.offBoxStartFree / 4, g.hostBoxStartFree.data(), g.hostBoxStartFree.size() * 4);
    }
    HIPCHECK(hipStreamSynchronize(g.stream)); /* pageable sources: complete for the caller when this returns */
    if (ok())
        dropFreeStage(false);
    phase.mark("geometry: upload");
    buildLeafRecords();
    phase.mark("geometry: leaf records");
    if (ok())
    {
        g.geometryDirty = false;
        g.freeDirty = false;
    }
}

/* the order-free lists exist for the resident scene and every condition of their use holds (rt_device.h closestHitWalk) */
bool orderFreeListsUsable()
{
    return g.nbBoxesFree > 0 && g.freeRows == 16 * (size_t)g.nbBoxesFree && g.primsContained && !g.freeStale &&
           g.nested && g.orderedCompact && g.variant != 6;
}

/* bounce rays on the order-free lists: the API's word, else SOLR_HIP_SHORT_RAY_LISTS=0|1 (experiments), else the engine's
 * own choice for this frame */
bool shortRayListsChoice()
{
    static const int fromEnv = getenv("SOLR_HIP_SHORT_RAY_LISTS") ? atoi(getenv("SOLR_HIP_SHORT_RAY_LISTS")) : -1;
    const int mode = g.shortRayListsMode >= 0 ? g.shortRayListsMode : fromEnv;
    /* The engine's own choice.  Bounce rays on the order-free lists save work in nearly every tile and add some to the
     * few whose lanes have to be walked again (the mesh's horizon tiles: + 13 %).  With frames in flight the next frame
     * fills the chip behind those tiles and the saving is what shows (the mesh delivered 0.368 -> 0.356 ms, a 136-row
     * frame of it 0.239 -> 0.222); one frame at a time is as long as its longest tile and gets longer (0.43 -> 0.48 ms). */
    return mode < 0 ? activeFlights() >= 2 : mode != 0;
}

SceneArgs makeScene(bool exactNodes)
{
    SceneArgs S;
    memset(&S, 0, sizeof(S));
    S.geometry = g.geometry.ptr;
    S.materials = g.materials.ptr;
    S.textures = g.textures.ptr;
    S.randoms = g.randoms.ptr;
    S.offBoxes = exactNodes ? g.offBoxes : g.offBoxesCompact;
    S.offBoxStart = exactNodes ? g.offBoxStart : g.offBoxStartCompact;
    S.offLeaf = exactNodes ? g.offLeaf : g.offLeafCompact;
    S.offPrims = g.offPrims;
    S.offLights = g.offLights;
    S.offMatCold = g.offMatCold;
    S.nbBoxes = exactNodes ? g.nbBoxes : g.nbBoxesCompact;
    S.nbPrimitives = g.nbPrimitives;
    S.nbLights = g.nbLights;
    S.nbLamps = g.nbLamps;
    S.nested = g.nested;
    S.orderedBoxes = exactNodes ? g.orderedExact : g.orderedCompact;
    S.nbRandoms = g.randoms.ptr ? g.nbRandoms : 0;
    if (!exactNodes && orderFreeListsUsable())
    {
        S.offBoxesFree = g.offBoxesFree;
        S.offLeafFree = g.offLeafFree;
        S.nbBoxesFree = g.nbBoxesFree; /* per list; the eight lists and their leaf records lie one behind the other */
        S.opaqueShadows = g.opaqueShadows ? 1 : 0;
        S.shortRayLists = shortRayListsChoice() ? 1 : 0;
    }
    /* the thin copies behind the lists this frame walks (set by tightListsFor: they also depend on the frame) */
    S.tightLists = 0;
    /* ... and the copies with sorted bounds behind those (variant 12: the walks take the lists as they are) */
    S.sortedLists = (S.nbBoxesFree > 0 && g.sortedFree && g.variant != 12) ? 1 : 0;
    return S;
}

/* may the walks of a frame with this SceneInfo take the thin copies of the lists S names (rt_device.h tightRay)? */
int tightListsFor(const SceneArgs &S, const SceneInfo &sceneInfo, bool exactNodes)
{
    if (exactNodes || g.variant == 8 || !g.tightCompact || !sceneInfo.extendedGeometry)
        return 0;
    if (S.nbBoxesFree > 0 && !g.tightFree)
        return 0;
    return (sceneInfo.viewDistance > 0.f && sceneInfo.viewDistance <= 64.f * g.sceneExtent) ? 1 : 0;
}

/* The texel fetch (rt_device.h fetchTexel, skyboxMapping) indexes the atlas with textureOffset + index % texels
 * and reads three bytes, for the diffuse map and, at the same index, for every secondary map of the
 * material.  The reference reads whatever lies there when the tables and the atlas disagree; on this
 * device that is a memory fault which ends the process's use of the GPU.  So the tables are checked against
 * the atlas once after either was uploaded, and a frame with a material that points outside is refused. */
void checkTextureTables()
{
    if (g.textureTablesChecked)
        return;
    g.textureTablesChecked = true;
    for (const Engine::TextureUse &use : g.textureUses)
    {
        ARGCHECK(use.texels > 0, "cudaRender: a textured material with an empty or negative texture mapping");
        ARGCHECK(g.textures.ptr != nullptr && g.atlasBytes > 0,
                 "cudaRender: textured materials but no texture atlas was uploaded (h2d_textures)");
        for (int t = 0; ok() && t < 7; ++t)
            if (use.offsets[t] >= 0 || t == 0)
                ARGCHECK(use.offsets[t] >= 0 && (size_t)(use.offsets[t] + use.texels + 2) <= g.atlasBytes,
                         "cudaRender: a material's texture table points outside the uploaded atlas");
        if (!ok())
        {
            g.textureTablesChecked = false; /* checked again once the caller has uploaded something else */
            return;
        }
    }
}

/* (defined with the list builders further down) */
void maybeBuildOrderFreeLists();

/* exchangeDepthHalo, agreedHaloRows, haveCommunicator: solr_rccl.hip (engine.h) */

/* Inner nodes that hardly ever cull are left out of the walk list.  An inner node - one of the reference's tree
 * whose children all lie inside it, or a grouping node, which is the union of its members - passes whenever one
 * of its children would (the argument of groupSiblings below, read the other way: slab values are monotonic in
 * the bounds, the cut-off only shrinks along a walk), so testing the children without it reaches the same
 * leaves in the same order.  What the node buys is the tests of its subtree for the rays that miss it; what it
 * costs is one test for those that do not.  A ray that is in the parent enters the node
 *   - because it starts there: the rays of a frame start on the geometry (and at the camera, which is in the
 *     room it looks at), so about the share of the parent's leaves whose centre lies in the node;
 *   - otherwise with the surface-area probability area(node) / area(parent).
 * The node stays if (1 - the larger of the two) x (nodes below it) is at least `threshold` tests.  Cornell's
 * upper cells and the groups around its walls hold every leaf centre of the room: they go, the groups of small
 * spheres on the floor stay.  Works on the walk-order rows in place; returns the new node count. */
static int pruneInnerNodes(std::vector<float4> &rows, std::vector<int> &start, std::vector<int> &origin, int *nbPruned,
                           bool everyInnerNode = false)
{
    const int n = (int)start.size();
    const double threshold = everyInnerNode ? 1e300 : (getenv("SOLR_HIP_PRUNE") ? atof(getenv("SOLR_HIP_PRUNE")) : 1.0);
    *nbPruned = 0;
    if (n < 2 || !(threshold > 0.0))
        return n;
    auto skipOf = [&](int i) { return std::max(bitsi(rows[2 * i + 1].w), 1); };
    auto countOf = [&](int i) { return bitsi(rows[2 * i + 1].z); };
    auto lo = [&](int i, int k) { return k == 0 ? rows[2 * i].x : (k == 1 ? rows[2 * i].y : rows[2 * i].z); };
    auto hi = [&](int i, int k) { return k == 0 ? rows[2 * i + 1].x : (k == 1 ? rows[2 * i + 1].y : rows[2 * i].w); };
    auto areaOf = [&](int i) {
        const double x = (double)hi(i, 0) - lo(i, 0), y = (double)hi(i, 1) - lo(i, 1), z = (double)hi(i, 2) - lo(i, 2);
        return x * y + y * z + z * x;
    };
    std::vector<char> keep(n, 1);
    /* the decisions: on the device (solr_lists.hip, one launch per depth of the list; the same arithmetic, the same
     * decisions) unless told otherwise or declined */
    int decided = -1;
    if (!everyInnerNode && g.initialized && !getenv("SOLR_HIP_LISTS_ON_HOST"))
        decided = solrPruneDecisionsOnDevice(rows.data(), n, threshold, keep, g.stream);
    if (decided >= 0)
        *nbPruned = decided;
    else
    {
        keep.assign(n, 1);
        std::vector<int> leaves; /* node indices of the leaves, in walk order */
        std::vector<int> leavesBefore(n + 1, 0);
        for (int i = 0; i < n; ++i)
        {
            leavesBefore[i + 1] = leavesBefore[i] + (countOf(i) > 0 ? 1 : 0);
            if (countOf(i) > 0)
                leaves.push_back(i);
        }
        struct Open
        {
            int node, end;
        };
        std::vector<Open> open; /* kept ancestors of node i */
        double sceneLo[3] = {1e300, 1e300, 1e300}, sceneHi[3] = {-1e300, -1e300, -1e300};
        for (int j = 0; j < n; j += skipOf(j))
            for (int k = 0; k < 3; ++k)
            {
                sceneLo[k] = std::min(sceneLo[k], (double)lo(j, k));
                sceneHi[k] = std::max(sceneHi[k], (double)hi(j, k));
            }
        const double sceneArea = (sceneHi[0] - sceneLo[0]) * (sceneHi[1] - sceneLo[1]) + (sceneHi[1] - sceneLo[1]) * (sceneHi[2] - sceneLo[2]) +
                                 (sceneHi[2] - sceneLo[2]) * (sceneHi[0] - sceneLo[0]);
        for (int i = 0; i < n; ++i)
        {
            while (!open.empty() && open.back().end <= i)
                open.pop_back();
            const int end = std::min(i + skipOf(i), n);
            if (countOf(i) == 0 && end > i + 1)
            {
                const int parentFrom = open.empty() ? 0 : open.back().node, parentTo = open.empty() ? n : open.back().end;
                const double parentArea = open.empty() ? sceneArea : areaOf(open.back().node);
                const double bySurface = parentArea > 0.0 ? std::min(1.0, areaOf(i) / parentArea) : 1.0;
                /* share of the parent's leaves whose centre lies in the node (sampled beyond 4096 leaves) */
                const int firstLeaf = leavesBefore[parentFrom], lastLeaf = leavesBefore[parentTo];
                const int stride = std::max(1, (lastLeaf - firstLeaf) / 4096);
                int sampled = 0, inside = 0;
                for (int q = firstLeaf; q < lastLeaf; q += stride)
                {
                    const int leaf = leaves[q];
                    bool in = true;
                    for (int k = 0; k < 3 && in; ++k)
                    {
                        const double c = 0.5 * ((double)lo(leaf, k) + hi(leaf, k));
                        in = c >= lo(i, k) && c <= hi(i, k);
                    }
                    ++sampled;
                    inside += in ? 1 : 0;
                }
                const double byOrigin = sampled ? (double)inside / sampled : 1.0;
                bool encloses = true; /* every child within the node: what the argument above rests on */
                for (int j = i + 1; j < end && encloses; j += skipOf(j))
                    for (int k = 0; k < 3; ++k)
                        encloses = encloses && lo(j, k) >= lo(i, k) && hi(j, k) <= hi(i, k);
                if (encloses && (1.0 - std::max(bySurface, byOrigin)) * (end - i - 1) < threshold)
                {
                    keep[i] = 0;
                    ++*nbPruned;
                    continue;
                }
            }
            open.push_back({i, end});
        }
    }
    if (*nbPruned == 0)
        return n;
    std::vector<int> newIndex(n + 1, 0);
    for (int i = 0; i < n; ++i)
        newIndex[i + 1] = newIndex[i] + (keep[i] ? 1 : 0);
    const int m = newIndex[n];
    std::vector<float4> outRows(2 * (size_t)m);
    std::vector<int> outStart(m), outOrigin(m);
    for (int i = 0; i < n; ++i)
        if (keep[i])
        {
            const int j = newIndex[i];
            const int end = std::min(i + skipOf(i), n);
            outRows[2 * j] = rows[2 * i];
            outRows[2 * j + 1] = rows[2 * i + 1];
            outRows[2 * j + 1].w = bitsf(newIndex[end] - j);
            outStart[j] = start[i];
            outOrigin[j] = origin[i];
        }
    rows.swap(outRows);
    start.swap(outStart);
    origin.swap(outOrigin);
    return m;
}

/* The order-free lists: the leaves of the scene - every node with primitives, whatever the reference put above
 * it - under a binary surface-area hierarchy of our own (binned SAH over the leaf boxes' centres, sixteen bins),
 * flattened depth-first with skip pointers like the other lists, EIGHT TIMES: once per sign octant of a ray's
 * direction, the child on the near side of each split first.  Closest-hit walks whose result does not depend on
 * the order of the leaves (rt_device.h closestHitWalk: rays longer than 2, ties to the smaller flattened index)
 * walk the list of their octant instead of the reference's order - near boxes first, so that the first hits
 * shrink the cut-off and the far side of the scene is culled, which no fixed order can do for every direction.
 * Any of the eight is correct for any ray; the choice is only speed.  Inner nodes that hardly cull are left
 * out as in the other lists (decided once, on the first flattening).  Valid only when every primitive lies
 * inside its leaf's box and every inner node of the reference's list encloses its children (the caller checks
 * both).  `rows` / `start`: a nested list.  Output: 8 x count nodes, list after list. */
static int buildFreeOrderLists(const std::vector<float4> &rows, const std::vector<int> &start, const std::vector<int> &origin,
                               std::vector<float4> &outRows, std::vector<int> &outStart, std::vector<int> &outOrigin,
                               int *nbPruned)
{
    struct Leaf
    {
        float lo[3], hi[3];
        int node;
    };
    struct TreeNode
    {
        float lo[3], hi[3];
        int left, right, axis, leaf; /* leaf: node of the input list, -1 for an inner node */
        int depth;
        bool keep;
    };
    const int n = (int)start.size();
    std::vector<Leaf> leaves;
    for (int i = 0; i < n; ++i)
        if (bitsi(rows[2 * i + 1].z) > 0)
        {
            Leaf l;
            l.lo[0] = rows[2 * i].x, l.lo[1] = rows[2 * i].y, l.lo[2] = rows[2 * i].z;
            l.hi[0] = rows[2 * i + 1].x, l.hi[1] = rows[2 * i + 1].y, l.hi[2] = rows[2 * i].w;
            l.node = i;
            leaves.push_back(l);
        }
    outRows.clear();
    outStart.clear();
    outOrigin.clear();
    *nbPruned = 0;
    if (leaves.size() < 2)
        return 0;
    auto area = [](const float *lo, const float *hi) {
        const double x = (double)hi[0] - lo[0], y = (double)hi[1] - lo[1], z = (double)hi[2] - lo[2];
        return x * y + y * z + z * x;
    };
    std::vector<TreeNode> tree;
    tree.reserve(2 * leaves.size());
    struct Range
    {
        int from, to, node;
    };
    std::vector<Range> todo;
    tree.push_back(TreeNode());
    tree[0].depth = 0;
    todo.push_back({0, (int)leaves.size(), 0});
    while (!todo.empty())
    {
        const Range r = todo.back();
        todo.pop_back();
        const int count = r.to - r.from;
        TreeNode t;
        t.left = t.right = -1;
        t.axis = 0;
        t.leaf = -1;
        t.depth = tree[r.node].depth;
        t.keep = true;
        if (count == 1)
        {
            for (int k = 0; k < 3; ++k)
                t.lo[k] = leaves[r.from].lo[k], t.hi[k] = leaves[r.from].hi[k];
            t.leaf = leaves[r.from].node;
            tree[r.node] = t;
            continue;
        }
        float clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
        for (int k = 0; k < 3; ++k)
            t.lo[k] = 1e30f, t.hi[k] = -1e30f;
        for (int q = r.from; q < r.to; ++q)
            for (int k = 0; k < 3; ++k)
            {
                t.lo[k] = std::min(t.lo[k], leaves[q].lo[k]);
                t.hi[k] = std::max(t.hi[k], leaves[q].hi[k]);
                const float c = 0.5f * (leaves[q].lo[k] + leaves[q].hi[k]);
                clo[k] = std::min(clo[k], c);
                chi[k] = std::max(chi[k], c);
            }
        /* zeros are +0 (std::min keeps whichever zero it met first; the device builder of solr_lists.hip, whose
         * minima are atomics, could not tell which that was) */
        for (int k = 0; k < 3; ++k)
            t.lo[k] += 0.f, t.hi[k] += 0.f;
        /* binned surface-area split: one pass over the leaves fills the bins of all three axes */
        const int BINS = 16;
        int bestAxis = -1, bestBin = 0;
        double bestCost = 1e300;
        {
            int counts[3][BINS];
            float blo[3][BINS][3], bhi[3][BINS][3];
            float scale[3];
            for (int axis = 0; axis < 3; ++axis)
            {
                const float extent = chi[axis] - clo[axis];
                scale[axis] = extent > 0.f ? BINS / extent : 0.f;
                for (int b = 0; b < BINS; ++b)
                {
                    counts[axis][b] = 0;
                    for (int k = 0; k < 3; ++k)
                        blo[axis][b][k] = 1e30f, bhi[axis][b][k] = -1e30f;
                }
            }
            for (int q = r.from; q < r.to; ++q)
            {
                const Leaf &l = leaves[q];
                for (int axis = 0; axis < 3; ++axis)
                {
                    if (!(scale[axis] > 0.f))
                        continue;
                    const float c = 0.5f * (l.lo[axis] + l.hi[axis]);
                    const int b = std::min(BINS - 1, std::max(0, (int)((c - clo[axis]) * scale[axis])));
                    ++counts[axis][b];
                    float *lo3 = blo[axis][b], *hi3 = bhi[axis][b];
                    lo3[0] = std::min(lo3[0], l.lo[0]), lo3[1] = std::min(lo3[1], l.lo[1]), lo3[2] = std::min(lo3[2], l.lo[2]);
                    hi3[0] = std::max(hi3[0], l.hi[0]), hi3[1] = std::max(hi3[1], l.hi[1]), hi3[2] = std::max(hi3[2], l.hi[2]);
                }
            }
            for (int axis = 0; axis < 3; ++axis)
            {
                if (!(scale[axis] > 0.f))
                    continue;
                double rightArea[BINS];
                int rightCount[BINS];
                float rlo[3] = {1e30f, 1e30f, 1e30f}, rhi[3] = {-1e30f, -1e30f, -1e30f};
                int rc = 0;
                for (int b = BINS - 1; b > 0; --b)
                {
                    rc += counts[axis][b];
                    for (int k = 0; k < 3; ++k)
                    {
                        rlo[k] = std::min(rlo[k], blo[axis][b][k]);
                        rhi[k] = std::max(rhi[k], bhi[axis][b][k]);
                    }
                    rightCount[b] = rc;
                    rightArea[b] = rc ? area(rlo, rhi) : 0.0;
                }
                float llo[3] = {1e30f, 1e30f, 1e30f}, lhi[3] = {-1e30f, -1e30f, -1e30f};
                int lc = 0;
                for (int b = 0; b + 1 < BINS; ++b)
                {
                    lc += counts[axis][b];
                    for (int k = 0; k < 3; ++k)
                    {
                        llo[k] = std::min(llo[k], blo[axis][b][k]);
                        lhi[k] = std::max(lhi[k], bhi[axis][b][k]);
                    }
                    if (lc == 0 || rightCount[b + 1] == 0)
                        continue;
                    const double cost = area(llo, lhi) * lc + rightArea[b + 1] * rightCount[b + 1];
                    if (cost < bestCost)
                    {
                        bestCost = cost;
                        bestAxis = axis;
                        bestBin = b;
                    }
                }
            }
        }
        int mid;
        if (bestAxis < 0)
            mid = r.from + count / 2; /* all centres coincide */
        else
        {
            const float scale = BINS / (chi[bestAxis] - clo[bestAxis]);
            const float origin = clo[bestAxis];
            const int axis = bestAxis, bin = bestBin;
            /* stable: the order inside a node stays the order of the leaf list (it decides the halving by position
             * below, and the device builder partitions the same way) */
            mid = (int)(std::stable_partition(leaves.begin() + r.from, leaves.begin() + r.to,
                                       [&](const Leaf &l) {
                                           const float c = 0.5f * (l.lo[axis] + l.hi[axis]);
                                           return std::min(BINS - 1, std::max(0, (int)((c - origin) * scale))) <= bin;
                                       }) -
                        leaves.begin());
            if (mid == r.from || mid == r.to)
                mid = r.from + count / 2;
            t.axis = bestAxis;
        }
        t.left = (int)tree.size(); /* the low side of the split */
        t.right = t.left + 1;
        tree.push_back(TreeNode());
        tree.push_back(TreeNode());
        tree[t.left].depth = tree[t.right].depth = t.depth + 1;
        tree[r.node] = t;
        todo.push_back({r.from, mid, t.left});
        todo.push_back({mid, r.to, t.right});
    }

    /* one flattening: depth-first, the child on the near side of a ray of this octant first */
    auto flatten = [&](int octant, std::vector<float4> &fr, std::vector<int> &fs, std::vector<int> *which,
                       std::vector<int> *from) {
        struct Visit
        {
            int node, slot; /* slot >= 0: close the inner node written at `slot` */
        };
        std::vector<Visit> stack;
        stack.push_back({0, -1});
        while (!stack.empty())
        {
            const Visit v = stack.back();
            stack.pop_back();
            if (v.slot >= 0)
            {
                fr[2 * v.slot + 1].w = bitsf((int)fs.size() - v.slot);
                continue;
            }
            const TreeNode &t = tree[v.node];
            if (t.leaf >= 0)
            {
                fr.push_back(rows[2 * t.leaf]);
                float4 second = rows[2 * t.leaf + 1];
                second.w = bitsf(1);
                fr.push_back(second);
                fs.push_back(start[t.leaf]);
                if (which)
                    which->push_back(v.node);
                if (from)
                    from->push_back(origin[t.leaf]); /* the node of the reference's list this leaf is */
                continue;
            }
            if (t.keep)
            {
                const int slot = (int)fs.size();
                fr.push_back(make_float4(t.lo[0], t.lo[1], t.lo[2], t.hi[2]));
                fr.push_back(make_float4(t.hi[0], t.hi[1], bitsf(0), bitsf(1)));
                fs.push_back(0);
                if (which)
                    which->push_back(v.node);
                if (from)
                    from->push_back(-1);
                stack.push_back({0, slot});
            }
            const bool highFirst = (octant >> t.axis) & 1; /* direction negative along the split axis */
            stack.push_back({highFirst ? t.left : t.right, -1});
            stack.push_back({highFirst ? t.right : t.left, -1}); /* popped first */
        }
    };
    /* which inner nodes stay: decided on the first flattening */
    {
        std::vector<float4> fr;
        std::vector<int> fs, which;
        flatten(0, fr, fs, &which, nullptr);
        std::vector<int> survivors(which);
        pruneInnerNodes(fr, fs, survivors, nbPruned);
        std::vector<char> kept(tree.size(), 0);
        for (int t : survivors)
            kept[t] = 1;
        const int wide = getenv("SOLR_HIP_FREE_WIDE") ? atoi(getenv("SOLR_HIP_FREE_WIDE")) : 0;
        for (size_t t = 0; t < tree.size(); ++t)
            if (tree[t].leaf < 0)
            {
                tree[t].keep = kept[t] != 0;
                if (wide > 1 && tree[t].depth % wide != 0) /* experiment: only every wide-th level keeps its nodes */
                    tree[t].keep = false;
            }
    }
    /* the eight lists: every node's place follows from the sizes of the subtrees before it (children are stored
     * behind their parent in `tree`, so one backward pass gives the sizes); skip pointers are relative, each list
     * is self-contained */
    std::vector<int> size(tree.size(), 0);
    for (int t = (int)tree.size() - 1; t >= 0; --t)
        size[t] = tree[t].leaf >= 0 ? 1 : (tree[t].keep ? 1 : 0) + size[tree[t].left] + size[tree[t].right];
    const int count = size[0];
    outRows.assign(16 * (size_t)count, make_float4(0.f, 0.f, 0.f, 0.f));
    outStart.assign(8 * (size_t)count, 0);
    outOrigin.assign(8 * (size_t)count, -1);
    struct Place
    {
        int node, at;
    };
    std::vector<Place> stack;
    for (int octant = 0; octant < 8; ++octant)
    {
        float4 *fr = outRows.data() + 2 * (size_t)octant * count;
        int *fs = outStart.data() + (size_t)octant * count, *fo = outOrigin.data() + (size_t)octant * count;
        stack.clear();
        stack.push_back({0, 0});
        while (!stack.empty())
        {
            const Place v = stack.back();
            stack.pop_back();
            const TreeNode &t = tree[v.node];
            if (t.leaf >= 0)
            {
                fr[2 * v.at] = rows[2 * t.leaf];
                float4 second = rows[2 * t.leaf + 1];
                second.w = bitsf(1);
                fr[2 * v.at + 1] = second;
                fs[v.at] = start[t.leaf];
                fo[v.at] = origin[t.leaf]; /* the node of the reference's list this leaf is */
                continue;
            }
            int at = v.at;
            if (t.keep)
            {
                fr[2 * at] = make_float4(t.lo[0], t.lo[1], t.lo[2], t.hi[2]);
                fr[2 * at + 1] = make_float4(t.hi[0], t.hi[1], bitsf(0), bitsf(size[v.node]));
                ++at;
            }
            const bool highFirst = (octant >> t.axis) & 1; /* direction negative along the split axis */
            const int first = highFirst ? t.right : t.left, second = highFirst ? t.left : t.right;
            stack.push_back({second, at + size[first]});
            stack.push_back({first, at});
        }
    }
    return count;
}

/* The scene has been rendered `freeCountdown` times since its upload: build the order-free lists now, from the
 * host images of the reference's list and the primitives as they are (brought up to date first if rotations ran
 * on the device), after checking what their use rests on - every inner node encloses its children, every leaf
 * holds its primitives (the reference's builder makes it so, GPUKernel.cpp:741-830; another host's boxes are
 * taken at their word only after this check; the types whose extent is not what the builder adds around p0 -
 * cones, ellipsoids ... - are sorted out by retagPrimitives). */
void maybeBuildOrderFreeLists()
{
    if (g.freeCountdown <= 0 || --g.freeCountdown > 0)
        return;
    if (!g.primsContained)
    {
        g.freeCountdown = 1; /* no walk would take them (orderFreeListsUsable): asked again with the next frame */
        return;
    }
    PhaseTimer phase;
    quiesce();
    pullGeometry();
    if (!ok())
        return;
    phase.mark("order-free: host images");
    const std::vector<float4> &rows = g.hostBoxes;
    const std::vector<int> &start = g.hostBoxStart;
    const int n = (int)start.size();
    if (n < 2 || rows.size() != 2 * (size_t)n || n > 16000000) /* (beyond that the eight lists pass a dozen GB) */
        return;
    auto skipOf = [&](int i) { return std::max(bitsi(rows[2 * i + 1].w), 1); };
    bool encloses = true;
    /* with the arena laid out as the host images are (the usual case: the scene has been rendered once), the checks
     * and the builder read the exact list and the primitive records there */
    const bool fromArena = !g.geometryDirty && g.geometry.ptr != nullptr && !g.exactStale && !g.deviceAhead &&
                           !getenv("SOLR_HIP_LISTS_ON_HOST") && !getenv("SOLR_HIP_LISTS_VIA_HOST") && !getenv("SOLR_HIP_FREE_WIDE");
    const float4 *arena = (const float4 *)g.geometry.ptr;
    if (fromArena)
    {
        HIPCHECK(hipSetDevice(g.device));
        int *bad = nullptr, found = 1;
        HIPCHECK(hipMalloc((void **)&bad, sizeof(int)));
        if (ok())
        {
            HIPCHECK(hipMemsetAsync(bad, 0, sizeof(int), g.stream));
            hipLaunchKernelGGL(k_listEncloses, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g.stream, arena, g.offBoxes, g.offBoxStart,
                               g.offPrims, n, (int)(g.hostPrims.size() / PRIM_ROWS), bad);
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipMemcpyAsync(&found, bad, sizeof(int), hipMemcpyDeviceToHost, g.stream));
            HIPCHECK(hipStreamSynchronize(g.stream));
            (void)hipFree(bad);
        }
        if (!ok())
            return;
        encloses = found == 0;
    }
    for (int i = 0; i < n && encloses && !fromArena; ++i)
    {
        const int end = std::min(i + skipOf(i), n);
        if (bitsi(rows[2 * i + 1].z) > 0 || end <= i + 1)
            continue;
        for (int j = i + 1; j < end && encloses; j += skipOf(j))
            encloses = rows[2 * j].x >= rows[2 * i].x && rows[2 * j].y >= rows[2 * i].y && rows[2 * j].z >= rows[2 * i].z &&
                       rows[2 * j + 1].x <= rows[2 * i + 1].x && rows[2 * j + 1].y <= rows[2 * i + 1].y &&
                       rows[2 * j].w <= rows[2 * i].w;
    }
    const size_t nbPrims = g.hostPrims.size() / PRIM_ROWS;
    for (int i = 0; i < n && encloses && !fromArena; ++i)
    {
        const int count = bitsi(rows[2 * i + 1].z);
        for (int k = 0; k < count && encloses; ++k)
        {
            const size_t pi = (size_t)start[i] + k;
            if (start[i] < 0 || pi >= nbPrims)
            {
                encloses = false;
                break;
            }
            const float4 *r = &g.hostPrims[PRIM_ROWS * pi];
            const int type = bitsi(r[ROW_P0_TYPE].w) & PRIM_TYPE_MASK;
            float lo[3] = {r[ROW_P0_TYPE].x, r[ROW_P0_TYPE].y, r[ROW_P0_TYPE].z};
            float hi[3] = {lo[0], lo[1], lo[2]};
            auto add = [&](const float4 &v) {
                lo[0] = std::min(lo[0], v.x), lo[1] = std::min(lo[1], v.y), lo[2] = std::min(lo[2], v.z);
                hi[0] = std::max(hi[0], v.x), hi[1] = std::max(hi[1], v.y), hi[2] = std::max(hi[2], v.z);
            };
            float grow[3] = {r[ROW_SIZE_MAT].x, r[ROW_SIZE_MAT].y, r[ROW_SIZE_MAT].z};
            if (type == ptTriangle)
            {
                add(r[ROW_P1_INDEX]);
                add(r[ROW_P2]);
                grow[0] = grow[1] = grow[2] = 0.f;
            }
            else if (type == ptCylinder)
            {
                add(r[ROW_P1_INDEX]);
                grow[1] = grow[2] = grow[0];
            }
            else if (type == ptSphere)
                grow[1] = grow[2] = grow[0];
            /* the builder subtracts and adds in another order: four ulps of the coordinates' magnitude of slack, per
             * axis - relative, so that it stays far below the order-free walks' cut-off margin (2e-4 of the distance
             * + 1e-4 of the origin's coordinates, rt_device.h) whatever the scale of the scene */
            auto slack = [&](int k) { return 4.f * 1.1920929e-7f * std::max(std::max(fabsf(lo[k]), fabsf(hi[k])), fabsf(grow[k])); };
            const float ex = slack(0), ey = slack(1), ez = slack(2);
            encloses = rows[2 * i].x <= lo[0] - fabsf(grow[0]) + ex && rows[2 * i].y <= lo[1] - fabsf(grow[1]) + ey &&
                       rows[2 * i].z <= lo[2] - fabsf(grow[2]) + ez && rows[2 * i + 1].x >= hi[0] + fabsf(grow[0]) - ex &&
                       rows[2 * i + 1].y >= hi[1] + fabsf(grow[1]) - ey && rows[2 * i].w >= hi[2] + fabsf(grow[2]) - ez;
        }
    }
    if (!encloses)
    {
        if (getenv("SOLR_HIP_DEBUG_TREE"))
            fprintf(stderr, "solr_hip: no order-free lists: a node does not hold its children or primitives\n");
        return;
    }
    phase.mark("order-free: checks");
    std::vector<int> origin;
    if (!fromArena)
    {
        origin.resize(n);
        for (int i = 0; i < n; ++i)
            origin[i] = i;
    }
    std::vector<float4> boxesF;
    std::vector<int> startF, originF;
    int prunedFree = 0;
    /* on the device (solr_lists.hip: the same tree level by level, the same lists bit for bit) unless told otherwise
     * or declined */
    int count = -1;
    const bool onHost = getenv("SOLR_HIP_LISTS_ON_HOST") != nullptr || getenv("SOLR_HIP_FREE_WIDE") != nullptr;
    if (!onHost)
    {
        HIPCHECK(hipSetDevice(g.device));
        const double threshold = getenv("SOLR_HIP_PRUNE") ? atof(getenv("SOLR_HIP_PRUNE")) : 1.0;
        if (ok() && threshold > 0.0)
        {
            dropFreeStage(true);
            if (fromArena)
                count = solrBuildOrderFreeListsOnDevice(arena + g.offBoxes, (const int *)arena + g.offBoxStart, nullptr, n, threshold, boxesF,
                                                        startF, originF, &prunedFree, g.stream, &g.freeStage);
            else
                count = solrBuildOrderFreeListsOnDevice(rows.data(), start.data(), origin.data(), n, threshold, boxesF, startF, originF,
                                                        &prunedFree, g.stream, getenv("SOLR_HIP_LISTS_VIA_HOST") ? nullptr : &g.freeStage);
        }
    }
    const bool stayed = count > 0 && g.freeStage.rows != nullptr;
    if (count < 0)
    {
        if (origin.empty())
        {
            origin.resize(n);
            for (int i = 0; i < n; ++i)
                origin[i] = i;
        }
        count = buildFreeOrderLists(rows, start, origin, boxesF, startF, originF, &prunedFree);
    }
    if (getenv("SOLR_HIP_DEBUG_TREE"))
        fprintf(stderr, "solr_hip: order-free lists: 8 x %d nodes (%d inner nodes that hardly cull left out)\n", count, prunedFree);
    if (count <= 0)
        return;
    phase.mark("order-free: tree, pruning, eight flattenings");
    g.hostBoxesFree.swap(boxesF);
    g.hostBoxStartFree.swap(startF);
    g.nbBoxesFree = count;
    g.freeRows = 16 * (size_t)count;
    g.freeHostValid = !stayed;
    g.freeStale = false;
    g.hostOriginFree.swap(originF);
    g.refitReady = false;
    g.refitPlanPending = true; /* 8-12 ms for 100 k primitives: only scenes that are rotated on the device pay them */
    /* the lists join the arena: added behind what it holds when they are on the device and it is up to date, else
     * laid out and uploaded again */
    if (stayed && fromArena)
        g.freeDirty = true;
    else
        g.geometryDirty = true;
}

/* Grouping nodes.  The reference's grid builder produces wide levels - 31 sibling leaves under the root of
 * the Cornell scene, 134 top-level cells for the 100k-primitive molecule - and a walk tests every sibling
 * of every node it enters.  Here runs of CONSECUTIVE siblings are wrapped in nodes of our own whose bounds
 * are the union of the siblings' bounds (up to four parts per level, split points by the surface-area
 * heuristic, recursively while a part has more than four members; members about as large as their whole
 * run are left out).  No result can change:
 *   - the depth-first order of the original nodes, hence of every primitive test, is untouched (only
 *     consecutive runs are wrapped), so ties and the shadow accumulation resolve as before;
 *   - a walk reaches an original node only through nodes whose tests it passed, and a group passes
 *     whenever one of its members does: the slab values (b - o) * inv are monotonic in b under IEEE
 *     rounding, so the union's near values are <= and its far values >= the member's on every axis, and
 *     the member's three conditions tnear <= tfar, tnear < far, tfar > 0 carry over (for the sign-selected
 *     form with an infinite reciprocal as well: a member can only pass an axis whose slab contains the
 *     origin coordinate, and then so does the union); the closest-distance cut-off a group is tested
 *     with is never smaller than the one its members will see;
 *   - groups hold no primitives and have no side effects.
 * Requires nested skip pointers and ordered finite bounds (checked by the caller).  Rewrites the node
 * rows and the first-primitive plane in place; returns the new node count. */
static int groupSiblings(std::vector<float4> &rows, std::vector<int> &start, std::vector<int> &origin)
{
    const int n = (int)start.size();
    auto skipOf = [&](int i) { return bitsi(rows[2 * i + 1].w); };
    struct Bounds
    {
        float lo[3], hi[3];
    };
    auto boundsOf = [&](int i) {
        Bounds b;
        b.lo[0] = rows[2 * i].x, b.lo[1] = rows[2 * i].y, b.lo[2] = rows[2 * i].z;
        b.hi[0] = rows[2 * i + 1].x, b.hi[1] = rows[2 * i + 1].y, b.hi[2] = rows[2 * i].w;
        return b;
    };
    auto merge = [](Bounds a, const Bounds &b) {
        for (int k = 0; k < 3; ++k)
        {
            a.lo[k] = std::min(a.lo[k], b.lo[k]);
            a.hi[k] = std::max(a.hi[k], b.hi[k]);
        }
        return a;
    };
    auto area = [](const Bounds &b) {
        const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
        return x * y + y * z + z * x;
    };
    std::vector<float4> outRows;
    std::vector<int> outStart, outOrigin; /* origin: the caller's tag of each node, -1 for the nodes made here */
    outRows.reserve(rows.size() + rows.size() / 2);
    outStart.reserve(start.size() + start.size() / 2);
    outOrigin.reserve(start.size() + start.size() / 2);

    /* best split of sib[from, to) into two consecutive parts */
    std::vector<Bounds> suffix;
    auto splitPoint = [&](const std::vector<int> &sib, int from, int to) {
        const int count = to - from;
        suffix.resize((size_t)count);
        Bounds acc = boundsOf(sib[to - 1]);
        suffix[count - 1] = acc;
        for (int k = count - 2; k >= 0; --k)
        {
            acc = merge(acc, boundsOf(sib[from + k]));
            suffix[k] = acc;
        }
        Bounds left = boundsOf(sib[from]);
        double best = 1e300;
        int bestAt = from + count / 2;
        for (int k = 1; k < count; ++k)
        {
            const double cost = area(left) * k + area(suffix[k]) * (count - k);
            if (cost < best)
            {
                best = cost;
                bestAt = from + k;
            }
            left = merge(left, boundsOf(sib[from + k]));
        }
        return bestAt;
    };

    /* tuning knobs (tools/group_sweep.sh); parts[] / next[] below hold at most 2^4 parts */
    const int flatMax = std::max(1, getenv("SOLR_HIP_GROUP_FLAT") ? atoi(getenv("SOLR_HIP_GROUP_FLAT")) : 4);
    struct Emit
    {
        std::function<void(const std::vector<int> &, int, int)> siblings;
        std::function<void(int)> node;
    } emit;
    emit.node = [&](int i) {
        const size_t at = outStart.size();
        outRows.push_back(rows[2 * i]);
        outRows.push_back(rows[2 * i + 1]);
        outStart.push_back(start[i]);
        outOrigin.push_back(origin[i]);
        /* (most inner nodes have a handful of children, which siblings() would emit as they are: no list is made for
         * them - a vector per inner node was two thirds of this function's time for a 100k-primitive scene) */
        const int end = std::min(i + skipOf(i), n);
        int few = 0;
        for (int j = i + 1; j < end && few <= flatMax; j += std::max(skipOf(j), 1))
            ++few;
        if (few > flatMax)
        {
            std::vector<int> children;
            for (int j = i + 1; j < end; j += std::max(skipOf(j), 1))
                children.push_back(j);
            emit.siblings(children, 0, (int)children.size());
        }
        else
            for (int j = i + 1; j < end;)
            {
                const int next = j + std::max(skipOf(j), 1); /* (read before the node is emitted: rows are not touched, but so it stays) */
                emit.node(j);
                j = next;
            }
        outRows[2 * at + 1].w = bitsf((int)(outStart.size() - at));
    };
    /* (a list of a few dozen nodes - the Cornell room - gains 2 % from a third round of splits, lists of
     * thousands lose 7 %: profiles/r2/group_sweep.txt) */
    /* (at least one round: with none a run longer than flatMax would be wrapped in a node around itself, for ever - no
     * grouping at all is solr_hip_set_variant(5)) */
    const int levels = std::min(
        4, std::max(1, getenv("SOLR_HIP_GROUP_LEVELS") ? atoi(getenv("SOLR_HIP_GROUP_LEVELS")) : (n <= 64 ? 3 : 2)));
    emit.siblings = [&](const std::vector<int> &sib, int from, int to) {
        if (to - from <= flatMax)
        {
            for (int k = from; k < to; ++k)
                emit.node(sib[k]);
            return;
        }
        /* a member about as large as the whole run (a wall of the room, the light cell that spans the
         * view distance) would make every group around it as large as itself and never culled: such
         * members stay where they are, ungrouped, and the runs between them are grouped on their own */
        {
            Bounds u = boundsOf(sib[from]);
            for (int k = from + 1; k < to; ++k)
                u = merge(u, boundsOf(sib[k]));
            const double limit = 0.5 * area(u);
            bool dominant = false;
            for (int k = from; k < to && !dominant; ++k)
                dominant = area(boundsOf(sib[k])) > limit;
            if (dominant)
            {
                int runStart = from;
                for (int k = from; k <= to; ++k)
                    if (k == to || area(boundsOf(sib[k])) > limit)
                    {
                        if (k > runStart)
                            emit.siblings(sib, runStart, k);
                        if (k < to)
                            emit.node(sib[k]);
                        runStart = k + 1;
                    }
                return;
            }
        }
        /* `levels` rounds of binary splits without intermediate nodes: up to 2^levels parts */
        int parts[17];
        int np = 1;
        parts[0] = from;
        parts[1] = to;
        for (int level = 0; level < levels; ++level)
        {
            int next[17];
            int nn = 0;
            for (int q = 0; q < np; ++q)
            {
                next[nn++] = parts[q];
                if (parts[q + 1] - parts[q] > 2)
                    next[nn++] = splitPoint(sib, parts[q], parts[q + 1]);
            }
            next[nn] = to;
            np = nn;
            for (int q = 0; q <= np; ++q)
                parts[q] = next[q];
        }
        for (int q = 0; q < np; ++q)
        {
            const int a = parts[q], b = parts[q + 1];
            if (b - a == 1)
            {
                emit.node(sib[a]);
                continue;
            }
            Bounds u = boundsOf(sib[a]);
            for (int k = a + 1; k < b; ++k)
                u = merge(u, boundsOf(sib[k]));
            const size_t at = outStart.size();
            outRows.push_back(make_float4(u.lo[0], u.lo[1], u.lo[2], u.hi[2]));
            outRows.push_back(make_float4(u.hi[0], u.hi[1], bitsf(0), bitsf(1)));
            outStart.push_back(0);
            outOrigin.push_back(-1);
            emit.siblings(sib, a, b);
            outRows[2 * at + 1].w = bitsf((int)(outStart.size() - at));
        }
    };
    std::vector<int> top;
    for (int j = 0; j < n; j += std::max(skipOf(j), 1))
        top.push_back(j);
    emit.siblings(top, 0, (int)top.size());
    rows.swap(outRows);
    start.swap(outStart);
    origin.swap(outOrigin);
    return (int)start.size();
}

void h2dSceneOne(BoundingBox *boundingBoxes, int nbActiveBoxes, Primitive *primitives, int nbPrimitives, Lamp *lamps,
                        int nbLamps)
{
    if (!ready("h2d_scene"))
        return;
    quiesce();
    ARGCHECK(nbActiveBoxes >= 0 && nbPrimitives >= 0 && nbLamps >= 0, "h2d_scene: negative count");
    ARGCHECK(nbActiveBoxes == 0 || boundingBoxes, "h2d_scene: null boxes");
    ARGCHECK(nbPrimitives == 0 || primitives, "h2d_scene: null primitives");
    if (!ok())
        return;
    PhaseTimer phase;
    std::vector<float4> boxes(2 * (size_t)nbActiveBoxes);
    std::vector<int> start(nbActiveBoxes);
    for (int i = 0; i < nbActiveBoxes; ++i)
    {
        const BoundingBox &b = boundingBoxes[i];
        ARGCHECK(b.nbPrimitives >= 0 && (b.nbPrimitives == 0 || (b.startIndex >= 0 &&
                                                                  (long)b.startIndex + b.nbPrimitives <= nbPrimitives)),
                 "h2d_scene: box primitive range outside the primitive array");
        /* node record, scene_layout.h: { min.xyz, max.z } { max.xy, nbPrimitives, skip } */
        boxes[2 * i] = make_float4(b.parameters[0].x, b.parameters[0].y, b.parameters[0].z, b.parameters[1].z);
        boxes[2 * i + 1] =
            make_float4(b.parameters[1].x, b.parameters[1].y, bitsf(b.nbPrimitives), bitsf(b.indexForNextBox.x));
        start[i] = b.startIndex;
    }
    if (!ok())
        return;
    phase.mark("h2d_scene: node rows");
    g.nested = validateNesting(boundingBoxes, nbActiveBoxes);
    phase.mark("h2d_scene: nesting check");
    if (!g.nested)
    {
        /* the general walk needs at least forward progress */
        for (int i = 0; i < nbActiveBoxes; ++i)
            ARGCHECK(boundingBoxes[i].indexForNextBox.x >= 1, "h2d_scene: skip pointer < 1");
        if (!ok())
            return;
    }

    /* Collapsed walk order.  The reference's grid builder wraps most leaves in
     * a chain of inner nodes with bit-identical bounds (one per tree level,
     * GPUKernel.cpp:1008-1035).  A ray that enters the first node of such a
     * chain enters all of them - same slabs, same ray, same minDistance since
     * no primitive is tested in between - and a ray that misses it skips all
     * of them, so dropping every inner node whose only child has the same
     * bounds changes no result.  Skip pointers are recomputed in the compacted
     * numbering and stay nested. */
    /* (one pass: which nodes stay, whether every bound is ordered and finite, the compacted numbering) */
    std::vector<char> keep(nbActiveBoxes, 1);
    std::vector<int> newIndex((size_t)nbActiveBoxes + 1);
    newIndex[0] = 0;
    g.orderedExact = 1;
    g.orderedCompact = 1;
    for (int i = 0; i < nbActiveBoxes; ++i)
    {
        const BoundingBox &a = boundingBoxes[i];
        if (g.nested && a.nbPrimitives == 0)
        {
            if (i + 1 < nbActiveBoxes)
            {
                const BoundingBox &b = boundingBoxes[i + 1];
                if (a.indexForNextBox.x >= 2 && b.indexForNextBox.x == a.indexForNextBox.x - 1 &&
                    memcmp(a.parameters, b.parameters, sizeof(a.parameters)) == 0)
                    keep[i] = 0;
            }
            /* an inner node without emitted children (its cell held only lights or nothing,
             * GPUKernel.cpp:1096) leads nowhere: entering or missing it changes nothing */
            if (a.indexForNextBox.x == 1)
                keep[i] = 0;
        }
        const float *lo = &a.parameters[0].x, *hi = &a.parameters[1].x;
        bool ordered = true;
        for (int k = 0; k < 3; ++k)
            ordered = ordered && (lo[k] <= hi[k]) && (fabsf(lo[k]) < 1.0e30f) && (fabsf(hi[k]) < 1.0e30f);
        if (!ordered)
        {
            g.orderedExact = 0;
            if (keep[i])
                g.orderedCompact = 0;
        }
        newIndex[(size_t)i + 1] = newIndex[i] + (keep[i] ? 1 : 0);
    }
    const int nc = newIndex[nbActiveBoxes];
    std::vector<float4> boxesC(2 * (size_t)nc);
    std::vector<int> startC(nc), originC(nc);
    for (int i = 0; i < nbActiveBoxes; ++i)
        if (keep[i])
        {
            const int j = newIndex[i];
            originC[j] = i;
            const int end = std::min(i + boundingBoxes[i].indexForNextBox.x, nbActiveBoxes);
            boxesC[2 * j] = boxes[2 * i];
            boxesC[2 * j + 1] = boxes[2 * i + 1];
            boxesC[2 * j + 1].w = bitsf(newIndex[end] - j);
            startC[j] = start[i];
        }

    /* the order-free lists are built when the scene has stayed for a frame (maybeBuildOrderFreeLists): a host that
     * uploads the scene again for every frame - the reference's own way of animating - never pays for them */
    std::vector<float4> boxesF;
    std::vector<int> startF, originF;
    const int nbFreeNodes = 0;
    g.freeCountdown = 0;
    if (g.nested && g.orderedCompact && nc > 1 && g.grouping && !getenv("SOLR_HIP_NO_FREE_ORDER"))
        g.freeCountdown = std::max(1, getenv("SOLR_HIP_FREE_AFTER") ? atoi(getenv("SOLR_HIP_FREE_AFTER")) : 2);

    int nbWalkNodes = nc, prunedBefore = 0, prunedAfter = 0;
    if (g.nested && g.orderedCompact && nc > 0 && g.grouping)
    {
        phase.mark("h2d_scene: chain collapse");
        pruneInnerNodes(boxesC, startC, originC, &prunedBefore, getenv("SOLR_HIP_REBUILD") != nullptr); /* cells that do not cull: their children join the run above */
        phase.mark("h2d_scene: prune");
        groupSiblings(boxesC, startC, originC);
        phase.mark("h2d_scene: grouping");
        nbWalkNodes = pruneInnerNodes(boxesC, startC, originC, &prunedAfter); /* groups that do not cull either */
        phase.mark("h2d_scene: prune groups");
    }
    if (getenv("SOLR_HIP_DEBUG_TREE"))
    {
        fprintf(stderr, "solr_hip: %d nodes uploaded, %d after collapsing chains, %d in the walk list (%d + %d inner nodes that hardly cull left out)\n",
                nbActiveBoxes, nc, nbWalkNodes, prunedBefore, prunedAfter);
        if (nbWalkNodes <= 80)
            for (int i = 0; i < nbWalkNodes; ++i)
                fprintf(stderr, "  node %2d: prims %d skip %d  [%g %g %g .. %g %g %g]\n", i, bitsi(boxesC[2 * i + 1].z),
                        bitsi(boxesC[2 * i + 1].w), boxesC[2 * i].x, boxesC[2 * i].y, boxesC[2 * i].z,
                        boxesC[2 * i + 1].x, boxesC[2 * i + 1].y, boxesC[2 * i].w);
    }

    std::vector<float4> prims(8 * (size_t)nbPrimitives);
    for (int i = 0; i < nbPrimitives; ++i)
    {
        const Primitive &p = primitives[i];
        float4 *r = &prims[8 * (size_t)i];
        r[ROW_P0_TYPE] = make_float4(p.p0.x, p.p0.y, p.p0.z, bitsf(p.type & PRIM_TYPE_MASK));
        r[ROW_SIZE_MAT] = make_float4(p.size.x, p.size.y, p.size.z, bitsf(p.materialId));
        r[ROW_P1_INDEX] = make_float4(p.p1.x, p.p1.y, p.p1.z, bitsf(p.index));
        r[ROW_P2] = make_float4(p.p2.x, p.p2.y, p.p2.z, 0.f);
        r[ROW_N0] = make_float4(p.n0.x, p.n0.y, p.n0.z, p.vt0.x);
        r[ROW_N1] = make_float4(p.n1.x, p.n1.y, p.n1.z, p.vt0.y);
        r[ROW_N2] = make_float4(p.n2.x, p.n2.y, p.n2.z, p.vt1.x);
        r[ROW_UV] = make_float4(p.vt1.y, p.vt2.x, p.vt2.y, 0.f);
    }
    if (prims.empty())
        prims.assign(8, make_float4(0.f, 0.f, 0.f, 0.f)); /* inactive lanes read record 0 */
    phase.mark("h2d_scene: primitive rows");
    g.refitReady = false;
    g.exactStale = false;
    g.refitPlanPending = true;
    g.hostOriginFree.swap(originF);
    g.deviceAhead = false;
    g.nbMovable = -1;
    g.hostBoxes.swap(boxes);
    g.hostBoxesCompact.swap(boxesC);
    g.hostBoxStart.swap(start);
    g.hostBoxStartCompact.swap(startC);
    g.hostOriginCompact = originC;
    g.hostBoxesFree.swap(boxesF);
    g.hostBoxStartFree.swap(startF);
    g.freeRows = 0;
    g.freeHostValid = true;
    g.freeDirty = false;
    dropFreeStage(true);
    g.nbBoxesFree = nbFreeNodes;
    g.freeStale = false;
    g.hostPrims.swap(prims);
    retagPrimitives();
    phase.mark("h2d_scene: tags");
    HIPCHECK(hipSetDevice(g.device));
    std::vector<int> l(lamps, lamps + (lamps ? nbLamps : 0));
    upload(g.lamps, l);
    if (ok())
    {
        g.nbBoxes = nbActiveBoxes;
        g.nbBoxesCompact = nbWalkNodes;
        g.nbPrimitives = nbPrimitives;
        g.nbLamps = nbLamps;
    }
}

/* Extension: per flattened primitive, whether GPUKernel::rotatePrimitives would move it (it sits in a
 * level-0 box, is movable and is not the camera primitive).  Valid until the next h2d_scene. */
void setMovableOne(const unsigned char *flags, int nbPrimitives)
{
    if (!ready("solr_hip_set_movable"))
        return;
    ARGCHECK(nbPrimitives >= 0 && (nbPrimitives == 0 || flags), "solr_hip_set_movable: null flags");
    if (!ok())
        return;
    g.nbMovable = -1;
    if (nbPrimitives != g.nbPrimitives)
        return;
    quiesce();
    HIPCHECK(hipSetDevice(g.device));
    std::vector<unsigned char> f(flags, flags + nbPrimitives);
    if (f.empty())
        f.push_back(0);
    upload(g.movable, f);
    if (ok())
        g.nbMovable = nbPrimitives;
}

/* Extension: GPUKernel::rotatePrimitives + compactBoxes(false) + h2d_scene on the resident scene
 * (GPUKernel.cpp:1378-1460, 1151-1281 of the reference), see k_rotatePrimitives.  Returns 1 when the
 * arena now holds the rotated scene, 0 when the request cannot be served here and the caller has to
 * take the host route (nothing was changed). */
/* can this engine rotate its resident scene?  (makes the refit plan when the lists changed; changes nothing else) */
bool canRotateOne(const float center[3], const float cosAngles[3], const float sinAngles[3], float viewDistance)
{
    if (!ready("solr_hip_rotate_primitives") || !ok())
        return false;
    if (g.refitPlanPending)
    {
        /* which nodes to refit, in which order: made for the first rotation after the lists changed */
        g.refitPlanPending = false;
        ensureHostFreeLists();
        buildRefitPlan(g.hostBoxes, g.hostBoxesCompact, g.hostOriginCompact, g.hostBoxesFree, g.hostOriginFree);
    }
    /* the seeds of the two box updates only commute with the unions while viewDistance <= 1e6, and a
     * tree cut off at NB_MAX_BOXES has host-side children the flattened list does not show */
    if (!g.refitReady || g.nbMovable != g.nbPrimitives || g.nbPrimitives <= 0 || !(viewDistance <= 1000000.f) ||
        !(viewDistance > 0.f) || g.nbBoxes >= NB_MAX_BOXES || !center || !cosAngles || !sinAngles)
    {
        if (getenv("SOLR_HIP_DEBUG_TREE"))
            fprintf(stderr, "solr_hip_rotate_primitives refused: plan %d, flags for %d of %d primitives, viewDistance %g, %d nodes\n",
                    (int)g.refitReady, g.nbMovable, g.nbPrimitives, viewDistance, g.nbBoxes);
        return false;
    }
    return true;
}

int rotatePrimitivesOne(const float center[3], const float cosAngles[3], const float sinAngles[3], float viewDistance)
{
    if (!canRotateOne(center, cosAngles, sinAngles, viewDistance))
        return 0;
    HIPCHECK(hipSetDevice(g.device));
    flushGeometry();
    if (!ok())
        return 0;
    quiesce();
    RotationArgs R;
    R.cx = center[0], R.cy = center[1], R.cz = center[2];
    R.cosx = cosAngles[0], R.cosy = cosAngles[1], R.cosz = cosAngles[2];
    R.sinx = sinAngles[0], R.siny = sinAngles[1], R.sinz = sinAngles[2];
    hipLaunchKernelGGL(k_rotatePrimitives, dim3((unsigned)((g.nbPrimitives + 255) / 256)), dim3(256), 0, g.stream,
                       (float4 *)g.geometry.ptr, g.offPrims, g.nbPrimitives, (const unsigned char *)g.movable.ptr, R);
    refitList(g.refitWalkLevels, g.offBoxesCompact, g.offBoxStartCompact, viewDistance);
    if (g.nbBoxesFree > 0 && !g.refitFreeLevels.empty())
        refitList(g.refitFreeLevels, g.offBoxesFree, g.offBoxStartFree, viewDistance);
    else
        g.freeStale = g.nbBoxesFree > 0; /* no plan: rotated scenes walk the reference's order until the next upload */
    buildLeafRecords(); /* the leaves' copies of their first primitive follow the primitives */
    g.exactStale = true;
    g.exactStaleViewDistance = viewDistance;
    HIPCHECK(hipGetLastError());
    /* the other flights' streams start their next frame only after this */
    HIPCHECK(hipStreamSynchronize(g.stream));
    if (!ok())
        return 0;
    g.deviceAhead = true;
    ++g.nbDeviceRotations;
    return 1;
}

void h2dMaterialsOne(Material *materials, int nbActiveMaterials)
{
    if (!ready("h2d_materials"))
        return;
    quiesce();
    ARGCHECK(nbActiveMaterials >= 0 && (nbActiveMaterials == 0 || materials), "h2d_materials: bad arguments");
    if (!ok())
        return;
    /* always NB_MAX_MATERIALS + 1 records on the device (zero beyond the
     * active ones) so that every id a primitive or the box-debug view can
     * produce stays inside the allocation */
    const int capacity = NB_MAX_MATERIALS + 1;
    const int active = std::min(nbActiveMaterials, capacity);
    /* (only the active records are built and copied: the 12.6 MB of the full table took 10 ms per call, and the
     * reference's hosts call this whenever one material changes) */
    std::vector<MaterialHot> hot((size_t)std::max(active, 1));
    std::vector<MaterialCold> cold((size_t)std::max(active, 1));
    memset(hot.data(), 0, hot.size() * sizeof(MaterialHot));
    memset(cold.data(), 0, cold.size() * sizeof(MaterialCold));
    g.materialTags.assign(capacity, PRIM_FAST0 | (1 << PRIM_WIDTH_SHIFT));
    g.materialAverage.assign(capacity, 0.f);
    g.textureUses.clear();
    g.textureTablesChecked = false;
    for (int i = 0; i < nbActiveMaterials && i < capacity; ++i)
    {
        Material m = materials[i];
        /* A diffuse texture id that was never loaded: GPUKernel::setMaterial then leaves the "computed texture"
         * mapping (40000 x 40000 at offset 0, GPUKernel.cpp:1893-1896) next to the id, and the mappers would
         * index gigabytes past the atlas (the reference reads whatever is there; a memory fault here).  Such a
         * material is untextured on the device. */
        if (m.textureIds.x >= 0 && m.textureMapping.x == 40000 && m.textureMapping.y == 40000 && m.textureOffset.x == 0)
            m.textureIds.x = TEXTURE_NONE;
        g.materialTags[i] = materialTag(m);
        /* the mappers fetch only for 0 <= u < mapping.x (rt_device.h): a mapping without columns - what
         * realignTexturesAndMaterials gives a material whose texture nobody loaded - never reaches the atlas */
        if (m.textureIds.x >= 0 && m.textureMapping.x > 0) /* procedural ids (Mandelbrot, Julia) are negative */
        {
            Engine::TextureUse use;
            use.material = i;
            use.texels = (long)m.textureMapping.x * (long)m.textureMapping.y * (long)m.textureMapping.w;
            if (m.textureMapping.y <= 0 || m.textureMapping.w <= 0 || m.textureOffset.x < 0)
                use.texels = 0; /* fetchTexel takes an index modulo this: refused by checkTextureTables */
            const int ids[7] = {m.textureIds.x, m.textureIds.y, m.textureIds.z, m.textureIds.w,
                                m.advancedTextureIds.x, m.advancedTextureIds.y, m.advancedTextureIds.z};
            const int offs[7] = {m.textureOffset.x, m.textureOffset.y, m.textureOffset.z, m.textureOffset.w,
                                 m.advancedTextureOffset.x, m.advancedTextureOffset.y, m.advancedTextureOffset.z};
            for (int t = 0; t < 7; ++t)
                use.offsets[t] = ids[t] != TEXTURE_NONE ? (long)offs[t] : -1L;
            g.textureUses.push_back(use);
        }
        g.materialAverage[i] = (m.color.x + m.color.y + m.color.z) / 3.f; /* same expression, same rounding */
        MaterialHot &h = hot[i];
        h.innerIllumination = make_float4(m.innerIllumination.x, m.innerIllumination.y, m.innerIllumination.z,
                                          m.innerIllumination.w);
        h.color = make_float4(m.color.x, m.color.y, m.color.z, m.color.w);
        h.specular = make_float4(m.specular.x, m.specular.y, m.specular.z, m.specular.w);
        h.reflection = m.reflection;
        h.refraction = m.refraction;
        h.transparency = m.transparency;
        h.opacity = m.opacity;
        h.attributes = make_int4(m.attributes.x, m.attributes.y, m.attributes.z, m.attributes.w);
        h.ids = make_int4(m.textureIds.x, m.advancedTextureIds.z, 0, 0);
        MaterialCold &c = cold[i];
        c.textureMapping = make_int4(m.textureMapping.x, m.textureMapping.y, m.textureMapping.z, m.textureMapping.w);
        c.textureOffset = make_int4(m.textureOffset.x, m.textureOffset.y, m.textureOffset.z, m.textureOffset.w);
        c.textureIds = make_int4(m.textureIds.x, m.textureIds.y, m.textureIds.z, m.textureIds.w);
        c.advancedTextureOffset = make_int4(m.advancedTextureOffset.x, m.advancedTextureOffset.y,
                                            m.advancedTextureOffset.z, m.advancedTextureOffset.w);
        c.advancedTextureIds = make_int4(m.advancedTextureIds.x, m.advancedTextureIds.y, m.advancedTextureIds.z,
                                         m.advancedTextureIds.w);
        c.mappingOffset = make_float2(m.mappingOffset.x, m.mappingOffset.y);
        c.pad = make_float2(0.f, 0.f);
    }
    HIPCHECK(hipSetDevice(g.device));
    const size_t tableBytes = 12 * (size_t)capacity * sizeof(float4);
    const bool fresh = !g.materials.ptr || g.materials.bytes < tableBytes;
    reserve(g.materials, tableBytes);
    if (!ok())
        return;
    char *table = (char *)g.materials.ptr;
    const size_t coldAt = 6 * (size_t)capacity * sizeof(float4);
    /* zeros beyond the active records: the whole table when it is new, else what the last call left behind */
    const int stale = fresh ? capacity : std::min(std::max(g.nbMaterials, 0), capacity);
    if (fresh)
        HIPCHECK(hipMemsetAsync(table, 0, tableBytes, g.stream));
    else if (stale > active)
    {
        HIPCHECK(hipMemsetAsync(table + (size_t)active * sizeof(MaterialHot), 0, (size_t)(stale - active) * sizeof(MaterialHot),
                                g.stream));
        HIPCHECK(hipMemsetAsync(table + coldAt + (size_t)active * sizeof(MaterialCold), 0,
                                (size_t)(stale - active) * sizeof(MaterialCold), g.stream));
    }
    if (active > 0)
    {
        HIPCHECK(hipMemcpyAsync(table, hot.data(), (size_t)active * sizeof(MaterialHot), hipMemcpyHostToDevice, g.stream));
        HIPCHECK(hipMemcpyAsync(table + coldAt, cold.data(), (size_t)active * sizeof(MaterialCold), hipMemcpyHostToDevice,
                                g.stream));
    }
    HIPCHECK(hipStreamSynchronize(g.stream)); /* pageable sources: complete for the caller when this returns */
    if (ok())
    {
        g.offMatCold = 6u * (unsigned)capacity;
        g.nbMaterials = nbActiveMaterials;
        retagPrimitives();
    }
}

static void noteRandomsReach(const std::vector<float> &r)
{
    /* the ambient-occlusion taps read randoms[i] and randoms[i + 100], i < 256 (CRT:1146-1153) */
    float reach = 0.f;
    for (size_t i = 0; i < r.size() && i < 356; ++i)
        reach = std::max(reach, fabsf(r[i]));
    g.randomsReach = reach;
}

static void uploadRandoms(const float *randoms, long count, const char *who)
{
    if (ready(who))
    {
        quiesce();
        std::vector<float> r(randoms, randoms + count);
        HIPCHECK(hipSetDevice(g.device));
        upload(g.randoms, r);
        if (ok())
            g.nbRandoms = count;
        noteRandomsReach(r);
    }
    /* with a communicator rank 0's buffer is everybody's: every rank ends its upload here, in whatever state */
    shareRandoms();
}

void h2dRandomsOne(float *randoms)
{
    if (g.initialized && ok())
        ARGCHECK(randoms != nullptr, "h2d_randoms: null buffer");
    uploadRandoms(randoms, MAX_BITMAP_SIZE, "h2d_randoms");
}

/* Frames larger than the reference's 1920 x 1080 limit: its natural depth of field indexes the buffer with
 * `pixel index + timestamp % (MAX_BITMAP_SIZE - 2)` (CRT:475, the precedence as written), i.e. up to
 * W * H + 9999 + 1 - beyond MAX_BITMAP_SIZE floats as soon as the frame is larger (and by up to 9 999 floats
 * even at that size).  A host that renders such frames hands over as many values
 * as the expression can reach; reads beyond what was handed over return 0 (rt_device.h rnd()). */
void h2dRandomsSizedOne(const float *randoms, long count)
{
    if (g.initialized && ok())
        ARGCHECK(randoms != nullptr && count >= MAX_BITMAP_SIZE && count <= (1L << 30),
                 "solr_hip_h2d_randoms_sized: needs at least MAX_BITMAP_SIZE values");
    uploadRandoms(randoms, count, "solr_hip_h2d_randoms_sized");
}

void h2dTexturesOne(int activeTextures, TextureInfo *textureInfos)
{
    if (!ready("h2d_textures"))
        return;
    quiesce();
    ARGCHECK(activeTextures >= 0 && (activeTextures == 0 || textureInfos), "h2d_textures: bad arguments");
    for (int i = 0; ok() && i < activeTextures; ++i)
        if (textureInfos[i].buffer)
            ARGCHECK(textureInfos[i].offset >= 0 && textureInfos[i].size.x >= 0 && textureInfos[i].size.y >= 0 &&
                         textureInfos[i].size.z >= 0 &&
                         (double)textureInfos[i].size.x * textureInfos[i].size.y * textureInfos[i].size.z < 2147483648.0,
                     "h2d_textures: a texture with a negative offset or size, or larger than 2 GB");
    if (!ok())
        return;
    size_t total = 0, largest = 0;
    for (int i = 0; i < activeTextures; ++i)
        if (textureInfos[i].buffer)
        {
            size_t sz = (size_t)textureInfos[i].size.x * textureInfos[i].size.y * textureInfos[i].size.z;
            size_t end = (size_t)textureInfos[i].offset + sz;
            total = end > total ? end : total;
            largest = sz > largest ? sz : largest;
        }
    /* Slack: a texel fetch reads index .. index+2, and the secondary maps of a material (normal, bump,
     * specular ...) are read at the texel index of its DIFFUSE texture (TextureMapping.cuh:30-116): a map
     * smaller than the diffuse texture is read up to `largest` bytes past its own end.  Inside the atlas that
     * is the next texture, as in the reference; past the atlas the reference reads whatever follows its
     * buffer - here zeros, always. */
    std::vector<unsigned char> atlas(total + largest + 4, 0);
    for (int i = 0; i < activeTextures; ++i)
        if (textureInfos[i].buffer)
        {
            size_t sz = (size_t)textureInfos[i].size.x * textureInfos[i].size.y * textureInfos[i].size.z;
            memcpy(atlas.data() + textureInfos[i].offset, textureInfos[i].buffer, sz);
        }
    HIPCHECK(hipSetDevice(g.device));
    upload(g.textures, atlas);
    g.atlasBytes = ok() ? atlas.size() : 0;
    g.textureTablesChecked = false;
}

void h2dLightInformationOne(LightInformation *lightInformation, int lightInformationSize)
{
    if (!ready("h2d_lightInformation"))
        return;
    quiesce();
    ARGCHECK(lightInformationSize >= 0 && (lightInformationSize == 0 || lightInformation),
             "h2d_lightInformation: bad arguments");
    if (!ok())
        return;
    std::vector<float4> l(3 * (size_t)lightInformationSize);
    for (int i = 0; i < lightInformationSize; ++i)
    {
        const LightInformation &s = lightInformation[i];
        l[3 * i] = make_float4(s.location.x, s.location.y, s.location.z, bitsf(s.primitiveId));
        l[3 * i + 1] = make_float4(s.color.x, s.color.y, s.color.z, s.color.w);
        l[3 * i + 2] = make_float4(bitsf(s.materialId), 0.f, 0.f, 0.f);
    }
    g.hostLights.swap(l);
    g.geometryDirty = true;
    g.nbLights = lightInformationSize;
}

} // namespace solreng

extern "C" {
int solr_hip_device_rotations(void)
{
    return g.nbDeviceRotations;
}

/* Diagnostics / tests: the resident arena's node lists and primitive records as the device holds them
 * now.  exact != 0: the reference's list, else the walk-order list.  Returns the number of float4 rows
 * written (2 per node, 8 per primitive), -1 if the capacity is too small. */
int solr_hip_read_nodes(int exact, float *rows, int capacityRows)
{
    if (!ready("solr_hip_read_nodes") || !g.geometry.ptr)
        return -1;
    flushGeometry();
    if (exact)
        refreshExactList();
    quiesce();
    int n = 2 * (exact ? g.nbBoxes : g.nbBoxesCompact);
    unsigned at = exact ? g.offBoxes : g.offBoxesCompact;
    if (exact >= 2) /* 2 ... 9: the order-free list of octant exact - 2 (0 rows when there are none) */
    {
        const bool have = exact <= 9 && g.nbBoxesFree > 0 && !g.freeStale && g.freeRows == 16 * (size_t)g.nbBoxesFree;
        n = have ? 2 * g.nbBoxesFree : 0;
        at = g.offBoxesFree + 2u * (unsigned)((exact - 2) * g.nbBoxesFree);
    }
    if (!rows)
        return n; /* size query */
    if (n > capacityRows)
        return -1;
    if (n)
        HIPCHECK(hipMemcpy(rows, (const char *)g.geometry.ptr + (size_t)at * 16, (size_t)n * 16, hipMemcpyDeviceToHost));
    return ok() ? n : -1;
}

int solr_hip_read_primitives(float *rows, int capacityRows)
{
    if (!ready("solr_hip_read_primitives") || !g.geometry.ptr)
        return -1;
    flushGeometry();
    quiesce();
    const int n = PRIM_ROWS * g.nbPrimitives;
    if (!rows)
        return n;
    if (n > capacityRows)
        return -1;
    if (n)
        HIPCHECK(hipMemcpy(rows, (const char *)g.geometry.ptr + (size_t)g.offPrims * 16, (size_t)n * 16,
                           hipMemcpyDeviceToHost));
    return ok() ? n : -1;
}

} // extern "C"

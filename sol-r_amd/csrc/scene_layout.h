/*
 * scene_layout.h - how the scene lives in HBM.
 *
 * The host hands over arrays of structures (BoundingBox 48 B, Primitive 128 B,
 * Material 176 B, LightInformation 48 B; include/solr_types.h).  h2d_scene /
 * h2d_materials / h2d_lightInformation re-pack them into ROWS of 16 bytes
 * (one dwordx4 each) inside two arenas, grouped by the phase of the kernel
 * that reads them together:
 *
 *   geometry arena
 *     box node i        row 2i   = { min.x, min.y, min.z, max.z }        (x,y) (z,z) (x,y) pairs feed the
 *                       row 2i+1 = { max.x, max.y, bits(nbPrimitives), bits(skip) }   packed slab test
 *     boxStart[i]       int plane: first primitive of a leaf
 *     leaf record i     rows 4i..4i+3 (one 64-byte line, one s_load_dwordx16), for a leaf node i the first
 *                       primitive's test data and its index, so that entering a leaf costs ONE scalar-load
 *                       latency instead of a chain of three (start index -> head record -> rows behind it):
 *                         row 0 = { p0.xyz, bits(tag) }, row 1 = { size.xyz, bits(materialId) },
 *                         row 2 = { p1.xyz, bits(index) }, row 3 = { p2.xyz, bits(start) };
 *                       plane-class primitives (their tests read n0 and the colour-key average, not p1 / p2):
 *                         row 2 = { n0.xyz, bits(index) }, row 3 = { average colour, 0, 0, bits(start) }.
 *                       Built on the device from the primitive records (k_buildLeafRecords), after every
 *                       upload and every device-side rotation.
 *     primitive i       row 8i   = { p0.xyz,   bits(tag) }         \ sphere / ellipsoid / plane tests
 *                       row 8i+1 = { size.xyz, bits(materialId) }  / read these 32 bytes only
 *                       row 8i+2 = { p1.xyz,   bits(index) }       \ + cylinder, triangle: one 64-byte
 *                       row 8i+3 = { p2.xyz,   avg colour }        /   scalar-cache line in total
 *                       row 8i+4 = { n0.xyz, vt0.x }               \
 *                       row 8i+5 = { n1.xyz, vt0.y }                | normals / texture coordinates:
 *                       row 8i+6 = { n2.xyz, vt1.x }                | second cache line, triangle and
 *                       row 8i+7 = { vt1.y, vt2.x, vt2.y, 0 }      /  textured shading only
 *     light l           rows 3l..3l+2 = location+bits(primitiveId), colour, materialId
 *   material arena
 *     hot record m      rows 6m..6m+5: illumination, colour, specular, {reflection, refraction,
 *                       transparency, opacity}, attributes, {diffuse id, ambient-occlusion id}
 *     cold record m     rows 6m..6m+5 (after the hot block): texture mapping tables, only touched
 *                       when a textured material is hit
 *
 * tag = primitive type in bits 0-7 plus the material facts the walks need, joined in at
 * upload time (PRIM_* below) so that the hot loops never chase materialId -> material record
 * with a second, dependent scalar load.  The join is redone whenever either side changes.
 *
 * Why rows in an arena and not one plane per field: every access of the walk
 * is wave-uniform (one node / one primitive for the whole wave) and goes
 * through the scalar cache, so what matters is (a) the number of 64-byte
 * scalar-cache lines per test - a sphere is half a line, a triangle two lines -
 * and (b) the number of SGPRs that address the scene: ONE 64-bit base per arena
 * plus 32-bit row offsets, instead of a 64-bit base per plane (16 planes cost
 * 32 of the 102 SGPRs and pushed the kernel into v_readlane/v_writelane spill
 * traffic: 19 % of its vector instructions).  Per-lane gathers (the shading
 * phase reads the record of the lane's own hit) fetch rows with
 * global_load_dwordx4, one or two cache lines per record.
 *
 * Bytes per element: box 32 + 4, primitive 128, material 192, light 48.
 */
#pragma once

#include <hip/hip_runtime.h>

namespace solrdev
{
struct alignas(16) MaterialHot
{
    float4 innerIllumination; /* x emission, y diffusion, z range, w noise */
    float4 color;             /* rgb, w view noise */
    float4 specular;          /* x value, y power, z, w */
    float reflection, refraction, transparency, opacity;
    int4 attributes; /* x fast transparency, y procedural, z wireframe, w wireframe width */
    int4 ids;        /* x diffuse texture id, y ambient-occlusion texture id */
};
static_assert(sizeof(MaterialHot) == 96, "MaterialHot");

struct alignas(16) MaterialCold
{
    int4 textureMapping;        /* x width, y height, z -, w depth */
    int4 textureOffset;         /* diffuse, normal, bump, specular */
    int4 textureIds;            /* diffuse, normal, bump, specular */
    int4 advancedTextureOffset; /* reflection, transparency, ambient occlusion */
    int4 advancedTextureIds;
    float2 mappingOffset;
    float2 pad;
};
static_assert(sizeof(MaterialCold) == 96, "MaterialCold");

struct alignas(16) LightPlane
{
    float4 location; /* xyz, w = bits(primitiveId) */
    float4 color;    /* rgb, w = intensity */
    int materialId;
    int pad[3];
};
static_assert(sizeof(LightPlane) == 48, "LightPlane");

/* bits of the primitive tag (row 0, w) */
enum PrimTag
{
    PRIM_TYPE_MASK = 0xff,
    PRIM_FAST0 = 1 << 8,        /* material.attributes.x == 0 */
    PRIM_FAST1 = 1 << 9,        /* material.attributes.x == 1 (fast transparency) */
    PRIM_PROCEDURAL = 1 << 10,  /* material.attributes.y != 0 */
    PRIM_TRANSPARENT = 1 << 11, /* material.transparency != 0 */
    PRIM_WIRE1 = 1 << 12,       /* material.attributes.z == 1 */
    PRIM_WIRE2 = 1 << 13,       /* material.attributes.z == 2 */
    PRIM_EMISSIVE = 1 << 14,    /* material.innerIllumination.x != 0 */
    PRIM_TEXTURED = 1 << 15,    /* material.textureIds.x != TEXTURE_NONE */
    PRIM_WIDTH_SHIFT = 16,      /* clamp(material.attributes.w, -1, 100) + 1: bits 16-22 */
    PRIM_KIND_SHIFT = 24        /* PrimKind: what the walks may assume about the primitive and its material */
};

/* The common primitives in their plain form get a short path through the walks (rt_device.h): the type and the
 * material facts the tests branch on are settled at upload, the tests themselves are the general ones with
 * those facts as compile-time constants.  Only meaningful with extended geometry (without it every primitive is
 * tested as a triangle).  Every kind but KIND_GENERAL implies a material with attributes.x == 0 (PRIM_FAST0). */
enum PrimKind
{
    KIND_GENERAL = 0,
    KIND_SPHERE = 1,   /* ptSphere, not procedural */
    KIND_PLANE_XY = 2, /* ptXYPlane / ptYZPlane / ptXZPlane: not textured, no wireframe mode 2, */
    KIND_PLANE_YZ = 3, /* (YZ) not emissive */
    KIND_PLANE_XZ = 4,
    KIND_TRIANGLE = 5, /* ptTriangle */
    KIND_CYLINDER = 6  /* ptCylinder, ptCone (one test, GI:293-416) */
};

enum PrimRow
{
    ROW_P0_TYPE = 0,
    ROW_SIZE_MAT = 1,
    ROW_P1_INDEX = 2,
    ROW_P2 = 3,
    ROW_N0 = 4,
    ROW_N1 = 5,
    ROW_N2 = 6,
    ROW_UV = 7,
    PRIM_ROWS = 8
};

/* What the host passes to the kernel. Offsets are in rows of 16 bytes from the
 * arena base (offBoxStart: in ints). */
struct SceneArgs
{
    const void *geometry;
    const void *materials;
    const void *textures;
    const void *randoms;
    unsigned offBoxes, offBoxStart, offPrims, offLights, offMatCold;
    int nbBoxes;
    int nbPrimitives;
    int nbLights;
    int nbLamps;
    int nested;
    int orderedBoxes; /* every node has finite bounds with min <= max (sign-free slab test allowed) */
    long nbRandoms;
    unsigned offLeaf; /* leaf records of the node list in use, rows */
    /* the order-free list (closest-hit walks of rays longer than 2, rt_device.h): 0 nodes when there is none */
    unsigned offBoxesFree, offLeafFree;
    int nbBoxesFree;
    int opaqueShadows; /* no primitive is transparent or a textured plane: any occluder saturates a shadow */
    int shortRayLists; /* bounce rays (shorter than 1) take the order-free lists, checked (rt_device.h closestHitWalk) */
    /* behind the walk-order list and behind the eight order-free lists lies a copy of their node rows in which every
     * leaf that holds nothing but plain axis planes is as thin as its planes (solr_scene.hip tightenList): long rays with
     * no zero direction component walk the copy (rt_device.h tightRay) */
    int tightLists;
    /* behind the thin copies lies a third copy of the eight order-free lists' node rows in which every node's bounds
     * are sorted for the octant its list was flattened for - per axis (the bound a ray of that octant reaches first, the
     * other one): {n.x, n.y, n.z, f.z} {f.x, f.y, count, 32 x skip: bytes} (solr_scene.hip sortFreeLists).  A walk whose rays all
     * have that octant's signs - or all the opposite ones - takes it with a node loop that has no min / max per axis
     * (rt_device.h SOLR_ORDER_SORTED / _REVERSED).  Row offset: offBoxesFree + 2 (16 nbBoxesFree + 2) */
    int sortedLists;
};

/* Device view: everything is read through the CONSTANT address space.  The
 * scene is immutable for the lifetime of a launch, and for loads from this
 * address space hipcc always selects the scalar path (s_load_dwordx4/x8 through
 * the scalar cache into SGPRs) when the index is wave-uniform and the vector
 * path when it is per-lane - without depending on alias analysis, which gives
 * up on a function of this size and silently falls back to vector loads. */
#define SOLR_CONST_AS __attribute__((address_space(4)))
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef const SOLR_CONST_AS f4v *cf4p;
typedef const SOLR_CONST_AS i4v *ci4p;
typedef const SOLR_CONST_AS int *cip;
typedef const SOLR_CONST_AS float *cfp;
typedef const SOLR_CONST_AS unsigned char *cbp;

struct Scene
{
    cf4p geo;
    cf4p mat;
    cbp textures;
    cfp randoms;
    unsigned offBoxes, offBoxStart, offPrims, offLights, offMatCold;
    int nbBoxes;
    int nbPrimitives;
    int nbLights;
    int nbLamps;
    int nested; /* 1: skip pointers form nested intervals (validated on upload) */
    int orderedBoxes;
    long nbRandoms;
    unsigned offLeaf;
    unsigned offBoxesFree, offLeafFree;
    int nbBoxesFree;
    int opaqueShadows;
    int shortRayLists;
    int tightLists;
    int sortedLists;
};

__device__ __forceinline__ Scene makeScene(const SceneArgs &a)
{
    Scene s;
    s.geo = (cf4p)a.geometry;
    s.mat = (cf4p)a.materials;
    s.textures = (cbp)a.textures;
    s.randoms = (cfp)a.randoms;
    s.offBoxes = a.offBoxes;
    s.offBoxStart = a.offBoxStart;
    s.offPrims = a.offPrims;
    s.offLights = a.offLights;
    s.offMatCold = a.offMatCold;
    s.nbBoxes = a.nbBoxes;
    s.nbPrimitives = a.nbPrimitives;
    s.nbLights = a.nbLights;
    s.nbLamps = a.nbLamps;
    s.nested = a.nested;
    s.orderedBoxes = a.orderedBoxes;
    s.nbRandoms = a.nbRandoms;
    s.offLeaf = a.offLeaf;
    s.offBoxesFree = a.offBoxesFree;
    s.offLeafFree = a.offLeafFree;
    s.nbBoxesFree = a.nbBoxesFree;
    s.opaqueShadows = a.opaqueShadows;
    s.shortRayLists = a.shortRayLists;
    s.tightLists = a.tightLists;
    s.sortedLists = a.sortedLists;
    return s;
}

typedef float f8v __attribute__((ext_vector_type(8)));
typedef const SOLR_CONST_AS f8v *cf8p;
typedef const SOLR_CONST_AS char *ccp;

/* 32-bit byte offset from the arena base: selects the s_load sbase+soffset form
 * without 64-bit address arithmetic (arenas are < 4 GiB: 2.5 M primitives = 320 MB) */
__device__ __forceinline__ float4 ld4(cf4p p, unsigned row)
{
    const f4v v = *(cf4p)((ccp)p + (row << 4));
    return make_float4(v.x, v.y, v.z, v.w);
}
struct Row2
{
    float4 a, b;
};
/* two consecutive rows (32 bytes, 32-byte aligned): one s_load_dwordx8 */
__device__ __forceinline__ Row2 ld8(cf4p p, unsigned row)
{
    const f8v v = *(cf8p)((ccp)p + (row << 4));
    Row2 r;
    r.a = make_float4(v[0], v[1], v[2], v[3]);
    r.b = make_float4(v[4], v[5], v[6], v[7]);
    return r;
}
struct Row4
{
    float4 a, b, c, d;
};
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const SOLR_CONST_AS f16v *cf16p;
/* four consecutive rows (64 bytes, 64-byte aligned): one s_load_dwordx16 */
__device__ __forceinline__ Row4 ld16(cf4p p, unsigned row)
{
    const f16v v = *(cf16p)((ccp)p + (row << 4));
    Row4 r;
    r.a = make_float4(v[0], v[1], v[2], v[3]);
    r.b = make_float4(v[4], v[5], v[6], v[7]);
    r.c = make_float4(v[8], v[9], v[10], v[11]);
    r.d = make_float4(v[12], v[13], v[14], v[15]);
    return r;
}
__device__ __forceinline__ int4 asint4(const float4 &v)
{
    return make_int4(__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w));
}

__device__ __forceinline__ Row2 boxNode(const Scene &s, int i) { return ld8(s.geo, s.offBoxes + 2u * (unsigned)i); }
__device__ __forceinline__ Row2 primHead(const Scene &s, int i) { return ld8(s.geo, s.offPrims + 8u * (unsigned)i); }
__device__ __forceinline__ float4 nodeLo(const Row2 &n) { return make_float4(n.a.x, n.a.y, n.a.z, 0.f); }
__device__ __forceinline__ float4 nodeHi(const Row2 &n) { return make_float4(n.b.x, n.b.y, n.a.w, 0.f); }
__device__ __forceinline__ int nodeCount(const Row2 &n) { return __float_as_int(n.b.z); }
__device__ __forceinline__ int nodeSkip(const Row2 &n) { return __float_as_int(n.b.w); }
__device__ __forceinline__ int boxStart(const Scene &s, int i) { return ((cip)s.geo)[s.offBoxStart + (unsigned)i]; }
__device__ __forceinline__ Row4 leafRecord(const Scene &s, int i) { return ld16(s.geo, s.offLeaf + 4u * (unsigned)i); }
__device__ __forceinline__ float4 primRow(const Scene &s, int i, int row)
{
    return ld4(s.geo, s.offPrims + 8u * (unsigned)i + (unsigned)row);
}

/* unused fields are never loaded: each row is an independent 16-byte load */
__device__ __forceinline__ MaterialHot loadMaterialHot(const Scene &s, int id)
{
    MaterialHot m;
    const unsigned r = 6u * (unsigned)id;
    m.innerIllumination = ld4(s.mat, r);
    m.color = ld4(s.mat, r + 1);
    m.specular = ld4(s.mat, r + 2);
    const float4 a = ld4(s.mat, r + 3);
    m.reflection = a.x;
    m.refraction = a.y;
    m.transparency = a.z;
    m.opacity = a.w;
    m.attributes = asint4(ld4(s.mat, r + 4));
    m.ids = asint4(ld4(s.mat, r + 5));
    return m;
}

__device__ __forceinline__ MaterialCold loadMaterialCold(const Scene &s, int id)
{
    MaterialCold m;
    const unsigned r = s.offMatCold + 6u * (unsigned)id;
    m.textureMapping = asint4(ld4(s.mat, r));
    m.textureOffset = asint4(ld4(s.mat, r + 1));
    m.textureIds = asint4(ld4(s.mat, r + 2));
    m.advancedTextureOffset = asint4(ld4(s.mat, r + 3));
    m.advancedTextureIds = asint4(ld4(s.mat, r + 4));
    const float4 t = ld4(s.mat, r + 5);
    m.mappingOffset = make_float2(t.x, t.y);
    m.pad = make_float2(0.f, 0.f);
    return m;
}

__device__ __forceinline__ LightPlane loadLight(const Scene &s, int i)
{
    LightPlane l;
    const unsigned r = s.offLights + 3u * (unsigned)i;
    l.location = ld4(s.geo, r);
    l.color = ld4(s.geo, r + 1);
    const float4 t = ld4(s.geo, r + 2);
    l.materialId = __float_as_int(t.x);
    l.pad[0] = l.pad[1] = l.pad[2] = 0;
    return l;
}
}

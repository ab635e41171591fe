/*
 * scene_layout.h - how the scene lives in HBM (and, for small scenes, LDS).
 *
 * The host hands over arrays of structures (BoundingBox 48 B, Primitive 128 B,
 * Material 176 B, LightInformation 48 B; include/solr_types.h).  h2d_scene /
 * h2d_materials / h2d_lightInformation re-pack them into planes of float4,
 * one plane per group of fields that a phase of the kernel reads together:
 *
 *   box tree      boxLo[i]   = { min.x, min.y, min.z, bits(nbPrimitives) }
 *                 boxHi[i]   = { max.x, max.y, max.z, bits(skip) }
 *                 boxStart[i]= startIndex                     (leaves only)
 *   primitives,   primA[i]   = { p0.xyz,   bits(type) }
 *   traversal     primB[i]   = { p1.xyz,   bits(materialId) }
 *                 primC[i]   = { p2.xyz,   bits(index) }
 *                 primD[i]   = { size.xyz, 0 }
 *   primitives,   primN0[i]  = { n0.xyz, vt0.x }
 *   normals / uv  primN1[i]  = { n1.xyz, vt0.y }
 *                 primN2[i]  = { n2.xyz, vt1.x }
 *                 primT[i]   = { vt1.y, vt2.x, vt2.y, 0 }
 *   materials     matHot[m]  = 96-byte record: everything traversal and
 *                              untextured shading read
 *                 matCold[m] = 96-byte record: texture mapping tables, only
 *                              touched when a textured material is hit
 *   lights        lights[l]  = 48-byte record
 *
 * A sphere test touches primA + primD (32 B) instead of a 128-byte record; a
 * box visit touches 32 B instead of 48 B, and consecutive nodes of the
 * depth-first order are consecutive in both planes, so the scalar cache line
 * fetched for node i already holds nodes i+1..i+3.  All plane elements are
 * 16-byte aligned so that every access is one dwordx4 (s_load_dwordx4 for
 * wave-uniform indices, global_load_dwordx4 / ds_read_b128 for per-lane
 * gathers in the shading phase).
 *
 * Bytes per element: box 36, primitive 128 (64 traversal + 64 shading),
 * material 192, light 48.
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

/* What the host passes to the kernel: untyped device pointers. */
struct ScenePointers
{
    const void *boxLo, *boxHi, *boxStart;
    const void *primA, *primB, *primC, *primD, *primN0, *primN1, *primN2, *primT;
    const void *matHot, *matCold, *lights, *textures, *randoms;
};

/* Device view: every plane is read through the CONSTANT address space.  The
 * scene is immutable for the lifetime of a launch, and for loads from this
 * address space hipcc always selects the scalar path (s_load_dwordx4 through
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

struct ScenePlanes
{
    cf4p boxLo, boxHi;
    cip boxStart;
    cf4p primA, primB, primC, primD, primN0, primN1, primN2, primT;
    cf4p matHot;  /* 6 rows of 16 bytes per material */
    ci4p matCold; /* 6 rows of 16 bytes per material */
    cf4p lights;  /* 3 rows of 16 bytes per light */
    cbp textures;
    cfp randoms;
};

__device__ __forceinline__ ScenePlanes makePlanes(const ScenePointers &q)
{
    ScenePlanes p;
    p.boxLo = (cf4p)q.boxLo;
    p.boxHi = (cf4p)q.boxHi;
    p.boxStart = (cip)q.boxStart;
    p.primA = (cf4p)q.primA;
    p.primB = (cf4p)q.primB;
    p.primC = (cf4p)q.primC;
    p.primD = (cf4p)q.primD;
    p.primN0 = (cf4p)q.primN0;
    p.primN1 = (cf4p)q.primN1;
    p.primN2 = (cf4p)q.primN2;
    p.primT = (cf4p)q.primT;
    p.matHot = (cf4p)q.matHot;
    p.matCold = (ci4p)q.matCold;
    p.lights = (cf4p)q.lights;
    p.textures = (cbp)q.textures;
    p.randoms = (cfp)q.randoms;
    return p;
}

__device__ __forceinline__ float4 ld4(cf4p p, int i)
{
    const f4v v = p[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 ld4i(ci4p p, int i)
{
    const i4v v = p[i];
    return make_int4(v.x, v.y, v.z, v.w);
}

/* unused fields are never loaded: each row is an independent 16-byte load */
__device__ __forceinline__ MaterialHot loadMaterialHot(const ScenePlanes &p, int id)
{
    MaterialHot m;
    const int r = id * 6;
    m.innerIllumination = ld4(p.matHot, r);
    m.color = ld4(p.matHot, r + 1);
    m.specular = ld4(p.matHot, r + 2);
    const float4 a = ld4(p.matHot, r + 3);
    m.reflection = a.x;
    m.refraction = a.y;
    m.transparency = a.z;
    m.opacity = a.w;
    const float4 b = ld4(p.matHot, r + 4);
    m.attributes = make_int4(__float_as_int(b.x), __float_as_int(b.y), __float_as_int(b.z), __float_as_int(b.w));
    const float4 c = ld4(p.matHot, r + 5);
    m.ids = make_int4(__float_as_int(c.x), __float_as_int(c.y), __float_as_int(c.z), __float_as_int(c.w));
    return m;
}

__device__ __forceinline__ MaterialCold loadMaterialCold(const ScenePlanes &p, int id)
{
    MaterialCold m;
    const int r = id * 6;
    m.textureMapping = ld4i(p.matCold, r);
    m.textureOffset = ld4i(p.matCold, r + 1);
    m.textureIds = ld4i(p.matCold, r + 2);
    m.advancedTextureOffset = ld4i(p.matCold, r + 3);
    m.advancedTextureIds = ld4i(p.matCold, r + 4);
    const int4 t = ld4i(p.matCold, r + 5);
    m.mappingOffset = make_float2(__int_as_float(t.x), __int_as_float(t.y));
    m.pad = make_float2(0.f, 0.f);
    return m;
}

__device__ __forceinline__ LightPlane loadLight(const ScenePlanes &p, int i)
{
    LightPlane l;
    l.location = ld4(p.lights, i * 3);
    l.color = ld4(p.lights, i * 3 + 1);
    const float4 t = ld4(p.lights, i * 3 + 2);
    l.materialId = __float_as_int(t.x);
    l.pad[0] = l.pad[1] = l.pad[2] = 0;
    return l;
}
}

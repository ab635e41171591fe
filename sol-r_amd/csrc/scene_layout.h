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

struct ScenePlanes
{
    const float4 *__restrict__ boxLo;
    const float4 *__restrict__ boxHi;
    const int *__restrict__ boxStart;
    const float4 *__restrict__ primA;
    const float4 *__restrict__ primB;
    const float4 *__restrict__ primC;
    const float4 *__restrict__ primD;
    const float4 *__restrict__ primN0;
    const float4 *__restrict__ primN1;
    const float4 *__restrict__ primN2;
    const float4 *__restrict__ primT;
    const MaterialHot *__restrict__ matHot;
    const MaterialCold *__restrict__ matCold;
    const LightPlane *__restrict__ lights;
    const unsigned char *__restrict__ textures;
    const float *__restrict__ randoms;
};
}

/*
 * OBJReader.h - Wavefront OBJ / MTL models (reference: solr/io/OBJReader.{h,cpp}).
 *
 * What the reference's reader does with a file, restated:
 *   pass 1  "v x y z" -> vertex (x, y, -z), numbered from 1, and the model's bounds;
 *           "vn"      -> normal (x, y, -z); "vt u v" -> texture coordinate, a negative component replaced
 *           by the fractional part of its magnitude; "mtllib f" -> the material library next to the model.
 *   scale   autoScale: scale / largest extent, autoCenter (only with autoScale): the bounds' centre is
 *           moved to the origin; every point becomes position + scale * (point - centre).
 *   pass 2  "usemtl name" selects the material, "f a/b/c ..." adds one triangle (two for a quad: 0 1 2 and
 *           3 2 0; anything beyond four corners is dropped), with texture coordinates and normals looked
 *           up by the face's indices - an index that was never defined reads as zeros, as the reference's
 *           std::map look-ups do.  allSpheres replaces triangles by ellipsoids / spheres.  Groups whose "g"
 *           line contains SoL_R are SketchUp light components: their faces add no geometry, each group
 *           becomes one sphere (centre and half diagonal of the bounds of the face centres) with the
 *           group's material; a face of such a group still writes its texture coordinates and normals
 *           to primitive 0 (OBJReader.cpp:700-707, nbPrimitives is 0 there).
 *   MTL     newmtl / Kd / Ks / Tr / illum / SoL_R_Light as OBJReader.cpp:158-375; material ids are handed
 *           out in file order from `materialId`.  map_Kd / map_bump / map_norm / map_spec name image files:
 *           image codecs are outside this engine, the maps are reported and skipped.
 */
#pragma once

#include <map>
#include <string>
#include <vector>

#include "GPUKernel.h"

namespace solr
{
/* reference: OBJReader.h:29-50 */
struct MaterialMTL
{
    unsigned int index;
    vec4f Ka, Kd, Ks;
    float Ns, reflection, transparency, opacity, refraction, noise;
    int diffuseTextureId, normalTextureId, bumpTextureId, specularTextureId, reflectionTextureId,
        transparencyTextureId, ambientOcclusionTextureId;
    float illumination;
    bool isSketchupLightMaterial;
};

class OBJReader
{
public:
    unsigned int loadMaterialsFromFile(const std::string &filename, std::map<std::string, MaterialMTL> &materials,
                                       GPUKernel &kernel, int materialId);
    /* returns the size of the scaled model; aabb receives the bounds of the file's vertices */
    vec4f loadModelFromFile(const std::string &filename, GPUKernel &kernel, const vec4f &center, const bool autoScale,
                            const vec4f &scale, bool loadMaterials, int materialId, bool allSpheres, bool autoCenter,
                            CPUBoundingBox &aabb, const bool &checkInAABB, const CPUBoundingBox &inAABB);

private:
    void addLightComponent(GPUKernel &kernel, std::vector<vec4f> &faceCenters, const vec4f &center,
                           const vec4f &objectCenter, const vec4f &objectScale, const int material,
                           CPUBoundingBox &aabb);
};
}

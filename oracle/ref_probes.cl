/*
 * ref_probes.cl - TEST INFRASTRUCTURE ONLY (see oracle/solr_oracle.h).
 *
 * Function-level probes of the REFERENCE'S OWN per-pixel code.  This file holds no rendering code: it
 * #includes the reference's OpenCL engine where it lies (REFERENCE_CL is given by oracle/Makefile as
 * /root/reference/solr/engines/opencl/RayTracer.cl; nothing of it is copied) and adds kernels that call its
 * functions - boxIntersection, sphere/ellipsoid/cylinder/plane/triangleIntersection, intersectionWithPrimitives,
 * processShadows, primitiveShader, intersectionShader and the texture mappers behind it, skyboxMapping,
 * vectorRefraction / vectorReflection, makeColor, launchRayTracing - once per element of arrays of inputs,
 * and store everything the function returns or writes.  tests/test_reference_probes.py compares
 * oracle/solr_oracle.c with these outputs element for element (live on the GPU box; from the fixtures under
 * tests/golden/ on CPU).  The reference's post-processing kernels (k_default, k_depthOfField,
 * k_ambientOcclusion) need no probe: the runner launches them as they are.
 *
 * Two code objects are built from this file (oracle/Makefile):
 *   ref_probes_gfx950.co           the engine as ROCm's OpenCL compiler builds it;
 *   ref_probes_srcorder_gfx950.co  -DSOLR_PROBE_SOURCE_ORDER: the four geometric builtins the engine calls
 *       (dot, cross, length, normalize - library code, not the reference's) are evaluated as the CUDA engine's
 *       helper_math.h defines them: products summed left to right over x, y, z, no fused multiply-add,
 *       length = sqrt(dot), normalize = v * (1.f / sqrt(dot)) (helper_math.h:1248,1291,1309-1313 with the host
 *       rsqrtf of :62-65; cross as VectorUtils.cuh:45-52).  ROCm's versions fuse the sums and use the
 *       approximate reciprocal square root, which moves results by an ULP or two and cannot be compared bit for
 *       bit.  Every statement of the reference's own code is the same in both builds.
 * Both are compiled with -ftrivial-auto-var-init=zero: the engine reads the never-written .w of float4
 * locals (RayTracer.cl:1151-1290 writes .xyz of the normal and hit point, the float4 dot products and
 * lengths read .w; :1817-1818 declares them uninitialised) - with the flag those are zero, without it
 * whatever the register held.  Callers' float4s are zeroed here explicitly as well.
 */
#ifdef SOLR_PROBE_SOURCE_ORDER
static float so_dot(float4 a, float4 b)
{
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
static float so_length(float4 v)
{
    return sqrt(so_dot(v, v));
}
static float4 so_normalize(float4 v)
{
    float invLen = 1.0f / sqrt(so_dot(v, v));
    return v * invLen;
}
static float4 so_cross(float4 b, float4 c)
{
    float4 a;
    a.x = b.y * c.z - b.z * c.y;
    a.y = b.z * c.x - b.x * c.z;
    a.z = b.x * c.y - b.y * c.x;
    a.w = 0.f;
    return a;
}
#define dot(a, b) so_dot(a, b)
#define length(a) so_length(a)
#define normalize(a) so_normalize(a)
#define cross(a, b) so_cross(a, b)
#endif

#include REFERENCE_CL

#define ZERO4 ((float4)(0.f, 0.f, 0.f, 0.f))

static Ray makeRay(float4 origin, float4 direction)
{
    Ray r;
    r.origin = origin;
    r.direction = direction;
    r.inv_direction = ZERO4;
    r.signs = (int4)(0, 0, 0, 0);
    computeRayAttributes(&r);
    return r;
}

/* boxIntersection (RayTracer.cl:847-874) behind computeRayAttributes (:363-371) */
__kernel void probe_box(CONST BoundingBox* boxes, CONST float4* origins, CONST float4* directions, CONST float* t0,
                        CONST float* t1, CONST int* hit)
{
    const int i = get_global_id(0);
    Ray r = makeRay(origins[i], directions[i]);
    hit[i] = boxIntersection(&boxes[i], &r, t0[i], t1[i]) ? 1 : 0;
}

/* one primitive test, dispatched as the closest-hit walk (:1844-1873) or the shadow walk (:1542-1575) does;
 * out: 4 float4 per element = intersection, normal, areas, (hit, shadowIntensity, 0, 0) */
__kernel void probe_primitive(const SceneInfo sceneInfo, CONST Primitive* primitives, CONST Material* materials,
                              CONST BitmapBuffer* textures, CONST float4* origins, CONST float4* directions,
                              CONST int* shadows, CONST float4* initial, CONST float4* out)
{
    const int i = get_global_id(0);
    Ray r = makeRay(origins[i], directions[i]);
    CONST Primitive* primitive = &primitives[i];
    float4 intersection = initial[2 * i];
    float4 normal = initial[2 * i + 1];
    float4 areas = ZERO4;
    float shadowIntensity = 0.f;
    bool hit = false;
    const bool shadow = shadows[i] != 0;
    if (sceneInfo.extendedGeometry)
    {
        switch ((*primitive).type)
        {
        case ptEnvironment:
            if (shadow)
                hit = planeIntersection(&sceneInfo, primitive, materials, textures, &r, &intersection, &normal,
                                        &shadowIntensity, false);
            else
                hit = sphereIntersection(&sceneInfo, primitive, materials, &r, &intersection, &normal, &shadowIntensity);
            break;
        case ptSphere:
            hit = sphereIntersection(&sceneInfo, primitive, materials, &r, &intersection, &normal, &shadowIntensity);
            break;
        case ptCylinder:
            hit = cylinderIntersection(&sceneInfo, primitive, materials, &r, &intersection, &normal, &shadowIntensity);
            break;
        case ptEllipsoid:
            hit = ellipsoidIntersection(&sceneInfo, primitive, materials, &r, &intersection, &normal, &shadowIntensity);
            break;
        case ptTriangle:
            hit = triangleIntersection(&sceneInfo, primitive, &r, &intersection, &normal, &areas, &shadowIntensity,
                                       shadow);
            break;
        case ptCamera:
            if (shadow)
            {
                hit = false;
                break;
            }
        default:
            hit = planeIntersection(&sceneInfo, primitive, materials, textures, &r, &intersection, &normal,
                                    &shadowIntensity, false);
            break;
        }
    }
    else
        hit = triangleIntersection(&sceneInfo, primitive, &r, &intersection, &normal, &areas, &shadowIntensity, shadow);
    out[4 * i] = intersection;
    out[4 * i + 1] = normal;
    out[4 * i + 2] = areas;
    out[4 * i + 3] = (float4)(hit ? 1.f : 0.f, shadowIntensity, 0.f, 0.f);
}

/* intersectionWithPrimitives (:1802-1901); out: 4 float4 per element = closestIntersection, closestNormal,
 * closestAreas, colorBox; ids: (hit, closestPrimitive) */
__kernel void probe_closest(const SceneInfo sceneInfo, CONST BoundingBox* boxes, int nbBoxes,
                            CONST Primitive* primitives, int nbPrimitives, CONST Material* materials,
                            CONST BitmapBuffer* textures, CONST float4* origins, CONST float4* targets,
                            CONST int* iteration, CONST int* currentMaterialId, CONST float4* out, CONST int2* ids)
{
    const int i = get_global_id(0);
    Ray ray;
    ray.origin = origins[i];
    ray.direction = targets[i];
    ray.inv_direction = ZERO4;
    ray.signs = (int4)(0, 0, 0, 0);
    int closestPrimitive = -1;
    float4 closestIntersection = ZERO4, closestNormal = ZERO4, closestAreas = ZERO4, colorBox = ZERO4;
    bool hit = intersectionWithPrimitives(&sceneInfo, boxes, nbBoxes, primitives, nbPrimitives, materials, textures, &ray,
                                          iteration[i], &closestPrimitive, &closestIntersection, &closestNormal,
                                          &closestAreas, &colorBox, currentMaterialId[i]);
    out[4 * i] = closestIntersection;
    out[4 * i + 1] = closestNormal;
    out[4 * i + 2] = closestAreas;
    out[4 * i + 3] = colorBox;
    ids[i] = (int2)(hit ? 1 : 0, closestPrimitive);
}

/* processShadows (:1509-1612); out: (result, color.x, color.y, color.z) */
__kernel void probe_shadow(const SceneInfo sceneInfo, CONST BoundingBox* boxes, int nbBoxes,
                           CONST Primitive* primitives, int nbPrimitives, CONST Material* materials,
                           CONST BitmapBuffer* textures, CONST float4* lampCenters, CONST float4* origins,
                           CONST int* objectId, CONST int* iteration, CONST float4* out)
{
    const int i = get_global_id(0);
    float4 color = ZERO4;
    float result = processShadows(&sceneInfo, boxes, nbBoxes, primitives, materials, textures, nbPrimitives,
                                  lampCenters[i], origins[i], objectId[i], iteration[i], &color);
    out[i] = (float4)(result, color.x, color.y, color.z);
}

/* primitiveShader (:1620-1794); inout: 5 float4 per element = normal, intersection, closestColor, totalBlinn,
 * attributes (read, then overwritten with their values after the call); out: 3 float4 = returned colour,
 * refractionFromColor, (shadowIntensity, 0, 0, 0) */
__kernel void probe_shader(const SceneInfo sceneInfo, const PostProcessingInfo postProcessingInfo,
                           CONST BoundingBox* boxes, int nbBoxes, CONST Primitive* primitives, int nbPrimitives,
                           CONST LightInformation* lightInformation, int lightInformationSize, int nbLamps,
                           CONST Material* materials, CONST BitmapBuffer* textures, CONST RandomBuffer* randoms,
                           CONST int* index, CONST float4* origins, CONST int* objectId, CONST float4* areas,
                           CONST int* iteration, CONST float4* inout, CONST float4* out)
{
    const int i = get_global_id(0);
    float4 normal = inout[5 * i];
    float4 intersection = inout[5 * i + 1];
    float4 closestColor = inout[5 * i + 2];
    float4 totalBlinn = inout[5 * i + 3];
    float4 attributes = inout[5 * i + 4];
    float4 refractionFromColor = ZERO4;
    float shadowIntensity = 0.f;
    float4 returned =
        primitiveShader(index[i], &sceneInfo, &postProcessingInfo, boxes, nbBoxes, primitives, nbPrimitives,
                        lightInformation, lightInformationSize, nbLamps, materials, textures, randoms, origins[i], &normal,
                        objectId[i], &intersection, areas[i], &closestColor, iteration[i], &refractionFromColor,
                        &shadowIntensity, &totalBlinn, &attributes);
    inout[5 * i] = normal;
    inout[5 * i + 1] = intersection;
    inout[5 * i + 2] = closestColor;
    inout[5 * i + 3] = totalBlinn;
    inout[5 * i + 4] = attributes;
    out[3 * i] = returned;
    out[3 * i + 1] = refractionFromColor;
    out[3 * i + 2] = (float4)(shadowIntensity, 0.f, 0.f, 0.f);
}

/* intersectionShader (:1413-1494) with the mappers and maps behind it (:505-839), set up as primitiveShader
 * does (:1636-1648); attributes in/out; out: 4 float4 = colour, bump normal, specular, advanced attributes */
__kernel void probe_intersection_shader(const SceneInfo sceneInfo, CONST Primitive* primitives,
                                        CONST Material* materials, CONST BitmapBuffer* textures,
                                        CONST float4* intersections, CONST float4* areas, CONST float4* attributes,
                                        CONST float4* out)
{
    const int i = get_global_id(0);
    CONST Primitive* primitive = &primitives[i];
    CONST Material* material = &materials[(*primitive).materialId];
    float4 bumpNormal = ZERO4;
    float4 advancedAttributes = ZERO4;
    float4 specular = ZERO4;
    specular.x = (*material).specular.x;
    specular.y = (*material).specular.y;
    specular.z = (*material).specular.z;
    float4 intersection = intersections[i];
    float4 attr = attributes[i];
    float4 color = intersectionShader(&sceneInfo, primitive, materials, textures, &intersection, areas[i], &bumpNormal,
                                      &specular, &attr, &advancedAttributes);
    attributes[i] = attr;
    out[4 * i] = color;
    out[4 * i + 1] = bumpNormal;
    out[4 * i + 2] = specular;
    out[4 * i + 3] = advancedAttributes;
}

/* skyboxMapping (:941-1006) */
__kernel void probe_skybox(const SceneInfo sceneInfo, CONST Material* materials, CONST BitmapBuffer* textures,
                           CONST float4* origins, CONST float4* targets, CONST float4* out)
{
    const int i = get_global_id(0);
    Ray r;
    r.origin = origins[i];
    r.direction = targets[i];
    r.inv_direction = ZERO4;
    r.signs = (int4)(0, 0, 0, 0);
    out[i] = skyboxMapping(&sceneInfo, materials, textures, &r);
}

/* vectorRefraction (:322-334) and the vectorReflection macro (:312); out: 2 float4 = refracted, reflected */
__kernel void probe_vectors(CONST float4* incident, CONST float4* normals, CONST float* n1, CONST float* n2,
                            CONST float4* out)
{
    const int i = get_global_id(0);
    float4 refracted = ZERO4, reflected = ZERO4;
    vectorRefraction(&refracted, incident[i], n1[i], normals[i], n2[i]);
    vectorReflection(reflected, incident[i], normals[i]);
    out[2 * i] = refracted;
    out[2 * i + 1] = reflected;
}

/* makeColor (:379-412): pixel i of an n x 1 ... image as sceneInfo.size says */
__kernel void probe_make_color(const SceneInfo sceneInfo, CONST float4* colors, CONST BitmapBuffer* bitmap)
{
    const int i = get_global_id(0);
    float4 color = colors[i];
    makeColor(&sceneInfo, &color, bitmap, i);
}

/* launchRayTracing (:2104-2440) for the ray origins[i] -> targets[i] of pixel index[i]: the whole bounce loop with
 * its closest-hit walks, shading, shadow rays, refraction / reflection, deferred reflection ray, blend and fog,
 * without the camera code of k_standardRenderer around it (which jitters the ray on every pass and rotates
 * with half_cos / half_sin).  out: (colour.xyz, depth); ids in/out */
__kernel void probe_launch(CONST BoundingBox* boxes, int nbBoxes, CONST Primitive* primitives, int nbPrimitives,
                           CONST LightInformation* lightInformation, int lightInformationSize, int nbLamps,
                           CONST Material* materials, CONST BitmapBuffer* textures, CONST RandomBuffer* randoms,
                           const SceneInfo sceneInfo, const PostProcessingInfo postProcessingInfo,
                           CONST float4* origins, CONST float4* targets, CONST int* index, CONST float4* out,
                           CONST PrimitiveXYIdBuffer* ids)
{
    const int i = get_global_id(0);
    Ray r;
    r.origin = origins[i];
    r.direction = targets[i];
    r.inv_direction = ZERO4;
    r.signs = (int4)(0, 0, 0, 0);
    float dof = 0.f;
    float4 color = launchRayTracing(index[i], boxes, nbBoxes, primitives, nbPrimitives, lightInformation,
                                    lightInformationSize, nbLamps, materials, textures, randoms, &r, &sceneInfo,
                                    &postProcessingInfo, &dof, &ids[i]);
    out[i] = (float4)(color.x, color.y, color.z, dof);
}

/*
 * solr_hip_probes.h - TEST-ONLY entry points of libsolr_hip.so (sol-r_amd/csrc/solr_probes.hip).
 *
 * Not part of the drop-in boundary (include/solr_hip.h) and not used by the product: they exist so that the
 * engine's OWN device functions - the slab test in its three forms (the reference's compare chain, the sign-free
 * form, the hand-scheduled node loop), the primitive tests as both walks dispatch them, the two walks themselves
 * over the resident scene, primitiveShader, refraction / reflection, makeColor, skyboxMapping, intersectionShader with
 * the texture mappers, the post-processing kernels - can be evaluated once per element of arrays of inputs, the arrays that oracle/ref_probes.cl feeds to
 * THE REFERENCE'S OWN functions (RayTracer.cl:847-874, 1151-1394, 1528-1591 ...), and compared with the reference's
 * outputs bit for bit WITH NO ORACLE IN BETWEEN (tests/test_engine_probes_gpu.py against
 * tests/golden/reference_probes.npz).  Every kernel calls the very functions the renderer is built from
 * (rt_device.h), in the instantiation the renderer would launch for the resident scene.
 *
 * Vectors are packed xyz floats.  `features`: 0 = the instantiation renderImpl would pick for the resident scene
 * and this SceneInfo (enum Feature of rt_device.h, F_DEEP for a list of more than 1 024 nodes), or one of the
 * masks this file instantiates.  Every function returns the mask it ran (> 0), or -1 with solr_hip_last_error set.
 * The scene probes need a resident scene (h2d_scene / h2d_materials ..., or a frame rendered through the host
 * protocol); `exactNodes` != 0 walks the reference's own node list instead of the engine's walk-order list.
 */
#ifndef SOLR_HIP_PROBES_H
#define SOLR_HIP_PROBES_H

#include "solr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* boxIntersection (GI:52-79) behind computeRayAttributes (GI:36-44), element i = box i against ray i:
 * hitExact - the reference's compare chain (rt_device.h boxIntersectionExact); hitFast - the sign-free form
 * (boxIntersectionFast), -1 where the ray does not meet its precondition (finiteRay) */
int solr_hip_probe_box(int n, const BoundingBox *boxes, const float *origins, const float *directions, const float *t0,
                       const float *t1, int *hitExact, int *hitFast);

/* the hand-scheduled node loop (rt_device.h advanceTidy): the resident scene is a flat list of n leaf boxes; ray i
 * walks the whole list with far distance t1[i] (the loop's near distance is 0, as in both walks) and hit[i] says
 * whether it entered leaf i */
int solr_hip_probe_box_walk(const SceneInfo *sceneInfo, int n, const float *origins, const float *directions,
                            const float *t1, int features, int *hit);

/* one primitive test as the closest-hit walk (GI:712-747) or, shadows[i] != 0, the shadow walk (GI:835-864)
 * dispatches it (rt_device.h testPrimitive): element i = primitive i of the resident scene against ray i.
 * intersection / normal are in/out (the walks hand the tests whatever their locals held). */
int solr_hip_probe_primitive(const SceneInfo *sceneInfo, int n, const float *origins, const float *directions,
                             const int *shadows, int features, float *intersection, float *normal, float *areas,
                             float *shadowIntensity, int *hit);

/* intersectionWithPrimitives (GI:667-772; rt_device.h closestHitWalk) over the resident scene, 64 rays per wave */
int solr_hip_probe_closest(const SceneInfo *sceneInfo, int n, const float *origins, const float *targets,
                           const int *iteration, const int *currentMaterialId, int features, int exactNodes, int *hit,
                           int *primitive, float *intersection, float *normal, float *areas);

/* processShadows (GI:798-908; rt_device.h shadowWalk): lightId is the lamp's Primitive.index, objectId the shaded
 * primitive's (both left out of the walk, GI:829; pass an index nobody has to leave out the lamp only) */
int solr_hip_probe_shadow(const SceneInfo *sceneInfo, int n, const float *lampCenters, const float *origins,
                          const int *lightId, const int *objectId, const int *iteration, int features, int exactNodes,
                          float *result, float *color);

/* primitiveShader (GI:916-1080; rt_device.h primitiveShader) with the resident scene, lights and random buffer: element i
 * shades primitive objectId[i] (index into the flattened array) at intersections[i] as bounce iteration[i] of pixel
 * index[i], seen from origins[i].  normal, closestColor, totalBlinn (3 per element) and attributes (4) are in/out, as
 * in the reference (they persist across bounces); returned 3, shadowIntensity 1 per element.  Lamp 0's shadow rays walk
 * the resident node list (exactNodes as above). */
int solr_hip_probe_shader(const SceneInfo *sceneInfo, int n, const int *index, const float *origins, const int *objectId,
                          const float *intersections, const float *areas, const int *iteration, int features, int exactNodes,
                          float *normal, float *closestColor, float *totalBlinn, float *attributes, float *returned,
                          float *shadowIntensity);

/* the post-processing stage of cudaRender (CRT:1857-1890) over a frame buffer of the caller's (sceneInfo.size pixels):
 * k_depthOfField / k_ambientOcclusion / k_radiosity / k_filter / k_cartoon exactly as renderImpl launches them behind the
 * renderer, or, for ppe_none, the stand-alone k_default (CRT:1057-1073: the conversion the renderer otherwise fuses into
 * its epilogue).  The random buffer is the resident one (h2d_randoms).  bitmap: 3 bytes per pixel. */
int solr_hip_probe_postprocess(const SceneInfo *sceneInfo, const PostProcessingInfo *postProcessingInfo,
                               const PostProcessingBuffer *frame, unsigned char *bitmap);

/* The read-back tickets of solr_hip_d2h_image_async (include/solr_hip.h): the ticket the serial-th frame of a process
 * gets - never negative (0 once a period): (serial mod period) x ring + slot - with its slot and the period (no GPU needed); and
 * the engine's serial counter, returned, and set first when setTo >= 0 (to put a running engine just before 2^31 / 6
 * tickets - 27 hours of frames - and render across) */
int solr_hip_probe_ticket(long long serial, int *slot, long long *period);
long long solr_hip_probe_image_serial(long long setTo);

/* vectorRefraction (VU:73-87) and vectorReflection (VU:61-64) */
int solr_hip_probe_vectors(int n, const float *incident, const float *normals, const float *n1, const float *n2,
                           float *refracted, float *reflected);

/* makeColor (GS:132-165) for pixel i of an image as sceneInfo.size says; bitmap: 3 n bytes */
int solr_hip_probe_make_color(const SceneInfo *sceneInfo, int n, const float *colors, unsigned char *bitmap);

/* skyboxMapping (GI:87-151) with the resident materials and textures */
int solr_hip_probe_skybox(const SceneInfo *sceneInfo, int n, const float *origins, const float *targets, float *color);

/* intersectionShader (GS:36-124) with the mappers and maps behind it, set up as primitiveShader does (GI:933-945):
 * element i = primitive i of the resident scene.  attributes (4 per element) in/out; color 4, bump 3, specular 3,
 * ambientOcclusion 1 per element */
int solr_hip_probe_intersection_shader(const SceneInfo *sceneInfo, int n, const float *intersections, const float *areas,
                                       float *attributes, float *color, float *bump, float *specular,
                                       float *ambientOcclusion);

#ifdef __cplusplus
}
#endif
#endif

/*
 * solr_oracle.h - CPU oracle for the Sol-R per-pixel rendering path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS - pinned to outputs of the reference itself, function by function and bit for bit.
 * The reference ships no tests, golden vectors or fixtures for this path; its CPU
 * engine does not compile and its CUDA engine cannot be built in this image without writing
 * stand-ins for the CUDA SDK headers and the cmake-generated defines.h, which the build rules forbid.  Its
 * OpenCL engine keeps the same path in one self-contained file, and ROCm's clang compiles that file for
 * gfx950 as it lies (oracle/Makefile, target `ref` -> oracle/_ref/):
 *   - oracle/ref_probes.cl wraps the reference's OWN functions (box and primitive tests, the closest-hit and
 *     shadow walks, primitiveShader, intersectionShader and the texture mappers, skyboxMapping, refraction /
 *     reflection, makeColor, the whole of launchRayTracing; plus its post-processing kernels as they are) in
 *     kernels over arrays of inputs.  This file has an OpenCL DIALECT (oracle_set_dialect(1)): the few dozen
 *     statements in which the reference's two engines differ, each an `if (g_cl)` citing both files.  In that
 *     dialect it reproduces every output of those probes BIT FOR BIT (tests/test_reference_probes.py: live on
 *     the GPU box, and on CPU from tests/golden/reference_probes.npz), the library pow of the Blinn term aside
 *     (<= 4 ULP) - planes, glass, transparent shadows, textures, accumulation passes and whole 3-bounce frames
 *     of the Cornell room included.  Everything outside those switches is shared with dialect 0, the CUDA
 *     engine's form that the product is held to; what a switch selects in dialect 0 is the cited CUDA statement.
 *   - tests/test_reference_opencl.py / test_golden_reference.py compare whole frames of the reference's
 *     k_standardRenderer with this oracle at image level (the renderer as built fuses its dot products, so
 *     that comparison is statistical, and every differing pixel is shown to sit on a silhouette, crease or
 *     shadow edge).
 * The CUDA engine itself has no runnable form here: the dialect-0 statements under the switches (listed in
 * DESIGN.md section 2) are pinned by reading only, and by hand-derived known answers
 * (tests/test_oracle_known_answers.py).
 */
#ifndef SOLR_ORACLE_H
#define SOLR_ORACLE_H

#include "../include/solr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleScene_s
{
    const BoundingBox *boxes;
    int nbBoxes;
    const Primitive *primitives;
    int nbPrimitives;
    const LightInformation *lights;
    int nbLights; /* lightInformationSize */
    int nbLamps;
    const Material *materials;
    const BitmapBuffer *textures; /* may be NULL when no material is textured */
    const float *randoms;         /* may be NULL when the frame does not read randoms */
    long nbRandoms;               /* number of floats behind randoms (bounds checked) */
} OracleScene;

/* counts[0] closest-hit walks, [1] shadow walks, [2] box nodes visited,
 * [3] primitive tests; may be NULL */
typedef unsigned long long oracle_counts_t[4];

/* One frame of the reference's k_standardRenderer (CudaRayTracer.cu:437-563)
 * followed by its post-processing stage (k_default / k_ambiantOcclusion /
 * k_depthOfField, CudaRayTracer.cu:1057-1181) for rows
 * [firstRow, firstRow+nbRows) of the image.  pp / ids / bitmap are strip
 * sized (nbRows*W records; W*3 bytes per row for bitmap) and in/out: the
 * progressive-refinement and accumulation modes read the previous pass.
 * nthreads <= 0 uses every core (OpenMP over rows).  Returns 0, or -1 when a
 * random index would fall outside [0, nbRandoms). */
int oracle_render(const OracleScene *scene, const SceneInfo *sceneInfo, const PostProcessingInfo *ppInfo,
                  const float origin[3], const float direction[3], const float angles[4], int firstRow, int nbRows,
                  PostProcessingBuffer *pp, PrimitiveXYIdBuffer *ids, BitmapBuffer *bitmap, oracle_counts_t counts,
                  int nthreads);

/* Function-level entry points for known-answer tests ------------------- */

/* boxIntersection (GeometryIntersections.cuh:52-79) for the ray
 * origin -> origin + direction (direction is NOT normalised, as in the
 * reference's walk) */
int oracle_box_intersection(const BoundingBox *box, const float origin[3], const float direction[3], float t0,
                            float t1);

/* One primitive test exactly as the closest-hit walk dispatches it
 * (GeometryIntersections.cuh:712-747).  Returns hit; writes intersection,
 * normal, areas and the primitive's shadow intensity. */
int oracle_primitive_intersection(const SceneInfo *sceneInfo, const Primitive *primitive, const Material *materials,
                                  const BitmapBuffer *textures, const float origin[3], const float direction[3],
                                  int processingShadows, float intersection[3], float normal[3], float areas[3],
                                  float *shadowIntensity);

/* intersectionWithPrimitives (GeometryIntersections.cuh:667-772) */
int oracle_closest_hit(const OracleScene *scene, const SceneInfo *sceneInfo, const float origin[3],
                       const float target[3], int iteration, int currentMaterialId, int *closestPrimitive,
                       float closestIntersection[3], float closestNormal[3], float closestAreas[3]);

/* processShadows (GeometryIntersections.cuh:798-908) */
float oracle_shadow(const OracleScene *scene, const SceneInfo *sceneInfo, const float lampCenter[3],
                    const float origin[3], int lightId, int iteration, int objectId, float color[3]);

/* vectorRotation (VectorUtils.cuh:104-142) */
void oracle_vector_rotation(float v[3], const float center[3], const float angles[3]);

/* makeColor (GeometryShaders.cuh:132-165) for one pixel */
void oracle_make_color(const SceneInfo *sceneInfo, const float color[3], BitmapBuffer *bitmap, int index);

int oracle_max_threads(void);

/* 0 (default): the CUDA engine's statements; 1: the OpenCL engine's, for the comparison with oracle/ref_probes.cl */
void oracle_set_dialect(int openclEngine);
/* powf / sinf / cosf / atan2f / asinf evaluated in binary64 and rounded once, as the engine does (they are
 * specified to an error bound only); off by default = libm's binary32 routines */
void oracle_set_rounded_transcendentals(int on);
int oracle_get_rounded_transcendentals(void);
/* mask: one byte per pixel of the strip the next oracle_render calls render (NULL ends it).  While set and the
 * transcendentals are libm's, every powf (bit 0) and sinf / cosf / atan2f / asinf (bit 1) is also evaluated in
 * binary64 and rounded once; a pixel whose libm result is not that value gets the bit.  tests/: the pixels where
 * the engine may be a second ULP from the oracle as pinned are exactly such pixels, and they are counted. */
void oracle_set_misround_mask(unsigned char *mask, long n);
int oracle_get_dialect(void);
/* batched function-level entry points (oracle_probe_*) and oracle_postprocess: see the end of solr_oracle.c
 * and oracle/probes.py */

#ifdef __cplusplus
}
#endif
#endif

/*
 * SolRStub.h - flat extern "C" facade over the singleton engine.
 *
 * Same names, argument order and return conventions as the reference's
 * solr/SolRStub.h:36-135 (0 / -1 ints, doubles converted to floats, no C++
 * exceptions across the boundary) for every call that feeds or runs the
 * rendering path, including the scene-file loaders (SolR_LoadMolecule,
 * SolR_LoadOBJModel, SolR_SaveToFile, SolR_LoadFromFile).  Not provided (out
 * of scope): the OpenCL queries, SolR_LoadTextureFromFile
 * and SolR_GenerateScreenshot (image codecs) and the Kinect call.
 *
 * SolRx_* are extensions used by the test-suite and bench.py: engine
 * selection, deterministic timestamps/randoms, access to the flattened arrays
 * and to the float framebuffer.
 */
#pragma once

#include "../../include/solr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------- Scene (SolRStub.h:44-60) ---------- */
int SolR_SetSceneInfo(int width, int height, int graphicsLevel, int nbRayIterations, double transparentColor,
                      double viewDistance, double shadowIntensity, double eyeSeparation, double bgColorR,
                      double bgColorG, double bgColorB, double bgColorA, int renderBoxes, int pathTracingIteration,
                      int maxPathTracingIterations, int frameBufferType, int timestamp, int atmosphericEffect,
                      int cameraType, int doubleSidedTriangles, int extendedGeometry, int advancedIllumination,
                      int skyboxSize, int skyboxMaterialId, double geometryEpsilon, double rayEpsilon);
int SolR_SetPostProcessingInfo(int type, double param1, double param2, int param3);
int SolR_SetDraftMode(int draft);
int SolR_InitializeKernel(bool activeLogging, int platform, int device);
int SolR_FinalizeKernel();
int SolR_ResetKernel();

/* ---------- Camera (SolRStub.h:65-66) ---------- */
void SolR_SetCamera(double eye_x, double eye_y, double eye_z, double dir_x, double dir_y, double dir_z,
                    double angle_x, double angle_y, double angle_z);

/* ---------- Rendering (SolRStub.h:69) ---------- */
int SolR_RunKernel(double timer, BitmapBuffer *image);

/* ---------- Primitives (SolRStub.h:72-103) ---------- */
int SolR_AddPrimitive(int type, int movable);
int SolR_SetPrimitive(int index, double p0_x, double p0_y, double p0_z, double p1_x, double p1_y, double p1_z,
                      double p2_x, double p2_y, double p2_z, double size_x, double size_y, double size_z,
                      int materialId);
int SolR_GetPrimitive(int index, double *p0_x, double *p0_y, double *p0_z, double *p1_x, double *p1_y, double *p1_z,
                      double *p2_x, double *p2_y, double *p2_z, double *size_x, double *size_y, double *size_z,
                      int *materialId);
int SolR_GetPrimitiveAt(int x, int y);
int SolR_GetPrimitiveCenter(int index, double *x, double *y, double *z);
int SolR_RotatePrimitives(int fromBoxId, int toBoxId, double rx, double ry, double rz, double ax, double ay,
                          double az);
int SolR_SetPrimitiveMaterial(int index, int materialId);
int SolR_GetPrimitiveMaterial(int index);
int SolR_SetPrimitiveNormals(int index, double n0_x, double n0_y, double n0_z, double n1_x, double n1_y, double n1_z,
                             double n2_x, double n2_y, double n2_z);
int SolR_SetPrimitiveTextureCoordinates(int index, double t0_x, double t0_y, double t1_x, double t1_y, double t2_x,
                                        double t2_y);

/* ---------- Materials (SolRStub.h:106-125) ---------- */
int SolR_AddMaterial();
int SolR_SetMaterial(int index, double color_r, double color_g, double color_b, double noise, double reflection,
                     double refraction, int procedural, int wireframe, int wireframeDepth, double transparency,
                     double opacity, int diffuseTextureId, int normalTextureId, int bumpTextureId,
                     int specularTextureId, int reflectionTextureId, int transparencyTextureId,
                     int ambientOcclusionTextureId, double specValue, double specPower, double specCoef,
                     double innerIllumination, double illuminationDiffusion, double illuminationPropagation,
                     int fastTransparency);

/* ---------- Read-back of materials and textures (SolRStub.h:114-122,133) ---------- */
/* the reference's `type &out` parameters are pointers here: the same ABI.  -1 (outputs untouched) for an
 * index beyond the active materials */
int SolR_GetMaterial(int index, double *color_r, double *color_g, double *color_b, double *noise, double *reflection,
                     double *refraction, int *procedural, int *wireframe, int *wireframeDepth, double *transparency,
                     double *opacity, int *diffuseTextureId, int *normalTextureId, int *bumpTextureId,
                     int *specularTextureId, int *reflectionTextureId, int *transparencyTextureId,
                     int *ambientOcclusionTextureId, double *specValue, double *specPower, double *specCoef,
                     double *innerIllumination, double *illuminationDiffusion, double *illuminationPropagation,
                     int *fastTransparency);
/* the texture's pixels with the first and third channel swapped (SolRStub.cpp:351-373); 0 = done, 1 = no
 * such texture */
int SolR_GetTexture(int index, BitmapBuffer *image);
/* accepted for compatibility: the reference's SolR_RotatePrimitive is an empty TODO (SolRStub.cpp:232-240),
 * and there is nothing to recompile at run time - the kernels are built ahead of time for gfx950 */
int SolR_RotatePrimitive(int index, double rx, double ry, double rz, double ax, double ay, double az);
int SolR_RecompileKernels(char *filename);

/* ---------- Scene files (SolRStub.h:145-146) ---------- */
/* .irt scene dumps, host/FileMarshaller.h; both return the number of active (flattened) primitives,
 * which is what the reference returns: 0 until the next SolR_CompactBoxes */
/* Molecules from PDB files, host/PDBReader.h (SolRStub.h:137-138): geometryType 0 atoms, 1 fixed-size atoms,
 * 2 sticks, 3 atoms and sticks, 4 iso-surface, 5 backbone; atomMaterialType 0 by element, 1 by chain,
 * 2 by residue.  Materials 0..118 are overwritten with the element colours. */
int SolR_LoadMolecule(char *filename, int geometryType, double defaultAtomSize, double defaultStickSize,
                      int atomMaterialType, double scale);
/* Wavefront OBJ + MTL, host/OBJReader.h (SolRStub.h:141-143; the reference passes `double &height`, the
 * same ABI).  The model's materials take the ids materialId, materialId + 1, ... in the order of the MTL
 * file; *height receives -(scaled model height) / 2. */
int SolR_LoadOBJModel(char *filename, int materialId, int autoScale, double scale, int autoCenter, double *height);
/* SWC neuron morphology, host/SWCReader.h.  The reference has the reader (solr/io/SWCReader.cpp, used by
 * its SwcScene) but no stub entry for it; this one passes loadMorphologyFromFile's arguments through and
 * returns the number of samples read. */
int SolRx_LoadSWCMorphology(const char *filename, double px, double py, double pz, double sx, double sy, double sz,
                            double sw, int materialId);
int SolR_SaveToFile(char *filename);
int SolR_LoadFromFile(char *filename, double scale);

/* ---------- Boxes / lights (SolRStub.h:128-131) ---------- */
int SolR_CompactBoxes(bool update);
int SolR_GetLight(int index);

/* ---------- Textures (SolRStub.h:134-138).  The reference's SolR_SetTexture
 * is an empty TODO (SolRStub.cpp:357-366); this one takes the pixels. */
int SolR_SetTexture(int index, const unsigned char *pixels, int width, int height, int depth, int textureType);
int SolR_GetTextureSize(int index, int *width, int *height, int *depth);
int SolR_GetNbTextures(int *nbTextures);

/* ---------- Extensions ---------- */
/* "hip" (default) or "host-only"; destroys the current singleton */
int SolRx_SelectEngine(const char *name);
/* seed >= 0: keep SceneInfo.timestamp and fill the random buffer from a seeded
 * LCG; seed < 0: reference behaviour (rand()/time(0)) */
int SolRx_SetDeterministic(long seed);
/* 1: compactBoxes(true) always builds the box tree on the host; 0 (default): on the device when the engine
 * offers it (include/solr_hip.h solr_hip_build_tree) - the same tree either way.  Also SOLR_HOST_BUILD=1 */
int SolRx_HostBuild(int hostOnly);
/* pending engine error: 0 = none; copies the text into buf when non-null */
int SolRx_LastError(char *buf, int len);
/* render_begin + render_end without copying the bitmap */
int SolRx_Render(double timer);
/* frames in flight through SolR_RunKernel / SolRx_Render (GPUKernel::setFramesInFlight, 1 ... 4; 1 = the reference's
 * protocol): a call delivers the image of the frame n - 1 calls back while the newer ones render;
 * SolRx_FlushFrames waits for all of them; SolRx_GetBitmap is the image delivered last, without a copy */
int SolRx_SetFramesInFlight(int n);
int SolRx_FlushFrames(void);
/* the frame shared out over n devices of this process (occupancyParameters.x of the boundary, include/solr_hip.h);
 * returns the number in use, or -1.  With frames in flight the ids behind SolR_GetPrimitiveAt are the newest
 * frame's, up to n - 1 frames ahead of the image on show. */
int SolRx_SetGpuCount(int n);
const BitmapBuffer *SolRx_GetBitmap(void);
/* flattened arrays of the current frame (owned by the engine, valid until
 * the next compactBoxes / material change) */
int SolRx_GetBoxes(const BoundingBox **boxes, int *nbBoxes);
int SolRx_GetPrimitives(const Primitive **primitives, int *nbPrimitives);
int SolRx_GetLights(const LightInformation **lights, int *nbLights, int *nbLamps);
int SolRx_GetMaterials(const Material **materials, int *nbMaterials);
int SolRx_GetRandoms(const float **randoms, int *nbRandoms);
int SolRx_GetTextureAtlas(const unsigned char **atlas, long *nbBytes);
int SolRx_GetPrimitiveIds(const PrimitiveXYIdBuffer **ids, int *nbPixels);
int SolRx_GetSceneInfo(SceneInfo *sceneInfo, PostProcessingInfo *postProcessingInfo, float eye[3], float dir[3],
                       float angles[4]);
int SolRx_GetTreeDepth();
/* rotations applied on the device that the host scene store has not replayed yet; SolRx_SyncHost replays them */
/* key frames (GPUKernel::setNbFrames / setFrame / morphPrimitives; the reference's scenes call them directly) */
int SolRx_SetNbFrames(int nbFrames);
int SolRx_SetFrame(int frame);
int SolRx_MorphPrimitives();
int SolRx_PendingRotations();
int SolRx_SyncHost();
int SolRx_GetMovable(const unsigned char **flags, int *nbPrimitives);
/* float framebuffer of the last frame, W*H records */
int SolRx_GetPostProcessingBuffer(PostProcessingBuffer *buffer);
/* camera with an explicit angles.w (SolR_SetCamera forces 6400) */
void SolRx_SetCameraW(double eye_x, double eye_y, double eye_z, double dir_x, double dir_y, double dir_z,
                      double angle_x, double angle_y, double angle_z, double angle_w);
int SolRx_AddRectangle(double x, double y, double z, double w, double h, double d, int materialId);
int SolRx_SetSceneInfoExtras(int gradientBackground, int draftMode);

#ifdef __cplusplus
}
#endif

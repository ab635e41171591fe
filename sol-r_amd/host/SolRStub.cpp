/*
 * SolRStub.cpp - see SolRStub.h.  Call sequences follow the reference's
 * solr/SolRStub.cpp:48-164 (scene info staged in a global and applied in
 * SolR_InitializeKernel / SolR_RunKernel), :166-330 (primitives),
 * :368-424 (materials), :533-544 (boxes, lights).
 */
#include "SolRStub.h"

#include <cstring>
#include <string>

#include "../../include/solr_hip.h"
#include "FileMarshaller.h"
#include "GPUKernel.h"
#include "OBJReader.h"
#include "PDBReader.h"
#include "SWCReader.h"

using solr::SingletonKernel;

static SceneInfo gSceneInfoStub;
static PostProcessingInfo gPostProcessingInfoStub;

static int engineStatus()
{
    return SingletonKernel::kernel()->lastError() == 0 ? 0 : -1;
}

extern "C" {

int SolR_SetSceneInfo(int width, int height, int graphicsLevel, int nbRayIterations, double transparentColor,
                      double viewDistance, double shadowIntensity, double eyeSeparation, double bgColorR,
                      double bgColorG, double bgColorB, double bgColorA, int renderBoxes, int pathTracingIteration,
                      int maxPathTracingIterations, int frameBufferType, int timestamp, int atmosphericEffect,
                      int cameraType, int doubleSidedTriangles, int extendedGeometry, int advancedIllumination,
                      int skyboxSize, int skyboxMaterialId, double geometryEpsilon, double rayEpsilon)
{
    gSceneInfoStub.size.x = width;
    gSceneInfoStub.size.y = height;
    gSceneInfoStub.graphicsLevel = graphicsLevel;
    gSceneInfoStub.nbRayIterations = nbRayIterations;
    gSceneInfoStub.transparentColor = static_cast<float>(transparentColor);
    gSceneInfoStub.viewDistance = static_cast<float>(viewDistance);
    gSceneInfoStub.shadowIntensity = static_cast<float>(shadowIntensity);
    gSceneInfoStub.eyeSeparation = static_cast<float>(eyeSeparation);
    gSceneInfoStub.backgroundColor.x = static_cast<float>(bgColorR);
    gSceneInfoStub.backgroundColor.y = static_cast<float>(bgColorG);
    gSceneInfoStub.backgroundColor.z = static_cast<float>(bgColorB);
    gSceneInfoStub.backgroundColor.w = static_cast<float>(bgColorA);
    gSceneInfoStub.renderBoxes = renderBoxes;
    gSceneInfoStub.pathTracingIteration = pathTracingIteration;
    gSceneInfoStub.maxPathTracingIterations = maxPathTracingIterations;
    gSceneInfoStub.frameBufferType = frameBufferType;
    gSceneInfoStub.timestamp = timestamp;
    gSceneInfoStub.atmosphericEffect = atmosphericEffect;
    gSceneInfoStub.cameraType = cameraType;
    gSceneInfoStub.doubleSidedTriangles = doubleSidedTriangles;
    gSceneInfoStub.extendedGeometry = extendedGeometry;
    gSceneInfoStub.advancedIllumination = advancedIllumination;
    gSceneInfoStub.skyboxRadius = skyboxSize;
    gSceneInfoStub.skyboxMaterialId = skyboxMaterialId;
    gSceneInfoStub.geometryEpsilon = static_cast<float>(geometryEpsilon);
    gSceneInfoStub.rayEpsilon = static_cast<float>(rayEpsilon);
    return 0;
}

int SolR_SetPostProcessingInfo(int type, double param1, double param2, int param3)
{
    gPostProcessingInfoStub.type = type;
    gPostProcessingInfoStub.param1 = static_cast<float>(param1);
    gPostProcessingInfoStub.param2 = static_cast<float>(param2);
    gPostProcessingInfoStub.param3 = param3;
    return 0;
}

int SolR_SetDraftMode(int draft)
{
    gSceneInfoStub.draftMode = draft;
    return 0;
}

int SolRx_SetSceneInfoExtras(int gradientBackground, int draftMode)
{
    gSceneInfoStub.gradientBackground = gradientBackground;
    gSceneInfoStub.draftMode = draftMode;
    return 0;
}

int SolR_InitializeKernel(bool, int, int device)
{
    solr::GPUKernel *kernel = SingletonKernel::kernel();
    if (!kernel)
        return -1;
    gSceneInfoStub.pathTracingIteration = 0;
    kernel->setSceneInfo(gSceneInfoStub);
    kernel->setDeviceId(device);
    kernel->initBuffers();
    kernel->setFrame(0);
    return engineStatus();
}

int SolR_FinalizeKernel()
{
    SingletonKernel::destroy();
    return 0;
}

int SolR_ResetKernel()
{
    SingletonKernel::kernel()->resetAll();
    return 0;
}

void SolR_SetCamera(double eye_x, double eye_y, double eye_z, double dir_x, double dir_y, double dir_z,
                    double angle_x, double angle_y, double angle_z)
{
    SolRx_SetCameraW(eye_x, eye_y, eye_z, dir_x, dir_y, dir_z, angle_x, angle_y, angle_z, 6400.0);
}

void SolRx_SetCameraW(double eye_x, double eye_y, double eye_z, double dir_x, double dir_y, double dir_z,
                      double angle_x, double angle_y, double angle_z, double angle_w)
{
    const vec3f eye = solr::make_vec3f((float)eye_x, (float)eye_y, (float)eye_z);
    const vec3f dir = solr::make_vec3f((float)dir_x, (float)dir_y, (float)dir_z);
    const vec4f angles = solr::make_vec4f((float)angle_x, (float)angle_y, (float)angle_z, (float)angle_w);
    SingletonKernel::kernel()->setCamera(eye, dir, angles);
}

int SolRx_Render(double timer)
{
    solr::GPUKernel *kernel = SingletonKernel::kernel();
    kernel->setSceneInfo(gSceneInfoStub);
    kernel->setPostProcessingInfo(gPostProcessingInfoStub);
    kernel->render_begin(static_cast<float>(timer));
    kernel->render_end();
    return engineStatus();
}

/* Extension: frames in flight through SolR_RunKernel / SolRx_Render (GPUKernel::setFramesInFlight): with n > 1 a
 * call delivers the image of the frame n - 1 calls back while the newer ones render; SolRx_FlushFrames waits for
 * all of them, the next getBitmap / SolRx_GetBitmap is then the newest frame's. */
int SolRx_SetFramesInFlight(int n)
{
    SingletonKernel::kernel()->setFramesInFlight(n);
    return engineStatus();
}

/* Extension: the frame shared out over n devices of THIS process (the reference's occupancyParameters.x, which its
 * API cannot set: CudaKernel.cpp:90); returns the number in use after the change (fewer when fewer are there) or -1 */
int SolRx_SetGpuCount(int n)
{
    SingletonKernel::kernel()->setGpuCount(n);
    return engineStatus() == 0 ? SingletonKernel::kernel()->getGpuCount() : -1;
}

int SolRx_FlushFrames(void)
{
    SingletonKernel::kernel()->flushFrames();
    return engineStatus();
}

/* the image render_end delivered last, without a copy (valid until four more frames were rendered) */
const BitmapBuffer *SolRx_GetBitmap(void)
{
    return SingletonKernel::kernel()->getBitmap();
}

int SolR_RunKernel(double timer, BitmapBuffer *image)
{
    /* reference: SolRStub.cpp:154-164 - render_begin, render_end, then m_bitmap copied to the caller.  render_end(image)
     * is those last two steps (an engine with a device delivers straight into `image`, GPUKernel.h) */
    if (!image)
        return -1;
    solr::GPUKernel *kernel = SingletonKernel::kernel();
    kernel->setSceneInfo(gSceneInfoStub);
    kernel->setPostProcessingInfo(gPostProcessingInfoStub);
    kernel->render_begin(static_cast<float>(timer));
    kernel->render_end(image);
    return engineStatus();
}

int SolR_AddPrimitive(int type, int movable)
{
    int id = SingletonKernel::kernel()->addPrimitive(static_cast<PrimitiveType>(type));
    SingletonKernel::kernel()->setPrimitiveIsMovable(id, (movable == 1));
    return id;
}

int SolR_SetPrimitive(int index, double p0_x, double p0_y, double p0_z, double p1_x, double p1_y, double p1_z,
                      double p2_x, double p2_y, double p2_z, double size_x, double size_y, double size_z,
                      int materialId)
{
    SingletonKernel::kernel()->setPrimitive(index, (float)p0_x, (float)p0_y, (float)p0_z, (float)p1_x, (float)p1_y,
                                            (float)p1_z, (float)p2_x, (float)p2_y, (float)p2_z, (float)size_x,
                                            (float)size_y, (float)size_z, materialId);
    return 0;
}

int SolR_GetPrimitive(int index, double *p0_x, double *p0_y, double *p0_z, double *p1_x, double *p1_y, double *p1_z,
                      double *p2_x, double *p2_y, double *p2_z, double *size_x, double *size_y, double *size_z,
                      int *materialId)
{
    solr::CPUPrimitive *p = SingletonKernel::kernel()->getPrimitive(index);
    if (!p)
        return -1;
    *p0_x = p->p0.x; *p0_y = p->p0.y; *p0_z = p->p0.z;
    *p1_x = p->p1.x; *p1_y = p->p1.y; *p1_z = p->p1.z;
    *p2_x = p->p2.x; *p2_y = p->p2.y; *p2_z = p->p2.z;
    *size_x = p->size.x; *size_y = p->size.y; *size_z = p->size.z;
    *materialId = p->materialId;
    return 0;
}

int SolR_GetPrimitiveAt(int x, int y)
{
    return SingletonKernel::kernel()->getPrimitiveAt(x, y);
}

int SolR_GetPrimitiveCenter(int index, double *x, double *y, double *z)
{
    vec4f center = SingletonKernel::kernel()->getPrimitiveCenter(index);
    *x = center.x;
    *y = center.y;
    *z = center.z;
    return 0;
}

int SolR_RotatePrimitives(int, int, double rx, double ry, double rz, double ax, double ay, double az)
{
    vec3f rotationCenter = solr::make_vec3f((float)rx, (float)ry, (float)rz);
    vec4f angles = solr::make_vec4f((float)ax, (float)ay, (float)az);
    SingletonKernel::kernel()->rotatePrimitives(rotationCenter, angles);
    SingletonKernel::kernel()->compactBoxes(false);
    return 0;
}

int SolR_LoadMolecule(char *filename, int geometryType, double defaultAtomSize, double defaultStickSize,
                      int atomMaterialType, double scale)
{
    solr::PDBReader reader;
    const float s = static_cast<float>(scale);
    reader.loadAtomsFromFile(filename ? filename : "", *SingletonKernel::kernel(),
                             static_cast<solr::GeometryType>(geometryType), static_cast<float>(defaultAtomSize),
                             static_cast<float>(defaultStickSize), atomMaterialType, solr::make_vec4f(s, s, s));
    return (int)SingletonKernel::kernel()->getNbActivePrimitives();
}

int SolR_LoadOBJModel(char *filename, int materialId, int autoScale, double scale, int autoCenter, double *height)
{
    solr::OBJReader reader;
    const float s = static_cast<float>(scale);
    solr::CPUBoundingBox aabb, inAABB;
    const vec4f size = reader.loadModelFromFile(filename ? filename : "", *SingletonKernel::kernel(),
                                                solr::make_vec4f(0.f, 0.f, 0.f), autoScale == 1,
                                                solr::make_vec4f(s, s, s), true, materialId, false, autoCenter == 1, aabb,
                                                false, inAABB);
    if (height)
        *height = -size.y / 2.f;
    return (int)SingletonKernel::kernel()->getNbActivePrimitives();
}

int SolRx_LoadSWCMorphology(const char *filename, double px, double py, double pz, double sx, double sy, double sz,
                            double sw, int materialId)
{
    solr::SWCReader reader;
    reader.loadMorphologyFromFile(filename ? filename : "", *SingletonKernel::kernel(),
                                  solr::make_vec4f((float)px, (float)py, (float)pz),
                                  solr::make_vec4f((float)sx, (float)sy, (float)sz, (float)sw), materialId);
    return (int)reader.getMorphologies().size();
}

int SolR_SaveToFile(char *filename)
{
    solr::FileMarshaller fm;
    fm.saveToFile(*SingletonKernel::kernel(), filename ? filename : "");
    return (int)SingletonKernel::kernel()->getNbActivePrimitives();
}

int SolR_LoadFromFile(char *filename, double scale)
{
    solr::FileMarshaller fm;
    fm.loadFromFile(*SingletonKernel::kernel(), filename ? filename : "", solr::make_vec4f(0.f, 0.f, 0.f),
                    static_cast<float>(scale));
    return (int)SingletonKernel::kernel()->getNbActivePrimitives();
}

int SolRx_SetNbFrames(int nbFrames)
{
    SingletonKernel::kernel()->setNbFrames(nbFrames);
    return 0;
}

int SolRx_SetFrame(int frame)
{
    SingletonKernel::kernel()->setFrame(frame);
    return SingletonKernel::kernel()->getFrame();
}

int SolRx_MorphPrimitives()
{
    SingletonKernel::kernel()->morphPrimitives();
    return SingletonKernel::kernel()->getFrame();
}

int SolRx_PendingRotations()
{
    return (int)SingletonKernel::kernel()->nbPendingRotations();
}

int SolRx_SyncHost()
{
    SingletonKernel::kernel()->syncHost();
    return 0;
}

int SolRx_GetMovable(const unsigned char **flags, int *nbPrimitives)
{
    const std::vector<unsigned char> &v = SingletonKernel::kernel()->hostMovable();
    *flags = v.data();
    *nbPrimitives = (int)v.size();
    return 0;
}

int SolR_SetPrimitiveMaterial(int index, int materialId)
{
    SingletonKernel::kernel()->setPrimitiveMaterial(index, materialId);
    return 0;
}

int SolR_GetPrimitiveMaterial(int index)
{
    return SingletonKernel::kernel()->getPrimitiveMaterial(index);
}

int SolR_SetPrimitiveNormals(int index, double n0_x, double n0_y, double n0_z, double n1_x, double n1_y, double n1_z,
                             double n2_x, double n2_y, double n2_z)
{
    SingletonKernel::kernel()->setPrimitiveNormals(index, solr::make_vec3f((float)n0_x, (float)n0_y, (float)n0_z),
                                                   solr::make_vec3f((float)n1_x, (float)n1_y, (float)n1_z),
                                                   solr::make_vec3f((float)n2_x, (float)n2_y, (float)n2_z));
    return 0;
}

int SolR_SetPrimitiveTextureCoordinates(int index, double t0_x, double t0_y, double t1_x, double t1_y, double t2_x,
                                        double t2_y)
{
    SingletonKernel::kernel()->setPrimitiveTextureCoordinates(index, solr::make_vec2f((float)t0_x, (float)t0_y),
                                                              solr::make_vec2f((float)t1_x, (float)t1_y),
                                                              solr::make_vec2f((float)t2_x, (float)t2_y));
    return 0;
}

int SolR_AddMaterial()
{
    return SingletonKernel::kernel()->addMaterial();
}

int SolR_SetMaterial(int index, double color_r, double color_g, double color_b, double noise, double reflection,
                     double refraction, int procedural, int wireframe, int wireframeDepth, double transparency,
                     double opacity, int diffuseTextureId, int normalTextureId, int bumpTextureId,
                     int specularTextureId, int reflectionTextureId, int transparencyTextureId,
                     int ambientOcclusionTextureId, double specValue, double specPower, double specCoef,
                     double innerIllumination, double illuminationDiffusion, double illuminationPropagation,
                     int fastTransparency)
{
    SingletonKernel::kernel()->setMaterial(
        index, (float)color_r, (float)color_g, (float)color_b, (float)noise, (float)reflection, (float)refraction,
        (procedural == 1), (wireframe == 1), wireframeDepth, (float)transparency, (float)opacity, diffuseTextureId,
        normalTextureId, bumpTextureId, specularTextureId, reflectionTextureId, transparencyTextureId,
        ambientOcclusionTextureId, (float)specValue, (float)specPower, (float)specCoef, (float)innerIllumination,
        (float)illuminationDiffusion, (float)illuminationPropagation, (fastTransparency == 1));
    return 0;
}

int SolR_CompactBoxes(bool update)
{
    return SingletonKernel::kernel()->compactBoxes(update);
}

int SolR_GetLight(int index)
{
    return SingletonKernel::kernel()->getLight(index);
}

int SolR_SetTexture(int index, const unsigned char *pixels, int width, int height, int depth, int textureType)
{
    if (!pixels || index < 0 || index >= NB_MAX_TEXTURES || width <= 0 || height <= 0 || depth <= 0)
        return -1;
    TextureInfo info;
    memset(&info, 0, sizeof(info));
    info.buffer = const_cast<unsigned char *>(pixels);
    info.size.x = width;
    info.size.y = height;
    info.size.z = depth;
    info.type = textureType;
    SingletonKernel::kernel()->setTexture(index, info);
    return 0;
}

int SolR_GetTextureSize(int index, int *width, int *height, int *depth)
{
    if (index < 0 || index >= NB_MAX_TEXTURES)
        return -1;
    TextureInfo &t = SingletonKernel::kernel()->getTextureInformation(index);
    *width = t.size.x;
    *height = t.size.y;
    *depth = t.size.z;
    return 0;
}

int SolR_GetNbTextures(int *nbTextures)
{
    *nbTextures = SingletonKernel::kernel()->getNbActiveTextures();
    return 0;
}

int SolR_GetMaterial(int index, double *color_r, double *color_g, double *color_b, double *noise, double *reflection,
                     double *refraction, int *procedural, int *wireframe, int *wireframeDepth, double *transparency,
                     double *opacity, int *diffuseTextureId, int *normalTextureId, int *bumpTextureId,
                     int *specularTextureId, int *reflectionTextureId, int *transparencyTextureId,
                     int *ambientOcclusionTextureId, double *specValue, double *specPower, double *specCoef,
                     double *innerIllumination, double *illuminationDiffusion, double *illuminationPropagation,
                     int *fastTransparency)
{
    /* where GPUKernel::setMaterial put each attribute (GPUKernel.cpp:1780-1909 of the reference) */
    const Material *m = SingletonKernel::kernel()->getMaterial(index);
    if (!m)
        return -1;
    auto d = [](double *out, float v) { if (out) *out = static_cast<double>(v); };
    auto n = [](int *out, int v) { if (out) *out = v; };
    d(color_r, m->color.x), d(color_g, m->color.y), d(color_b, m->color.z);
    d(noise, m->innerIllumination.w);
    d(reflection, m->reflection), d(refraction, m->refraction);
    d(transparency, m->transparency), d(opacity, m->opacity);
    n(fastTransparency, m->attributes.x == 1), n(procedural, m->attributes.y == 1);
    n(wireframe, m->attributes.z == 1), n(wireframeDepth, m->attributes.w);
    n(diffuseTextureId, m->textureIds.x), n(normalTextureId, m->textureIds.y);
    n(bumpTextureId, m->textureIds.z), n(specularTextureId, m->textureIds.w);
    n(reflectionTextureId, m->advancedTextureIds.x), n(transparencyTextureId, m->advancedTextureIds.y);
    n(ambientOcclusionTextureId, m->advancedTextureIds.z);
    d(specValue, m->specular.x), d(specPower, m->specular.y), d(specCoef, m->specular.w);
    d(innerIllumination, m->innerIllumination.x), d(illuminationDiffusion, m->innerIllumination.y);
    d(illuminationPropagation, m->innerIllumination.z);
    return 0;
}

int SolR_GetTexture(int index, BitmapBuffer *image)
{
    if (index < 0 || index >= (int)SingletonKernel::kernel()->getNbActiveTextures() || !image)
        return 1;
    TextureInfo info;
    memset(&info, 0, sizeof(info));
    SingletonKernel::kernel()->getTexture(index, info);
    if (info.buffer && info.size.z >= 3)
    {
        const int len = info.size.x * info.size.y * info.size.z;
        for (int i = 0; i + 2 < len; i += info.size.z)
        {
            image[i] = info.buffer[i + 2];
            image[i + 1] = info.buffer[i + 1];
            image[i + 2] = info.buffer[i];
        }
    }
    return 0;
}

int SolR_RotatePrimitive(int, double, double, double, double, double, double)
{
    return 0;
}

int SolR_RecompileKernels(char *)
{
    return 0;
}

/* ---------------------------------------------------------------- extensions */

int SolRx_SelectEngine(const char *name)
{
    SingletonKernel::selectEngine(name);
    return 0;
}

int SolRx_HostBuild(int hostOnly)
{
    SingletonKernel::kernel()->setHostBuildOnly(hostOnly != 0);
    return 0;
}

int SolRx_SetDeterministic(long seed)
{
    SingletonKernel::kernel()->setDeterministic(seed);
    return 0;
}

int SolRx_LastError(char *buf, int len)
{
    std::string message;
    int code = SingletonKernel::kernel()->lastError(&message);
    if (buf && len > 0)
    {
        strncpy(buf, message.c_str(), len - 1);
        buf[len - 1] = 0;
    }
    return code;
}

int SolRx_GetBoxes(const BoundingBox **boxes, int *nbBoxes)
{
    const std::vector<BoundingBox> &v = SingletonKernel::kernel()->hostBoxes();
    *boxes = v.data();
    *nbBoxes = (int)SingletonKernel::kernel()->getNbActiveBoxes();
    return 0;
}

int SolRx_GetPrimitives(const Primitive **primitives, int *nbPrimitives)
{
    const std::vector<Primitive> &v = SingletonKernel::kernel()->hostPrimitives();
    *primitives = v.data();
    *nbPrimitives = (int)SingletonKernel::kernel()->getNbActivePrimitives();
    return 0;
}

int SolRx_GetLights(const LightInformation **lights, int *nbLights, int *nbLamps)
{
    *lights = SingletonKernel::kernel()->hostLights().data();
    *nbLights = SingletonKernel::kernel()->lightInformationSize();
    *nbLamps = (int)SingletonKernel::kernel()->getNbActiveLamps();
    return 0;
}

int SolRx_GetMaterials(const Material **materials, int *nbMaterials)
{
    SingletonKernel::kernel()->realignTexturesAndMaterials();
    *materials = SingletonKernel::kernel()->hostMaterials();
    *nbMaterials = (int)SingletonKernel::kernel()->getNbActiveMaterials() + 1;
    return 0;
}

int SolRx_GetRandoms(const float **randoms, int *nbRandoms)
{
    const std::vector<RandomBuffer> &v = SingletonKernel::kernel()->hostRandoms();
    *randoms = v.data();
    *nbRandoms = (int)v.size();
    return 0;
}

int SolRx_GetTextureAtlas(const unsigned char **atlas, long *nbBytes)
{
    const std::vector<BitmapBuffer> &v = SingletonKernel::kernel()->hostTextureAtlas();
    *atlas = v.empty() ? nullptr : v.data();
    *nbBytes = (long)v.size();
    return 0;
}

int SolRx_GetPrimitiveIds(const PrimitiveXYIdBuffer **ids, int *nbPixels)
{
    *ids = SingletonKernel::kernel()->hostPrimitiveIds();
    SceneInfo &si = SingletonKernel::kernel()->getSceneInfo();
    *nbPixels = si.size.x * si.size.y;
    return 0;
}

int SolRx_GetSceneInfo(SceneInfo *sceneInfo, PostProcessingInfo *postProcessingInfo, float eye[3], float dir[3],
                       float angles[4])
{
    solr::GPUKernel *k = SingletonKernel::kernel();
    if (sceneInfo)
        *sceneInfo = gSceneInfoStub;
    if (postProcessingInfo)
        *postProcessingInfo = gPostProcessingInfoStub;
    if (eye)
    {
        eye[0] = k->getViewPos().x; eye[1] = k->getViewPos().y; eye[2] = k->getViewPos().z;
    }
    if (dir)
    {
        dir[0] = k->getViewDir().x; dir[1] = k->getViewDir().y; dir[2] = k->getViewDir().z;
    }
    if (angles)
    {
        angles[0] = k->getViewAngles().x; angles[1] = k->getViewAngles().y;
        angles[2] = k->getViewAngles().z; angles[3] = k->getViewAngles().w;
    }
    return 0;
}

int SolRx_GetTreeDepth()
{
    return (int)SingletonKernel::kernel()->treeDepth();
}

int SolRx_GetPostProcessingBuffer(PostProcessingBuffer *buffer)
{
    if (!buffer)
        return -1;
    solr_hip_d2h_postprocessing(buffer);
    return engineStatus();
}

int SolRx_AddRectangle(double x, double y, double z, double w, double h, double d, int materialId)
{
    return SingletonKernel::kernel()->addRectangle((float)x, (float)y, (float)z, (float)w, (float)h, (float)d,
                                                   materialId);
}
}

/*
 * solr_types.h - plain-old-data records shared by the host layer, the C-ABI
 * boundary (solr_hip.h), the HIP kernels and the CPU oracle.
 *
 * These are layout-compatible re-declarations of the records the reference
 * engine exchanges with its device code (reference: solr/types.h:93-97,
 * 140-189, 210-286, 301-329 and solr/Consts.h:27-48, CUDA flavour where
 * vec3f is a 12-byte float3 and vec4f/vec4i are 16-byte aligned).  Host
 * applications built against the reference headers can hand their arrays to
 * this library unchanged; static asserts at the bottom pin every size and the
 * offsets the kernels rely on.
 *
 * C and C++ compatible; no CUDA/HIP vector types are used so that the file can
 * be included from gcc (oracle), g++ (host) and hipcc (device) alike.
 */
#ifndef SOLR_TYPES_H
#define SOLR_TYPES_H

#include <stddef.h>

/* gcc, g++, clang and hipcc all accept the attribute between the struct
 * keyword and the tag */
#define SOLR_ALIGN(n) __attribute__((aligned(n)))

/* ---- vector records (reference: types.h:58-66, CUDA vector_types layout) */
typedef float vec1f;
typedef int vec1i;
typedef struct SOLR_ALIGN(8) vec2f_s { float x, y; } vec2f;
typedef struct vec3f_s { float x, y, z; } vec3f;
typedef struct SOLR_ALIGN(16) vec4f_s { float x, y, z, w; } vec4f;
typedef struct SOLR_ALIGN(8) vec2i_s { int x, y; } vec2i;
typedef struct vec3i_s { int x, y, z; } vec3i;
typedef struct SOLR_ALIGN(16) vec4i_s { int x, y, z, w; } vec4i;
typedef vec4i PrimitiveXYIdBuffer; /* x prim index, y iterations, z emissive*256, w shadow*255 */

typedef unsigned char BitmapBuffer;
typedef float RandomBuffer;
typedef int Lamp;

/* ---- limits (reference: Consts.h:27-48) */
#define SOLR_MAX_GPU_COUNT 32
#define NB_MAX_ITERATIONS 10
#define BOUNDING_BOXES_TREE_DEPTH 64
#define NB_MAX_BOXES 2500000
#define NB_MAX_PRIMITIVES 2500000
#define NB_MAX_LAMPS 512
#define NB_MAX_MATERIALS (65506 + 30)
#define NB_MAX_TEXTURES 512
#define NB_MAX_FRAMES 512
#define NB_MAX_LIGHTINFORMATIONS 512
#define MAX_BITMAP_WIDTH 1920
#define MAX_BITMAP_HEIGHT 1080
#define MAX_BITMAP_SIZE (MAX_BITMAP_WIDTH * MAX_BITMAP_HEIGHT)
#define MATERIAL_NONE (-1)
#define TEXTURE_NONE (-1)
#define TEXTURE_MANDELBROT (-2)
#define TEXTURE_JULIA (-3)
#define SOLR_COLOR_DEPTH 3 /* gColorDepth */
#define SOLR_PI 3.14159265358979323846f
#define STANDARD_LUNINANCE_STRENGTH 0.1f /* sic, Consts.h:52 */
#define SKYBOX_LUNINANCE_STRENGTH 0.2f

/* named materials used by the scene helpers (reference: Consts.h:60-116) */
#define RANDOM_MATERIALS_OFFSET 1000
#define DEFAULT_LIGHT_MATERIAL (NB_MAX_MATERIALS - 34)
#define WHITE_MATERIAL (NB_MAX_MATERIALS - 35)
#define RED_MATERIAL (NB_MAX_MATERIALS - 36)
#define GREEN_MATERIAL (NB_MAX_MATERIALS - 37)
#define BLUE_MATERIAL (NB_MAX_MATERIALS - 38)

/* ---- enums (reference: types.h:99-137, 192-207, 289-320); stored as int */
enum CameraType { ctPerspective = 0, ctOrthographic = 1, ctAnaglyph = 2, ctVR = 3, ctPanoramic = 4,
                  ctAntialiazed = 5, ctVolumeRendering = 6 };
enum FrameBufferType { ftRGB = 0, ftBGR = 1 };
enum AdvancedIllumination { aiNone = 0, aiBasic = 1, aiFull = 2, aiRandomIllumination = 3 };
enum GraphicsLevel { glNoShading = 0, glPhong = 1, glPhongAndBlinn = 2, glReflectionsAndRefractions = 3, glFull = 4 };
enum AtmosphericEffect { aeNone = 0, aeFog = 1 };
enum PrimitiveType { ptSphere = 0, ptCylinder = 1, ptTriangle = 2, ptCheckboard = 3, ptCamera = 4, ptXYPlane = 5,
                     ptYZPlane = 6, ptXZPlane = 7, ptMagicCarpet = 8, ptEnvironment = 9, ptEllipsoid = 10,
                     ptQuad = 11, ptCone = 12 };
enum TextureType { tex_diffuse = 0, tex_bump, tex_normal, tex_ambient_occlusion, tex_reflective, tex_specular,
                   tex_transparent };
enum PostProcessingType { ppe_none = 0, ppe_depthOfField, ppe_ambientOcclusion, ppe_radiosity, ppe_filter,
                          ppe_cartoon };

/* ---- per-pixel float framebuffer record (reference: types.h:93-97) */
typedef struct PostProcessingBuffer_s
{
    vec4f colorInfo; /* xyz colour (or accumulated sum), w first-hit distance */
    vec4f sceneInfo; /* xyz last sample */
} PostProcessingBuffer;

/* ---- scene description, passed by value into every kernel (types.h:140-168) */
typedef struct SOLR_ALIGN(16) SceneInfo_s
{
    vec2i size;
    int cameraType;
    int graphicsLevel;
    vec1i nbRayIterations;
    vec1f transparentColor;
    vec1f viewDistance;
    vec1f shadowIntensity;
    vec1f eyeSeparation;
    vec1i renderBoxes;
    vec1i pathTracingIteration;
    vec1i maxPathTracingIterations;
    int frameBufferType;
    vec1i timestamp;
    int atmosphericEffect;
    vec1i doubleSidedTriangles;
    vec1i extendedGeometry;
    int advancedIllumination;
    vec1i draftMode;
    vec1i skyboxRadius;
    vec1i skyboxMaterialId;
    vec1i gradientBackground;
    vec1f geometryEpsilon;
    vec1f rayEpsilon;
    vec4f backgroundColor;
} SceneInfo;

/* ---- light record (types.h:183-189) */
typedef struct SOLR_ALIGN(16) LightInformation_s
{
    vec1i primitiveId;
    vec1i materialId;
    vec3f location;
    vec4f color; /* w = intensity */
} LightInformation;

/* ---- material record (types.h:210-251) */
typedef struct SOLR_ALIGN(16) Material_s
{
    vec4f innerIllumination; /* x emission, y diffusion, z range, w noise */
    vec4f color;             /* rgb, w view noise */
    vec4f specular;          /* x value, y power, z -, w coef */
    vec1f reflection;
    vec1f refraction;
    vec1f transparency;
    vec1f opacity;
    vec4i attributes;            /* x fast transparency, y procedural, z wireframe, w wireframe width */
    vec4i textureMapping;        /* x width, y height, z deprecated, w depth */
    vec4i textureOffset;         /* diffuse, normal, bump, specular */
    vec4i textureIds;            /* diffuse, normal, bump, specular */
    vec4i advancedTextureOffset; /* reflection, transparency, ambient occlusion, - */
    vec4i advancedTextureIds;
    vec2f mappingOffset;
} Material;

/* ---- flattened box-tree node (types.h:254-260) */
typedef struct SOLR_ALIGN(16) BoundingBox_s
{
    vec3f parameters[2];   /* min corner, max corner */
    vec1i nbPrimitives;    /* 0 for inner nodes */
    vec1i startIndex;      /* first primitive (leaves) or depth (inner nodes) */
    vec2i indexForNextBox; /* .x = size of the node's subtree (1 for leaves) */
} BoundingBox;

/* ---- primitive record (types.h:264-286) */
typedef struct SOLR_ALIGN(16) Primitive_s
{
    vec3f p0, p1, p2;
    vec3f n0, n1, n2;
    vec3f size;
    vec1i type;
    vec1i index;
    vec1i materialId;
    vec2f vt0, vt1, vt2;
} Primitive;

/* ---- texture descriptor (types.h:301-308) */
typedef struct SOLR_ALIGN(16) TextureInfo_s
{
    unsigned char *buffer;
    vec1i offset;
    vec3i size;
    int type;
} TextureInfo;

/* ---- post-processing request (types.h:323-329) */
typedef struct SOLR_ALIGN(16) PostProcessingInfo_s
{
    vec1i type;
    vec1f param1;
    vec1f param2;
    vec1i param3;
} PostProcessingInfo;

#ifdef __cplusplus
#define SOLR_SA(c, m) static_assert(c, m)
#else
#define SOLR_SA(c, m) _Static_assert(c, m)
#endif
SOLR_SA(sizeof(SceneInfo) == 112, "SceneInfo layout");
SOLR_SA(sizeof(LightInformation) == 48, "LightInformation layout");
SOLR_SA(sizeof(Material) == 176, "Material layout");
SOLR_SA(sizeof(BoundingBox) == 48, "BoundingBox layout");
SOLR_SA(sizeof(Primitive) == 128, "Primitive layout");
SOLR_SA(sizeof(PostProcessingInfo) == 16, "PostProcessingInfo layout");
SOLR_SA(sizeof(PostProcessingBuffer) == 32, "PostProcessingBuffer layout");
SOLR_SA(sizeof(PrimitiveXYIdBuffer) == 16, "PrimitiveXYIdBuffer layout");
SOLR_SA(offsetof(SceneInfo, backgroundColor) == 96, "SceneInfo.backgroundColor");
SOLR_SA(offsetof(SceneInfo, geometryEpsilon) == 88, "SceneInfo.geometryEpsilon");
SOLR_SA(offsetof(LightInformation, location) == 8 && offsetof(LightInformation, color) == 32, "LightInformation");
SOLR_SA(offsetof(Material, attributes) == 64 && offsetof(Material, mappingOffset) == 160, "Material offsets");
SOLR_SA(offsetof(BoundingBox, nbPrimitives) == 24 && offsetof(BoundingBox, indexForNextBox) == 32, "BoundingBox");
SOLR_SA(offsetof(Primitive, size) == 72 && offsetof(Primitive, type) == 84 && offsetof(Primitive, vt0) == 96,
        "Primitive offsets");

#endif /* SOLR_TYPES_H */

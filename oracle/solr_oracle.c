/*
 * solr_oracle.c - CPU restatement of Sol-R's per-pixel rendering path.
 *
 * TEST INFRASTRUCTURE ONLY (see solr_oracle.h).  Parity status: pinned bit for
 * bit, function by function, to outputs of the reference's own OpenCL engine
 * (oracle/ref_probes.cl, tests/test_reference_probes.py) through the OpenCL
 * dialect below; the statements that differ in the CUDA engine it restates by
 * default are marked `g_cl` and cite both files (solr_oracle.h).
 *
 * Every function cites the reference file:line it restates.  "ref:" paths are
 * relative to the reference tree, with
 *   CRT = solr/engines/cuda/CudaRayTracer.cu
 *   GI  = solr/engines/cuda/GeometryIntersections.cuh
 *   GS  = solr/engines/cuda/GeometryShaders.cuh
 *   TM  = solr/engines/cuda/TextureMapping.cuh
 *   VU  = solr/engines/cuda/VectorUtils.cuh
 *   HM  = solr/engines/cuda/helper_math.h
 *
 * Numerical definition (DESIGN.md "numerics"): IEEE-754 binary32 for every
 * operation in source order, no fused multiply-add (build with
 * -ffp-contract=off), correctly rounded division and square root,
 * rsqrtf(x) = 1.0f / sqrtf(x) (HM:62-65, the host definition; the CUDA
 * build's --use_fast_math approximations are not reproducible and are not the
 * target), min/max on floats = fminf/fmaxf (the CUDA overloads), libm
 * transcendentals in binary32 (powf, sinf, cosf, atan2f, asinf).
 */
#include "solr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* The specular power.  powf is only specified to an error bound: glibc's is within 1 ULP but not always
 * the correctly rounded value, CUDA's within 2 ULP; the engine evaluates it in binary64 and rounds once.
 * SOLR_ORACLE_CORRECTLY_ROUNDED_POW=1 makes the restatement do the same, which tells a powf rounding
 * difference (gone with it) from a real one when a frame is 2 ULP off (tools/fuzz_parity.py). */
/* float -> int as the reference's device code converts (cvt.rzi.s32.f32 on its GPU, v_cvt_i32_f32 on this
 * one): truncation, saturating at the ends of the range, NaN -> 0.  A C cast of such a value is undefined
 * (x86 yields INT_MIN for all of them), and the texture mappers do reach them: a grazing triangle's
 * barycentric sum is 0. */
static int f2i(float v)
{
    if (v != v)
        return 0;
    if (v >= 2147483648.f)
        return 2147483647;
    if (v <= -2147483648.f)
        return (-2147483647 - 1);
    return (int)v;
}

/* Which pixels of a frame met a libm result that is not the correctly rounded one (oracle_set_misround_mask):
 * a whole-frame comparison against the oracle AS PINNED (libm's binary32 routines) can then hold the engine to
 * <= 1 ULP everywhere except on a counted set of pixels, each of which is shown - by evaluating the same call in
 * binary64 and rounding once, here, at the call - to have gone through such a result (bit 0: powf of the Blinn
 * term, bit 1: sinf / cosf / atan2f / asinf).  Off (NULL) by default: the timed CPU baseline pays nothing. */
static unsigned char *g_misroundMask = NULL;
static long g_misroundLen = 0;
static _Thread_local long t_pixel = -1; /* strip-local index of the pixel being rendered */
void oracle_set_misround_mask(unsigned char *mask, long n)
{
    g_misroundMask = mask;
    g_misroundLen = mask ? n : 0;
}
static inline float noteMisround(float libm, double exact, int bit)
{
    const float rounded = (float)exact;
    if (memcmp(&libm, &rounded, sizeof(float)) != 0 && !(libm != libm && rounded != rounded) && t_pixel >= 0 &&
        t_pixel < g_misroundLen)
        g_misroundMask[t_pixel] |= (unsigned char)bit; /* a byte per pixel, written by that pixel's thread alone */
    return libm;
}

static int correctlyRoundedPow = -1;
static float specularPower(float base, float exponent)
{
    if (correctlyRoundedPow < 0)
    {
        const char *e = getenv("SOLR_ORACLE_CORRECTLY_ROUNDED_POW");
        correctlyRoundedPow = (e && atoi(e) != 0) ? 1 : 0;
    }
    if (correctlyRoundedPow)
        return (float)pow((double)base, (double)exponent);
    if (g_misroundMask)
        return noteMisround(powf(base, exponent), pow((double)base, (double)exponent), 1);
    return powf(base, exponent);
}

/* The same for the trigonometry of the procedural sphere, the Julia/Mandelbrot parameters and the sphere / skybox
 * UV maps: sinf, cosf, atan2f, asinf are specified to an error bound only (CUDA: 2 ULP, glibc: < 1 ULP, not
 * always the nearest value); the engine evaluates them in binary64 and rounds once.  With
 * oracle_set_rounded_transcendentals(1) (or SOLR_ORACLE_CORRECTLY_ROUNDED_TRIG=1; the call also switches the
 * power above) the restatement does the same and the frames that depend on them can be compared EXACTLY
 * (tests/test_gpu_parity.py::test_every_primitive_type, test_textures) instead of within a handful of pixels.
 * The camera's six values (makeTrig) are computed by the host on both sides and are not affected. */
static int roundedTrig = -1;
static int roundedTrigOn(void)
{
    if (roundedTrig < 0)
    {
        const char *e = getenv("SOLR_ORACLE_CORRECTLY_ROUNDED_TRIG");
        roundedTrig = (e && atoi(e) != 0) ? 1 : 0;
    }
    return roundedTrig;
}
static float sin_o(float a)
{
    if (roundedTrigOn())
        return (float)sin((double)a);
    return g_misroundMask ? noteMisround(sinf(a), sin((double)a), 2) : sinf(a);
}
static float cos_o(float a)
{
    if (roundedTrigOn())
        return (float)cos((double)a);
    return g_misroundMask ? noteMisround(cosf(a), cos((double)a), 2) : cosf(a);
}
static float atan2_o(float a, float b)
{
    if (roundedTrigOn())
        return (float)atan2((double)a, (double)b);
    return g_misroundMask ? noteMisround(atan2f(a, b), atan2((double)a, (double)b), 2) : atan2f(a, b);
}
static float asin_o(float a)
{
    if (roundedTrigOn())
        return (float)asin((double)a);
    return g_misroundMask ? noteMisround(asinf(a), asin((double)a), 2) : asinf(a);
}
void oracle_set_rounded_transcendentals(int on)
{
    roundedTrig = on ? 1 : 0;
    correctlyRoundedPow = on ? 1 : 0;
}
int oracle_get_rounded_transcendentals(void) { return roundedTrigOn() && correctlyRoundedPow == 1; }

/* ---- dialect ------------------------------------------------------------------------------------
 * The reference keeps the same per-pixel path twice: the CUDA engine, which this file restates and the
 * product matches (dialect 0, the default), and an older sibling, the OpenCL engine
 * (solr/engines/opencl/RayTracer.cl, "CL" below), which is the only form of the path that can be BUILT
 * in this image and therefore the only source of reference-produced outputs (oracle/_ref, oracle/ref_probes.cl).
 * The two have drifted apart in a few dozen statements.  oracle_set_dialect(1) switches exactly those
 * statements to the OpenCL engine's form - every switch is an `if (g_cl)` next to the CUDA form, citing both
 * files - so that the restatement can be compared with the reference's own functions BIT FOR BIT
 * (tests/test_reference_probes.py).  Everything not under such a switch is shared by both dialects and is
 * thereby pinned to reference output; what a switch selects in dialect 0 is the cited CUDA statement.
 * Test infrastructure only; never set by the parity tests of the product. */
static int g_cl = 0;
/* Every place where the two dialects part reads the switch through DIALECT(site).  A build with
 * -DORACLE_SITE_COVERAGE (oracle/Makefile, target `coverage`: libsolr_oracle_cov.so, used by
 * tests/test_cuda_text_model.py alone) counts, per site, how often it was evaluated in each dialect, so that a test
 * can show that its cases execute the CUDA arm of every switch; the normal build reads the flag and nothing else. */
#define ORACLE_NB_SITES 36
#ifdef ORACLE_SITE_COVERAGE
static unsigned long g_siteHits[ORACLE_NB_SITES][2];
static int g_siteFlipped = -1; /* that one site reads the other dialect (oracle_flip_site): does any test notice? */
#define DIALECT(site) (g_siteHits[site][g_cl ? 1 : 0]++, ((site) == g_siteFlipped) ? !g_cl : g_cl)
#else
#define DIALECT(site) g_cl
#endif
/* counting build only: site >= 0 makes that one switch read the other dialect, -1 ends it.  The tests flip every
 * switch in turn and demand that some case then differs from the model: a switch no case can tell apart is not pinned. */
void oracle_flip_site(int site)
{
#ifdef ORACLE_SITE_COVERAGE
    g_siteFlipped = site;
#else
    (void)site;
#endif
}
/* hits[2 * site + dialect]; returns the number of sites, or 0 in a build without the counters */
int oracle_site_hits(unsigned long *hits, int capacity, int reset)
{
#ifdef ORACLE_SITE_COVERAGE
    for (int i = 0; i < ORACLE_NB_SITES && 2 * i + 1 < capacity; ++i)
    {
        hits[2 * i] = g_siteHits[i][0];
        hits[2 * i + 1] = g_siteHits[i][1];
    }
    if (reset)
        memset(g_siteHits, 0, sizeof(g_siteHits));
    return ORACLE_NB_SITES;
#else
    (void)hits;
    (void)capacity;
    (void)reset;
    return 0;
#endif
}
void oracle_set_dialect(int openclEngine)
{
    g_cl = openclEngine ? 1 : 0;
}
int oracle_get_dialect(void)
{
    return g_cl;
}

typedef vec3f v3;

typedef struct
{
    float x, y, z;
} c3; /* colour triple; the reference carries float4 colours whose w never reaches the output */

typedef struct
{
    v3 origin;
    v3 direction;
    v3 inv_direction;
    int sx, sy, sz;
} Ray; /* ref: solr/types.h:171-178 */

typedef struct
{
    unsigned long long closest, shadow, boxes, prims;
    int randomFault;
} Stats;

/* ---- helper_math.h vector operators (HM:349-365,579-593,804-818,997,1248,1291,1309) */
static inline v3 V(float x, float y, float z)
{
    v3 r = {x, y, z};
    return r;
}
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float b) { return V(a.x * b, a.y * b, a.z * b); }
static inline v3 vdivs(v3 a, float b) { return V(a.x / b, a.y / b, a.z / b); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float vlength(v3 v) { return sqrtf(vdot(v, v)); }
static inline float rsqrt_host(float x) { return 1.0f / sqrtf(x); } /* HM:62-65 */
static inline v3 vnormalize(v3 v)
{
    float invLen = rsqrt_host(vdot(v, v)); /* HM:1309-1313 */
    return vscale(v, invLen);
}

/* ref VU:45-52 */
static inline v3 crossProduct(v3 b, v3 c)
{
    v3 a;
    a.x = b.y * c.z - b.z * c.y;
    a.y = b.z * c.x - b.x * c.z;
    a.z = b.x * c.y - b.y * c.x;
    return a;
}

/* ref VU:31-42 (colour part) */
static inline float sat1(float v)
{
    v = (v < 0.f) ? 0.f : v;
    v = (v > 1.f) ? 1.f : v;
    return v;
}
static inline void saturate3(c3 *v)
{
    v->x = sat1(v->x);
    v->y = sat1(v->y);
    v->z = sat1(v->z);
}

/* ref VU:61-64: r = i - 2.f * dot(i, n) * n */
static inline v3 vectorReflection(v3 i, v3 n)
{
    float k = 2.f * vdot(i, n);
    return vsub(i, vscale(n, k));
}

/* ref VU:73-87 */
static inline v3 vectorRefraction(v3 incident, float n1, v3 normal, float n2)
{
    v3 refracted = incident;
    if (n2 != 0.f)
    {
        float eta = n1 / n2;
        float c1 = -vdot(incident, normal);
        float cs2 = 1.f - eta * eta * (1.f - c1 * c1);
        if (cs2 >= 0.f)
        {
            float k = eta * c1 - sqrtf(cs2);
            refracted = vadd(vscale(incident, eta), vscale(normal, k));
        }
    }
    return refracted;
}

/* ref VU:92-95 */
static inline v3 project(v3 A, v3 B) { return vscale(B, vdot(A, B) / vdot(B, B)); }

/* ref VU:104-142; the six sinf/cosf values depend only on the camera angles */
typedef struct
{
    float cx, cy, cz, sx, sy, sz;
} Trig;
static Trig makeTrig(const float angles[3])
{
    Trig t;
    t.cx = cosf(angles[0]);
    t.cy = cosf(angles[1]);
    t.cz = cosf(angles[2]);
    t.sx = sinf(angles[0]);
    t.sy = sinf(angles[1]);
    t.sz = sinf(angles[2]);
    return t;
}
static v3 vectorRotation(v3 v, v3 c, const Trig *t)
{
    float vx = v.x - c.x, vy = v.y - c.y, vz = v.z - c.z;
    float rx = vx, ry, rz;
    /* X axis */
    ry = vy * t->cx - vz * t->sx;
    rz = vy * t->sx + vz * t->cx;
    vy = ry;
    vz = rz;
    /* Y axis */
    rz = vz * t->cy - vx * t->sy;
    rx = vz * t->sy + vx * t->cy;
    vz = rz;
    vx = rx;
    /* Z axis */
    rx = vx * t->cz - vy * t->sz;
    ry = vx * t->sz + vy * t->cz;
    return V(rx + c.x, ry + c.y, rz + c.z);
}

void oracle_vector_rotation(float v[3], const float center[3], const float angles[3])
{
    Trig t = makeTrig(angles);
    v3 r = vectorRotation(V(v[0], v[1], v[2]), V(center[0], center[1], center[2]), &t);
    v[0] = r.x;
    v[1] = r.y;
    v[2] = r.z;
}

/* ---- random buffer access with bounds check (the reference reads past the
 * end at full HD; the oracle reports that instead) */
static inline float rnd(const OracleScene *s, long i, Stats *st)
{
    if (!s->randoms || i < 0 || i >= s->nbRandoms)
    {
        st->randomFault = 1;
        return 0.f;
    }
    return s->randoms[i];
}

/* ref GI:36-44 */
static inline void computeRayAttributes(Ray *ray)
{
    const float zero = DIALECT(0) ? 0.f : 1.f; /* CL:365-367 gives a zero component 0.f, GI:39-41 1.f */
    ray->inv_direction.x = ray->direction.x != 0.f ? 1.f / ray->direction.x : zero;
    ray->inv_direction.y = ray->direction.y != 0.f ? 1.f / ray->direction.y : zero;
    ray->inv_direction.z = ray->direction.z != 0.f ? 1.f / ray->direction.z : zero;
    ray->sx = (ray->inv_direction.x < 0);
    ray->sy = (ray->inv_direction.y < 0);
    ray->sz = (ray->inv_direction.z < 0);
}

/* ref GI:52-79 */
static inline int boxIntersection(const BoundingBox *box, const Ray *ray, float t0, float t1)
{
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    tmin = (box->parameters[ray->sx].x - ray->origin.x) * ray->inv_direction.x;
    tmax = (box->parameters[1 - ray->sx].x - ray->origin.x) * ray->inv_direction.x;
    tymin = (box->parameters[ray->sy].y - ray->origin.y) * ray->inv_direction.y;
    tymax = (box->parameters[1 - ray->sy].y - ray->origin.y) * ray->inv_direction.y;
    if ((tmin > tymax) || (tymin > tmax))
        return 0;
    if (tymin > tmin)
        tmin = tymin;
    if (tymax < tmax)
        tmax = tymax;
    tzmin = (box->parameters[ray->sz].z - ray->origin.z) * ray->inv_direction.z;
    tzmax = (box->parameters[1 - ray->sz].z - ray->origin.z) * ray->inv_direction.z;
    if ((tmin > tzmax) || (tzmin > tmax))
        return 0;
    if (tzmin > tmin)
        tmin = tzmin;
    if (tzmax < tmax)
        tmax = tzmax;
    return ((tmin < t1) && (tmax > t0));
}

int oracle_box_intersection(const BoundingBox *box, const float origin[3], const float direction[3], float t0,
                            float t1)
{
    Ray r;
    r.origin = V(origin[0], origin[1], origin[2]);
    r.direction = V(direction[0], direction[1], direction[2]);
    computeRayAttributes(&r);
    return boxIntersection(box, &r, t0, t1);
}

/* ---- texture tier ---------------------------------------------------- */

typedef struct
{
    float x, y, z, w;
} f4;

/* ref TM:30-40 */
static inline void normalMap(int index, const Material *m, const BitmapBuffer *tex, v3 *normal, float strength)
{
    int i = m->textureOffset.y + index;
    BitmapBuffer r = tex[i], g = tex[i + 1];
    normal->x -= strength * (r / 256.f - 0.5f);
    normal->y -= strength * (g / 256.f - 0.5f);
    if (!DIALECT(1)) /* TM:39; CL:505-514 leaves z alone */
        normal->z = 0.f;
}
/* ref TM:45-57 */
static inline void bumpMap(int index, const Material *m, const BitmapBuffer *tex, float *value)
{
    int i = m->textureOffset.z + index;
    BitmapBuffer r = tex[i], g = tex[i + 1], b = tex[i + 2];
    *value = 10.f * (r + g + b) / 768.f;
}
/* ref TM:62-73 */
static inline void specularMap(int index, const Material *m, const BitmapBuffer *tex, f4 *specular)
{
    int i = m->textureOffset.w + index;
    BitmapBuffer r = tex[i], g = tex[i + 1], b = tex[i + 2];
    if (DIALECT(2)) /* CL:533-541 scales the specular value by the texel's brightness */
    {
        specular->x *= (r + g + b) / 768.f;
        return;
    }
    specular->x = r / 256.f;
    specular->y = 1000.f * g / 256.f;
    specular->z = b / 256.f;
}
/* ref TM:78-87 */
static inline void reflectionMap(int index, const Material *m, const BitmapBuffer *tex, f4 *attributes)
{
    int i = m->advancedTextureOffset.x + index;
    BitmapBuffer r = tex[i], g = tex[i + 1], b = tex[i + 2];
    attributes->x *= (r + g + b) / 768.f;
}
/* ref TM:92-102 */
static inline void transparencyMap(int index, const Material *m, const BitmapBuffer *tex, f4 *attributes)
{
    int i = m->advancedTextureOffset.y + index;
    BitmapBuffer r = tex[i], g = tex[i + 1], b = tex[i + 2];
    attributes->y *= (r + g + b) / 768.f;
}
/* ref TM:107-116 */
static inline void ambientOcclusionMap(int index, const Material *m, const BitmapBuffer *tex, f4 *adv)
{
    int i = m->advancedTextureOffset.z + index;
    BitmapBuffer r = tex[i], g = tex[i + 1], b = tex[i + 2];
    adv->x = (r + g + b) / 768.f;
}

/* ref TM:118-158 */
static void juliaSet(const Material *material, const SceneInfo *si, float x, float y, f4 *color)
{
    float W = (float)material->textureMapping.x;
    float H = (float)material->textureMapping.y;
    float cRe = -0.7f + 0.4f * sin_o(si->timestamp / 1500.f);
    float cIm = 0.27015f + 0.4f * cos_o(si->timestamp / 2000.f);
    float newRe = 1.5f * (x - W / 2.f) / (0.5f * W);
    float newIm = (y - H / 2.f) / (0.5f * H);
    int n;
    float maxIterations = 40.f + si->pathTracingIteration;
    for (n = 0; n < maxIterations; n++)
    {
        float oldRe = newRe;
        float oldIm = newIm;
        newRe = oldRe * oldRe - oldIm * oldIm + cRe;
        newIm = 2.f * oldRe * oldIm + cIm;
        if ((newRe * newRe + newIm * newIm) > 4.f)
            break;
    }
    color->x = 1.f - color->x * (n / maxIterations);
    color->y = 1.f - color->y * (n / maxIterations);
    color->z = 1.f - color->z * (n / maxIterations);
    color->w = 1.f - (n / maxIterations);
}

/* ref TM:160-197 (note the double Im_factor, TM:172-175) */
static void mandelbrotSet(const Material *material, const SceneInfo *si, float x, float y, f4 *color)
{
    float W = (float)material->textureMapping.x;
    float H = (float)material->textureMapping.y;
    float MinRe = -2.f;
    float MaxRe = 1.f;
    float MinIm = -1.2f;
    float MaxIm = MinIm + (MaxRe - MinRe) * H / W;
    float Re_factor = (MaxRe - MinRe) / (W - 1.f);
    double Im_factor = (MaxIm - MinIm) / (H - 1.f);
    float maxIterations = NB_MAX_ITERATIONS + si->pathTracingIteration;
    float c_im = (float)(MaxIm - y * Im_factor);
    float c_re = MinRe + x * Re_factor;
    float Z_re = c_re;
    float Z_im = c_im;
    int isInside = 1;
    unsigned n;
    for (n = 0; isInside && n < maxIterations; ++n)
    {
        float Z_re2 = Z_re * Z_re;
        float Z_im2 = Z_im * Z_im;
        if (Z_re2 + Z_im2 > 4.f)
            isInside = 0;
        Z_im = 2.f * Z_re * Z_im + c_im;
        Z_re = Z_re2 - Z_im2 + c_re;
    }
    color->x = 1.f - color->x * (n / maxIterations);
    color->y = 1.f - color->y * (n / maxIterations);
    color->z = 1.f - color->z * (n / maxIterations);
    color->w = 1.f - (n / maxIterations);
}

static inline f4 colorOf(const Material *m)
{
    f4 c = {m->color.x, m->color.y, m->color.z, m->color.w};
    return c;
}

/* common tail of the three mappers (TM:238-279, 311-343, 410-441) */
static inline void fetchTexel(const Material *material, const BitmapBuffer *textures, int u, int v, f4 *result,
                              v3 *normal, f4 *specular, f4 *attributes, f4 *advancedAttributes, int triangleMapper)
{
    int A = (v * material->textureMapping.x + u) * material->textureMapping.w;
    int B = material->textureMapping.x * material->textureMapping.y * material->textureMapping.w;
    int index = A % B;
    int i = material->textureOffset.x + index;
    BitmapBuffer r = textures[i], g = textures[i + 1], b = textures[i + 2];
    result->x = r / 256.f;
    result->y = g / 256.f;
    result->z = b / 256.f;
    float strength = 3.f;
    if (material->textureIds.z != TEXTURE_NONE)
    {
        bumpMap(index, material, textures, &strength);
        if (DIALECT(3) && triangleMapper) /* CL:815-819: the triangle mapper also scales the opacity; TM:260-264 does not */
            attributes->w *= strength / 10.f;
    }
    if (material->textureIds.y != TEXTURE_NONE)
        normalMap(index, material, textures, normal, strength);
    if (material->textureIds.w != TEXTURE_NONE)
        specularMap(index, material, textures, specular);
    if (material->advancedTextureIds.x != TEXTURE_NONE)
        reflectionMap(index, material, textures, attributes);
    if (material->advancedTextureIds.y != TEXTURE_NONE)
        transparencyMap(index, material, textures, attributes);
    if (material->advancedTextureIds.z != TEXTURE_NONE)
        ambientOcclusionMap(index, material, textures, advancedAttributes);
}

/* ref TM:205-283 */
static f4 triangleUVMapping(const SceneInfo *si, const Primitive *primitive, const Material *materials,
                            const BitmapBuffer *textures, v3 areas, v3 *normal, f4 *specular, f4 *attributes,
                            f4 *advancedAttributes)
{
    const Material *material = &materials[primitive->materialId];
    f4 result = colorOf(material);
    float sum = areas.x + areas.y + areas.z;
    float Tx = (primitive->vt0.x * areas.x + primitive->vt1.x * areas.y + primitive->vt2.x * areas.z) / sum;
    float Ty = (primitive->vt0.y * areas.x + primitive->vt1.y * areas.y + primitive->vt2.y * areas.z) / sum;
    float mox = 0.f, moy = 0.f;
    if (material->attributes.y == 1)
    {
        mox = material->mappingOffset.x * si->timestamp;
        moy = material->mappingOffset.y * si->timestamp;
    }
    int u = f2i(Tx * material->textureMapping.x + mox);
    int v = f2i(Ty * material->textureMapping.y + moy);
    u = u % material->textureMapping.x;
    v = v % material->textureMapping.y;
    if (u >= 0 && u < material->textureMapping.x && v >= 0 && v < material->textureMapping.y)
    {
        switch (material->textureIds.x)
        {
        case TEXTURE_MANDELBROT:
            mandelbrotSet(material, si, (float)u, (float)v, &result);
            break;
        case TEXTURE_JULIA:
            juliaSet(material, si, (float)u, (float)v, &result);
            break;
        default:
            fetchTexel(material, textures, u, v, &result, normal, specular, attributes, advancedAttributes, 1);
        }
    }
    return result;
}

/* ref TM:291-346 */
static f4 sphereUVMapping(const Primitive *primitive, const Material *materials, const BitmapBuffer *textures,
                          v3 intersection, v3 *normal, f4 *specular, f4 *attributes, f4 *advancedAttributes)
{
    const Material *material = &materials[primitive->materialId];
    f4 result = colorOf(material);
    v3 I = vnormalize(vsub(intersection, primitive->p0));
    float U = ((atan2_o(I.x, I.z) / SOLR_PI) + 1.f) * .5f;
    float V_ = (asin_o(I.y) / SOLR_PI) + .5f;
    int u = f2i(material->textureMapping.x * (U * primitive->vt1.x));
    int v = f2i(material->textureMapping.y * (V_ * primitive->vt1.y));
    if (material->textureMapping.x != 0)
        u = u % material->textureMapping.x;
    if (material->textureMapping.y != 0)
        v = v % material->textureMapping.y;
    if (u >= 0 && u < material->textureMapping.x && v >= 0 && v < material->textureMapping.y)
        fetchTexel(material, textures, u, v, &result, normal, specular, attributes, advancedAttributes, 0);
    return result;
}

/* ref TM:354-447 (non-Kinect build) */
static f4 cubeMapping(const SceneInfo *si, const Primitive *primitive, const Material *materials,
                      const BitmapBuffer *textures, v3 intersection, v3 *normal, f4 *specular, f4 *attributes,
                      f4 *advancedAttributes)
{
    const Material *material = &materials[primitive->materialId];
    f4 result = colorOf(material);
    int u = f2i(((primitive->type == ptCheckboard) || (primitive->type == ptXZPlane) ||
                   (primitive->type == ptXYPlane))
                      ? (intersection.x - primitive->p0.x + primitive->size.x)
                      : (intersection.z - primitive->p0.z + primitive->size.z));
    int v = f2i(((primitive->type == ptCheckboard) || (primitive->type == ptXZPlane))
                      ? (intersection.z + primitive->p0.z + primitive->size.z)
                      : (intersection.y - primitive->p0.y + primitive->size.y));
    if (material->textureMapping.x != 0)
        u = u % material->textureMapping.x;
    if (material->textureMapping.y != 0)
        v = v % material->textureMapping.y;
    /* TM:398 compares v against textureMapping.x (sic) */
    if (u >= 0 && u < material->textureMapping.x && v >= 0 && v < material->textureMapping.x)
    {
        switch (material->textureIds.x)
        {
        case TEXTURE_MANDELBROT:
            mandelbrotSet(material, si, (float)u, (float)v, &result);
            break;
        case TEXTURE_JULIA:
            juliaSet(material, si, (float)u, (float)v, &result);
            break;
        default:
            fetchTexel(material, textures, u, v, &result, normal, specular, attributes, advancedAttributes, 0);
        }
    }
    return result;
}

/* ref TM:449-456 */
static inline int wireFrameMapping(float x, float y, int width)
{
    int X = f2i(fabsf(x));
    int Y = f2i(fabsf(y));
    int A = 100;
    int B = 100;
    return (X % A <= width) || (Y % B <= width);
}

/* ref GI:87-151 */
static c3 skyboxMapping(const SceneInfo *si, const Material *materials, const BitmapBuffer *textures,
                        const Ray *ray)
{
    const Material *material = &materials[si->skyboxMaterialId];
    c3 result = {material->color.x, material->color.y, material->color.z};
    v3 dir = vnormalize(vsub(ray->direction, ray->origin));
    float a = 2.f * vdot(dir, dir);
    float b = 2.f * vdot(ray->origin, dir);
    float c = vdot(ray->origin, ray->origin) - (float)(si->skyboxRadius * si->skyboxRadius);
    float d = b * b - 2.f * a * c;
    if (d <= 0.f || a == 0.f)
        return result;
    /* GI:87-151 fetches a texel whatever the material: without a diffuse texture the "computed texture"
     * mapping (40000 x 40000, GPUKernel.cpp:1893-1896) indexes gigabytes past the atlas.  Here, as in the
     * engine, such a skybox shows the material's colour. */
    if (material->textureIds.x < 0)
        return result;
    float r = sqrtf(d);
    float t1 = (-b - r) / a;
    float t2 = (-b + r) / a;
    if (t1 <= si->geometryEpsilon && t2 <= si->geometryEpsilon)
        return result;
    float t = 0.f;
    if (t1 <= si->geometryEpsilon)
        t = t2;
    else if (t2 <= si->geometryEpsilon)
        t = t1;
    else
        t = (t1 < t2) ? t1 : t2;
    if (t < si->geometryEpsilon)
        return result;
    v3 intersection = vnormalize(vadd(ray->origin, vscale(dir, t)));
    float U = ((atan2_o(intersection.x, intersection.z) / SOLR_PI) + 1.f) * .5f;
    float V_ = (asin_o(intersection.y) / SOLR_PI) + .5f;
    int u = f2i(material->textureMapping.x * U);
    int v = f2i(material->textureMapping.y * V_);
    if (material->textureMapping.x != 0)
        u %= material->textureMapping.x;
    if (material->textureMapping.y != 0)
        v %= material->textureMapping.y;
    if (u >= 0 && u < material->textureMapping.x && v >= 0 && v < material->textureMapping.y)
    {
        int A = (v * material->textureMapping.x + u) * material->textureMapping.w;
        int B = material->textureMapping.x * material->textureMapping.y * material->textureMapping.w;
        int index = A % B;
        int i = material->textureOffset.x + index;
        result.x = textures[i] / 256.f;
        result.y = textures[i + 1] / 256.f;
        result.z = textures[i + 2] / 256.f;
    }
    return result;
}

/* ---- primitive intersections ------------------------------------------ */

/* ref GI:159-212 */
static int ellipsoidIntersection(const SceneInfo *si, const Primitive *e, const Ray *ray, v3 *intersection,
                                 v3 *normal, float *shadowIntensity)
{
    *shadowIntensity = 1.f;
    v3 O_C = vsub(ray->origin, e->p0);
    v3 dir = vnormalize(ray->direction);
    float a = ((dir.x * dir.x) / (e->size.x * e->size.x)) + ((dir.y * dir.y) / (e->size.y * e->size.y)) +
              ((dir.z * dir.z) / (e->size.z * e->size.z));
    float b = ((2.f * O_C.x * dir.x) / (e->size.x * e->size.x)) + ((2.f * O_C.y * dir.y) / (e->size.y * e->size.y)) +
              ((2.f * O_C.z * dir.z) / (e->size.z * e->size.z));
    float c = ((O_C.x * O_C.x) / (e->size.x * e->size.x)) + ((O_C.y * O_C.y) / (e->size.y * e->size.y)) +
              ((O_C.z * O_C.z) / (e->size.z * e->size.z)) - 1.f;
    float d = ((b * b) - (4.f * a * c));
    if (d < 0.f || a == 0.f || b == 0.f || c == 0.f)
        return 0;
    d = sqrtf(d);
    float t1 = (-b + d) / (2.f * a);
    float t2 = (-b - d) / (2.f * a);
    if (t1 <= si->geometryEpsilon && t2 <= si->geometryEpsilon)
        return 0;
    float t = 0.f;
    if (t1 <= si->geometryEpsilon)
        t = t2;
    else if (t2 <= si->geometryEpsilon)
        t = t1;
    else
        t = (t1 < t2) ? t1 : t2;
    if (t < si->geometryEpsilon)
        return 0;
    *intersection = vadd(ray->origin, vscale(dir, t));
    v3 n = vsub(*intersection, e->p0);
    n.x = 2.f * n.x / (e->size.x * e->size.x);
    n.y = 2.f * n.y / (e->size.y * e->size.y);
    n.z = 2.f * n.z / (e->size.z * e->size.z);
    *normal = vnormalize(n);
    return 1;
}

/* ref GI:220-284 */
static int sphereIntersection(const SceneInfo *si, const Primitive *sphere, const Material *materials,
                              const Ray *ray, v3 *intersection, v3 *normal, float *shadowIntensity)
{
    int back = 0;
    v3 O_C = vsub(ray->origin, sphere->p0);
    v3 dir = vnormalize(ray->direction);
    float a = 2.f * vdot(dir, dir);
    float b = 2.f * vdot(O_C, dir);
    float c = vdot(O_C, O_C) - (sphere->size.x * sphere->size.x);
    float d = b * b - 2.f * a * c;
    if (d <= 0.f || a == 0.f)
        return 0;
    float r = sqrtf(d);
    float t1 = (-b - r) / a;
    float t2 = (-b + r) / a;
    if (t1 <= si->geometryEpsilon && t2 <= si->geometryEpsilon)
        return 0;
    float t = 0.f;
    if (t1 <= si->geometryEpsilon)
    {
        t = t2;
        back = 1;
    }
    else if (t2 <= si->geometryEpsilon)
        t = t1;
    else
        t = (t1 < t2) ? t1 : t2;
    if (t < si->geometryEpsilon)
        return 0;
    *intersection = vadd(ray->origin, vscale(dir, t));
    v3 n;
    if (materials[sphere->materialId].attributes.y == 0)
        n = vsub(*intersection, sphere->p0);
    else
    {
        /* procedural bumps, GI:268-273: timestamp (int) + coordinate (float) in binary32 */
        v3 newCenter;
        newCenter.x = sphere->p0.x + 0.008f * sphere->size.x * cos_o(si->timestamp + intersection->x);
        newCenter.y = sphere->p0.y + 0.008f * sphere->size.y * sin_o(si->timestamp + intersection->y);
        newCenter.z = sphere->p0.z + 0.008f * sphere->size.z * sin_o(cos_o(si->timestamp + intersection->z));
        n = vsub(*intersection, newCenter);
    }
    n = vnormalize(n);
    if (back)
        n = vscale(n, -1.f);
    *normal = n;
    r = vdot(dir, n);
    *shadowIntensity = (materials[sphere->materialId].transparency != 0.f) ? (1.f - fabsf(r)) : 1.f;
    return 1;
}

/* ref GI:293-349 (cylinder) and GI:358-416 (cone: identical arithmetic) */
static int cylinderIntersection(const SceneInfo *si, const Primitive *cyl, const Ray *ray, v3 *intersection,
                                v3 *normal, float *shadowIntensity)
{
    v3 O_C = vsub(ray->origin, cyl->p0);
    v3 dir = ray->direction;
    v3 n = crossProduct(dir, cyl->n1);
    float ln = vlength(n);
    if ((ln < si->geometryEpsilon) && (ln > -si->geometryEpsilon))
        return 0;
    n = vnormalize(n);
    float d = fabsf(vdot(O_C, n));
    if (d > cyl->size.y)
        return 0;
    v3 O = crossProduct(O_C, cyl->n1);
    float t = -vdot(O, n) / ln;
    if (t < 0.f)
        return 0;
    O = vnormalize(crossProduct(n, cyl->n1));
    float s = fabsf(sqrtf(cyl->size.x * cyl->size.x - d * d) / vdot(dir, O));
    float t1 = t - s;
    float t2 = t + s;
    *intersection = vadd(ray->origin, vscale(dir, t1));
    v3 HB1 = vsub(*intersection, cyl->p0);
    v3 HB2 = vsub(*intersection, cyl->p1);
    float scale1 = vdot(HB1, cyl->n1);
    float scale2 = vdot(HB2, cyl->n1);
    if (scale1 < si->geometryEpsilon || scale2 > si->geometryEpsilon)
    {
        *intersection = vadd(ray->origin, vscale(dir, t2));
        HB1 = vsub(*intersection, cyl->p0);
        HB2 = vsub(*intersection, cyl->p1);
        scale1 = vdot(HB1, cyl->n1);
        scale2 = vdot(HB2, cyl->n1);
        if (scale1 < si->geometryEpsilon || scale2 > si->geometryEpsilon)
            return 0;
    }
    v3 Vv = vsub(*intersection, cyl->p2);
    *normal = vnormalize(vsub(Vv, project(Vv, cyl->n1)));
    *shadowIntensity = 1.f;
    return 1;
}

/* one side of an axis plane (GI:439-445, 451-461, 479-494, 518-528) */
#define PLANE_HIT(U, V_, W_)                                                                                     \
    do                                                                                                           \
    {                                                                                                            \
        float k = ray->origin.W_ - primitive->p0.W_;                                                             \
        intersection->U = ray->origin.U + k * ray->direction.U / -ray->direction.W_;                             \
        intersection->W_ = primitive->p0.W_;                                                                     \
        intersection->V_ = ray->origin.V_ + k * ray->direction.V_ / -ray->direction.W_;                          \
        collision = fabsf(intersection->U - primitive->p0.U) < primitive->size.U &&                              \
                    fabsf(intersection->V_ - primitive->p0.V_) < primitive->size.V_;                             \
    } while (0)

/* ref GI:424-567.  OpenCL dialect (CL:1151-1305): the normal is written as the axis vector inside the branch
 * that is taken instead of being copied from primitive.n0; no wireframe and no "chessboard light" masks;
 * ptCamera has a front side only. */
#define PLANE_NORMAL(NX, NY, NZ, SIGN)                                                                           \
    do                                                                                                           \
    {                                                                                                            \
        if (DIALECT(4))                                                                                                \
            *normal = V((SIGN) * (NX), (SIGN) * (NY), (SIGN) * (NZ));                                            \
        else if ((SIGN) < 0.f)                                                                                   \
            *normal = vneg(*normal);                                                                             \
    } while (0)
static int planeIntersection(const SceneInfo *si, const Primitive *primitive, const Material *materials,
                             const BitmapBuffer *textures, const Ray *ray, v3 *intersection, v3 *normal,
                             float *shadowIntensity, int reverse)
{
    int collision = 0;
    float reverted = reverse ? -1.f : 1.f;
    const Material *mat = &materials[primitive->materialId];
    const int masks = !DIALECT(5); /* GI:447-449, 463-471 ... : wireframe / chessboard-light masks, absent from CL */
    if (!DIALECT(6))
        *normal = primitive->n0;
    switch (primitive->type)
    {
    case ptMagicCarpet:
    case ptCheckboard:
    {
        intersection->y = primitive->p0.y;
        float y = ray->origin.y - primitive->p0.y;
        if (reverted * ray->direction.y < 0.f && reverted * ray->origin.y > reverted * primitive->p0.y)
        {
            PLANE_NORMAL(0.f, 1.f, 0.f, 1.f);
            intersection->x = ray->origin.x + y * ray->direction.x / -ray->direction.y;
            intersection->z = ray->origin.z + y * ray->direction.z / -ray->direction.y;
            collision = fabsf(intersection->x - primitive->p0.x) < primitive->size.x &&
                        fabsf(intersection->z - primitive->p0.z) < primitive->size.z;
        }
        break;
    }
    case ptXZPlane:
    {
        if (reverted * ray->direction.y < 0.f && reverted * ray->origin.y > reverted * primitive->p0.y)
        {
            PLANE_NORMAL(0.f, 1.f, 0.f, 1.f);
            PLANE_HIT(x, z, y);
            if (masks && mat->attributes.z == 2)
                collision &= wireFrameMapping(intersection->x, intersection->z, mat->attributes.w);
        }
        if (!collision && reverted * ray->direction.y > 0.f && reverted * ray->origin.y < reverted * primitive->p0.y)
        {
            PLANE_NORMAL(0.f, 1.f, 0.f, -1.f);
            PLANE_HIT(x, z, y);
            if (masks && mat->attributes.z == 2)
                collision &= wireFrameMapping(intersection->x, intersection->z, mat->attributes.w);
        }
        break;
    }
    case ptYZPlane:
    {
        if (reverted * ray->direction.x < 0.f && reverted * ray->origin.x > reverted * primitive->p0.x)
        {
            PLANE_NORMAL(1.f, 0.f, 0.f, 1.f);
            PLANE_HIT(y, z, x);
            if (masks && mat->innerIllumination.x != 0.f)
                collision &= f2i(fabsf(intersection->z)) % 4000 < 2000 && f2i(fabsf(intersection->y)) % 4000 < 2000;
            if (masks && mat->attributes.z == 2)
                collision &= wireFrameMapping(intersection->y, intersection->z, mat->attributes.w);
        }
        if (!collision && reverted * ray->direction.x > 0.f && reverted * ray->origin.x < reverted * primitive->p0.x)
        {
            PLANE_NORMAL(1.f, 0.f, 0.f, -1.f);
            PLANE_HIT(y, z, x);
            if (masks && mat->innerIllumination.x != 0.f)
                collision &= f2i(fabsf(intersection->z)) % 4000 < 2000 && f2i(fabsf(intersection->y)) % 4000 < 2000;
            if (masks && mat->attributes.z == 2)
                collision &= wireFrameMapping(intersection->y, intersection->z, mat->attributes.w);
        }
        break;
    }
    case ptXYPlane:
    case ptCamera:
    {
        if (reverted * ray->direction.z < 0.f && reverted * ray->origin.z > reverted * primitive->p0.z)
        {
            PLANE_NORMAL(0.f, 0.f, 1.f, 1.f);
            PLANE_HIT(x, y, z);
            if (masks && mat->attributes.z == 2)
                collision &= wireFrameMapping(intersection->x, intersection->y, mat->attributes.w);
        }
        if (!(DIALECT(7) && primitive->type == ptCamera) /* CL:1266-1281 */ && !collision &&
            reverted * ray->direction.z > 0.f && reverted * ray->origin.z < reverted * primitive->p0.z)
        {
            PLANE_NORMAL(0.f, 0.f, 1.f, -1.f);
            PLANE_HIT(x, y, z);
            if (masks && mat->attributes.z == 2)
                collision &= wireFrameMapping(intersection->x, intersection->y, mat->attributes.w);
        }
        break;
    }
    default:
        break;
    }

    if (collision)
    {
        *shadowIntensity = 1.f;
        f4 color = colorOf(mat);
        if (primitive->type == ptCamera || mat->textureIds.x != TEXTURE_NONE)
        {
            f4 specular = {0.f, 0.f, 0.f, 0.f};
            f4 attributes = {0.f, 0.f, 0.f, 0.f};
            f4 advancedAttributes = {0.f, 0.f, 0.f, 0.f};
            color = cubeMapping(si, primitive, materials, textures, *intersection, normal, &specular, &attributes,
                                &advancedAttributes);
            *shadowIntensity = color.w;
        }
        if ((color.x + color.y + color.z) / 3.f >= si->transparentColor)
            collision = 0;
    }
    return collision;
}

/* ref GI:575-659 */
static int triangleIntersection(const SceneInfo *si, const Primitive *tri, const Ray *ray, v3 *intersection,
                                v3 *normal, v3 *areas, float *shadowIntensity, int processingShadows)
{
    v3 E01 = vsub(tri->p1, tri->p0);
    v3 E03 = vsub(tri->p2, tri->p0);
    v3 P = crossProduct(ray->direction, E03);
    float det = vdot(E01, P);
    if (fabsf(det) < si->geometryEpsilon)
        return 0;
    v3 T = vsub(ray->origin, tri->p0);
    float a = vdot(T, P) / det;
    if (a < 0.f || a > 1.f)
        return 0;
    v3 Q = crossProduct(T, E01);
    float b = vdot(ray->direction, Q) / det;
    if (b < 0.f || b > 1.f)
        return 0;
    if ((a + b) > 1.f)
    {
        /* GI:603-616: E21 = p1 - p1 is the zero vector (sic) */
        v3 E23 = vsub(tri->p0, tri->p1);
        v3 E21 = vsub(tri->p1, tri->p1);
        v3 P_ = crossProduct(ray->direction, E21);
        float det_ = vdot(E23, P_);
        if (fabsf(det_) < si->geometryEpsilon)
            return 0;
        v3 T_ = vsub(ray->origin, tri->p2);
        float a_ = vdot(T_, P_) / det_;
        if (a_ < 0.f)
            return 0;
        v3 Q_ = crossProduct(T_, E23);
        float b_ = vdot(ray->direction, Q_) / det_;
        if (b_ < 0.f)
            return 0;
    }
    float t = vdot(E03, Q) / det;
    if (t < 0)
        return 0;
    *intersection = vadd(ray->origin, vscale(ray->direction, t));
    v3 v0 = vsub(tri->p0, *intersection);
    v3 v1 = vsub(tri->p1, *intersection);
    v3 v2 = vsub(tri->p2, *intersection);
    areas->x = 0.5f * vlength(crossProduct(v1, v2));
    areas->y = 0.5f * vlength(crossProduct(v0, v2));
    areas->z = 0.5f * vlength(crossProduct(v0, v1));
    if (DIALECT(8)) /* CL:1374 normalises the areas; GI:626-628 keeps them */
        *areas = vnormalize(*areas);
    v3 wn = vadd(vadd(vscale(tri->n0, areas->x), vscale(tri->n1, areas->y)), vscale(tri->n2, areas->z));
    *normal = vdivs(wn, areas->x + areas->y + areas->z);
    if (!DIALECT(9)) /* GI:630-631 normalises the interpolated normal; CL:1376-1377 does not */
        *normal = vnormalize(*normal);
    if (si->doubleSidedTriangles)
    {
        v3 N = vnormalize(ray->direction);
        if (DIALECT(10))
        {
            /* CL:1379-1394: shadow rays keep the faces turned away from the lamp, view rays those turned to the eye */
            if (processingShadows)
            {
                if (vdot(N, *normal) <= 0.f)
                    return 0;
            }
            else if (vdot(N, *normal) >= 0.f)
                return 0;
        }
        else if (processingShadows)
        {
            /* GI:643-647: the else binds to the inner if */
            if (vdot(N, *normal) <= 0.f)
                return 0;
            else if (vdot(N, *normal) >= 0.f)
                return 0;
        }
    }
    v3 dir = vnormalize(ray->direction);
    float r = vdot(dir, *normal);
    if (r > 0.f)
        *normal = vscale(*normal, -1.f);
    *shadowIntensity = 1.f;
    return 1;
}

/* dispatch of the closest-hit walk, GI:712-747 */
static inline int testPrimitive(const SceneInfo *si, const Primitive *primitive, const Material *materials,
                                const BitmapBuffer *textures, const Ray *r, v3 *intersection, v3 *normal, v3 *areas,
                                float *shadowIntensity)
{
    if (si->extendedGeometry)
    {
        switch (primitive->type)
        {
        case ptEnvironment:
        case ptSphere:
            return sphereIntersection(si, primitive, materials, r, intersection, normal, shadowIntensity);
        case ptCone:
            if (DIALECT(11)) /* CL:1846-1869 has no cone: the type falls through to the plane test, which ignores it */
                return planeIntersection(si, primitive, materials, textures, r, intersection, normal, shadowIntensity,
                                         0);
            /* fall through */
        case ptCylinder:
            return cylinderIntersection(si, primitive, r, intersection, normal, shadowIntensity);
        case ptEllipsoid:
            return ellipsoidIntersection(si, primitive, r, intersection, normal, shadowIntensity);
        case ptTriangle:
            return triangleIntersection(si, primitive, r, intersection, normal, areas, shadowIntensity, 0);
        default:
            return planeIntersection(si, primitive, materials, textures, r, intersection, normal, shadowIntensity,
                                     0);
        }
    }
    return triangleIntersection(si, primitive, r, intersection, normal, areas, shadowIntensity, 0);
}

/* dispatch of the shadow walk, GI:833-870 */
static inline int testPrimitiveShadow(const SceneInfo *si, const Primitive *primitive, const Material *materials,
                                      const BitmapBuffer *textures, const Ray *r, v3 *intersection, v3 *normal,
                                      v3 *areas, float *shadowIntensity)
{
    if (si->extendedGeometry)
    {
        switch (primitive->type)
        {
        case ptSphere:
            return sphereIntersection(si, primitive, materials, r, intersection, normal, shadowIntensity);
        case ptEllipsoid:
            return ellipsoidIntersection(si, primitive, r, intersection, normal, shadowIntensity);
        case ptCone:
            if (DIALECT(12)) /* CL:1544-1569: no cone */
                return planeIntersection(si, primitive, materials, textures, r, intersection, normal, shadowIntensity,
                                         0);
            /* fall through */
        case ptCylinder:
            return cylinderIntersection(si, primitive, r, intersection, normal, shadowIntensity);
        case ptTriangle:
            return triangleIntersection(si, primitive, r, intersection, normal, areas, shadowIntensity, 1);
        case ptCamera:
            return 0;
        default:
            return planeIntersection(si, primitive, materials, textures, r, intersection, normal, shadowIntensity,
                                     0);
        }
    }
    return triangleIntersection(si, primitive, r, intersection, normal, areas, shadowIntensity, 1);
}

int oracle_primitive_intersection(const SceneInfo *sceneInfo, const Primitive *primitive, const Material *materials,
                                  const BitmapBuffer *textures, const float origin[3], const float direction[3],
                                  int processingShadows, float intersection[3], float normal[3], float areas[3],
                                  float *shadowIntensity)
{
    Ray r;
    r.origin = V(origin[0], origin[1], origin[2]);
    r.direction = V(direction[0], direction[1], direction[2]);
    computeRayAttributes(&r);
    v3 i = V(intersection[0], intersection[1], intersection[2]);
    v3 n = V(normal[0], normal[1], normal[2]);
    v3 a = V(0.f, 0.f, 0.f);
    int hit = processingShadows
                  ? testPrimitiveShadow(sceneInfo, primitive, materials, textures, &r, &i, &n, &a, shadowIntensity)
                  : testPrimitive(sceneInfo, primitive, materials, textures, &r, &i, &n, &a, shadowIntensity);
    intersection[0] = i.x;
    intersection[1] = i.y;
    intersection[2] = i.z;
    normal[0] = n.x;
    normal[1] = n.y;
    normal[2] = n.z;
    areas[0] = a.x;
    areas[1] = a.y;
    areas[2] = a.z;
    return hit;
}

/* ref GI:667-772 */
static int intersectionWithPrimitives(const OracleScene *s, const SceneInfo *si, const Ray *ray, int iteration,
                                      int *closestPrimitive, v3 *closestIntersection, v3 *closestNormal,
                                      v3 *closestAreas, c3 *colorBox, int currentMaterialId, Stats *st)
{
    int intersections = 0;
    float minDistance = (iteration < 2) ? si->viewDistance : si->viewDistance / (iteration + 1);
    Ray r;
    r.origin = ray->origin;
    r.direction = vsub(ray->direction, ray->origin);
    computeRayAttributes(&r);

    v3 intersection = {0.f, 0.f, 0.f};
    v3 normal = {0.f, 0.f, 0.f};
    int i = 0;
    float shadowIntensity = 0.f;
    st->closest++;

    int cptBoxes = 0;
    while (cptBoxes < s->nbBoxes)
    {
        const BoundingBox *box = &s->boxes[cptBoxes];
        st->boxes++;
        if (boxIntersection(box, &r, 0.f, minDistance))
        {
            if (si->renderBoxes != 0)
            {
                /* GI:695: NB_MAX_MATERIALS is unsigned in the reference */
                const Material *m = &s->materials[(unsigned)box->startIndex % (unsigned)NB_MAX_MATERIALS];
                const float share = DIALECT(13) ? 50.f : 200.f; /* GI:695; CL:1894 */
                colorBox->x += m->color.x / share;
                colorBox->y += m->color.y / share;
                colorBox->z += m->color.z / share;
            }
            else
            {
                for (int cptPrimitives = 0; cptPrimitives < box->nbPrimitives; ++cptPrimitives)
                {
                    const Primitive *primitive = &s->primitives[box->startIndex + cptPrimitives];
                    const Material *material = &s->materials[primitive->materialId];
                    if (material->attributes.x == 0 ||
                        (material->attributes.x == 1 && currentMaterialId != primitive->materialId))
                    {
                        v3 areas = {0.f, 0.f, 0.f};
                        st->prims++;
                        i = testPrimitive(si, primitive, s->materials, s->textures, &r, &intersection, &normal,
                                          &areas, &shadowIntensity);
                        float distance = vlength(vsub(intersection, r.origin));
                        if (i && distance > si->geometryEpsilon && distance < minDistance)
                        {
                            minDistance = distance;
                            *closestPrimitive = box->startIndex + cptPrimitives;
                            *closestIntersection = intersection;
                            *closestNormal = normal;
                            *closestAreas = areas;
                            intersections = 1;
                        }
                    }
                }
            }
            ++cptBoxes;
        }
        else
            cptBoxes += box->indexForNextBox.x;
    }
    return intersections;
}

int oracle_closest_hit(const OracleScene *scene, const SceneInfo *sceneInfo, const float origin[3],
                       const float target[3], int iteration, int currentMaterialId, int *closestPrimitive,
                       float closestIntersection[3], float closestNormal[3], float closestAreas[3])
{
    Ray ray;
    Stats st;
    memset(&st, 0, sizeof(st));
    ray.origin = V(origin[0], origin[1], origin[2]);
    ray.direction = V(target[0], target[1], target[2]);
    v3 ci = V(closestIntersection[0], closestIntersection[1], closestIntersection[2]);
    v3 cn = V(closestNormal[0], closestNormal[1], closestNormal[2]);
    v3 ca = V(closestAreas[0], closestAreas[1], closestAreas[2]);
    c3 colorBox = {0.f, 0.f, 0.f};
    int hit = intersectionWithPrimitives(scene, sceneInfo, &ray, iteration, closestPrimitive, &ci, &cn, &ca,
                                         &colorBox, currentMaterialId, &st);
    closestIntersection[0] = ci.x;
    closestIntersection[1] = ci.y;
    closestIntersection[2] = ci.z;
    closestNormal[0] = cn.x;
    closestNormal[1] = cn.y;
    closestNormal[2] = cn.z;
    closestAreas[0] = ca.x;
    closestAreas[1] = ca.y;
    closestAreas[2] = ca.z;
    return hit;
}

/* ref GI:798-908.  objectId is the FLATTENED index of the shaded primitive,
 * compared against Primitive.index (the original index), exactly as the
 * reference does (GI:829, call site GI:995-997). */
static float processShadows(const OracleScene *s, const SceneInfo *si, v3 lampCenter, v3 origin, int lightId,
                            int iteration, c3 *color, int objectId, Stats *st)
{
    float result = 0.f;
    int cptBoxes = 0;
    color->x = 0.f;
    color->y = 0.f;
    color->z = 0.f;
    Ray r;
    r.direction = vsub(lampCenter, origin);
    r.origin = vadd(origin, vscale(vnormalize(r.direction), si->rayEpsilon));
    computeRayAttributes(&r);
    const float minDistance = (iteration < 2) ? si->viewDistance : si->viewDistance / (iteration + 1);
    st->shadow++;

    while (result < (si->shadowIntensity) && cptBoxes < s->nbBoxes)
    {
        const BoundingBox *box = &s->boxes[cptBoxes];
        st->boxes++;
        /* GI:817 tests the node against [0, minDistance]; CL:1528 against [0.05, minDistance] */
        if (boxIntersection(box, &r, DIALECT(14) ? 0.05f : 0.f, minDistance))
        {
            int cptPrimitives = 0;
            while (result < si->shadowIntensity && cptPrimitives < box->nbPrimitives)
            {
                v3 intersection = {0.f, 0.f, 0.f};
                v3 normal = {0.f, 0.f, 0.f};
                v3 areas = {0.f, 0.f, 0.f};
                float shadowIntensity = 0.f;
                const Primitive *primitive = &s->primitives[box->startIndex + cptPrimitives];
                const Material *pm = &s->materials[primitive->materialId];
                /* GI:829 leaves out the lamp and the shaded primitive; CL:1539 only what its caller passes as
                 * objectId, which is the lamp's primitive (CL:1709-1711) */
                if (primitive->index != lightId && (DIALECT(15) || primitive->index != objectId) && pm->attributes.x == 0)
                {
                    st->prims++;
                    int hit = testPrimitiveShadow(si, primitive, s->materials, s->textures, &r, &intersection,
                                                  &normal, &areas, &shadowIntensity);
                    if (hit)
                    {
                        v3 O_I = vsub(intersection, r.origin);
                        v3 O_L = r.direction;
                        float l = vlength(O_I);
                        if (l > si->geometryEpsilon && l < vlength(O_L))
                        {
                            float ratio = shadowIntensity * si->shadowIntensity;
                            if (pm->transparency != 0.f)
                            {
                                O_L = vnormalize(O_L);
                                float a = fabsf(vdot(O_L, normal));
                                /* GI:885-886; CL:1589-1591 lets 20 % more through */
                                float rr = (pm->transparency == 0.f)
                                               ? 1.f
                                               : (DIALECT(16) ? (1.f - 0.8f * pm->transparency) : (1.f - pm->transparency));
                                ratio *= rr * a;
                                color->x += ratio * (0.3f - 0.3f * pm->color.x);
                                color->y += ratio * (0.3f - 0.3f * pm->color.y);
                                color->z += ratio * (0.3f - 0.3f * pm->color.z);
                            }
                            result += ratio;
                        }
                    }
                }
                ++cptPrimitives;
            }
            ++cptBoxes;
        }
        else
            cptBoxes += box->indexForNextBox.x;
    }
    /* GI:906: max/min are the float overloads on the device */
    result = fmaxf(0.f, fminf(result, si->shadowIntensity));
    return result;
}

float oracle_shadow(const OracleScene *scene, const SceneInfo *sceneInfo, const float lampCenter[3],
                    const float origin[3], int lightId, int iteration, int objectId, float color[3])
{
    Stats st;
    memset(&st, 0, sizeof(st));
    c3 c;
    float r = processShadows(scene, sceneInfo, V(lampCenter[0], lampCenter[1], lampCenter[2]),
                             V(origin[0], origin[1], origin[2]), lightId, iteration, &c, objectId, &st);
    color[0] = c.x;
    color[1] = c.y;
    color[2] = c.z;
    return r;
}

/* ref GS:36-124.  normal here is the shader's bumpNormal accumulator. */
static f4 intersectionShader(const SceneInfo *si, const Primitive *primitive, const Material *materials,
                             const BitmapBuffer *textures, v3 intersection, v3 areas, v3 *normal, f4 *specular,
                             f4 *attributes, f4 *advancedAttributes)
{
    const Material *m = &materials[primitive->materialId];
    f4 c = colorOf(m);
    c.w = 0.f;
    if (si->extendedGeometry)
    {
        switch (primitive->type)
        {
        case ptCone:
            if (DIALECT(17)) /* CL:1422-1487: no cone */
                break;
            /* fall through */
        case ptCylinder:
        case ptEnvironment:
        case ptSphere:
        case ptEllipsoid:
            if (m->textureIds.x != TEXTURE_NONE)
                c = sphereUVMapping(primitive, materials, textures, intersection, normal, specular, attributes,
                                    advancedAttributes);
            break;
        case ptCheckboard:
            if (m->textureIds.x != TEXTURE_NONE)
                c = cubeMapping(si, primitive, materials, textures, intersection, normal, specular, attributes,
                                advancedAttributes);
            else
            {
                int x = f2i(si->viewDistance + ((intersection.x - primitive->p0.x) / primitive->size.x));
                int z = f2i(si->viewDistance + ((intersection.z - primitive->p0.z) / primitive->size.x));
                if (x % 2 == 0)
                {
                    if (z % 2 == 0)
                    {
                        c.x = 1.f - c.x;
                        c.y = 1.f - c.y;
                        c.z = 1.f - c.z;
                    }
                }
                else
                {
                    if (z % 2 != 0)
                    {
                        c.x = 1.f - c.x;
                        c.y = 1.f - c.y;
                        c.z = 1.f - c.z;
                    }
                }
            }
            break;
        case ptXYPlane:
        case ptYZPlane:
        case ptXZPlane:
        case ptCamera:
            if (m->textureIds.x != TEXTURE_NONE)
                c = cubeMapping(si, primitive, materials, textures, intersection, normal, specular, attributes,
                                advancedAttributes);
            break;
        case ptTriangle:
            if (m->textureIds.x != TEXTURE_NONE)
                c = triangleUVMapping(si, primitive, materials, textures, areas, normal, specular, attributes,
                                      advancedAttributes);
            break;
        default:
            break;
        }
    }
    else
    {
        if (m->textureIds.x != TEXTURE_NONE)
            c = triangleUVMapping(si, primitive, materials, textures, areas, normal, specular, attributes,
                                  advancedAttributes);
    }
    return c;
}

/* ref GI:916-1080.  closestColor, totalBlinn and normal are in/out and
 * persist across bounces. */
static c3 primitiveShader(const OracleScene *s, int index, const SceneInfo *si, v3 origin, v3 *normal, int objectId,
                          v3 intersection, v3 areas, c3 *closestColor, int iteration, float *shadowIntensity,
                          c3 *totalBlinn, f4 *attributes, Stats *st)
{
    const Primitive *primitive = &s->primitives[objectId];
    const Material *material = &s->materials[primitive->materialId];
    c3 lampsColor = {0.f, 0.f, 0.f};
    *shadowIntensity = 0.f;
    v3 bumpNormal = {0.f, 0.f, 0.f};
    f4 advancedAttributes = {0.f, 0.f, 0.f, 0.f};
    f4 specular;
    specular.x = material->specular.x;
    specular.y = material->specular.y;
    specular.z = material->specular.z;
    specular.w = 0.f;

    f4 ic4 = intersectionShader(si, primitive, s->materials, s->textures, intersection, areas, &bumpNormal, &specular,
                                attributes, &advancedAttributes);
    c3 intersectionColor = {ic4.x, ic4.y, ic4.z};
    *normal = vadd(*normal, bumpNormal);
    *normal = vnormalize(*normal);

    if (DIALECT(18))
    {
        /* CL:1652-1659: unshaded frames, emissive and any wireframe material return the texel */
        if (si->graphicsLevel == glNoShading || material->innerIllumination.x != 0.f || material->attributes.z != 0)
            return intersectionColor;
    }
    else if (material->attributes.z == 1)
        return intersectionColor; /* wireframe: constant colour */

    if (si->graphicsLevel > glNoShading)
    {
        closestColor->x *= material->innerIllumination.x;
        closestColor->y *= material->innerIllumination.x;
        closestColor->z *= material->innerIllumination.x;
        /* GI:956: once per entry of the light list; CL:1664-1665: once */
        for (int cpt = 0; cpt < (DIALECT(19) ? 1 : s->nbLights); ++cpt)
        {
            /* GI:958-960: lamp 0 is applied nbLights times below iteration 10 */
            int cptLamp = (si->pathTracingIteration >= NB_MAX_ITERATIONS) ? (si->pathTracingIteration % s->nbLights) : 0;
            const LightInformation *li = &s->lights[cptLamp];
            if (li->primitiveId != primitive->index)
            {
                v3 center = li->location;
                int t = (index + si->timestamp) % (MAX_BITMAP_SIZE - 3);
                const Material *m = &s->materials[li->materialId];
                /* CL:1678-1682 jitters only lamps that are primitives of the scene */
                if (si->pathTracingIteration >= NB_MAX_ITERATIONS &&
                    (!DIALECT(20) || (li->primitiveId >= 0 && li->primitiveId < s->nbPrimitives)))
                {
                    float a = m->innerIllumination.y * 10.f * si->pathTracingIteration / si->maxPathTracingIterations;
                    center.x += rnd(s, t, st) * a;
                    center.y += rnd(s, t + 1, st) * a;
                    center.z += rnd(s, t + 2, st) * a;
                }
                v3 lightRay = vsub(center, intersection);
                float lightRayLength = vlength(lightRay);
                if (lightRayLength < m->innerIllumination.z)
                {
                    c3 shadowColor = {0.f, 0.f, 0.f};
                    lightRay = vnormalize(lightRay);
                    /* GI:985; CL:1701 has the bare cosine */
                    float lambert = DIALECT(21) ? vdot(*normal, lightRay)
                                         : material->innerIllumination.x + vdot(*normal, lightRay);
                    if (lambert > 0.f && si->graphicsLevel > 3 && iteration < 4 && material->innerIllumination.x == 0.f)
                        *shadowIntensity = processShadows(s, si, center, intersection, li->primitiveId, iteration,
                                                          &shadowColor, objectId, st);
                    if (si->graphicsLevel > glNoShading)
                    {
                        float photonEnergy = sqrtf(lightRayLength / m->innerIllumination.z);
                        photonEnergy = (photonEnergy > 1.f) ? 1.f : photonEnergy;
                        photonEnergy = (photonEnergy < 0.f) ? 0.f : photonEnergy;
                        lambert *= (lambert < 0.f) ? -s->materials[primitive->materialId].transparency : 1.f;
                        if (li->materialId != MATERIAL_NONE)
                            lambert *= s->materials[li->materialId].innerIllumination.x;
                        else
                            lambert *= li->color.w;
                        if (material->innerIllumination.w != 0.f)
                            lambert *= (1.f + rnd(s, t, st) * material->innerIllumination.w * 100.f);
                        lambert *= (1.f - *shadowIntensity);
                        lambert += si->backgroundColor.w;
                        lambert *= (1.f - photonEnergy);
                        lampsColor.x += lambert * li->color.x - shadowColor.x;
                        lampsColor.y += lambert * li->color.y - shadowColor.y;
                        lampsColor.z += lambert * li->color.z - shadowColor.z;
                        if (si->graphicsLevel > 1 && *shadowIntensity < si->shadowIntensity)
                        {
                            v3 viewRay = vnormalize(vsub(intersection, origin));
                            v3 blinnDir = vsub(lightRay, viewRay);
                            float temp = sqrtf(vdot(blinnDir, blinnDir));
                            if (temp != 0.f)
                            {
                                blinnDir = vscale(blinnDir, 1.f / temp);
                                float blinnTerm = vdot(blinnDir, *normal);
                                blinnTerm = (blinnTerm < 0.f) ? 0.f : blinnTerm;
                                blinnTerm = specular.x * specularPower(blinnTerm, specular.y);
                                blinnTerm *= (1.f - photonEnergy);
                                totalBlinn->x += li->color.x * li->color.w * blinnTerm;
                                totalBlinn->y += li->color.y * li->color.w * blinnTerm;
                                totalBlinn->z += li->color.z * li->color.w * blinnTerm;
                            }
                        }
                    }
                }
            }
            closestColor->x += intersectionColor.x * lampsColor.x;
            closestColor->y += intersectionColor.y * lampsColor.y;
            closestColor->z += intersectionColor.z * lampsColor.z;
            if (material->advancedTextureIds.z != TEXTURE_NONE)
            {
                closestColor->x *= advancedAttributes.x;
                closestColor->y *= advancedAttributes.x;
                closestColor->z *= advancedAttributes.x;
            }
            saturate3(closestColor);
            saturate3(totalBlinn);
        }
    }
    else
        *closestColor = intersectionColor;
    return *closestColor;
}

/* ref CRT:69-408 */
static c3 launchRayTracing(const OracleScene *s, int index, const Ray *ray, const SceneInfo *si, float *depthOfField,
                           PrimitiveXYIdBuffer *primitiveXYId, Stats *st)
{
    c3 intersectionColor = {0.f, 0.f, 0.f};
    v3 closestIntersection = {0.f, 0.f, 0.f};
    v3 firstIntersection = {0.f, 0.f, 0.f};
    v3 normal = {0.f, 0.f, 0.f};
    int closestPrimitive = DIALECT(22) ? 0 : -1; /* CRT:76; CL:2117 */
    int carryon = 1;
    Ray rayOrigin = *ray;
    float initialRefraction = 1.f;
    int iteration = 0;
    primitiveXYId->x = -1;
    primitiveXYId->z = 0;
    if (!DIALECT(23)) /* CRT:83; CL:2122-2123 leaves w as the caller's buffer holds it */
        primitiveXYId->w = 0;
    int currentMaterialId = -2;

    float colorContributions[NB_MAX_ITERATIONS + 1];
    c3 colors[NB_MAX_ITERATIONS + 1];
    memset(colorContributions, 0, sizeof(colorContributions));
    memset(colors, 0, sizeof(colors));

    c3 recursiveBlinn = {0.f, 0.f, 0.f};
    float shadowIntensity = 0.f;
    v3 reflectedTarget = {0.f, 0.f, 0.f};
    c3 closestColor = {0.f, 0.f, 0.f};
    c3 colorBox = {0.f, 0.f, 0.f};
    v3 latestIntersection = ray->origin;
    float rayLength = 0.f;
    *depthOfField = si->viewDistance;

    int reflectedRays = -1;
    Ray reflectedRay;
    float reflectedRatio = 0.f;
    memset(&reflectedRay, 0, sizeof(reflectedRay));

    Ray pathTracingRay;
    memset(&pathTracingRay, 0, sizeof(pathTracingRay)); /* uninitialised in the reference (UB); zero here */
    float pathTracingRatio = 0.f;
    c3 pathTracingColor = {0.f, 0.f, 0.f};
    int useGlobalIllumination = 0;

    c3 rBlinn = {0.f, 0.f, 0.f};
    int currentMaxIteration =
        (si->graphicsLevel < glReflectionsAndRefractions) ? 1 : si->nbRayIterations + si->pathTracingIteration;
    currentMaxIteration = (currentMaxIteration > NB_MAX_ITERATIONS) ? NB_MAX_ITERATIONS : currentMaxIteration;

    while (iteration < currentMaxIteration && rayLength < si->viewDistance && carryon)
    {
        v3 areas = {0.f, 0.f, 0.f};
        if (carryon)
            carryon = intersectionWithPrimitives(s, si, &rayOrigin, iteration, &closestPrimitive, &closestIntersection,
                                                 &normal, &areas, &colorBox, currentMaterialId, st);
        if (carryon)
        {
            const Primitive *cp = &s->primitives[closestPrimitive];
            const Material *cm = &s->materials[cp->materialId];
            currentMaterialId = cp->materialId;

            f4 attributes;
            attributes.x = cm->reflection;
            attributes.y = cm->transparency;
            attributes.z = cm->refraction;
            attributes.w = cm->opacity;

            if (iteration == 0)
            {
                colors[iteration].x = 0.f;
                colors[iteration].y = 0.f;
                colors[iteration].z = 0.f;
                colorContributions[iteration] = 1.f;
                firstIntersection = closestIntersection;
                latestIntersection = closestIntersection;
                if (!DIALECT(24)) /* CRT:139; CL:2411-2412 takes the length after the loop, hit or not */
                    *depthOfField = vlength(vsub(firstIntersection, ray->origin));

                if (!DIALECT(25) && cm->innerIllumination.x == 0.f && /* the CL global-illumination ray is not restated */
                    (si->advancedIllumination == aiBasic || si->advancedIllumination == aiFull))
                {
                    int t = (index + si->pathTracingIteration * 100 + si->timestamp) % (MAX_BITMAP_SIZE - 3);
                    pathTracingRay.origin = vadd(closestIntersection, vscale(normal, si->rayEpsilon));
                    pathTracingRay.direction.x = normal.x + 100.f * rnd(s, t, st);
                    pathTracingRay.direction.y = normal.y + 100.f * rnd(s, t + 1, st);
                    pathTracingRay.direction.z = normal.z + 100.f * rnd(s, t + 2, st);
                    float cos_theta = vdot(vnormalize(pathTracingRay.direction), normal);
                    if (cos_theta < 0.f)
                        pathTracingRay.direction = vneg(pathTracingRay.direction);
                    pathTracingRay.direction = vadd(pathTracingRay.direction, closestIntersection);
                    pathTracingRatio = (1.f - attributes.y) * fabsf(cos_theta);
                    useGlobalIllumination = 1;
                }
                primitiveXYId->x = cp->index;
            }

            colors[iteration] = primitiveShader(s, index, si, rayOrigin.origin, &normal, closestPrimitive,
                                                closestIntersection, areas, &closestColor, iteration,
                                                &shadowIntensity, &rBlinn, &attributes, st);

            if (DIALECT(26))
            {
                /* CL:2226-2227: sixteen per bounce whose shaded colour is brighter than the colour key */
                float colorLight = colors[iteration].x + colors[iteration].y + colors[iteration].z;
                primitiveXYId->z += (colorLight > si->transparentColor) ? 16 : 0;
            }
            else
                /* CRT:190: int += float*int, evaluated in float then truncated */
                primitiveXYId->z = f2i((float)primitiveXYId->z + cm->innerIllumination.x * 256);

            float segmentLength = vlength(vsub(closestIntersection, latestIntersection));
            latestIntersection = closestIntersection;

            float transparency = attributes.y;
            float a = 0.f;
            if (attributes.y != 0.f)
            {
                float refraction = attributes.z;
                if (initialRefraction == refraction)
                {
                    refraction = 1.f;
                    float length = segmentLength * (attributes.w * (1.f - transparency));
                    rayLength += length;
                    rayLength = (rayLength > si->viewDistance) ? si->viewDistance : rayLength;
                    a = (rayLength / si->viewDistance);
                    colors[iteration].x -= a;
                    colors[iteration].y -= a;
                    colors[iteration].z -= a;
                }
                v3 O_E = vnormalize(vsub(closestIntersection, rayOrigin.origin));
                reflectedTarget = vectorRefraction(O_E, refraction, normal, initialRefraction);
                colorContributions[iteration] = transparency - a;
                initialRefraction = refraction;
                if (reflectedRays == -1 && attributes.x != 0.f)
                {
                    reflectedRay.direction = vectorReflection(O_E, normal);
                    reflectedRay.origin = vadd(closestIntersection, vscale(reflectedRay.direction, si->rayEpsilon));
                    reflectedRay.direction = vadd(closestIntersection, reflectedRay.direction);
                    reflectedRatio = attributes.x;
                    reflectedRays = iteration;
                }
            }
            else if (attributes.x != 0.f)
            {
                v3 O_E = vnormalize(vsub(closestIntersection, rayOrigin.origin));
                reflectedTarget = vectorReflection(O_E, normal);
                colorContributions[iteration] = attributes.x;
            }
            else
            {
                carryon = 0;
                colorContributions[iteration] = 1.f;
            }

            /* CRT:248: float4 /= int -> division by (float)(iteration+1) */
            rBlinn.x /= (float)(iteration + 1);
            rBlinn.y /= (float)(iteration + 1);
            rBlinn.z /= (float)(iteration + 1);
            recursiveBlinn.x = (rBlinn.x > recursiveBlinn.x) ? rBlinn.x : recursiveBlinn.x;
            recursiveBlinn.y = (rBlinn.y > recursiveBlinn.y) ? rBlinn.y : recursiveBlinn.y;
            recursiveBlinn.z = (rBlinn.z > recursiveBlinn.z) ? rBlinn.z : recursiveBlinn.z;

            rayOrigin.origin = vadd(closestIntersection, vscale(reflectedTarget, si->rayEpsilon));
            rayOrigin.direction = vadd(closestIntersection, reflectedTarget);

            if (si->pathTracingIteration != 0 && cm->color.w != 0.f)
            {
                float ratio = cm->color.w;
                ratio *= (attributes.y == 0.f) ? 1000.f : 1.f;
                int rindex = (index + si->timestamp) % (MAX_BITMAP_SIZE - 3);
                rayOrigin.direction.x += rnd(s, rindex, st) * ratio;
                rayOrigin.direction.y += rnd(s, rindex + 1, st) * ratio;
                rayOrigin.direction.z += rnd(s, rindex + 2, st) * ratio;
            }
        }
        else
        {
            if (si->skyboxMaterialId != MATERIAL_NONE)
            {
                colors[iteration] = skyboxMapping(si, s->materials, s->textures, &rayOrigin);
                if (!DIALECT(27)) /* CRT:274-275; absent from CL:2314-2315 */
                {
                    float rad = colors[iteration].x + colors[iteration].y + colors[iteration].z;
                    primitiveXYId->z = f2i((float)primitiveXYId->z + ((rad > 2.5f) ? rad * 256.f : 0.f));
                }
            }
            else if (DIALECT(28) ? (si->extendedGeometry == 2) /* CL:2318 */ : si->gradientBackground)
            {
                v3 up = {0.f, 1.f, 0.f};
                v3 dir = vnormalize(vsub(rayOrigin.direction, rayOrigin.origin));
                float angle = 0.5f - vdot(up, dir);
                angle = (angle > 1.f) ? 1.f : angle;
                colors[iteration].x = (1.f - angle) * si->backgroundColor.x;
                colors[iteration].y = (1.f - angle) * si->backgroundColor.y;
                colors[iteration].z = (1.f - angle) * si->backgroundColor.z;
            }
            else
            {
                colors[iteration].x = si->backgroundColor.x;
                colors[iteration].y = si->backgroundColor.y;
                colors[iteration].z = si->backgroundColor.z;
            }
            colorContributions[iteration] = 1.f;
        }
        iteration++;
    }

    v3 areas = {0.f, 0.f, 0.f};
    if (si->graphicsLevel >= glReflectionsAndRefractions && reflectedRays != -1)
        if (intersectionWithPrimitives(s, si, &reflectedRay, reflectedRays, &closestPrimitive, &closestIntersection,
                                       &normal, &areas, &colorBox, currentMaterialId, st))
        {
            f4 attributes = {0.f, 0.f, 0.f, 0.f}; /* only .x is set in the reference (CRT:305-306) */
            attributes.x = s->materials[s->primitives[closestPrimitive].materialId].reflection;
            /* CRT:307-311 shades it as bounce `reflectedRays`, CL:2345-2349 as bounce `iteration` */
            c3 color = primitiveShader(s, index, si, reflectedRay.origin, &normal, closestPrimitive,
                                       closestIntersection, areas, &closestColor, DIALECT(29) ? iteration : reflectedRays,
                                       &shadowIntensity, &rBlinn, &attributes, st);
            colors[reflectedRays].x += color.x * reflectedRatio;
            colors[reflectedRays].y += color.y * reflectedRatio;
            colors[reflectedRays].z += color.z * reflectedRatio;
            primitiveXYId->w = f2i(shadowIntensity * 255);
        }

    int test = 1;
    if ((si->advancedIllumination == aiBasic || si->advancedIllumination == aiFull) &&
        si->pathTracingIteration >= NB_MAX_ITERATIONS)
    {
        if (useGlobalIllumination && si->advancedIllumination == aiFull)
        {
            if (intersectionWithPrimitives(s, si, &pathTracingRay, 30, &closestPrimitive, &closestIntersection,
                                           &normal, &areas, &colorBox, MATERIAL_NONE, st))
            {
                if (s->primitives[closestPrimitive].materialId != MATERIAL_NONE)
                {
                    const Material *material = &s->materials[s->primitives[closestPrimitive].materialId];
                    if (material->innerIllumination.x == 0.f)
                    {
                        colors[0].x = material->color.x * material->innerIllumination.x * pathTracingRatio;
                        colors[0].y = material->color.y * material->innerIllumination.x * pathTracingRatio;
                        colors[0].z = material->color.z * material->innerIllumination.x * pathTracingRatio;
                        test = 0;
                    }
                    else
                    {
                        colors[0].x = material->color.x * pathTracingRatio;
                        colors[0].y = material->color.y * pathTracingRatio;
                        colors[0].z = material->color.z * pathTracingRatio;
                    }
                }
                if (test)
                {
                    pathTracingRatio *= STANDARD_LUNINANCE_STRENGTH;
                    f4 attributes = {0.f, 0.f, 0.f, 0.f};
                    const Material *material = &s->materials[s->primitives[closestPrimitive].materialId];
                    if (material->innerIllumination.x == 0.f)
                    {
                        colors[0].x -= si->shadowIntensity;
                        colors[0].y -= si->shadowIntensity;
                        colors[0].z -= si->shadowIntensity;
                    }
                    else
                        pathTracingColor =
                            primitiveShader(s, index, si, pathTracingRay.origin, &normal, closestPrimitive,
                                            closestIntersection, areas, &closestColor, iteration, &shadowIntensity,
                                            &rBlinn, &attributes, st);
                }
            }
            else if (si->skyboxMaterialId != MATERIAL_NONE)
            {
                pathTracingColor = skyboxMapping(si, s->materials, s->textures, &pathTracingRay);
                pathTracingRatio *= SKYBOX_LUNINANCE_STRENGTH;
            }
        }
        else if (si->skyboxMaterialId != MATERIAL_NONE)
        {
            pathTracingColor = skyboxMapping(si, s->materials, s->textures, &pathTracingRay);
            pathTracingRatio *= SKYBOX_LUNINANCE_STRENGTH;
        }
        if (test)
        {
            colors[0].x += pathTracingColor.x * pathTracingRatio;
            colors[0].y += pathTracingColor.y * pathTracingRatio;
            colors[0].z += pathTracingColor.z * pathTracingRatio;
        }
    }

    if (test)
    {
        for (int i = iteration - 2; i >= 0; --i)
        {
            colors[i].x = colors[i].x * (1.f - colorContributions[i]) + colors[i + 1].x * colorContributions[i];
            colors[i].y = colors[i].y * (1.f - colorContributions[i]) + colors[i + 1].y * colorContributions[i];
            colors[i].z = colors[i].z * (1.f - colorContributions[i]) + colors[i + 1].z * colorContributions[i];
        }
        intersectionColor = colors[0];
        intersectionColor.x += recursiveBlinn.x;
        intersectionColor.y += recursiveBlinn.y;
        intersectionColor.z += recursiveBlinn.z;
    }
    else
        intersectionColor = colors[0];

    float len = *depthOfField;
    if (DIALECT(30))
    {
        /* CL:2411-2418: the depth is taken here, from the first hit or from the origin of the frame of reference
         * when there was none; a wireframe material under the last hit pushes the fog to the horizon */
        len = vlength(vsub(firstIntersection, ray->origin));
        *depthOfField = len;
        if (closestPrimitive != -1 && s->materials[s->primitives[closestPrimitive].materialId].attributes.z == 1)
            len = si->viewDistance;
    }
    float D1 = si->viewDistance * 0.95f;
    if (si->atmosphericEffect == aeFog && len > D1)
    {
        float D2 = si->viewDistance * 0.05f;
        float a = len - D1;
        float b = 1.f - (a / D2);
        intersectionColor.x = intersectionColor.x * b + si->backgroundColor.x * (1.f - b);
        intersectionColor.y = intersectionColor.y * b + si->backgroundColor.y * (1.f - b);
        intersectionColor.z = intersectionColor.z * b + si->backgroundColor.z * (1.f - b);
    }

    primitiveXYId->y = iteration;
    intersectionColor.x -= colorBox.x;
    intersectionColor.y -= colorBox.y;
    intersectionColor.z -= colorBox.z;
    if (DIALECT(31)) /* CL:2438; CRT:404-407 returns the colour as it is */
        saturate3(&intersectionColor);
    return intersectionColor;
}

/* ref GS:132-165 */
static void makeColor(const SceneInfo *si, c3 color, BitmapBuffer *bitmap, int index)
{
    int mdc_index = index * SOLR_COLOR_DEPTH;
    color.x = (color.x > 1.f) ? 1.f : color.x;
    color.y = (color.y > 1.f) ? 1.f : color.y;
    color.z = (color.z > 1.f) ? 1.f : color.z;
    color.x = (color.x < 0.f) ? 0.f : color.x;
    color.y = (color.y < 0.f) ? 0.f : color.y;
    color.z = (color.z < 0.f) ? 0.f : color.z;
    switch (si->frameBufferType)
    {
    case ftBGR:
    {
        int y = index / si->size.y;
        int x = index % si->size.x;
        int i = (y + 1) * si->size.y - x - 1;
        i *= SOLR_COLOR_DEPTH;
        bitmap[i] = (BitmapBuffer)(color.z * 255.f);
        bitmap[i + 1] = (BitmapBuffer)(color.y * 255.f);
        bitmap[i + 2] = (BitmapBuffer)(color.x * 255.f);
        break;
    }
    default:
        bitmap[mdc_index] = (BitmapBuffer)(color.x * 255.f);
        bitmap[mdc_index + 1] = (BitmapBuffer)(color.y * 255.f);
        bitmap[mdc_index + 2] = (BitmapBuffer)(color.z * 255.f);
        break;
    }
}

void oracle_make_color(const SceneInfo *sceneInfo, const float color[3], BitmapBuffer *bitmap, int index)
{
    c3 c = {color[0], color[1], color[2]};
    makeColor(sceneInfo, c, bitmap, index);
}

/* ref CRT:437-563 for one pixel.  yLocal is the row inside the strip,
 * firstRow the reference's device_split.  Random indices use the GLOBAL pixel
 * index so that an N-strip render equals the 1-strip render (deviation from
 * the reference's strip-local index, which only matters when nbGPUs > 1). */
static void standardRendererPixel(const OracleScene *s, const SceneInfo *si, const PostProcessingInfo *ppi, v3 origin,
                                  v3 direction, const float angles[4], const Trig *trig, int x, int yLocal,
                                  int firstRow, PostProcessingBuffer *pp, PrimitiveXYIdBuffer *ids, Stats *st)
{
    static const float AAx[4] = {3.f, 5.f, -3.f, -5.f};
    static const float AAy[4] = {5.f, -3.f, -5.f, 3.f};
    int index = yLocal * si->size.x + x;
    int gindex = (firstRow + yLocal) * si->size.x + x;

    if (si->pathTracingIteration > ids[index].y && ids[index].w == 0 && si->pathTracingIteration > 0 &&
        si->pathTracingIteration <= NB_MAX_ITERATIONS)
        return;

    Ray ray;
    memset(&ray, 0, sizeof(ray));
    ray.origin = origin;
    ray.direction = direction;
    v3 rotationCenter = {0.f, 0.f, 0.f};
    if (si->cameraType == ctVR)
        rotationCenter = origin;
    int antialiasingActivated = (si->cameraType == ctAntialiazed);

    if (ppi->type != ppe_depthOfField && si->pathTracingIteration >= NB_MAX_ITERATIONS)
    {
        float a = (ppi->param1 / 20000.f);
        long rindex = (long)gindex + si->timestamp % (MAX_BITMAP_SIZE - 2); /* CRT:475 precedence */
        ray.origin.x += rnd(s, rindex, st) * pp[index].colorInfo.w * a;
        ray.origin.y += rnd(s, rindex + 1, st) * pp[index].colorInfo.w * a;
    }

    float dof = 0.f;
    int yGlobal = firstRow + yLocal;
    if (si->cameraType == ctOrthographic)
    {
        ray.direction.x = ray.origin.z * 0.001f * (float)(x - (si->size.x / 2));
        ray.direction.y = -ray.origin.z * 0.001f * (float)(yGlobal - (si->size.y / 2));
        ray.origin.x = ray.direction.x;
        ray.origin.y = ray.direction.y;
    }
    else
    {
        float ratio = (float)si->size.x / (float)si->size.y;
        float stepx = ratio * angles[3] / (float)si->size.x;
        float stepy = angles[3] / (float)si->size.y;
        ray.direction.x = ray.direction.x - stepx * (float)(x - (si->size.x / 2));
        ray.direction.y = ray.direction.y + stepy * (float)(yGlobal - (si->size.y / 2));
    }

    ray.origin = vectorRotation(ray.origin, rotationCenter, trig);
    ray.direction = vectorRotation(ray.direction, rotationCenter, trig);

    c3 color = {0.f, 0.f, 0.f};
    Ray r = ray;
    if (antialiasingActivated)
    {
        for (int I = 0; I < 4; ++I)
        {
            r.origin.x += AAx[I];
            r.origin.y += AAy[I];
            c3 c = launchRayTracing(s, gindex, &r, si, &dof, &ids[index], st);
            color.x += c.x;
            color.y += c.y;
            color.z += c.z;
        }
    }
    else if (si->pathTracingIteration >= NB_MAX_ITERATIONS)
    {
        r.direction.x += AAx[si->pathTracingIteration % 4];
        r.direction.y += AAy[si->pathTracingIteration % 4];
    }
    {
        c3 c = launchRayTracing(s, gindex, &r, si, &dof, &ids[index], st);
        color.x += c.x;
        color.y += c.y;
        color.z += c.z;
    }

    if (si->advancedIllumination == aiRandomIllumination)
    {
        int rindex = (gindex + si->timestamp) % MAX_BITMAP_SIZE;
        float rv = rnd(s, rindex, st);
        color.x += si->backgroundColor.x * rv * 5.f;
        color.y += si->backgroundColor.y * rv * 5.f;
        color.z += si->backgroundColor.z * rv * 5.f;
    }

    if (antialiasingActivated)
    {
        color.x /= 5.f;
        color.y /= 5.f;
        color.z /= 5.f;
    }

    if (si->pathTracingIteration == 0)
        pp[index].colorInfo.w = dof;

    if (si->pathTracingIteration <= NB_MAX_ITERATIONS)
    {
        pp[index].colorInfo.x = color.x;
        pp[index].colorInfo.y = color.y;
        pp[index].colorInfo.z = color.z;
        pp[index].sceneInfo.x = color.x;
        pp[index].sceneInfo.y = color.y;
        pp[index].sceneInfo.z = color.z;
    }
    else
    {
        pp[index].sceneInfo.x = (ids[index].z > 0) ? fmaxf(pp[index].sceneInfo.x, color.x) : color.x;
        pp[index].sceneInfo.y = (ids[index].z > 0) ? fmaxf(pp[index].sceneInfo.y, color.y) : color.y;
        pp[index].sceneInfo.z = (ids[index].z > 0) ? fmaxf(pp[index].sceneInfo.z, color.z) : color.z;
        pp[index].colorInfo.x += pp[index].sceneInfo.x;
        pp[index].colorInfo.y += pp[index].sceneInfo.y;
        pp[index].colorInfo.z += pp[index].sceneInfo.z;
    }
}

/* ref GI:1088-1265 intersectionsWithPrimitives, the body of launchVolumeRendering (CRT:50-67): EVERY primitive the
 * ray meets beyond postProcessingInfo.param1 is shaded (primitiveShader as for a first hit: iteration 0, its own
 * shadow rays) and kept, nearest first, in ten layers; the layers are then composited front to back along the ray in
 * 500 steps of viewDistance / 500, each step adding the layer it is in with weight 1 / param2.  The walk is the
 * closest-hit walk's (skip pointers, boxIntersection against [0, viewDistance]) without its cut-off: nothing hides
 * anything.  As written there:
 *   - the insertion shifts colors[9] into colors[10], one element behind the array (GI:1198-1200, MAXDEPTH = 10):
 *     a store nobody reads; eleven elements here;
 *   - VOLUME_RENDERING_NORMALS is not defined (GI:22, Consts.h:57): N stays 0;
 *   - `normalize(color)` (GI:1254) discards its result;
 *   - the first value of `color` is colors[0] * backgroundColor.w, all four components (GI:1227), and .w is
 *     overwritten with the nearest layer's distance at the end (GI:1263).
 * The OpenCL engine's function of the same name (CL:1909-2071) is another effect (no shading, no threshold, bands
 * subtracted from the background colour, a normalised result): it is not restated, and this one does not depend on
 * the dialect switch beyond what primitiveShader and the intersection tests do. */
static f4 volumeIntersectionsWithPrimitives(const OracleScene *s, int index, const SceneInfo *si,
                                            const PostProcessingInfo *ppi, const Ray *ray, Stats *st)
{
    enum { MAXDEPTH = 10 };
    Ray r;
    r.origin = ray->origin;
    r.direction = vsub(ray->direction, ray->origin);
    computeRayAttributes(&r);

    v3 intersection = ray->origin;
    v3 normal = {0.f, 0.f, 0.f};
    float shadowIntensity = 0.f;
    f4 colors[MAXDEPTH + 1];
    for (int k = 0; k < MAXDEPTH; ++k)
    {
        colors[k].x = colors[k].y = colors[k].z = 0.f;
        colors[k].w = si->viewDistance;
    }
    colors[MAXDEPTH] = colors[MAXDEPTH - 1];
    st->closest++;

    int nbIntersections = 0;
    int cptBoxes = 0;
    while (cptBoxes < s->nbBoxes)
    {
        const BoundingBox *box = &s->boxes[cptBoxes];
        st->boxes++;
        if (boxIntersection(box, &r, 0.f, si->viewDistance))
        {
            for (int cptPrimitives = 0; cptPrimitives < box->nbPrimitives; ++cptPrimitives)
            {
                const Primitive *primitive = &s->primitives[box->startIndex + cptPrimitives];
                const Material *material = &s->materials[primitive->materialId];
                v3 areas = {0.f, 0.f, 0.f};
                st->prims++;
                const int hit = testPrimitive(si, primitive, s->materials, s->textures, &r, &intersection, &normal, &areas,
                                              &shadowIntensity);
                if (!hit)
                    continue;
                const float dist = vlength(vsub(intersection, r.origin));
                if (!(dist > ppi->param1))
                    continue;
                ++nbIntersections;
                f4 color = colorOf(material);
                if (si->graphicsLevel != glNoShading)
                {
                    f4 attributes;
                    attributes.x = material->reflection;
                    attributes.y = material->transparency;
                    attributes.z = material->refraction;
                    attributes.w = material->opacity;
                    c3 rBlinn = {0.f, 0.f, 0.f};
                    c3 closestColor = {material->color.x, material->color.y, material->color.z};
                    shadowIntensity = 0.f;
                    const c3 shaded = primitiveShader(s, index, si, r.origin, &normal, box->startIndex + cptPrimitives,
                                                      intersection, areas, &closestColor, 0, &shadowIntensity, &rBlinn,
                                                      &attributes, st);
                    color.x = shaded.x;
                    color.y = shaded.y;
                    color.z = shaded.z;
                }
                for (int k = 0; k < MAXDEPTH; ++k)
                    if (dist < colors[k].w)
                    {
                        const float a = vdot(vnormalize(vsub(ray->direction, ray->origin)), normal);
                        for (int j = MAXDEPTH - 1; j >= k; --j)
                            colors[j + 1] = colors[j];
                        colors[k].x = color.x * fabsf(a);
                        colors[k].y = color.y * fabsf(a);
                        colors[k].z = color.z * fabsf(a);
                        colors[k].w = dist;
                        break;
                    }
            }
            ++cptBoxes;
        }
        else
            cptBoxes += box->indexForNextBox.x;
    }

    f4 color;
    color.x = colors[0].x * si->backgroundColor.w;
    color.y = colors[0].y * si->backgroundColor.w;
    color.z = colors[0].z * si->backgroundColor.w;
    if (nbIntersections > 0)
    {
        float D = colors[0].w;
        const int precision = 500;
        const float step = si->viewDistance / (float)precision;
        const float alpha = 1.f / ppi->param2;
        int c = 0;
        for (int k = 0; k < precision && c < MAXDEPTH - 1; ++k)
        {
            if (D > colors[c].w)
            {
                color.x += colors[c].x * alpha;
                color.y += colors[c].y * alpha;
                color.z += colors[c].z * alpha;
            }
            D += step;
            if (D >= colors[c + 1].w)
                ++c;
        }
    }
    color.w = colors[0].w;
    return color;
}

/* ref CRT:592-713, k_volumeRenderer for one pixel (dispatched for cameraType == ctVolumeRendering, CRT:1777-1806,
 * which rules the kernel's own five-ray, orthographic and VR branches out: they are not restated).  What differs
 * from k_standardRenderer's frame: the rotated-grid offset is added to the DIRECTION on every pass (CRT:670-671: no
 * test of the pass number), the ids are (-1, 1, 0, unchanged) (CRT:59-61), the depth written on pass 0 is the 0 the
 * kernel initialised it with (CRT:633, 686-687), and the random index of the depth-of-field jitter is an int
 * (CRT:626). */
static void volumeRendererPixel(const OracleScene *s, const SceneInfo *si, const PostProcessingInfo *ppi, v3 origin,
                                v3 direction, const float angles[4], const Trig *trig, int x, int yLocal, int firstRow,
                                PostProcessingBuffer *pp, PrimitiveXYIdBuffer *ids, Stats *st)
{
    static const float AAx[4] = {3.f, 5.f, -3.f, -5.f};
    static const float AAy[4] = {5.f, -3.f, -5.f, 3.f};
    const int index = yLocal * si->size.x + x;
    const int gindex = (firstRow + yLocal) * si->size.x + x;

    if (si->pathTracingIteration > ids[index].y && ids[index].w == 0 && si->pathTracingIteration > 0 &&
        si->pathTracingIteration <= NB_MAX_ITERATIONS)
        return;

    Ray ray;
    memset(&ray, 0, sizeof(ray));
    ray.origin = origin;
    ray.direction = direction;
    const v3 rotationCenter = {0.f, 0.f, 0.f};

    if (ppi->type != ppe_depthOfField && si->pathTracingIteration >= NB_MAX_ITERATIONS)
    {
        const float a = (ppi->param1 / 20000.f);
        const int rindex = gindex + si->timestamp % (MAX_BITMAP_SIZE - 2);
        ray.origin.x += rnd(s, rindex, st) * pp[index].colorInfo.w * a;
        ray.origin.y += rnd(s, rindex + 1, st) * pp[index].colorInfo.w * a;
    }

    const float dof = 0.f;
    const int yGlobal = firstRow + yLocal;
    {
        const float ratio = (float)si->size.x / (float)si->size.y;
        const float stepx = ratio * angles[3] / (float)si->size.x;
        const float stepy = angles[3] / (float)si->size.y;
        ray.direction.x = ray.direction.x - stepx * (float)(x - (si->size.x / 2));
        ray.direction.y = ray.direction.y + stepy * (float)(yGlobal - (si->size.y / 2));
    }
    ray.origin = vectorRotation(ray.origin, rotationCenter, trig);
    ray.direction = vectorRotation(ray.direction, rotationCenter, trig);

    Ray r = ray;
    r.direction.x = ray.direction.x + AAx[si->pathTracingIteration % 4];
    r.direction.y = ray.direction.y + AAy[si->pathTracingIteration % 4];

    ids[index].x = -1;
    ids[index].y = 1;
    ids[index].z = 0;
    const f4 traced = volumeIntersectionsWithPrimitives(s, gindex, si, ppi, &r, st);
    c3 color = {0.f + traced.x, 0.f + traced.y, 0.f + traced.z};

    if (si->advancedIllumination == aiRandomIllumination)
    {
        const int rindex = (gindex + si->timestamp) % MAX_BITMAP_SIZE;
        const float rv = rnd(s, rindex, st);
        color.x += si->backgroundColor.x * rv * 5.f;
        color.y += si->backgroundColor.y * rv * 5.f;
        color.z += si->backgroundColor.z * rv * 5.f;
    }

    if (si->pathTracingIteration == 0)
        pp[index].colorInfo.w = dof;

    if (si->pathTracingIteration <= NB_MAX_ITERATIONS)
    {
        pp[index].colorInfo.x = color.x;
        pp[index].colorInfo.y = color.y;
        pp[index].colorInfo.z = color.z;
        pp[index].sceneInfo.x = color.x;
        pp[index].sceneInfo.y = color.y;
        pp[index].sceneInfo.z = color.z;
    }
    else
    {
        pp[index].sceneInfo.x = (ids[index].z > 0) ? fmaxf(pp[index].sceneInfo.x, color.x) : color.x;
        pp[index].sceneInfo.y = (ids[index].z > 0) ? fmaxf(pp[index].sceneInfo.y, color.y) : color.y;
        pp[index].sceneInfo.z = (ids[index].z > 0) ? fmaxf(pp[index].sceneInfo.z, color.z) : color.z;
        pp[index].colorInfo.x += pp[index].sceneInfo.x;
        pp[index].colorInfo.y += pp[index].sceneInfo.y;
        pp[index].colorInfo.z += pp[index].sceneInfo.z;
    }
}

/* ref CRT:840-950, k_anaglyphRenderer for one pixel: one trace per eye (origin.x -+ eyeSeparation), red from
 * the left eye's luminance, green and blue from the right eye.  No jitter, no random-illumination term,
 * plain store / accumulate.  The row is the global one (the reference's kernel has no split argument). */
static void anaglyphRendererPixel(const OracleScene *s, const SceneInfo *si, v3 origin, v3 direction,
                                  const float angles[4], const Trig *trig, int x, int yLocal, int firstRow,
                                  PostProcessingBuffer *pp, PrimitiveXYIdBuffer *ids, Stats *st)
{
    int index = yLocal * si->size.x + x;
    int gindex = (firstRow + yLocal) * si->size.x + x;
    int yGlobal = firstRow + yLocal;
    if (si->pathTracingIteration > ids[index].y && ids[index].w == 0 && si->pathTracingIteration > 0 &&
        si->pathTracingIteration <= NB_MAX_ITERATIONS)
        return;
    v3 rotationCenter = {0.f, 0.f, 0.f};
    float dof = 0.f;
    float ratio = (float)si->size.x / (float)si->size.y;
    float stepx = ratio * angles[3] / (float)si->size.x;
    float stepy = angles[3] / (float)si->size.y;
    c3 eye[2];
    for (int e = 0; e < 2; ++e)
    {
        Ray r;
        memset(&r, 0, sizeof(r));
        r.origin.x = (e == 0) ? origin.x - si->eyeSeparation : origin.x + si->eyeSeparation;
        r.origin.y = origin.y;
        r.origin.z = origin.z;
        r.direction.x = direction.x - stepx * (float)(x - (si->size.x / 2));
        r.direction.y = direction.y + stepy * (float)(yGlobal - (si->size.y / 2));
        r.direction.z = direction.z;
        r.origin = vectorRotation(r.origin, rotationCenter, trig);
        r.direction = vectorRotation(r.direction, rotationCenter, trig);
        eye[e] = launchRayTracing(s, gindex, &r, si, &dof, &ids[index], st);
    }
    float r1 = eye[0].x * 0.299f + eye[0].y * 0.587f + eye[0].z * 0.114f;
    float g2 = eye[1].y;
    float b2 = eye[1].z;
    if (si->pathTracingIteration == 0)
        pp[index].colorInfo.w = dof;
    if (si->pathTracingIteration <= NB_MAX_ITERATIONS)
    {
        pp[index].colorInfo.x = r1 + 0.f;
        pp[index].colorInfo.y = 0.f + g2;
        pp[index].colorInfo.z = 0.f + b2;
    }
    else
    {
        pp[index].colorInfo.x += r1 + 0.f;
        pp[index].colorInfo.y += 0.f + g2;
        pp[index].colorInfo.z += 0.f + b2;
    }
}

/* ref CRT:953-1043, k_3DVisionRenderer for one pixel - what cudaRender dispatches for cameraType == ctVR
 * (CRT:1737-1755): the left half of the image is the left eye's view, the right half the right eye's, the
 * eyes' distance scaled by the look-at depth over the depth of the frame's focus pixel.  That depth is read
 * from postProcessingBuffer[size.x / 2 * size.y / 2].colorInfo.w while, on pass 0, the thread of that pixel
 * writes it: a race in the reference.  `focusDepth` is the value before the frame - the outcome in which the
 * focus pixel is written last - and on later passes the only one. */
static void visionRendererPixel(const OracleScene *s, const SceneInfo *si, const PostProcessingInfo *ppi, v3 origin,
                                v3 direction, const float angles[4], const Trig *trig, float focusDepth, int x,
                                int yLocal, int firstRow, PostProcessingBuffer *pp, PrimitiveXYIdBuffer *ids, Stats *st)
{
    int index = yLocal * si->size.x + x;
    int gindex = (firstRow + yLocal) * si->size.x + x;
    int yGlobal = firstRow + yLocal;
    if (si->pathTracingIteration > ids[index].y && ids[index].w == 0 && si->pathTracingIteration > 0 &&
        si->pathTracingIteration <= NB_MAX_ITERATIONS)
        return;
    float focus = fabsf(focusDepth - origin.z);
    float eyeSeparation = si->eyeSeparation * (direction.z / focus);
    v3 rotationCenter = origin; /* cameraType is ctVR here */
    float dof = ppi->param1;
    int halfWidth = si->size.x / 2;
    float ratio = (float)si->size.x / (float)si->size.y;
    float stepx = ratio * angles[3] / (float)si->size.x;
    float stepy = angles[3] / (float)si->size.y;
    Ray r;
    memset(&r, 0, sizeof(r));
    if (x < halfWidth)
    {
        r.origin.x = origin.x + eyeSeparation;
        r.direction.x = direction.x - stepx * (float)(x - (si->size.x / 2) + halfWidth / 2) + si->eyeSeparation;
    }
    else
    {
        r.origin.x = origin.x - eyeSeparation;
        r.direction.x = direction.x - stepx * (float)(x - (si->size.x / 2) - halfWidth / 2) - si->eyeSeparation;
    }
    r.origin.y = origin.y;
    r.origin.z = origin.z;
    r.direction.y = direction.y + stepy * (float)(yGlobal - (si->size.y / 2));
    r.direction.z = direction.z;
    r.origin = vectorRotation(r.origin, rotationCenter, trig);
    r.direction = vectorRotation(r.direction, rotationCenter, trig);
    c3 color = launchRayTracing(s, gindex, &r, si, &dof, &ids[index], st);
    if (si->advancedIllumination == aiRandomIllumination)
    {
        int rindex = (gindex + si->timestamp) % MAX_BITMAP_SIZE;
        float rv = rnd(s, rindex, st);
        color.x += si->backgroundColor.x * rv * 5.f;
        color.y += si->backgroundColor.y * rv * 5.f;
        color.z += si->backgroundColor.z * rv * 5.f;
    }
    if (si->pathTracingIteration == 0)
        pp[index].colorInfo.w = dof;
    if (si->pathTracingIteration <= NB_MAX_ITERATIONS)
    {
        pp[index].colorInfo.x = color.x;
        pp[index].colorInfo.y = color.y;
        pp[index].colorInfo.z = color.z;
    }
    else
    {
        pp[index].colorInfo.x += color.x;
        pp[index].colorInfo.y += color.y;
        pp[index].colorInfo.z += color.z;
    }
}

/* ref CRT:741-815, k_fishEyeRenderer for one pixel: 360 degrees around the Y axis across the image width -
 * the look-at point is rotated about the eye by angles.y + 2 pi x / W - plain store / accumulate.  The
 * rotation's cos / sin are per pixel.  cosf / sinf are only specified to an error bound (CUDA's: 2 ULP, glibc's: <1 ULP),
 * so no two conforming libraries agree bit for bit on 2048 different angles per row; the restatement takes the
 * correctly rounded value (binary64 libm rounded once), which every such bound admits, and so does the engine. */
static void fishEyeRendererPixel(const OracleScene *s, const SceneInfo *si, const PostProcessingInfo *ppi, v3 origin,
                                 v3 direction, const float angles[4], int x, int yLocal, int firstRow,
                                 PostProcessingBuffer *pp, PrimitiveXYIdBuffer *ids, Stats *st)
{
    int index = yLocal * si->size.x + x;
    int gindex = (firstRow + yLocal) * si->size.x + x;
    int yGlobal = firstRow + yLocal;
    if (si->pathTracingIteration > ids[index].y && ids[index].w == 0 && si->pathTracingIteration > 0 &&
        si->pathTracingIteration <= NB_MAX_ITERATIONS)
        return;
    Ray ray;
    memset(&ray, 0, sizeof(ray));
    ray.origin = origin;
    ray.direction = direction;
    if (si->pathTracingIteration >= NB_MAX_ITERATIONS)
    {
        int rindex = (gindex + si->timestamp) % (MAX_BITMAP_SIZE - 3);
        float a = (float)si->pathTracingIteration / (float)si->maxPathTracingIterations;
        ray.direction.x += rnd(s, rindex, st) * pp[index].colorInfo.w * ppi->param2 * a;
        ray.direction.y += rnd(s, rindex + 1, st) * pp[index].colorInfo.w * ppi->param2 * a;
        ray.direction.z += rnd(s, rindex + 2, st) * pp[index].colorInfo.w * ppi->param2 * a;
    }
    float dof = 0.f;
    float stepy = angles[3] / (float)si->size.y;
    ray.direction.y = ray.direction.y + stepy * (float)(yGlobal - (si->size.y / 2));
    float stepx = 2.f * 3.14159265358979323846f / (float)si->size.x;
    float fishEyeAngles[3] = {0.f, angles[1] + stepx * (float)x, 0.f};
    Trig t = makeTrig(fishEyeAngles); /* x and z: cos(0) = 1, sin(0) = 0 exactly */
    t.cy = (float)cos((double)fishEyeAngles[1]);
    t.sy = (float)sin((double)fishEyeAngles[1]);
    ray.direction = vectorRotation(ray.direction, ray.origin, &t);
    c3 color = launchRayTracing(s, gindex, &ray, si, &dof, &ids[index], st);
    if (si->pathTracingIteration == 0)
        pp[index].colorInfo.w = dof;
    if (si->pathTracingIteration <= NB_MAX_ITERATIONS)
    {
        pp[index].colorInfo.x = color.x;
        pp[index].colorInfo.y = color.y;
        pp[index].colorInfo.z = color.z;
    }
    else
    {
        pp[index].colorInfo.x += color.x;
        pp[index].colorInfo.y += color.y;
        pp[index].colorInfo.z += color.z;
    }
}

/* ---- post-processing stage --------------------------------------------- */

/* ref CRT:1057-1073 */
static void postDefault(const SceneInfo *si, const PostProcessingBuffer *pp, BitmapBuffer *bitmap, int index)
{
    c3 c = {pp[index].colorInfo.x, pp[index].colorInfo.y, pp[index].colorInfo.z};
    if (si->pathTracingIteration > NB_MAX_ITERATIONS)
    {
        float d = (float)(si->pathTracingIteration - NB_MAX_ITERATIONS + 1);
        c.x /= d;
        c.y /= d;
        c.z /= d;
    }
    makeColor(si, c, bitmap, index);
}

/* ref CRT:1081-1120.  Gathers stay inside the strip that owns the pixel: the
 * oracle is only asked for this effect on full frames (firstRow == 0). */
static void postDepthOfField(const OracleScene *s, const SceneInfo *si, const PostProcessingInfo *ppi,
                             const PostProcessingBuffer *pp, BitmapBuffer *bitmap, int x, int y, int rows, Stats *st)
{
    int index = y * si->size.x + x;
    c3 localColor = {0.f, 0.f, 0.f};
    float depth = fabsf(pp[index].colorInfo.w - ppi->param1) / si->viewDistance;
    int wh = si->size.x * rows;
    for (int i = 0; i < ppi->param3; ++i)
    {
        int ix = i % wh;
        int iy = (i + (DIALECT(32) ? 100 : 1000)) % wh; /* CRT:1101; CL:2968 */
        int xx = f2i(x + depth * rnd(s, ix, st) * ppi->param2);
        int yy = f2i(y + depth * rnd(s, iy, st) * ppi->param2);
        if (xx >= 0 && xx < si->size.x && yy >= 0 && yy < rows)
        {
            int localIndex = yy * si->size.x + xx;
            if (localIndex >= 0 && localIndex < wh)
            {
                localColor.x += pp[localIndex].colorInfo.x;
                localColor.y += pp[localIndex].colorInfo.y;
                localColor.z += pp[localIndex].colorInfo.z;
            }
        }
        else
        {
            localColor.x += pp[index].colorInfo.x;
            localColor.y += pp[index].colorInfo.y;
            localColor.z += pp[index].colorInfo.z;
        }
    }
    localColor.x /= (float)ppi->param3;
    localColor.y /= (float)ppi->param3;
    localColor.z /= (float)ppi->param3;
    if (si->pathTracingIteration > NB_MAX_ITERATIONS)
    {
        float d = (float)(si->pathTracingIteration - NB_MAX_ITERATIONS + 1);
        localColor.x /= d;
        localColor.y /= d;
        localColor.z /= d;
    }
    makeColor(si, localColor, bitmap, index);
}

/* ref CRT:1128-1181 */
static void postAmbientOcclusion(const OracleScene *s, const SceneInfo *si, const PostProcessingInfo *ppi,
                                 const PostProcessingBuffer *pp, BitmapBuffer *bitmap, int x, int y, int rows,
                                 int firstRow, Stats *st)
{
    /* wh is the frame's (CRT:1140).  A strip (oracle_render's firstRow / nbRows, the multi-GPU split) is rows
     * of the frame: a tap's row is evaluated with the frame's y, so that strips whose neighbours' rows are at
     * hand assemble to the one-GPU frame (the engine's depth halo); here rows outside the strip read as outside
     * the frame.  (The reference's split uses the device-local y, CRT:1133: its frames have seams.)  With
     * firstRow == 0 - every whole frame - this is the reference's statement as written. */
    int index = y * si->size.x + x;
    int wh = si->size.x * si->size.y;
    float occ = 0.f;
    c3 localColor = {pp[index].colorInfo.x, pp[index].colorInfo.y, pp[index].colorInfo.z};
    float depth = pp[index].colorInfo.w;
    const int step = 16;
    int i = 0;
    float c = 0.f;
    for (int X = -step; X < step; X += 2)
        for (int Y = -step; Y < step; Y += 2)
        {
            int ix = i % wh;
            int iy = (i + 100) % wh;
            ++i;
            c += 1.f;
            int xx = f2i(x + (X * ppi->param2 * rnd(s, ix, st) / 10.f));
            int yy = f2i((y + firstRow) + (Y * ppi->param2 * rnd(s, iy, st) / 10.f)) - firstRow;
            if (xx >= 0 && xx < si->size.x && yy >= 0 && yy < rows)
            {
                int localIndex = yy * si->size.x + xx;
                if (DIALECT(33))
                {
                    /* CL:3026-3027: nearer neighbours darken, by more the nearer they are */
                    if (pp[localIndex].colorInfo.w < depth)
                        occ += 1.f - (pp[localIndex].colorInfo.w - depth) / si->viewDistance;
                }
                else if (pp[localIndex].colorInfo.w >= depth)
                    occ += 1.f;
            }
            else
                occ += 1.f;
        }
    if (DIALECT(34))
    {
        /* CL:3033-3043: the occlusion is subtracted, after the division by the number of samples */
        occ /= 5.f * c;
        occ = (occ > 1.f) ? 1.f : occ;
        occ = (occ < 0.f) ? 0.f : occ;
    }
    else
    {
        occ /= (float)c;
        occ += 0.3f;
        if (occ < 1.f)
        {
            localColor.x *= occ;
            localColor.y *= occ;
            localColor.z *= occ;
        }
    }
    if (si->pathTracingIteration > NB_MAX_ITERATIONS)
    {
        float d = (float)(si->pathTracingIteration - NB_MAX_ITERATIONS + 1);
        localColor.x /= d;
        localColor.y /= d;
        localColor.z /= d;
    }
    if (DIALECT(35))
    {
        localColor.x -= occ;
        localColor.y -= occ;
        localColor.z -= occ;
    }
    saturate3(&localColor);
    makeColor(si, localColor, bitmap, index);
}

/* ref CRT:1189-1228.  Random gathers over the frame, weighted by the emissive word of the pixel
 * they land on.  randoms is indexed with the reference's expressions (rnd() returns 0 out of range). */
static void postRadiosity(const OracleScene *s, const SceneInfo *si, const PostProcessingInfo *ppi,
                          const PostProcessingBuffer *pp, const PrimitiveXYIdBuffer *ids, BitmapBuffer *bitmap, int x,
                          int y, int rows, Stats *st)
{
    int index = y * si->size.x + x;
    int wh = si->size.x * rows;
    int div = (si->pathTracingIteration > NB_MAX_ITERATIONS) ? (si->pathTracingIteration - NB_MAX_ITERATIONS + 1) : 1;
    c3 localColor = {0.f, 0.f, 0.f};
    for (int i = 0; i < ppi->param3; ++i)
    {
        int ix = (i + si->pathTracingIteration) % wh;
        int iy = (i + 100 + si->pathTracingIteration) % wh;
        int xx = f2i((float)x + rnd(s, ix, st) * ppi->param2);
        int yy = f2i((float)y + rnd(s, iy, st) * ppi->param2);
        localColor.x += pp[index].colorInfo.x;
        localColor.y += pp[index].colorInfo.y;
        localColor.z += pp[index].colorInfo.z;
        if (xx >= 0 && xx < si->size.x && yy >= 0 && yy < rows)
        {
            int localIndex = yy * si->size.x + xx;
            float w = (float)ids[localIndex].z;
            localColor.x += pp[localIndex].colorInfo.x * w / 256.f;
            localColor.y += pp[localIndex].colorInfo.y * w / 256.f;
            localColor.z += pp[localIndex].colorInfo.z * w / 256.f;
        }
    }
    localColor.x /= (float)ppi->param3;
    localColor.y /= (float)ppi->param3;
    localColor.z /= (float)ppi->param3;
    localColor.x /= (float)div;
    localColor.y /= (float)div;
    localColor.z /= (float)div;
    saturate3(&localColor);
    makeColor(si, localColor, bitmap, index);
}

/* ref CRT:1236-1333: six convolution filters selected by param3, wrapping around the frame */
static void postFilter(const SceneInfo *si, const PostProcessingInfo *ppi, const PostProcessingBuffer *pp,
                       BitmapBuffer *bitmap, int x, int y, int rows)
{
    enum { NB_FILTERS = 6 };
    static const int filterSize[NB_FILTERS][2] = {{3, 3}, {5, 5}, {3, 3}, {3, 3}, {5, 5}, {5, 5}};
    static const float filterFactors[NB_FILTERS][2] = {{1.f, 128.f}, {1.f, 0.f}, {1.f, 0.f},
                                                       {1.f, 0.f},   {0.2f, 0.f}, {0.125f, 0.f}};
    static const float filterInfo[NB_FILTERS][5][5] = {
        {{-1.f, -1.f, 0.f, 0.f, 0.f}, {-1.f, 0.f, 1.f, 0.f, 0.f}, {0.f, 1.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
        {{0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {-1.f, -1.f, 2.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
        {{-1.f, -1.f, -1.f, 0.f, 0.f}, {-1.f, 9.f, -1.f, 0.f, 0.f}, {-1.f, -1.f, -1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
        {{0.f, 0.2f, 0.f, 0.f, 0.f}, {0.2f, 0.2f, 0.2f, 0.f, 0.f}, {0.f, 0.2f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}},
        {{1.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 1.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 1.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 1.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 1.f}},
        {{-1.f, -1.f, -1.f, -1.f, -1.f}, {-1.f, 2.f, 2.f, 2.f, -1.f}, {-1.f, 2.f, 8.f, 2.f, -1.f}, {-1.f, 2.f, 2.f, 2.f, -1.f}, {-1.f, -1.f, -1.f, -1.f, -1.f}}};
    int index = y * si->size.x + x;
    c3 localColor = {0.f, 0.f, 0.f};
    c3 color = {0.f, 0.f, 0.f};
    const int f = ppi->param3;
    if (f >= 0 && f < NB_FILTERS) /* the reference compares as unsigned: negative values select nothing either */
    {
        for (int filterX = 0; filterX < filterSize[f][0]; filterX++)
            for (int filterY = 0; filterY < filterSize[f][1]; filterY++)
            {
                int imageX = (x - filterSize[f][0] / 2 + filterX + si->size.x) % si->size.x;
                int imageY = (y - filterSize[f][1] / 2 + filterY + rows) % rows;
                int localIndex = imageY * si->size.x + imageX;
                c3 c = {pp[localIndex].colorInfo.x, pp[localIndex].colorInfo.y, pp[localIndex].colorInfo.z};
                if (si->pathTracingIteration > NB_MAX_ITERATIONS)
                {
                    float d = (float)(si->pathTracingIteration - NB_MAX_ITERATIONS + 1);
                    c.x /= d;
                    c.y /= d;
                    c.z /= d;
                }
                localColor.x += c.x * filterInfo[f][filterX][filterY];
                localColor.y += c.y * filterInfo[f][filterX][filterY];
                localColor.z += c.z * filterInfo[f][filterX][filterY];
            }
        color.x += fminf(fmaxf(filterFactors[f][0] * localColor.x + filterFactors[f][1] / 255.f, 0.f), 1.f);
        color.y += fminf(fmaxf(filterFactors[f][0] * localColor.y + filterFactors[f][1] / 255.f, 0.f), 1.f);
        color.z += fminf(fmaxf(filterFactors[f][0] * localColor.z + filterFactors[f][1] / 255.f, 0.f), 1.f);
    }
    saturate3(&color);
    makeColor(si, color, bitmap, index);
}

/* ref CRT:1341-1358: depth shown as grey */
static void postCartoon(const SceneInfo *si, const PostProcessingInfo *ppi, const PostProcessingBuffer *pp,
                        BitmapBuffer *bitmap, int index)
{
    float depth = si->viewDistance / fabsf(pp[index].colorInfo.w - ppi->param1);
    c3 color = {depth, depth, depth};
    saturate3(&color);
    makeColor(si, color, bitmap, index);
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_render(const OracleScene *scene, const SceneInfo *sceneInfo, const PostProcessingInfo *ppInfo,
                  const float origin[3], const float direction[3], const float angles[4], int firstRow, int nbRows,
                  PostProcessingBuffer *pp, PrimitiveXYIdBuffer *ids, BitmapBuffer *bitmap, oracle_counts_t counts,
                  int nthreads)
{
    const int W = sceneInfo->size.x;
    Trig trig = makeTrig(angles);
    v3 o = V(origin[0], origin[1], origin[2]);
    v3 d = V(direction[0], direction[1], direction[2]);
    Stats total;
    memset(&total, 0, sizeof(total));
#ifdef _OPENMP
    if (nthreads <= 0)
        nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif

    /* k_3DVisionRenderer's focus pixel, CRT:973 (integer expression as written there), as it is before the
     * frame; a strip that does not hold it reads 0 */
    float focusDepth = 0.f;
    if (sceneInfo->cameraType == ctVR)
    {
        const int focusIndex = sceneInfo->size.x / 2 * sceneInfo->size.y / 2;
        const int focusRow = focusIndex / W - firstRow;
        if (focusRow >= 0 && focusRow < nbRows)
            focusDepth = pp[focusRow * W + focusIndex % W].colorInfo.w;
    }

#pragma omp parallel num_threads(nthreads)
    {
        Stats st;
        memset(&st, 0, sizeof(st));
#pragma omp for schedule(dynamic, 2)
        for (int y = 0; y < nbRows; ++y)
            for (int x = 0; x < W; ++x)
            {
                t_pixel = (long)y * W + x;
                /* the camera types the engine renders with a kernel of their own (CRT:1714-1836) */
                if (sceneInfo->cameraType == ctVR)
                    visionRendererPixel(scene, sceneInfo, ppInfo, o, d, angles, &trig, focusDepth, x, y, firstRow, pp,
                                        ids, &st);
                else if (sceneInfo->cameraType == ctAnaglyph)
                    anaglyphRendererPixel(scene, sceneInfo, o, d, angles, &trig, x, y, firstRow, pp, ids, &st);
                else if (sceneInfo->cameraType == ctPanoramic)
                    fishEyeRendererPixel(scene, sceneInfo, ppInfo, o, d, angles, x, y, firstRow, pp, ids, &st);
                else if (sceneInfo->cameraType == ctVolumeRendering)
                    volumeRendererPixel(scene, sceneInfo, ppInfo, o, d, angles, &trig, x, y, firstRow, pp, ids, &st);
                else
                    standardRendererPixel(scene, sceneInfo, ppInfo, o, d, angles, &trig, x, y, firstRow, pp, ids, &st);
            }
#pragma omp for schedule(dynamic, 2)
        for (int y = 0; y < nbRows; ++y)
            for (int x = 0; x < W; ++x)
            {
                int index = y * W + x;
                switch (ppInfo->type)
                {
                case ppe_depthOfField:
                    postDepthOfField(scene, sceneInfo, ppInfo, pp, bitmap, x, y, nbRows, &st);
                    break;
                case ppe_ambientOcclusion:
                    postAmbientOcclusion(scene, sceneInfo, ppInfo, pp, bitmap, x, y, nbRows, firstRow, &st);
                    break;
                case ppe_radiosity:
                    postRadiosity(scene, sceneInfo, ppInfo, pp, ids, bitmap, x, y, nbRows, &st);
                    break;
                case ppe_filter:
                    postFilter(sceneInfo, ppInfo, pp, bitmap, x, y, nbRows);
                    break;
                case ppe_cartoon:
                    postCartoon(sceneInfo, ppInfo, pp, bitmap, index);
                    break;
                default:
                    postDefault(sceneInfo, pp, bitmap, index);
                    break;
                }
            }
#pragma omp critical
        {
            total.closest += st.closest;
            total.shadow += st.shadow;
            total.boxes += st.boxes;
            total.prims += st.prims;
            total.randomFault |= st.randomFault;
        }
    }
    if (counts)
    {
        counts[0] = total.closest;
        counts[1] = total.shadow;
        counts[2] = total.boxes;
        counts[3] = total.prims;
    }
    return total.randomFault ? -1 : 0;
}

/* ---- batched function-level entry points ----------------------------------------------------------
 * One call = n independent evaluations of one function of the path, on the arrays the reference probes
 * (oracle/ref_probes.cl) are given: tests/test_reference_probes.py compares the two element for element.
 * Vectors are packed xyz (3 floats) unless said otherwise. */
static inline v3 at3(const float *a, int i) { return V(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
static inline void put3(float *a, int i, v3 v)
{
    a[3 * i] = v.x;
    a[3 * i + 1] = v.y;
    a[3 * i + 2] = v.z;
}

void oracle_probe_box(int n, const BoundingBox *boxes, const float *origins, const float *directions, const float *t0,
                      const float *t1, int *hit)
{
    for (int i = 0; i < n; ++i)
    {
        Ray r;
        r.origin = at3(origins, i);
        r.direction = at3(directions, i);
        computeRayAttributes(&r);
        hit[i] = boxIntersection(&boxes[i], &r, t0[i], t1[i]);
    }
}

void oracle_probe_primitive(int n, const SceneInfo *si, const Primitive *prims, const Material *materials,
                            const BitmapBuffer *textures, const float *origins, const float *directions,
                            const int *shadows, float *intersection, float *normal, float *areas,
                            float *shadowIntensity, int *hit)
{
    for (int i = 0; i < n; ++i)
    {
        Ray r;
        r.origin = at3(origins, i);
        r.direction = at3(directions, i);
        computeRayAttributes(&r);
        v3 in = at3(intersection, i), no = at3(normal, i), ar = V(0.f, 0.f, 0.f);
        float sh = 0.f;
        hit[i] = shadows[i] ? testPrimitiveShadow(si, &prims[i], materials, textures, &r, &in, &no, &ar, &sh)
                            : testPrimitive(si, &prims[i], materials, textures, &r, &in, &no, &ar, &sh);
        put3(intersection, i, in);
        put3(normal, i, no);
        put3(areas, i, ar);
        shadowIntensity[i] = sh;
    }
}

void oracle_probe_closest(int n, const OracleScene *scene, const SceneInfo *si, const float *origins,
                          const float *targets, const int *iteration, const int *currentMaterialId, int *hit,
                          int *closestPrimitive, float *intersection, float *normal, float *areas)
{
    for (int i = 0; i < n; ++i)
    {
        Ray ray;
        Stats st;
        memset(&st, 0, sizeof(st));
        ray.origin = at3(origins, i);
        ray.direction = at3(targets, i);
        v3 ci = V(0.f, 0.f, 0.f), cn = V(0.f, 0.f, 0.f), ca = V(0.f, 0.f, 0.f);
        c3 colorBox = {0.f, 0.f, 0.f};
        int cp = -1;
        hit[i] = intersectionWithPrimitives(scene, si, &ray, iteration[i], &cp, &ci, &cn, &ca, &colorBox,
                                            currentMaterialId[i], &st);
        closestPrimitive[i] = cp;
        put3(intersection, i, ci);
        put3(normal, i, cn);
        put3(areas, i, ca);
    }
}

void oracle_probe_shadow(int n, const OracleScene *scene, const SceneInfo *si, const float *lampCenters,
                         const float *origins, const int *lightId, const int *objectId, const int *iteration,
                         float *result, float *color)
{
    for (int i = 0; i < n; ++i)
    {
        Stats st;
        memset(&st, 0, sizeof(st));
        c3 c;
        result[i] = processShadows(scene, si, at3(lampCenters, i), at3(origins, i), lightId[i], iteration[i], &c,
                                   objectId[i], &st);
        color[3 * i] = c.x;
        color[3 * i + 1] = c.y;
        color[3 * i + 2] = c.z;
    }
}

/* primitiveShader: normal, closestColor, totalBlinn (xyz) and attributes (xyzw) are in/out */
int oracle_probe_shader(int n, const OracleScene *scene, const SceneInfo *si, const int *index, const float *origins,
                        float *normal, const int *objectId, const float *intersection, const float *areas,
                        float *closestColor, const int *iteration, float *totalBlinn, float *attributes,
                        float *returned, float *shadowIntensity)
{
    int fault = 0;
    for (int i = 0; i < n; ++i)
    {
        Stats st;
        memset(&st, 0, sizeof(st));
        v3 no = at3(normal, i);
        c3 cc = {closestColor[3 * i], closestColor[3 * i + 1], closestColor[3 * i + 2]};
        c3 tb = {totalBlinn[3 * i], totalBlinn[3 * i + 1], totalBlinn[3 * i + 2]};
        f4 at = {attributes[4 * i], attributes[4 * i + 1], attributes[4 * i + 2], attributes[4 * i + 3]};
        float sh = 0.f;
        c3 ret = primitiveShader(scene, index[i], si, at3(origins, i), &no, objectId[i], at3(intersection, i),
                                 at3(areas, i), &cc, iteration[i], &sh, &tb, &at, &st);
        put3(normal, i, no);
        closestColor[3 * i] = cc.x, closestColor[3 * i + 1] = cc.y, closestColor[3 * i + 2] = cc.z;
        totalBlinn[3 * i] = tb.x, totalBlinn[3 * i + 1] = tb.y, totalBlinn[3 * i + 2] = tb.z;
        attributes[4 * i] = at.x, attributes[4 * i + 1] = at.y, attributes[4 * i + 2] = at.z, attributes[4 * i + 3] = at.w;
        returned[3 * i] = ret.x, returned[3 * i + 1] = ret.y, returned[3 * i + 2] = ret.z;
        shadowIntensity[i] = sh;
        fault |= st.randomFault;
    }
    return fault ? -1 : 0;
}

/* intersectionShader on primitive i with its material's starting specular (as primitiveShader sets it up):
 * colour (xyzw), the bump-normal accumulator, specular, attributes (in/out), advanced attributes */
void oracle_probe_intersection_shader(int n, const SceneInfo *si, const Primitive *prims, const Material *materials,
                                      const BitmapBuffer *textures, const float *intersection, const float *areas,
                                      float *attributes, float *color, float *bumpNormal, float *specular,
                                      float *advancedAttributes)
{
    for (int i = 0; i < n; ++i)
    {
        const Material *m = &materials[prims[i].materialId];
        v3 bump = V(0.f, 0.f, 0.f);
        f4 adv = {0.f, 0.f, 0.f, 0.f};
        f4 spec = {m->specular.x, m->specular.y, m->specular.z, 0.f};
        f4 at = {attributes[4 * i], attributes[4 * i + 1], attributes[4 * i + 2], attributes[4 * i + 3]};
        f4 c = intersectionShader(si, &prims[i], materials, textures, at3(intersection, i), at3(areas, i), &bump, &spec,
                                  &at, &adv);
        color[4 * i] = c.x, color[4 * i + 1] = c.y, color[4 * i + 2] = c.z, color[4 * i + 3] = c.w;
        put3(bumpNormal, i, bump);
        specular[4 * i] = spec.x, specular[4 * i + 1] = spec.y, specular[4 * i + 2] = spec.z, specular[4 * i + 3] = spec.w;
        attributes[4 * i] = at.x, attributes[4 * i + 1] = at.y, attributes[4 * i + 2] = at.z, attributes[4 * i + 3] = at.w;
        advancedAttributes[4 * i] = adv.x, advancedAttributes[4 * i + 1] = adv.y, advancedAttributes[4 * i + 2] = adv.z,
                               advancedAttributes[4 * i + 3] = adv.w;
    }
}

void oracle_probe_skybox(int n, const SceneInfo *si, const Material *materials, const BitmapBuffer *textures,
                         const float *origins, const float *targets, float *color)
{
    for (int i = 0; i < n; ++i)
    {
        Ray r;
        r.origin = at3(origins, i);
        r.direction = at3(targets, i);
        c3 c = skyboxMapping(si, materials, textures, &r);
        color[3 * i] = c.x, color[3 * i + 1] = c.y, color[3 * i + 2] = c.z;
    }
}

void oracle_probe_vectors(int n, const float *incident, const float *normal, const float *n1, const float *n2,
                          float *refracted, float *reflected)
{
    for (int i = 0; i < n; ++i)
    {
        put3(refracted, i, vectorRefraction(at3(incident, i), n1[i], at3(normal, i), n2[i]));
        put3(reflected, i, vectorReflection(at3(incident, i), at3(normal, i)));
    }
}

void oracle_probe_make_color(int n, const SceneInfo *si, const float *colors, BitmapBuffer *bitmap)
{
    for (int i = 0; i < n; ++i)
    {
        c3 c = {colors[3 * i], colors[3 * i + 1], colors[3 * i + 2]};
        makeColor(si, c, bitmap, i);
    }
}

/* launchRayTracing for the ray origins[i] -> targets[i] of pixel index[i]; ids is in/out (the OpenCL dialect
 * leaves .w as it finds it) */
int oracle_probe_launch(int n, const OracleScene *scene, const SceneInfo *si, const float *origins,
                        const float *targets, const int *index, float *color, float *depth,
                        PrimitiveXYIdBuffer *ids)
{
    int fault = 0;
#pragma omp parallel for schedule(dynamic, 64) reduction(| : fault)
    for (int i = 0; i < n; ++i)
    {
        Stats st;
        memset(&st, 0, sizeof(st));
        Ray r;
        memset(&r, 0, sizeof(r));
        r.origin = at3(origins, i);
        r.direction = at3(targets, i);
        float dof = 0.f;
        c3 c = launchRayTracing(scene, index[i], &r, si, &dof, &ids[i], &st);
        color[3 * i] = c.x, color[3 * i + 1] = c.y, color[3 * i + 2] = c.z;
        depth[i] = dof;
        fault |= st.randomFault;
    }
    return fault ? -1 : 0;
}

/* the post-processing stage alone (CRT:1853-1906) on a full frame of pp / ids */
int oracle_postprocess(const OracleScene *scene, const SceneInfo *si, const PostProcessingInfo *ppi,
                       const PostProcessingBuffer *pp, const PrimitiveXYIdBuffer *ids, BitmapBuffer *bitmap)
{
    Stats st;
    memset(&st, 0, sizeof(st));
    const int W = si->size.x, H = si->size.y;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
        {
            int index = y * W + x;
            switch (ppi->type)
            {
            case ppe_depthOfField:
                postDepthOfField(scene, si, ppi, pp, bitmap, x, y, H, &st);
                break;
            case ppe_ambientOcclusion:
                postAmbientOcclusion(scene, si, ppi, pp, bitmap, x, y, H, 0, &st);
                break;
            case ppe_radiosity:
                postRadiosity(scene, si, ppi, pp, ids, bitmap, x, y, H, &st);
                break;
            case ppe_filter:
                postFilter(si, ppi, pp, bitmap, x, y, H);
                break;
            case ppe_cartoon:
                postCartoon(si, ppi, pp, bitmap, index);
                break;
            default:
                postDefault(si, pp, bitmap, index);
                break;
            }
        }
    return st.randomFault ? -1 : 0;
}

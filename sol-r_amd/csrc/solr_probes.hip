/*
 * solr_probes.hip - TEST-ONLY entry points (include/solr_hip_probes.h): the engine's own device functions evaluated
 * once per element of arrays of inputs, for the comparison with the reference's own functions on the same arrays
 * (tests/test_engine_probes_gpu.py; the reference side is oracle/ref_probes.cl around RayTracer.cl).  No rendering
 * code lives here: every kernel calls the functions of rt_device.h that k_standardRenderer is built from, in the
 * instantiations the renderer launches.  gfx950 only.
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/solr_hip.h"
#include "../../include/solr_hip_probes.h"
#include "rt_device.h"

using namespace solrdev;

/* solr_diag.hip: the resident scene as renderImpl hands it to the renderer */
namespace solrprobe
{
int residentScene(const SceneInfo &sceneInfo, bool exactNodes, SceneArgs *S, int *features, int *deepList, hipStream_t *stream);
void fail(int code, const char *what);
int ticketOfSerial(long long serial, int *slot, long long *period);
long long imageSerial(long long setTo);
int postProcess(const SceneInfo &sceneInfo, const PostProcessingInfo &ppInfo, const PostProcessingBuffer *frame,
                unsigned char *bitmapOut);
} // namespace solrprobe

namespace
{
#define PROBE_HIP(expr)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess && !failed)                                                                               \
        {                                                                                                              \
            failed = true;                                                                                             \
            solrprobe::fail((int)e_, (std::string("solr_hip_probe: " #expr ": ") + hipGetErrorString(e_)).c_str());    \
        }                                                                                                              \
    } while (0)

/* device copies of the callers' arrays for the duration of one probe */
struct Arrays
{
    std::vector<void *> owned;
    struct Out
    {
        void *host, *device;
        size_t bytes;
    };
    std::vector<Out> outs;
    bool failed = false;
    template <class T>
    const T *in(const T *host, size_t count)
    {
        void *d = nullptr;
        const size_t bytes = (count ? count : 1) * sizeof(T);
        PROBE_HIP(hipMalloc(&d, bytes));
        if (d && count)
            PROBE_HIP(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
        owned.push_back(d);
        return (const T *)d;
    }
    /* in/out: the device copy starts as the host's contents and is copied back by finish() */
    template <class T>
    T *out(T *host, size_t count)
    {
        T *d = (T *)in(host, count);
        outs.push_back({host, d, count * sizeof(T)});
        return d;
    }
    bool finish(hipStream_t stream)
    {
        PROBE_HIP(hipGetLastError());
        PROBE_HIP(hipStreamSynchronize(stream));
        for (const Out &o : outs)
            if (o.device && o.bytes)
                PROBE_HIP(hipMemcpy(o.host, o.device, o.bytes, hipMemcpyDeviceToHost));
        return !failed;
    }
    ~Arrays()
    {
        for (void *p : owned)
            if (p)
                (void)hipFree(p);
    }
};

SOLR_DEV v3 at3(const float *a, int i) { return V(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }
SOLR_DEV void put3(float *a, int i, v3 v)
{
    a[3 * i] = v.x;
    a[3 * i + 1] = v.y;
    a[3 * i + 2] = v.z;
}

/* ---- the slab test, per element ---------------------------------------------------------------------------- */
__global__ __launch_bounds__(64) void k_probeBox(int n, const BoundingBox *boxes, const float *origins, const float *directions,
                                                const float *t0, const float *t1, int *hitExact, int *hitFast)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n)
        return;
    const WalkRay r = makeWalkRay(at3(origins, e), at3(directions, e));
    const BoundingBox b = boxes[e];
    const float4 lo = make_float4(b.parameters[0].x, b.parameters[0].y, b.parameters[0].z, 0.f);
    const float4 hi = make_float4(b.parameters[1].x, b.parameters[1].y, b.parameters[1].z, 0.f);
    hitExact[e] = boxIntersectionExact(lo, hi, r, t0[e], t1[e]) ? 1 : 0;
    hitFast[e] = finiteRay(r) ? (boxIntersectionFast(lo, hi, r, t0[e], t1[e]) ? 1 : 0) : -1;
}

/* ---- the hand-scheduled node loop over a flat list of leaves ------------------------------------------------ */
template <int FEAT>
__global__ __launch_bounds__(64) void k_probeBoxWalk(const SceneArgs SA, int n, const float *origins, const float *directions,
                                                    const float *t1, int *hit)
{
    const Scene S = makeScene(SA);
    const int e = blockIdx.x * 64 + threadIdx.x;
    const bool active = e < n;
    const int q = active ? e : 0;
    const WalkRay r = makeWalkRay(at3(origins, q), at3(directions, q));
    const PackedRay pr = packRay(r);
    const float farDistance = t1[q];
    int cursor = active ? 0 : SOLR_CURSOR_DONE;
    int cur = 0;
    int entered_mine = 0;
    while (cur < S.nbBoxes)
    {
        int nbPrimitives;
        bool entered;
        const int leaf = advanceTidy<FEAT>(S, pr, farDistance, cursor, cur, nbPrimitives, entered);
        if (leaf < 0)
            break;
        if (entered && leaf == e)
            entered_mine = 1;
    }
    if (active)
        hit[e] = entered_mine;
}

/* ---- one primitive test, as either walk dispatches it ------------------------------------------------------- */
template <int FEAT>
__global__ __launch_bounds__(64) void k_probePrimitive(const SceneArgs SA, const SceneInfo si, int n, const float *origins,
                                                      const float *directions, const int *shadows, float *intersection,
                                                      float *normal, float *areas, float *shadowIntensity, int *hit)
{
    const Scene S = makeScene(SA);
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n)
        return;
    const WalkRay r = makeWalkRay(at3(origins, e), at3(directions, e));
    PrimRec rec;
    rec.pi = e;
    rec.head = primHead(S, e);
    rec.c = rec.d = make_float4(0.f, 0.f, 0.f, 0.f);
    rec.packed = false;
    const int tag = asint(rec.head.a.w);
    Hit h;
    h.intersection = at3(intersection, e);
    h.normal = at3(normal, e);
    h.areas = V(0.f, 0.f, 0.f);
    h.shadowIntensity = 0.f;
    bool i;
    if (shadows[e])
        i = testPrimitive<true, FEAT>(S, si, rec, tag, r, h);
    else
        i = testPrimitive<false, FEAT>(S, si, rec, tag, r, h);
    put3(intersection, e, h.intersection);
    put3(normal, e, h.normal);
    put3(areas, e, h.areas);
    shadowIntensity[e] = h.shadowIntensity;
    hit[e] = i ? 1 : 0;
}

/* ---- the two walks over the resident scene ------------------------------------------------------------------- */
template <int FEAT>
__global__ __launch_bounds__(64) void k_probeClosest(const SceneArgs SA, const SceneInfo si, int n, const float *origins,
                                                    const float *targets, const int *iteration, const int *currentMaterialId,
                                                    int *hit, int *primitive, float *intersection, float *normal, float *areas)
{
    const Scene S = makeScene(SA);
    const int e = blockIdx.x * 64 + threadIdx.x;
    const bool active = e < n;
    const int q = active ? e : 0;
    Counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    int closestPrimitive = -1;
    v3 ci = V(0.f, 0.f, 0.f), cn = V(0.f, 0.f, 0.f), ca = V(0.f, 0.f, 0.f), colorBox = V(0.f, 0.f, 0.f);
    /* (the bounce number decides the initial cut-off, GI:674: the walk is called once per bounce number present) */
    bool found = false;
    for (int it = 0; it < 16; ++it)
    {
        const bool lanes = active && iteration[q] == it;
        if (ballot(lanes) == 0ull)
            continue;
        int p = -1;
        v3 i3 = V(0.f, 0.f, 0.f), n3 = V(0.f, 0.f, 0.f), a3 = V(0.f, 0.f, 0.f);
        const bool f = closestHitWalk<false, FEAT>(S, si, lanes, at3(origins, q), at3(targets, q), it, currentMaterialId[q], p,
                                                   i3, n3, a3, colorBox, cnt);
        if (lanes)
        {
            found = f;
            closestPrimitive = p;
            ci = i3;
            cn = n3;
            ca = a3;
        }
    }
    if (active)
    {
        hit[e] = found ? 1 : 0;
        primitive[e] = closestPrimitive;
        put3(intersection, e, ci);
        put3(normal, e, cn);
        put3(areas, e, ca);
    }
}

template <int FEAT>
__global__ __launch_bounds__(64) void k_probeShadow(const SceneArgs SA, const SceneInfo si, int n, const float *lampCenters,
                                                   const float *origins, const int *lightId, const int *objectId,
                                                   const int *iteration, float *result, float *color)
{
    const Scene S = makeScene(SA);
    const int e = blockIdx.x * 64 + threadIdx.x;
    const bool active = e < n;
    const int q = active ? e : 0;
    Counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    float value = 0.f;
    v3 tint = V(0.f, 0.f, 0.f);
    /* the lamp and the bounce number are the same for every lane of a shadow walk in the renderer (the light loop
     * and the bounce loop are wave-uniform): one call per (lamp, bounce number) present in the wave */
    unsigned long long todo = ballot(active);
    while (todo)
    {
        const int first = (int)__builtin_ctzll(todo);
        const int lamp = __builtin_amdgcn_readlane(lightId[q], first);
        const int it = __builtin_amdgcn_readlane(iteration[q], first);
        const bool lanes = active && lightId[q] == lamp && iteration[q] == it;
        v3 c = V(0.f, 0.f, 0.f);
        const float s = shadowWalk<false, FEAT>(S, si, lanes, at3(lampCenters, q), at3(origins, q), lamp, it, c, objectId[q], cnt);
        if (lanes)
        {
            value = s;
            tint = c;
        }
        todo &= ~ballot(lanes);
    }
    if (active)
    {
        result[e] = value;
        put3(color, e, tint);
    }
}

/* ---- primitiveShader, once per element --------------------------------------------------------------------------
 * The renderer calls it with the whole wave (the lamp loop is wave-uniform, the shadow walk inside it wave-synchronous)
 * and a bounce number per lane (the deferred reflection ray of phase 1 is shaded as bounce `reflectedRays`): so does
 * this kernel, 64 unrelated elements per wave. */
template <int FEAT>
__global__ __launch_bounds__(64) void k_probeShader(const SceneArgs SA, const SceneInfo si, int n, const int *index,
                                                   const float *origins, const int *objectId, const float *intersections,
                                                   const float *areas, const int *iteration, float *normal,
                                                   float *closestColor, float *totalBlinn, float *attributes, float *returned,
                                                   float *shadowIntensity)
{
    const Scene S = makeScene(SA);
    const int e = blockIdx.x * 64 + threadIdx.x;
    const bool active = e < n;
    const int q = active ? e : 0;
    Counters cnt;
    memset(&cnt, 0, sizeof(cnt));
    v3 nrm = at3(normal, q), cc = at3(closestColor, q), tb = at3(totalBlinn, q);
    float4 attr = make_float4(attributes[4 * q], attributes[4 * q + 1], attributes[4 * q + 2], attributes[4 * q + 3]);
    float shadow = 0.f;
    const v3 r = primitiveShader<false, FEAT>(S, active, index[q], si, at3(origins, q), nrm, objectId[q], at3(intersections, q),
                                              at3(areas, q), cc, iteration[q], shadow, tb, attr, cnt);
    if (active)
    {
        put3(normal, e, nrm);
        put3(closestColor, e, cc);
        put3(totalBlinn, e, tb);
        attributes[4 * e] = attr.x;
        attributes[4 * e + 1] = attr.y;
        attributes[4 * e + 2] = attr.z;
        attributes[4 * e + 3] = attr.w;
        put3(returned, e, r);
        shadowIntensity[e] = shadow;
    }
}

/* ---- plain per-element functions ------------------------------------------------------------------------------ */
__global__ __launch_bounds__(64) void k_probeVectors(int n, const float *incident, const float *normals, const float *n1,
                                                    const float *n2, float *refracted, float *reflected)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n)
        return;
    put3(refracted, e, vectorRefraction(at3(incident, e), n1[e], at3(normals, e), n2[e]));
    put3(reflected, e, vectorReflection(at3(incident, e), at3(normals, e)));
}

__global__ __launch_bounds__(64) void k_probeMakeColor(const SceneInfo si, int n, const float *colors, unsigned char *bitmap)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n)
        return;
    makeColor(si, at3(colors, e), bitmap, e);
}

template <int FEAT>
__global__ __launch_bounds__(64) void k_probeSkybox(const SceneArgs SA, const SceneInfo si, int n, const float *origins,
                                                   const float *targets, float *color)
{
    const Scene S = makeScene(SA);
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n)
        return;
    put3(color, e, skyboxMapping<FEAT>(S, si, at3(origins, e), at3(targets, e)));
}

template <int FEAT>
__global__ __launch_bounds__(64) void k_probeIntersectionShader(const SceneArgs SA, const SceneInfo si, int n,
                                                               const float *intersections, const float *areas, float *attributes,
                                                               float *color, float *bump, float *specularOut, float *ambient)
{
    const Scene S = makeScene(SA);
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= n)
        return;
    /* as primitiveShader sets it up (rt_device.h; GI:933-945) */
    const int pi = e;
    const int type = asint(primRow(S, pi, ROW_P0_TYPE).w) & PRIM_TYPE_MASK;
    const int materialId = asint(primRow(S, pi, ROW_SIZE_MAT).w);
    const MaterialHot mh = loadMaterialHot(S, materialId);
    float4 specular = make_float4(mh.specular.x, mh.specular.y, mh.specular.z, 0.f);
    float4 attr = make_float4(attributes[4 * e], attributes[4 * e + 1], attributes[4 * e + 2], attributes[4 * e + 3]);
    float ambientOcclusion = 0.f;
    v3 bumpNormal = V(0.f, 0.f, 0.f);
    TexOut o = {&bumpNormal, &specular, &attr, &ambientOcclusion};
    const float4 c = intersectionShader<FEAT>(S, si, pi, type, materialId, mh, at3(intersections, e), at3(areas, e), o);
    color[4 * e] = c.x;
    color[4 * e + 1] = c.y;
    color[4 * e + 2] = c.z;
    color[4 * e + 3] = c.w;
    put3(bump, e, bumpNormal);
    put3(specularOut, e, V(specular.x, specular.y, specular.z));
    ambient[e] = ambientOcclusion;
    attributes[4 * e] = attr.x;
    attributes[4 * e + 1] = attr.y;
    attributes[4 * e + 2] = attr.z;
    attributes[4 * e + 3] = attr.w;
}

/* ---- which instantiation ----------------------------------------------------------------------------------------
 * The lean instantiations the renderer launches for untextured scenes of the usual primitives (solr_launch.hip,
 * renderImpl's table, first four rows, each with the two-bank and the three-bank node loop) and the all-features one
 * that covers everything else here (the renderer has three more textured / special-camera rows between them). */
constexpr int LEAN[4] = {F_SPHERE | F_PLANE, F_SPHERE | F_TRI, F_SPHERE | F_CYL, F_SPHERE | F_PLANE | F_TRI | F_CYL};
constexpr int EVERYTHING = (F_ALL & ~F_FULL) | F_DEEP;

int chooseFeatures(int asked, int need, int deepList)
{
    if (asked > 0)
        return asked;
    for (int v = 0; v < 4; ++v)
        if ((need & ~LEAN[v]) == 0)
            return LEAN[v] | ((deepList || v == 3) ? F_DEEP : 0);
    return EVERYTHING;
}

/* launches KERNEL<features> for the masks instantiated here; anything else is an argument error */
#define PROBE_DISPATCH(KERNEL, features, grid, stream, ...)                                                            \
    do                                                                                                                 \
    {                                                                                                                  \
        switch (features)                                                                                              \
        {                                                                                                              \
        case LEAN[0]: hipLaunchKernelGGL(KERNEL<LEAN[0]>, grid, dim3(64), 0, stream, __VA_ARGS__); break;              \
        case LEAN[0] | F_DEEP: hipLaunchKernelGGL(KERNEL<LEAN[0] | F_DEEP>, grid, dim3(64), 0, stream, __VA_ARGS__); break; \
        case LEAN[1]: hipLaunchKernelGGL(KERNEL<LEAN[1]>, grid, dim3(64), 0, stream, __VA_ARGS__); break;              \
        case LEAN[1] | F_DEEP: hipLaunchKernelGGL(KERNEL<LEAN[1] | F_DEEP>, grid, dim3(64), 0, stream, __VA_ARGS__); break; \
        case LEAN[2]: hipLaunchKernelGGL(KERNEL<LEAN[2]>, grid, dim3(64), 0, stream, __VA_ARGS__); break;              \
        case LEAN[2] | F_DEEP: hipLaunchKernelGGL(KERNEL<LEAN[2] | F_DEEP>, grid, dim3(64), 0, stream, __VA_ARGS__); break; \
        case LEAN[3] | F_DEEP: hipLaunchKernelGGL(KERNEL<LEAN[3] | F_DEEP>, grid, dim3(64), 0, stream, __VA_ARGS__); break; \
        case EVERYTHING: hipLaunchKernelGGL(KERNEL<EVERYTHING>, grid, dim3(64), 0, stream, __VA_ARGS__); break;        \
        default: solrprobe::fail(-1, "solr_hip_probe: no instantiation with these features"); return -1;               \
        }                                                                                                              \
    } while (0)

dim3 waves(int n) { return dim3((unsigned)((n + 63) / 64)); }
} // namespace

extern "C" {

int solr_hip_probe_box(int n, const BoundingBox *boxes, const float *origins, const float *directions, const float *t0,
                       const float *t1, int *hitExact, int *hitFast)
{
    if (n <= 0)
        return -1;
    Arrays a;
    const BoundingBox *dBoxes = a.in(boxes, n);
    const float *dO = a.in(origins, 3 * (size_t)n), *dD = a.in(directions, 3 * (size_t)n);
    const float *dT0 = a.in(t0, n), *dT1 = a.in(t1, n);
    int *dExact = a.out(hitExact, n), *dFast = a.out(hitFast, n);
    if (a.failed)
        return -1;
    hipLaunchKernelGGL(k_probeBox, waves(n), dim3(64), 0, 0, n, dBoxes, dO, dD, dT0, dT1, dExact, dFast);
    return a.finish(0) ? 1 : -1;
}

int solr_hip_probe_box_walk(const SceneInfo *sceneInfo, int n, const float *origins, const float *directions,
                            const float *t1, int features, int *hit)
{
    SceneArgs S;
    int need, deep;
    hipStream_t stream;
    if (n <= 0 || solrprobe::residentScene(*sceneInfo, true, &S, &need, &deep, &stream) != 0)
        return -1;
    if (S.nbBoxes != n || !S.nested || !S.orderedBoxes)
    {
        solrprobe::fail(-1, "solr_hip_probe_box_walk: the resident scene is not a flat list of n ordered leaf boxes");
        return -1;
    }
    features = chooseFeatures(features, need, deep);
    Arrays a;
    const float *dO = a.in(origins, 3 * (size_t)n), *dD = a.in(directions, 3 * (size_t)n), *dT1 = a.in(t1, n);
    int *dHit = a.out(hit, n);
    if (a.failed)
        return -1;
    PROBE_DISPATCH(k_probeBoxWalk, features, waves(n), stream, S, n, dO, dD, dT1, dHit);
    return a.finish(stream) ? features : -1;
}

int solr_hip_probe_primitive(const SceneInfo *sceneInfo, int n, const float *origins, const float *directions,
                             const int *shadows, int features, float *intersection, float *normal, float *areas,
                             float *shadowIntensity, int *hit)
{
    SceneArgs S;
    int need, deep;
    hipStream_t stream;
    if (n <= 0 || solrprobe::residentScene(*sceneInfo, false, &S, &need, &deep, &stream) != 0)
        return -1;
    if (S.nbPrimitives < n)
    {
        solrprobe::fail(-1, "solr_hip_probe_primitive: fewer primitives resident than elements");
        return -1;
    }
    features = chooseFeatures(features, need, deep);
    Arrays a;
    const float *dO = a.in(origins, 3 * (size_t)n), *dD = a.in(directions, 3 * (size_t)n);
    const int *dShadows = a.in(shadows, n);
    float *dI = a.out(intersection, 3 * (size_t)n), *dN = a.out(normal, 3 * (size_t)n), *dA = a.out(areas, 3 * (size_t)n);
    float *dS = a.out(shadowIntensity, n);
    int *dHit = a.out(hit, n);
    if (a.failed)
        return -1;
    PROBE_DISPATCH(k_probePrimitive, features, waves(n), stream, S, *sceneInfo, n, dO, dD, dShadows, dI, dN, dA, dS, dHit);
    return a.finish(stream) ? features : -1;
}

int solr_hip_probe_closest(const SceneInfo *sceneInfo, int n, const float *origins, const float *targets,
                           const int *iteration, const int *currentMaterialId, int features, int exactNodes, int *hit,
                           int *primitive, float *intersection, float *normal, float *areas)
{
    SceneArgs S;
    int need, deep;
    hipStream_t stream;
    if (n <= 0 || solrprobe::residentScene(*sceneInfo, exactNodes != 0, &S, &need, &deep, &stream) != 0)
        return -1;
    features = chooseFeatures(features, need, deep);
    Arrays a;
    const float *dO = a.in(origins, 3 * (size_t)n), *dT = a.in(targets, 3 * (size_t)n);
    const int *dIt = a.in(iteration, n), *dCur = a.in(currentMaterialId, n);
    int *dHit = a.out(hit, n), *dPrim = a.out(primitive, n);
    float *dI = a.out(intersection, 3 * (size_t)n), *dN = a.out(normal, 3 * (size_t)n), *dA = a.out(areas, 3 * (size_t)n);
    if (a.failed)
        return -1;
    PROBE_DISPATCH(k_probeClosest, features, waves(n), stream, S, *sceneInfo, n, dO, dT, dIt, dCur, dHit, dPrim, dI, dN, dA);
    return a.finish(stream) ? features : -1;
}

int solr_hip_probe_shadow(const SceneInfo *sceneInfo, int n, const float *lampCenters, const float *origins,
                          const int *lightId, const int *objectId, const int *iteration, int features, int exactNodes,
                          float *result, float *color)
{
    SceneArgs S;
    int need, deep;
    hipStream_t stream;
    if (n <= 0 || solrprobe::residentScene(*sceneInfo, exactNodes != 0, &S, &need, &deep, &stream) != 0)
        return -1;
    features = chooseFeatures(features, need, deep);
    Arrays a;
    const float *dL = a.in(lampCenters, 3 * (size_t)n), *dO = a.in(origins, 3 * (size_t)n);
    const int *dLight = a.in(lightId, n), *dObject = a.in(objectId, n), *dIt = a.in(iteration, n);
    float *dR = a.out(result, n), *dC = a.out(color, 3 * (size_t)n);
    if (a.failed)
        return -1;
    PROBE_DISPATCH(k_probeShadow, features, waves(n), stream, S, *sceneInfo, n, dL, dO, dLight, dObject, dIt, dR, dC);
    return a.finish(stream) ? features : -1;
}

int solr_hip_probe_shader(const SceneInfo *sceneInfo, int n, const int *index, const float *origins, const int *objectId,
                          const float *intersections, const float *areas, const int *iteration, int features, int exactNodes,
                          float *normal, float *closestColor, float *totalBlinn, float *attributes, float *returned,
                          float *shadowIntensity)
{
    SceneArgs S;
    int need, deep;
    hipStream_t stream;
    if (n <= 0 || solrprobe::residentScene(*sceneInfo, exactNodes != 0, &S, &need, &deep, &stream) != 0)
        return -1;
    features = chooseFeatures(features, need, deep);
    Arrays a;
    const int *dIndex = a.in(index, n), *dObject = a.in(objectId, n), *dIt = a.in(iteration, n);
    const float *dO = a.in(origins, 3 * (size_t)n), *dI = a.in(intersections, 3 * (size_t)n), *dA = a.in(areas, 3 * (size_t)n);
    float *dN = a.out(normal, 3 * (size_t)n), *dC = a.out(closestColor, 3 * (size_t)n), *dB = a.out(totalBlinn, 3 * (size_t)n);
    float *dAttr = a.out(attributes, 4 * (size_t)n), *dR = a.out(returned, 3 * (size_t)n), *dS = a.out(shadowIntensity, n);
    if (a.failed)
        return -1;
    PROBE_DISPATCH(k_probeShader, features, waves(n), stream, S, *sceneInfo, n, dIndex, dO, dObject, dI, dA, dIt, dN, dC, dB,
                   dAttr, dR, dS);
    return a.finish(stream) ? features : -1;
}

int solr_hip_probe_postprocess(const SceneInfo *sceneInfo, const PostProcessingInfo *postProcessingInfo,
                               const PostProcessingBuffer *frame, unsigned char *bitmap)
{
    return solrprobe::postProcess(*sceneInfo, *postProcessingInfo, frame, bitmap) == 0 ? 1 : -1;
}

int solr_hip_probe_ticket(long long serial, int *slot, long long *period)
{
    return solrprobe::ticketOfSerial(serial, slot, period);
}

long long solr_hip_probe_image_serial(long long setTo) { return solrprobe::imageSerial(setTo); }

int solr_hip_probe_vectors(int n, const float *incident, const float *normals, const float *n1, const float *n2,
                           float *refracted, float *reflected)
{
    if (n <= 0)
        return -1;
    Arrays a;
    const float *dI = a.in(incident, 3 * (size_t)n), *dN = a.in(normals, 3 * (size_t)n), *d1 = a.in(n1, n), *d2 = a.in(n2, n);
    float *dR = a.out(refracted, 3 * (size_t)n), *dF = a.out(reflected, 3 * (size_t)n);
    if (a.failed)
        return -1;
    hipLaunchKernelGGL(k_probeVectors, waves(n), dim3(64), 0, 0, n, dI, dN, d1, d2, dR, dF);
    return a.finish(0) ? 1 : -1;
}

int solr_hip_probe_make_color(const SceneInfo *sceneInfo, int n, const float *colors, unsigned char *bitmap)
{
    if (n <= 0)
        return -1;
    Arrays a;
    const float *dC = a.in(colors, 3 * (size_t)n);
    unsigned char *dB = a.out(bitmap, 3 * (size_t)n);
    if (a.failed)
        return -1;
    hipLaunchKernelGGL(k_probeMakeColor, waves(n), dim3(64), 0, 0, *sceneInfo, n, dC, dB);
    return a.finish(0) ? 1 : -1;
}

int solr_hip_probe_skybox(const SceneInfo *sceneInfo, int n, const float *origins, const float *targets, float *color)
{
    SceneArgs S;
    int need, deep;
    hipStream_t stream;
    if (n <= 0 || solrprobe::residentScene(*sceneInfo, false, &S, &need, &deep, &stream) != 0)
        return -1;
    Arrays a;
    const float *dO = a.in(origins, 3 * (size_t)n), *dT = a.in(targets, 3 * (size_t)n);
    float *dC = a.out(color, 3 * (size_t)n);
    if (a.failed)
        return -1;
    hipLaunchKernelGGL(k_probeSkybox<EVERYTHING>, waves(n), dim3(64), 0, stream, S, *sceneInfo, n, dO, dT, dC);
    return a.finish(stream) ? EVERYTHING : -1;
}

int solr_hip_probe_intersection_shader(const SceneInfo *sceneInfo, int n, const float *intersections, const float *areas,
                                       float *attributes, float *color, float *bump, float *specular,
                                       float *ambientOcclusion)
{
    SceneArgs S;
    int need, deep;
    hipStream_t stream;
    if (n <= 0 || solrprobe::residentScene(*sceneInfo, false, &S, &need, &deep, &stream) != 0)
        return -1;
    if (S.nbPrimitives < n)
    {
        solrprobe::fail(-1, "solr_hip_probe_intersection_shader: fewer primitives resident than elements");
        return -1;
    }
    Arrays a;
    const float *dI = a.in(intersections, 3 * (size_t)n), *dA = a.in(areas, 3 * (size_t)n);
    float *dAttr = a.out(attributes, 4 * (size_t)n), *dC = a.out(color, 4 * (size_t)n), *dB = a.out(bump, 3 * (size_t)n);
    float *dS = a.out(specular, 3 * (size_t)n), *dAo = a.out(ambientOcclusion, n);
    if (a.failed)
        return -1;
    hipLaunchKernelGGL(k_probeIntersectionShader<EVERYTHING>, waves(n), dim3(64), 0, stream, S, *sceneInfo, n, dI, dA, dAttr,
                       dC, dB, dS, dAo);
    return a.finish(stream) ? EVERYTHING : -1;
}
}

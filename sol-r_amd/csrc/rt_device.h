/*
 * rt_device.h - gfx950 device code of the Sol-R per-pixel rendering path.
 *
 * Written for CDNA4 wave64 execution, not translated from the reference's
 * one-thread-one-pixel CUDA code.  The unit of work is a WAVE (an 8x8 pixel
 * tile):
 *
 *  - The flattened box tree (depth-first order, skip pointers = subtree
 *    sizes) is walked by the whole wave with ONE scalar cursor `cur`.  Every
 *    lane keeps its own cursor (the index of the next node ITS ray would
 *    visit); a lane takes part in node `cur` only when its cursor equals
 *    `cur`.  Because subtrees are nested intervals, the smallest cursor in
 *    the wave is always `cur + 1` when any lane entered the node (ballot) and
 *    `cur + skip` otherwise, so the next node is found with a single
 *    __ballot and no cross-lane reduction.  Each lane therefore visits exactly
 *    the nodes, in exactly the order, of the reference's scalar walk
 *    (GeometryIntersections.cuh:687-770) - closest-hit ties resolve the same
 *    way - while all node and primitive data is wave-uniform and arrives
 *    through the scalar cache (s_load_dwordx4) or as LDS broadcast reads,
 *    leaving the vector memory path to the framebuffer alone.
 *  - Scene records are stored as planes of float4 (SoA, see scene_layout.h),
 *    split by the phase that reads them: traversal planes and shading planes.
 *  - The per-bounce colour stack of launchRayTracing (CudaRayTracer.cu:92-95)
 *    lives in LDS, one column per lane, instead of in scratch memory.
 *
 * Arithmetic: every expression keeps the reference's operand order and is
 * compiled with -ffp-contract=off, IEEE division and square root, no
 * fast-math, so results are bit-identical to the CPU oracle except where a
 * libm transcendental is involved (pow, sin/cos of procedural spheres,
 * atan2/asin of UV maps), which are evaluated in binary64 and rounded once.
 *
 * Reference citations use the same abbreviations as oracle/solr_oracle.c:
 *   CRT CudaRayTracer.cu, GI GeometryIntersections.cuh, GS GeometryShaders.cuh,
 *   TM TextureMapping.cuh, VU VectorUtils.cuh, HM helper_math.h.
 */
#pragma once

#include <hip/hip_runtime.h>
#include "pow_table.h"

#include "../../include/solr_types.h"
#include "scene_layout.h"

namespace solrdev
{
#define SOLR_DEV __device__ __forceinline__

struct v3
{
    float x, y, z;
};

SOLR_DEV v3 V(float x, float y, float z)
{
    v3 r;
    r.x = x;
    r.y = y;
    r.z = z;
    return r;
}
SOLR_DEV v3 V4(const float4 &a) { return V(a.x, a.y, a.z); }
#ifdef SOLR_PACKED_V3
/* experiment (profiles/r4/packed_v3.txt): the x and y components of the vector operators as ONE packed instruction
 * (v_pk_add_f32 / v_pk_mul_f32: two IEEE binary32 operations, each rounded as the single one is - no contraction,
 * the same bits), the z component on its own.  The products of a dot product are summed in source order. */
typedef float pk2 __attribute__((ext_vector_type(2)));
SOLR_DEV v3 operator+(v3 a, v3 b)
{
    const pk2 r = (pk2){a.x, a.y} + (pk2){b.x, b.y};
    return V(r.x, r.y, a.z + b.z);
}
SOLR_DEV v3 operator-(v3 a, v3 b)
{
    const pk2 r = (pk2){a.x, a.y} - (pk2){b.x, b.y};
    return V(r.x, r.y, a.z - b.z);
}
SOLR_DEV v3 operator*(v3 a, float b)
{
    const pk2 r = (pk2){a.x, a.y} * (pk2){b, b};
    return V(r.x, r.y, a.z * b);
}
SOLR_DEV float dot(v3 a, v3 b)
{
    const pk2 p = (pk2){a.x, a.y} * (pk2){b.x, b.y};
    return p.x + p.y + a.z * b.z;
}
#else
SOLR_DEV v3 operator+(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
SOLR_DEV v3 operator-(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
SOLR_DEV v3 operator*(v3 a, float b) { return V(a.x * b, a.y * b, a.z * b); }
SOLR_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
#endif
SOLR_DEV v3 vdivs(v3 a, float b) { return V(a.x / b, a.y / b, a.z / b); }
SOLR_DEV v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* correctly rounded square root: __builtin_sqrtf gets hipcc's IEEE expansion
 * (v_sqrt_f32 + one fma-based correction step); __fsqrt_rn does NOT - it
 * lowers to the bare 1-ULP v_sqrt_f32 on gfx950 */
SOLR_DEV float sqrt_ieee(float x) { return __builtin_sqrtf(x); }
SOLR_DEV float length(v3 v) { return sqrt_ieee(dot(v, v)); }
/* HM:62-65,1309-1313: v * (1 / sqrt(dot)) - two roundings, as on the host */
SOLR_DEV v3 normalize(v3 v)
{
    float invLen = 1.0f / sqrt_ieee(dot(v, v));
    return v * invLen;
}
/* VU:45-52 */
SOLR_DEV v3 cross(v3 b, v3 c)
{
#if defined(SOLR_PACKED_V3) && SOLR_PACKED_V3 >= 2
    const pk2 m1 = (pk2){b.y, b.z} * (pk2){c.z, c.x};
    const pk2 m2 = (pk2){b.z, b.x} * (pk2){c.y, c.z};
    const pk2 r = m1 - m2;
    return V(r.x, r.y, b.x * c.y - b.y * c.x);
#else
    v3 a;
    a.x = b.y * c.z - b.z * c.y;
    a.y = b.z * c.x - b.x * c.z;
    a.z = b.x * c.y - b.y * c.x;
    return a;
#endif
}
SOLR_DEV float sat1(float v)
{
    v = (v < 0.f) ? 0.f : v;
    v = (v > 1.f) ? 1.f : v;
    return v;
}
SOLR_DEV void saturate3(v3 &v)
{
    v.x = sat1(v.x);
    v.y = sat1(v.y);
    v.z = sat1(v.z);
}
/* VU:61-64 */
SOLR_DEV v3 vectorReflection(v3 i, v3 n)
{
    float k = 2.f * dot(i, n);
    return i - n * k;
}
/* VU:73-87 */
SOLR_DEV v3 vectorRefraction(v3 incident, float n1, v3 normal, float n2)
{
    v3 refracted = incident;
    if (n2 != 0.f)
    {
        float eta = n1 / n2;
        float c1 = -dot(incident, normal);
        float cs2 = 1.f - eta * eta * (1.f - c1 * c1);
        if (cs2 >= 0.f)
        {
            float k = eta * c1 - sqrt_ieee(cs2);
            refracted = incident * eta + normal * k;
        }
    }
    return refracted;
}
/* VU:92-95 */
SOLR_DEV v3 project(v3 A, v3 B) { return B * (dot(A, B) / dot(B, B)); }

/* transcendental stand-ins: binary64 evaluation, one rounding to binary32 */
/* pow for the Blinn term (GS:1021), evaluated in binary64 and rounded once like the other stand-ins,
 * but without the generality of the library routine (which is two hundred instructions): for a normal
 * positive base and an exponent in (0, 4096], log2 by a 128-interval table + degree-8 polynomial,
 * exp2 by a degree-10 polynomial, relative error < 2^-41 (tools/gen_pow_table.py checks the binary32
 * results against pow() on 200 000 inputs: none differs).  +0 gives +0.  Anything else - negative,
 * infinite, NaN, subnormal bases, other exponents - sends the whole wave to the library routine. */
/* out of line: never taken on sane inputs, and inlining it doubles the live registers at its call site */
__device__ __attribute__((noinline)) float pow_general(float a, float b) { return (float)pow((double)a, (double)b); }

/* The polynomials' coefficients are materialised where they are used, in scalar registers, by instructions the
 * compiler cannot move: left to itself it hoists the nineteen binary64 constants out of the bounce loop - a 64-bit
 * immediate is a pair of moves it does not know how to re-materialise - and some thirty vector registers of the
 * kernel's 128 then hold them through every walk of the frame, six of them by way of scratch memory
 * (profiles/r3/isa_notes.txt).  Two s_mov_b32 per coefficient per call instead, and v_fma_f64 takes the pair as its
 * addend. */
template <unsigned long long BITS>
SOLR_DEV double scalarConstant()
{
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3"
                 : "=s"(lo), "=s"(hi)
                 : "n"((unsigned)(BITS & 0xffffffffull)), "n"((unsigned)(BITS >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
constexpr double POW_LOG_COEFFS[8] = SOLR_POW_LOG_COEFFS;
constexpr double POW_EXP_COEFFS[13] = SOLR_POW_EXP_COEFFS;
/* p * x + c[K], then on down to c[0] */
template <int K, const double *C>
SOLR_DEV double hornerDown(double p, double x)
{
    p = __builtin_fma(p, x, scalarConstant<__builtin_bit_cast(unsigned long long, C[K])>());
    if constexpr (K > 0)
        return hornerDown<K - 1, C>(p, x);
    else
        return p;
}

SOLR_DEV float pow_f(float a, float b)
{
#ifdef SOLR_LIBRARY_POW /* A/B builds: the library routine everywhere */
    return pow_general(a, b);
#endif
    const int ix = __float_as_int(a);
    const bool lean = (a >= 1.17549435e-38f) && (a < 3.0e38f) && (b > 0.f) && (b <= 4096.f);
    const bool zero = (ix == 0) && (b > 0.f) && (b <= 4096.f);
    if (__builtin_amdgcn_ballot_w64(!(lean || zero)) != 0ull)
        return pow_general(a, b);
    const int e = (ix >> 23) - 127;
    const int idx = (ix >> 16) & 0x7f;
    const double m = (double)__int_as_float((ix & 0x007fffff) | 0x3f800000);
    const double invc = SOLR_POW_LOG_TABLE[2 * idx], log2c = SOLR_POW_LOG_TABLE[2 * idx + 1];
    const double r = __builtin_fma(m, invc, -1.0); /* exact: 24-bit m times 28-bit invc */
    const double p = hornerDown<6, POW_LOG_COEFFS>(scalarConstant<__builtin_bit_cast(unsigned long long, POW_LOG_COEFFS[7])>(), r);
    const double L = ((double)e + log2c) + p * r;
    const double t = (double)b * L;
    const double kk = __builtin_rint(t);
    const double f = t - kk;
    const double q = hornerDown<9, POW_EXP_COEFFS>(scalarConstant<__builtin_bit_cast(unsigned long long, POW_EXP_COEFFS[10])>(), f);
    const float result = (float)__builtin_ldexp(q, (int)kk);
    return zero ? 0.f : result;
}
/* (out of line like pow_general: binary64 library routines of a few hundred instructions each, in the kernels of
 * procedural spheres, texture coordinates and the fish-eye camera only; inlined at every use they were what the
 * all-features instantiations spilled 300 registers around.  The two procedural textures out of line as well gave
 * the opposite: 51 -> 133 spills - a call costs its caller every register that is live across it) */
__device__ __attribute__((noinline)) float cos_f(float a) { return (float)cos((double)a); }
__device__ __attribute__((noinline)) float sin_f(float a) { return (float)sin((double)a); }
__device__ __attribute__((noinline)) float atan2_f(float a, float b) { return (float)atan2((double)a, (double)b); }
__device__ __attribute__((noinline)) float asin_f(float a) { return (float)asin((double)a); }

SOLR_DEV int asint(float f) { return __float_as_int(f); }
/* lane mask of a predicate, without the int round trip of HIP's ballot(int) */
SOLR_DEV unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

/* VU:104-142 with the six sin/cos values hoisted to the host (they depend on
 * the camera angles only) */
struct Trig
{
    float cx, cy, cz, sx, sy, sz;
};
SOLR_DEV v3 vectorRotation(v3 v, v3 c, const Trig &t)
{
    float vx = v.x - c.x, vy = v.y - c.y, vz = v.z - c.z;
    float rx = vx, ry, rz;
    ry = vy * t.cx - vz * t.sx;
    rz = vy * t.sx + vz * t.cx;
    vy = ry;
    vz = rz;
    rz = vz * t.cy - vx * t.sy;
    rx = vz * t.sy + vx * t.cy;
    vz = rz;
    vx = rx;
    rx = vx * t.cz - vy * t.sz;
    ry = vx * t.sz + vy * t.cz;
    return V(rx + c.x, ry + c.y, rz + c.z);
}

/* per-lane ray of a walk: origin, un-normalised direction, reciprocal, signs
 * (GI:36-44, types.h:171-178) */
struct WalkRay
{
    v3 o, d, inv;
    v3 dn;    /* normalize(d): the reference recomputes it in every sphere / ellipsoid /
                 triangle test of the walk (GI:168,227,642,650); it only depends on the ray */
    bool sx, sy, sz;
};
SOLR_DEV WalkRay makeWalkRay(v3 origin, v3 direction)
{
    WalkRay r;
    r.o = origin;
    r.d = direction;
    r.dn = normalize(direction);
    r.inv.x = direction.x != 0.f ? 1.f / direction.x : 1.f;
    r.inv.y = direction.y != 0.f ? 1.f / direction.y : 1.f;
    r.inv.z = direction.z != 0.f ? 1.f / direction.z : 1.f;
    r.sx = (r.inv.x < 0);
    r.sy = (r.inv.y < 0);
    r.sz = (r.inv.z < 0);
    return r;
}

/* GI:52-79; lo/hi are wave-uniform.  Exact form: sign-selected slabs and the
 * reference's compare/assign chain (keeps its NaN and inverted-box behaviour). */
SOLR_DEV bool boxIntersectionExact(const float4 &lo, const float4 &hi, const WalkRay &r, float t0, float t1)
{
    float ax = (lo.x - r.o.x) * r.inv.x, bx = (hi.x - r.o.x) * r.inv.x;
    float tmin = r.sx ? bx : ax;
    float tmax = r.sx ? ax : bx;
    float ay = (lo.y - r.o.y) * r.inv.y, by = (hi.y - r.o.y) * r.inv.y;
    float tymin = r.sy ? by : ay;
    float tymax = r.sy ? ay : by;
    bool ok = !((tmin > tymax) || (tymin > tmax));
    tmin = (tymin > tmin) ? tymin : tmin;
    tmax = (tymax < tmax) ? tymax : tmax;
    float az = (lo.z - r.o.z) * r.inv.z, bz = (hi.z - r.o.z) * r.inv.z;
    float tzmin = r.sz ? bz : az;
    float tzmax = r.sz ? az : bz;
    ok = ok && !((tmin > tzmax) || (tzmin > tmax));
    tmin = (tzmin > tmin) ? tzmin : tmin;
    tmax = (tzmax < tmax) ? tzmax : tmax;
    return ok && ((tmin < t1) && (tmax > t0));
}

/* Sign-free form.  Identical to the exact form whenever (a) lo <= hi on every
 * axis and all six bounds are finite (checked for the whole node list at
 * upload) and (b) the lane's three reciprocals are finite (checked once per
 * walk): then every slab product is a finite number, the sign-selected near /
 * far values are min / max of the pair, the reference's four early-outs are
 * exactly "some near value exceeds some far value" = max3(near) > min3(far),
 * and its running tmin / tmax are max3(near) / min3(far).  About half the
 * vector instructions and none of the exec-mask bookkeeping of the exact form. */
SOLR_DEV bool boxIntersectionFast(const float4 &lo, const float4 &hi, const WalkRay &r, float t0, float t1)
{
    const float ax = (lo.x - r.o.x) * r.inv.x, bx = (hi.x - r.o.x) * r.inv.x;
    const float ay = (lo.y - r.o.y) * r.inv.y, by = (hi.y - r.o.y) * r.inv.y;
    const float az = (lo.z - r.o.z) * r.inv.z, bz = (hi.z - r.o.z) * r.inv.z;
    const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    return (tmin <= tmax) && (tmin < t1) && (tmax > t0);
}


/* ---------------------------------------------------------------------- */
/* Texture tier (TM)                                                       */
/* ---------------------------------------------------------------------- */

struct TexOut
{
    v3 *normal;          /* bump accumulator */
    float4 *specular;    /* x value, y power, z transparency-from-specular */
    float4 *attributes;  /* x reflection, y transparency */
    float *ambientOcclusion;
};

/* TM:238-279 (and its two copies): texel fetch plus the optional maps */
SOLR_DEV void fetchTexel(const MaterialCold &mc, cbp tex, int u, int v, float4 &result,
                         const TexOut &o)
{
    int A = (v * mc.textureMapping.x + u) * mc.textureMapping.w;
    int B = mc.textureMapping.x * mc.textureMapping.y * mc.textureMapping.w;
    int index = A % B;
    int i = mc.textureOffset.x + index;
    unsigned char r = tex[i], g = tex[i + 1], b = tex[i + 2];
    result.x = r / 256.f;
    result.y = g / 256.f;
    result.z = b / 256.f;
    float strength = 3.f;
    if (mc.textureIds.z != TEXTURE_NONE) /* TM:45-57 */
    {
        int j = mc.textureOffset.z + index;
        unsigned char br = tex[j], bg = tex[j + 1], bb = tex[j + 2];
        strength = 10.f * (br + bg + bb) / 768.f;
    }
    if (mc.textureIds.y != TEXTURE_NONE) /* TM:30-40 */
    {
        int j = mc.textureOffset.y + index;
        unsigned char nr = tex[j], ng = tex[j + 1];
        o.normal->x -= strength * (nr / 256.f - 0.5f);
        o.normal->y -= strength * (ng / 256.f - 0.5f);
        o.normal->z = 0.f;
    }
    if (mc.textureIds.w != TEXTURE_NONE) /* TM:62-73 */
    {
        int j = mc.textureOffset.w + index;
        unsigned char sr = tex[j], sg = tex[j + 1], sb = tex[j + 2];
        o.specular->x = sr / 256.f;
        o.specular->y = 1000.f * sg / 256.f;
        o.specular->z = sb / 256.f;
    }
    if (mc.advancedTextureIds.x != TEXTURE_NONE) /* TM:78-87 */
    {
        int j = mc.advancedTextureOffset.x + index;
        unsigned char rr = tex[j], rg = tex[j + 1], rb = tex[j + 2];
        o.attributes->x *= (rr + rg + rb) / 768.f;
    }
    if (mc.advancedTextureIds.y != TEXTURE_NONE) /* TM:92-102 */
    {
        int j = mc.advancedTextureOffset.y + index;
        unsigned char tr = tex[j], tg = tex[j + 1], tb = tex[j + 2];
        o.attributes->y *= (tr + tg + tb) / 768.f;
    }
    if (mc.advancedTextureIds.z != TEXTURE_NONE) /* TM:107-116 */
    {
        int j = mc.advancedTextureOffset.z + index;
        unsigned char ar = tex[j], ag = tex[j + 1], ab = tex[j + 2];
        *o.ambientOcclusion = (ar + ag + ab) / 768.f;
    }
}

/* The two procedural textures (TM:118-197): escape-time fractals over the material's texture rectangle, a texel's
 * shade being 1 - colour x (steps taken / step limit).  Both iterate z <- z * z + c in binary32; the recurrence and
 * the shade are written once here, with the reference's order of operations, and the two functions keep what they
 * differ in: Julia looks at the NEW z and does not count the step that escapes, Mandelbrot looks at the OLD z, still
 * takes the step and counts it; Mandelbrot's imaginary pitch is held in binary64 (TM:172-175). */
struct Complex32
{
    float re, im;
};

SOLR_DEV Complex32 squaredPlus(Complex32 z, Complex32 c)
{
    Complex32 r;
    r.re = z.re * z.re - z.im * z.im + c.re;
    r.im = 2.f * z.re * z.im + c.im;
    return r;
}

SOLR_DEV bool escaped(Complex32 z) { return (z.re * z.re + z.im * z.im) > 4.f; }

SOLR_DEV void escapeShade(float4 &color, float steps, float limit)
{
    const float share = steps / limit;
    color.x = 1.f - color.x * share;
    color.y = 1.f - color.y * share;
    color.z = 1.f - color.z * share;
    color.w = 1.f - share;
}

/* TM:118-158: c drifts with the time stamp, z starts at the texel (1.5 : 1 over the rectangle's half extents) */
SOLR_DEV void juliaSet(const MaterialCold &mc, const SceneInfo &si, float x, float y, float4 &color)
{
    const float w = (float)mc.textureMapping.x, h = (float)mc.textureMapping.y;
    Complex32 c, z;
    c.re = -0.7f + 0.4f * sin_f(si.timestamp / 1500.f);
    c.im = 0.27015f + 0.4f * cos_f(si.timestamp / 2000.f);
    z.re = 1.5f * (x - w / 2.f) / (0.5f * w);
    z.im = (y - h / 2.f) / (0.5f * h);
    const float limit = 40.f + si.pathTracingIteration;
    int steps = 0;
    while (steps < limit)
    {
        z = squaredPlus(z, c);
        if (escaped(z))
            break;
        ++steps;
    }
    escapeShade(color, (float)steps, limit);
}

/* TM:160-197: c is the texel in the window re -2 ... 1, im -1.2 ... -1.2 + 3 h / w, z starts at c */
SOLR_DEV void mandelbrotSet(const MaterialCold &mc, const SceneInfo &si, float x, float y, float4 &color)
{
    const float w = (float)mc.textureMapping.x, h = (float)mc.textureMapping.y;
    const float left = -2.f, right = 1.f, bottom = -1.2f;
    const float top = bottom + (right - left) * h / w;
    const float pitchRe = (right - left) / (w - 1.f);
    const double pitchIm = (top - bottom) / (h - 1.f); /* (a binary32 quotient, widened) */
    const float limit = NB_MAX_ITERATIONS + si.pathTracingIteration;
    Complex32 c;
    c.im = (float)(top - y * pitchIm);
    c.re = left + x * pitchRe;
    Complex32 z = c;
    unsigned steps = 0;
    for (bool inside = true; inside && steps < limit; ++steps)
    {
        inside = !escaped(z);
        z = squaredPlus(z, c);
    }
    escapeShade(color, (float)steps, limit);
}

/* TM:354-447 (non-Kinect build) */
SOLR_DEV float4 cubeMapping(const SceneInfo &si, int type, v3 p0, v3 size, const float4 &matColor,
                            const MaterialCold &mc, cbp tex, v3 intersection,
                            const TexOut &o)
{
    float4 result = matColor;
    int u = (int)(((type == ptCheckboard) || (type == ptXZPlane) || (type == ptXYPlane))
                      ? (intersection.x - p0.x + size.x)
                      : (intersection.z - p0.z + size.z));
    int v = (int)(((type == ptCheckboard) || (type == ptXZPlane)) ? (intersection.z + p0.z + size.z)
                                                                  : (intersection.y - p0.y + size.y));
    if (mc.textureMapping.x != 0)
        u = u % mc.textureMapping.x;
    if (mc.textureMapping.y != 0)
        v = v % mc.textureMapping.y;
    if (u >= 0 && u < mc.textureMapping.x && v >= 0 && v < mc.textureMapping.x) /* sic, TM:398 */
    {
        switch (mc.textureIds.x)
        {
        case TEXTURE_MANDELBROT:
            mandelbrotSet(mc, si, (float)u, (float)v, result);
            break;
        case TEXTURE_JULIA:
            juliaSet(mc, si, (float)u, (float)v, result);
            break;
        default:
            fetchTexel(mc, tex, u, v, result, o);
        }
    }
    return result;
}

/* TM:291-346 */
SOLR_DEV float4 sphereUVMapping(v3 p0, float vt1x, float vt1y, const float4 &matColor, const MaterialCold &mc,
                                cbp tex, v3 intersection, const TexOut &o)
{
    float4 result = matColor;
    v3 I = normalize(intersection - p0);
    float U = ((atan2_f(I.x, I.z) / SOLR_PI) + 1.f) * .5f;
    float Vv = (asin_f(I.y) / SOLR_PI) + .5f;
    int u = (int)(mc.textureMapping.x * (U * vt1x));
    int v = (int)(mc.textureMapping.y * (Vv * vt1y));
    if (mc.textureMapping.x != 0)
        u = u % mc.textureMapping.x;
    if (mc.textureMapping.y != 0)
        v = v % mc.textureMapping.y;
    if (u >= 0 && u < mc.textureMapping.x && v >= 0 && v < mc.textureMapping.y)
        fetchTexel(mc, tex, u, v, result, o);
    return result;
}

/* TM:205-283 */
SOLR_DEV float4 triangleUVMapping(const SceneInfo &si, float2 vt0, float2 vt1, float2 vt2, int procedural,
                                  const float4 &matColor, const MaterialCold &mc,
                                  cbp tex, v3 areas, const TexOut &o)
{
    float4 result = matColor;
    float sum = areas.x + areas.y + areas.z;
    float Tx = (vt0.x * areas.x + vt1.x * areas.y + vt2.x * areas.z) / sum;
    float Ty = (vt0.y * areas.x + vt1.y * areas.y + vt2.y * areas.z) / sum;
    float mox = 0.f, moy = 0.f;
    if (procedural == 1)
    {
        mox = mc.mappingOffset.x * si.timestamp;
        moy = mc.mappingOffset.y * si.timestamp;
    }
    int u = (int)(Tx * mc.textureMapping.x + mox);
    int v = (int)(Ty * mc.textureMapping.y + moy);
    /* x % 0 traps on the host; the device returns the dividend unchanged */
    u = mc.textureMapping.x != 0 ? u % mc.textureMapping.x : u;
    v = mc.textureMapping.y != 0 ? v % mc.textureMapping.y : v;
    if (u >= 0 && u < mc.textureMapping.x && v >= 0 && v < mc.textureMapping.y)
    {
        switch (mc.textureIds.x)
        {
        case TEXTURE_MANDELBROT:
            mandelbrotSet(mc, si, (float)u, (float)v, result);
            break;
        case TEXTURE_JULIA:
            juliaSet(mc, si, (float)u, (float)v, result);
            break;
        default:
            fetchTexel(mc, tex, u, v, result, o);
        }
    }
    return result;
}

/* TM:449-456 */
SOLR_DEV bool wireFrameMapping(float x, float y, int width)
{
    int X = (int)fabsf(x);
    int Y = (int)fabsf(y);
    return (X % 100 <= width) || (Y % 100 <= width);
}

/* ---------------------------------------------------------------------- */
/* Primitive intersections (GI); primitive data is wave-uniform            */
/* ---------------------------------------------------------------------- */

/* Scene features a kernel instantiation supports.  The host scans the uploaded
 * scene and launches the smallest instantiation that covers it: code (and
 * registers) for primitive types and material features that do not occur in
 * the scene is not compiled in.  Keeping the kernel free of scratch spills this
 * way is worth far more than any single arithmetic optimisation (a variant with
 * the walks as real calls, i.e. state parked in scratch, ran 1.5-6x slower). */
enum Feature
{
    F_SPHERE = 1,   /* plain spheres / environment */
    F_PROC = 2,     /* procedural (bumpy) spheres: binary64 sin/cos */
    F_CYL = 4,      /* cylinders, cones */
    F_ELL = 8,      /* ellipsoids */
    F_TRI = 16,     /* triangles */
    F_PLANE = 32,   /* axis planes, checkerboards, magic carpet */
    F_TEX = 64,     /* textured materials, ptCamera planes, textured skybox */
    F_FULL = 128,   /* global illumination + box-debug view */
    F_ALL = 255,
    F_DEEP = 256,   /* not a feature of the scene but of its node list: walk it with the three-bank loop (advanceTidy) */
    F_STACK = 512,  /* nor of the scene but of the frame: more bounces than colour-stack slots are kept in LDS
                     * (SOLR_LDS_STACK_SLOTS) - the deeper slots live in a per-pixel buffer in HBM (ColorStack) */
    F_STREAM = 1024 /* of the frame: its image leaves in bands while it renders (renderer.h ImageStreaming) - the epilogue
                     * that counts tiles lives in instantiations of its own, so that every other frame runs the code it
                     * always ran (an epilogue with the branches in it cost the Cornell kernel 0.5 %: the compiler's
                     * allocation of the WHOLE kernel changed with it) */
};
/* Colour-stack slots (4 dwords each) a lane keeps in LDS.  With the 27-dword cold record that is 39 dwords per lane: what
 * 16 waves per CU - 4 per SIMD, the kernel's register budget - leave each lane of the 160 KB.  A frame that may bounce
 * deeper (the accumulation passes: up to NB_MAX_ITERATIONS = 10) used to size the LDS stack for it - 67 dwords, 9 waves
 * per CU - although hardly a ray goes that deep. */
#define SOLR_LDS_STACK_SLOTS 3

struct Hit
{
    v3 intersection;
    v3 normal;
    v3 areas;
    float shadowIntensity;
};

/* GI:159-212 */
SOLR_DEV bool ellipsoidIntersection(const SceneInfo &si, v3 p0, v3 size, const WalkRay &ray, Hit &h)
{
    h.shadowIntensity = 1.f;
    v3 O_C = ray.o - p0;
    v3 dir = ray.dn;
    float a = ((dir.x * dir.x) / (size.x * size.x)) + ((dir.y * dir.y) / (size.y * size.y)) +
              ((dir.z * dir.z) / (size.z * size.z));
    float b = ((2.f * O_C.x * dir.x) / (size.x * size.x)) + ((2.f * O_C.y * dir.y) / (size.y * size.y)) +
              ((2.f * O_C.z * dir.z) / (size.z * size.z));
    float c = ((O_C.x * O_C.x) / (size.x * size.x)) + ((O_C.y * O_C.y) / (size.y * size.y)) +
              ((O_C.z * O_C.z) / (size.z * size.z)) - 1.f;
    float d = ((b * b) - (4.f * a * c));
    if (d < 0.f || a == 0.f || b == 0.f || c == 0.f)
        return false;
    d = sqrt_ieee(d);
    float t1 = (-b + d) / (2.f * a);
    float t2 = (-b - d) / (2.f * a);
    if (t1 <= si.geometryEpsilon && t2 <= si.geometryEpsilon)
        return false;
    float t = 0.f;
    if (t1 <= si.geometryEpsilon)
        t = t2;
    else if (t2 <= si.geometryEpsilon)
        t = t1;
    else
        t = (t1 < t2) ? t1 : t2;
    if (t < si.geometryEpsilon)
        return false;
    h.intersection = ray.o + dir * t;
    v3 n = h.intersection - p0;
    n.x = 2.f * n.x / (size.x * size.x);
    n.y = 2.f * n.y / (size.y * size.y);
    n.z = 2.f * n.z / (size.z * size.z);
    h.normal = normalize(n);
    return true;
}

/* GI:220-284, split in two so that the walk can decide first and pay for the
 * normal only when it keeps the hit.  sphereHit: everything up to the
 * intersection point, branch-free apart from one wave-uniform early-out. */
SOLR_DEV bool sphereHit(const SceneInfo &si, v3 p0, float radius, const WalkRay &ray, bool lanes, v3 &intersection,
                        bool &back)
{
    const v3 O_C = ray.o - p0;
    const v3 dir = ray.dn;
    const float a = 2.f * dot(dir, dir);
    const float b = 2.f * dot(O_C, dir);
    const float c = dot(O_C, O_C) - (radius * radius);
    const float d = b * b - 2.f * a * c;
    bool ok = lanes && !(d <= 0.f || a == 0.f);
    back = false;
    if (ballot(ok) == 0ull)
        return false;
    const float r = sqrt_ieee(d);
    const float t1 = (-b - r) / a;
    const float t2 = (-b + r) / a;
    const bool b1 = t1 <= si.geometryEpsilon;
    const bool b2 = t2 <= si.geometryEpsilon;
    ok = ok && !(b1 && b2); /* both intersections behind the origin */
    const float t = b1 ? t2 : (b2 ? t1 : ((t1 < t2) ? t1 : t2));
    back = b1;
    ok = ok && !(t < si.geometryEpsilon);
    intersection = ray.o + dir * t;
    return ok;
}

/* the normal and the shadow intensity of a sphere hit (GI:261-281) */
template <bool PROCEDURAL>
SOLR_DEV void sphereNormal(const SceneInfo &si, v3 p0, v3 size, bool procedural, bool transparent, bool back,
                           const WalkRay &ray, Hit &h)
{
    v3 n;
    if (!(PROCEDURAL && procedural))
        n = h.intersection - p0;
    else
    {
        v3 newCenter;
        newCenter.x = p0.x + 0.008f * size.x * cos_f(si.timestamp + h.intersection.x);
        newCenter.y = p0.y + 0.008f * size.y * sin_f(si.timestamp + h.intersection.y);
        newCenter.z = p0.z + 0.008f * size.z * sin_f(cos_f(si.timestamp + h.intersection.z));
        n = h.intersection - newCenter;
    }
    n = normalize(n);
    n = back ? n * -1.f : n;
    h.normal = n;
    const float r = dot(ray.dn, n);
    h.shadowIntensity = transparent ? (1.f - fabsf(r)) : 1.f;
}

/* one-piece form for the code paths that want everything at once */
SOLR_DEV bool sphereIntersection(const SceneInfo &si, v3 p0, v3 size, bool procedural, bool transparent,
                                 const WalkRay &ray, Hit &h)
{
    bool back;
    const bool hit = sphereHit(si, p0, size.x, ray, true, h.intersection, back);
    if (hit)
        sphereNormal<true>(si, p0, size, procedural, transparent, back, ray, h);
    return hit;
}

/* GI:293-349 and GI:358-416 (the cone repeats the cylinder's arithmetic) */
SOLR_DEV bool cylinderIntersection(const SceneInfo &si, v3 p0, v3 p1, v3 p2, v3 n1, v3 size, const WalkRay &ray,
                                   Hit &h)
{
    v3 O_C = ray.o - p0;
    v3 dir = ray.d;
    v3 n = cross(dir, n1);
    float ln = length(n);
    if ((ln < si.geometryEpsilon) && (ln > -si.geometryEpsilon))
        return false;
    n = normalize(n);
    float d = fabsf(dot(O_C, n));
    if (d > size.y)
        return false;
    v3 O = cross(O_C, n1);
    float t = -dot(O, n) / ln;
    if (t < 0.f)
        return false;
    O = normalize(cross(n, n1));
    float s = fabsf(sqrt_ieee(size.x * size.x - d * d) / dot(dir, O));
    float t1 = t - s;
    float t2 = t + s;
    v3 I = ray.o + dir * t1;
    v3 HB1 = I - p0;
    v3 HB2 = I - p1;
    float scale1 = dot(HB1, n1);
    float scale2 = dot(HB2, n1);
    if (scale1 < si.geometryEpsilon || scale2 > si.geometryEpsilon)
    {
        I = ray.o + dir * t2;
        HB1 = I - p0;
        HB2 = I - p1;
        scale1 = dot(HB1, n1);
        scale2 = dot(HB2, n1);
        if (scale1 < si.geometryEpsilon || scale2 > si.geometryEpsilon)
            return false;
    }
    h.intersection = I;
    v3 Vv = I - p2;
    h.normal = normalize(Vv - project(Vv, n1));
    h.shadowIntensity = 1.f;
    return true;
}

/* GI:424-567.  The two faces of a rectangle share the intersection formula and
 * their side conditions exclude each other, so the reference's "try the front
 * face, else try the back face with the normal negated" is evaluated here
 * without branches: one point, two side predicates.  U,V in-plane axes, W the
 * normal axis. */
#define SOLR_PLANE_POINT(U, Vx, Wx)                                                                              \
    do                                                                                                           \
    {                                                                                                            \
        const float k = ray.o.Wx - p0.Wx;                                                                        \
        I.U = ray.o.U + k * ray.d.U / -ray.d.Wx;                                                                 \
        I.Wx = p0.Wx;                                                                                            \
        I.Vx = ray.o.Vx + k * ray.d.Vx / -ray.d.Wx;                                                              \
        inside = fabsf(I.U - p0.U) < size.U && fabsf(I.Vx - p0.Vx) < size.Vx;                                    \
        front = ray.d.Wx < 0.f && ray.o.Wx > p0.Wx;                                                              \
        rear = ray.d.Wx > 0.f && ray.o.Wx < p0.Wx;                                                               \
    } while (0)

/* pm carries the uniform material facts the test needs. */
struct PlaneMaterial
{
    int wireframe;      /* attributes.z */
    int wireframeWidth; /* attributes.w */
    bool emissive;      /* innerIllumination.x != 0 */
    bool textured;      /* textureIds.x != TEXTURE_NONE */
    int materialId;
    float averageColor; /* (r + g + b) / 3.f of the material colour, evaluated at upload */
};

template <bool TEX>
SOLR_DEV bool planeIntersection(const SceneInfo &si, int type, v3 p0, v3 size, v3 n0, const PlaneMaterial &pm,
                                const Scene &planes, int materialId, const WalkRay &ray, Hit &h)
{
    bool inside = false, front = false, rear = false;
    v3 I = h.intersection;
    switch (type)
    {
    case ptMagicCarpet:
    case ptCheckboard:
        SOLR_PLANE_POINT(x, z, y);
        rear = false; /* single sided, GI:435-446 */
        break;
    case ptXZPlane:
        SOLR_PLANE_POINT(x, z, y);
        if (pm.wireframe == 2)
            inside = inside && wireFrameMapping(I.x, I.z, pm.wireframeWidth);
        break;
    case ptYZPlane:
        SOLR_PLANE_POINT(y, z, x);
        if (pm.emissive) /* chessboard-like lights, GI:486-490 */
            inside = inside && ((int)fabsf(I.z) % 4000 < 2000 && (int)fabsf(I.y) % 4000 < 2000);
        if (pm.wireframe == 2)
            inside = inside && wireFrameMapping(I.y, I.z, pm.wireframeWidth);
        break;
    case ptXYPlane:
    case ptCamera:
        SOLR_PLANE_POINT(x, y, z);
        if (pm.wireframe == 2)
            inside = inside && wireFrameMapping(I.x, I.y, pm.wireframeWidth);
        break;
    default:
        break;
    }
    bool collision = (front || rear) && inside;
    v3 normal = rear ? vneg(n0) : n0;

    if (TEX && (type == ptCamera || pm.textured))
    {
        if (collision)
        {
            const float4 matColor = loadMaterialHot(planes, pm.materialId).color;
            float4 specular = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 attributes = make_float4(0.f, 0.f, 0.f, 0.f);
            float ao = 0.f;
            TexOut o = {&normal, &specular, &attributes, &ao};
            const float4 color =
                cubeMapping(si, type, p0, size, matColor, loadMaterialCold(planes, materialId), planes.textures, I, o);
            h.shadowIntensity = color.w;
            if ((color.x + color.y + color.z) / 3.f >= si.transparentColor)
                collision = false;
        }
    }
    else
    {
        h.shadowIntensity = 1.f;
        /* colour-key transparency, GI:561 */
        collision = collision && !(pm.averageColor >= si.transparentColor);
    }
    h.intersection = I;
    h.normal = normal;
    return collision;
}

/* GI:575-659 */
/* GI:575-626: does the ray hit the triangle, and where.  Nothing below this point of the reference's
 * function can turn a hit into a miss except the double-sided shadow case handled by the caller. */
SOLR_DEV bool triangleHit(const SceneInfo &si, v3 p0, v3 p1, v3 p2, const WalkRay &ray, v3 &intersection)
{
    v3 E01 = p1 - p0;
    v3 E03 = p2 - p0;
    v3 P = cross(ray.d, E03);
    float det = dot(E01, P);
    if (fabsf(det) < si.geometryEpsilon)
        return false;
    v3 T = ray.o - p0;
    float a = dot(T, P) / det;
    if (a < 0.f || a > 1.f)
        return false;
    v3 Q = cross(T, E01);
    float b = dot(ray.d, Q) / det;
    if (b < 0.f || b > 1.f)
        return false;
    if ((a + b) > 1.f)
    {
        /* GI:603-616 with E21 = p1 - p1 (sic) */
        v3 E23 = p0 - p1;
        v3 E21 = p1 - p1;
        v3 P_ = cross(ray.d, E21);
        float det_ = dot(E23, P_);
        if (fabsf(det_) < si.geometryEpsilon)
            return false;
        v3 T_ = ray.o - p2;
        float a_ = dot(T_, P_) / det_;
        if (a_ < 0.f)
            return false;
        v3 Q_ = cross(T_, E23);
        float b_ = dot(ray.d, Q_) / det_;
        if (b_ < 0.f)
            return false;
    }
    float t = dot(E03, Q) / det;
    if (t < 0)
        return false;
    intersection = ray.o + ray.d * t;
    return true;
}

/* GI:627-655: barycentric areas and the interpolated, ray-facing normal at the hit point.  The walks
 * call it only for hits they keep (closest-hit walk) or whose normal they use (shadows through a
 * transparent triangle): about two thirds of the arithmetic of a triangle test. */
SOLR_DEV void triangleNormal(v3 p0, v3 p1, v3 p2, v3 n0, v3 n1, v3 n2, const WalkRay &ray, Hit &h)
{
    v3 v0 = p0 - h.intersection;
    v3 v1 = p1 - h.intersection;
    v3 v2 = p2 - h.intersection;
    h.areas.x = 0.5f * length(cross(v1, v2));
    h.areas.y = 0.5f * length(cross(v0, v2));
    h.areas.z = 0.5f * length(cross(v0, v1));
    v3 wn = (n0 * h.areas.x + n1 * h.areas.y) + n2 * h.areas.z;
    h.normal = normalize(vdivs(wn, h.areas.x + h.areas.y + h.areas.z));
    v3 dir = ray.dn;
    float r = dot(dir, h.normal);
    if (r > 0.f)
        h.normal = h.normal * -1.f;
    h.shadowIntensity = 1.f;
}

SOLR_DEV bool triangleIntersection(const SceneInfo &si, v3 p0, v3 p1, v3 p2, v3 n0, v3 n1, v3 n2,
                                   const WalkRay &ray, Hit &h, bool processingShadows)
{
    v3 E01 = p1 - p0;
    v3 E03 = p2 - p0;
    v3 P = cross(ray.d, E03);
    float det = dot(E01, P);
    if (fabsf(det) < si.geometryEpsilon)
        return false;
    v3 T = ray.o - p0;
    float a = dot(T, P) / det;
    if (a < 0.f || a > 1.f)
        return false;
    v3 Q = cross(T, E01);
    float b = dot(ray.d, Q) / det;
    if (b < 0.f || b > 1.f)
        return false;
    if ((a + b) > 1.f)
    {
        /* GI:603-616 with E21 = p1 - p1 (sic) */
        v3 E23 = p0 - p1;
        v3 E21 = p1 - p1;
        v3 P_ = cross(ray.d, E21);
        float det_ = dot(E23, P_);
        if (fabsf(det_) < si.geometryEpsilon)
            return false;
        v3 T_ = ray.o - p2;
        float a_ = dot(T_, P_) / det_;
        if (a_ < 0.f)
            return false;
        v3 Q_ = cross(T_, E23);
        float b_ = dot(ray.d, Q_) / det_;
        if (b_ < 0.f)
            return false;
    }
    float t = dot(E03, Q) / det;
    if (t < 0)
        return false;
    h.intersection = ray.o + ray.d * t;
    v3 v0 = p0 - h.intersection;
    v3 v1 = p1 - h.intersection;
    v3 v2 = p2 - h.intersection;
    h.areas.x = 0.5f * length(cross(v1, v2));
    h.areas.y = 0.5f * length(cross(v0, v2));
    h.areas.z = 0.5f * length(cross(v0, v1));
    v3 wn = (n0 * h.areas.x + n1 * h.areas.y) + n2 * h.areas.z;
    h.normal = normalize(vdivs(wn, h.areas.x + h.areas.y + h.areas.z));
    if (si.doubleSidedTriangles)
    {
        /* GI:643-647: dangling else - shadow tests always miss */
        v3 N = ray.dn;
        if (processingShadows)
        {
            if (dot(N, h.normal) <= 0.f)
                return false;
            else if (dot(N, h.normal) >= 0.f)
                return false;
        }
    }
    v3 dir = ray.dn;
    float r = dot(dir, h.normal);
    if (r > 0.f)
        h.normal = h.normal * -1.f;
    h.shadowIntensity = 1.f;
    return true;
}

/* ---------------------------------------------------------------------- */
/* Scene access                                                            */
/* ---------------------------------------------------------------------- */

/* All pointers are read-only for the lifetime of the launch.  Indices that
 * are wave-uniform (box cursor, leaf primitive index, light index, material
 * of a uniform primitive) make the compiler select scalar loads. */
struct Counters
{
    unsigned int closest, shadow, boxes, prims; /* per lane */
    unsigned int wNodes, wPrims, wClosest, wShadow; /* per wave (only lane 0's copy is reported) */
    char *record;          /* COUNT == 2: this workgroup's slot of walk records */
    unsigned int ordinal;  /* ... and the number of walks it has recorded */
#ifdef SOLR_TIMING
    /* development build (make EXTRA_HIPFLAGS=-DSOLR_TIMING, tools/wave_time_split.py): shader-clock cycles a wave
     * spends in the node loop, at leaves, in either walk as a whole, and how often */
    unsigned long long tNode, tLeaf, tClosest, tShadow;
    unsigned long long tShade, tTrace; /* primitiveShader (its shadow walks included), launchRayTracing as a whole */
    unsigned int nAdvance, nLeaf;
    /* ... and finer, for the longest tiles (solr_hip_wave_cycle_slots): the primary ray's closest-hit walk, the second
     * attempts of the checked unit-ray walks (time, walks, lanes), the node-loop calls of closest-hit walks by attempt */
    unsigned long long tClosestPrimary, tAgain;
    unsigned int nAgain, nAgainLanes, nAdvanceFirst, nAdvanceAgain, nChecked;
#endif
};
#ifdef SOLR_TIMING
#define SOLR_T(...) __VA_ARGS__
#define SOLR_NOW() __builtin_readcyclecounter()
#else
#define SOLR_T(...)
#endif

/* COUNT: 0 a frame; 1 the ray census (solr_hip_render_counting: every walk in its general form, counted);
 * 2 a frame whose walks are RECORDED for the walk's own ceiling (WalkRecord below) - the frame itself is a frame
 * like any other */
template <int COUNT>
SOLR_DEV void countAdd(unsigned int &c, unsigned int v)
{
    if (COUNT == 1)
        c += v;
}

SOLR_DEV int waveMinInt(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        int o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

#define SOLR_CURSOR_DONE 0x7fffffff

/* ---- the walk's own ceiling ("achieved Mrays/s vs a measured empty-traversal upper bound") ----------
 * A frame rendered with COUNT == 2 leaves, per workgroup, a record of every walk its wave made: which list, and per
 * lane the ray, the cut-off and when the lane dropped out.  k_walkBound (renderer_kernel.h; solr_diag.hip) then replays the frame's walks
 * with NOTHING BUT THE NODE LOOP: the same waves, the same 64 rays together, the same lists, advanceTidy and nothing
 * else - no leaf record, no primitive test, no shading, no camera, no frame buffer.  Shadow walks replay exactly (their
 * cut-off is constant, and a lane leaves after the leaf visit it left after in the frame); a closest-hit walk is
 * replayed with its FINAL cut-off in place from the first node - the fewest nodes any walk that finds that hit can
 * visit.  The replay's time is what the walk structure alone costs this frame: rays / that time is the ceiling.
 * Slot layout: int4 head[SOLR_WALK_SLOTS + 1] (head[0].x = walks recorded; head[1 + j] = {kind, free list?, octant, thin copy? |
 * form of the node loop << 1}),
 * then float4 lanes[SOLR_WALK_SLOTS][64][2] = {origin.xyz, cut-off} {direction.xyz, bits: leaf visit after which the
 * lane is done, 0x7fffffff never, -1 the lane took no part}. */
#define SOLR_WALK_SLOTS 16
#define SOLR_WALK_SLOT_BYTES (16 * (SOLR_WALK_SLOTS + 1) + SOLR_WALK_SLOTS * 64 * 32)
enum WalkKind
{
    WALK_CLOSEST = 0,
    WALK_SHADOW = 1,
    WALK_GENERAL = 2 /* not through the node loop (a list that is not nested, non-finite rays): not replayed */
};
SOLR_DEV void recordWalk(Counters &cnt, int kind, bool freeList, int octant, bool took_part, const WalkRay &r, float cutOff,
                         int doneAfter, bool tight = false, int order = 0)
{
    const unsigned j = cnt.ordinal++;
    if (j >= SOLR_WALK_SLOTS || !cnt.record)
        return;
    const int lane = (int)threadIdx.x & 63;
    float4 *lanes = (float4 *)(cnt.record + 16 * (SOLR_WALK_SLOTS + 1)) + ((size_t)j * 64 + lane) * 2;
    lanes[0] = make_float4(r.o.x, r.o.y, r.o.z, cutOff);
    lanes[1] = make_float4(r.d.x, r.d.y, r.d.z, __int_as_float(took_part ? doneAfter : -1));
    if (lane == 0)
        ((int4 *)cnt.record)[1 + j] = make_int4(kind, freeList ? 1 : 0, octant, (tight ? 1 : 0) | (order << 1));
}

/* What a walk holds of the primitive it is testing.  The first primitive of a leaf comes with the leaf's record
 * (scene_layout.h: one 64-byte scalar load brings head, the two rows its test reads next, and the start
 * index); further primitives of the leaf bring their head and fetch the other rows when a test asks. */
template <int FEAT>
SOLR_DEV bool extendedGeometry(const SceneInfo &si);
struct PrimRec
{
    Row2 head;   /* rows 0-1: p0 | tag, size | materialId */
    float4 c, d; /* packed only: p1 | index and p2 | start - or, plane class, n0 | index and average colour */
    int pi;
    bool packed; /* wave-uniform */
};
SOLR_DEV bool planeClass(int type)
{
    return !(type == ptSphere || type == ptEnvironment || type == ptCylinder || type == ptCone || type == ptEllipsoid ||
             type == ptTriangle);
}
SOLR_DEV v3 recP1(const Scene &S, const PrimRec &r) { return r.packed ? V4(r.c) : V4(primRow(S, r.pi, ROW_P1_INDEX)); }
SOLR_DEV v3 recP2(const Scene &S, const PrimRec &r) { return r.packed ? V4(r.d) : V4(primRow(S, r.pi, ROW_P2)); }
SOLR_DEV v3 recPlaneNormal(const Scene &S, const PrimRec &r) { return r.packed ? V4(r.c) : V4(primRow(S, r.pi, ROW_N0)); }
SOLR_DEV float recPlaneAverage(const Scene &S, const PrimRec &r) { return r.packed ? r.d.x : primRow(S, r.pi, ROW_P2).w; }
SOLR_DEV int recIndex(const Scene &S, const PrimRec &r)
{
    return asint(r.packed ? r.c.w : primRow(S, r.pi, ROW_P1_INDEX).w);
}
/* primitive k of the leaf whose record is L (k == 0) or whose first primitive is `start`.  (Measured, round 4,
 * profiles/r4/leaf_loads.txt: the further primitives' head + p1 + p2 as ONE 64-byte request instead of the head and then
 * the rows the test asks for - mesh + 6 %, molecule + 1.5 % slower; the next primitive's record requested before this
 * one is tested - + 10 ... 18 % on all three scenes.  Sixteen more scalar registers live across a test cost more in
 * lane spills than the request they save: the kernels hold 540 v_readlane / v_writelane already) */
template <int FEAT>
SOLR_DEV PrimRec leafPrimitive(const Scene &S, const SceneInfo &si, const Row4 &L, int start, int k)
{
    PrimRec r;
    r.pi = start + k;
    if (k == 0)
    {
        r.head.a = L.a;
        r.head.b = L.b;
        r.c = L.c;
        r.d = L.d;
    }
    else
    {
        r.head = primHead(S, r.pi);
        r.c = r.d = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    /* without extended geometry every primitive is tested as a triangle (GI:743-747): a plane-class record's
     * packed rows hold its normal and colour, not p1 / p2 */
    r.packed = (k == 0) && (extendedGeometry<FEAT>(si) || !planeClass(asint(r.head.a.w) & PRIM_TYPE_MASK));
    return r;
}

/* Instantiations without F_TRI are only launched with extended geometry (solr_launch.hip, the choice of the
 * instantiation): there the flag is a compile-time fact. */
template <int FEAT>
SOLR_DEV bool extendedGeometry(const SceneInfo &si)
{
    return (FEAT & F_TRI) ? si.extendedGeometry != 0 : true;
}
/* PrimKind of the tag (scene_layout.h), KIND_GENERAL whenever the short paths do not apply */
template <int FEAT>
SOLR_DEV int primKind(const SceneInfo &si, int tag)
{
    const int kind = (int)((unsigned)tag >> PRIM_KIND_SHIFT);
    return extendedGeometry<FEAT>(si) ? kind : (int)KIND_GENERAL;
}
/* the plane test of a KIND_PLANE_* primitive: planeIntersection with the type and the material facts it branches
 * on as constants (no texture, no wireframe pattern, no chessboard light) */
SOLR_DEV bool planePlain(const Scene &S, const SceneInfo &si, int kind, v3 p0, v3 size, v3 n0, float averageColor,
                         const WalkRay &ray, Hit &h)
{
    PlaneMaterial pm;
    pm.wireframe = 0;
    pm.wireframeWidth = 0;
    pm.emissive = false;
    pm.textured = false;
    pm.materialId = 0;
    pm.averageColor = averageColor;
    if (kind == KIND_PLANE_XY)
        return planeIntersection<false>(si, ptXYPlane, p0, size, n0, pm, S, 0, ray, h);
    if (kind == KIND_PLANE_YZ)
        return planeIntersection<false>(si, ptYZPlane, p0, size, n0, pm, S, 0, ray, h);
    return planeIntersection<false>(si, ptXZPlane, p0, size, n0, pm, S, 0, ray, h);
}

/* Uniform primitive test shared by both walks. */
template <bool SHADOW, int FEAT>
SOLR_DEV bool testPrimitive(const Scene &S, const SceneInfo &si, const PrimRec &rec, int tag, const WalkRay &ray, Hit &h)
{
    const int pi = rec.pi;
    const Row2 &head = rec.head;
    const int type = tag & PRIM_TYPE_MASK;
    const v3 p0 = V4(head.a);
    const v3 size = V4(head.b);
    const bool extended = extendedGeometry<FEAT>(si);
    int t = extended ? type : (int)ptTriangle;
    if (SHADOW)
    {
        /* GI:835-864: ptEnvironment is not a sphere here; ptCamera never shadows */
        if (extended && type == ptCamera)
            return false;
        if (extended && type == ptEnvironment)
            t = ptQuad; /* falls to planeIntersection, which has no case for it */
    }
    switch (t)
    {
    case ptEnvironment:
    case ptSphere:
        if (!(FEAT & (F_SPHERE | F_PROC)))
            return false;
        return sphereIntersection(si, p0, size, (FEAT & F_PROC) && (tag & PRIM_PROCEDURAL) != 0,
                                  (tag & PRIM_TRANSPARENT) != 0, ray, h);
    case ptCylinder:
    case ptCone:
    {
        if (!(FEAT & F_CYL))
            return false;
        const v3 p1 = recP1(S, rec);
        const v3 p2 = recP2(S, rec);
        const v3 n1 = V4(primRow(S, pi, ROW_N1));
        return cylinderIntersection(si, p0, p1, p2, n1, size, ray, h);
    }
    case ptEllipsoid:
        if (!(FEAT & F_ELL))
            return false;
        return ellipsoidIntersection(si, p0, size, ray, h);
    case ptTriangle:
    {
        if (!(FEAT & F_TRI))
            return false;
        const v3 p1 = recP1(S, rec);
        const v3 p2 = recP2(S, rec);
        const v3 n0 = V4(primRow(S, pi, ROW_N0));
        const v3 n1 = V4(primRow(S, pi, ROW_N1));
        const v3 n2 = V4(primRow(S, pi, ROW_N2));
        return triangleIntersection(si, p0, p1, p2, n0, n1, n2, ray, h, SHADOW);
    }
    default:
    {
        if (!(FEAT & (F_PLANE | F_TEX)))
            return false;
        const v3 n0 = recPlaneNormal(S, rec);
        const int materialId = asint(head.b.w);
        PlaneMaterial pm;
        pm.wireframe = (tag & PRIM_WIRE2) ? 2 : ((tag & PRIM_WIRE1) ? 1 : 0);
        pm.wireframeWidth = ((tag >> PRIM_WIDTH_SHIFT) & 0xff) - 1;
        pm.emissive = (tag & PRIM_EMISSIVE) != 0;
        pm.textured = (tag & PRIM_TEXTURED) != 0;
        pm.materialId = materialId;
        pm.averageColor = recPlaneAverage(S, rec);
        return planeIntersection<(FEAT & F_TEX) != 0>(si, type, p0, size, n0, pm, S, materialId, ray, h);
    }
    }
}

/* next node for the wave: see the file header */
SOLR_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

SOLR_DEV int nextNode(const Scene &S, int cur, int skip, bool anyEntered, int cursor)
{
    /* readfirstlane pins the result to an SGPR: every access indexed with it
     * becomes a scalar load */
    if (S.nested)
        return uniform(anyEntered ? cur + 1 : cur + skip);
    return uniform(waveMinInt(cursor));
}

/* One node of the general walk: any skip-pointer list that makes forward
 * progress, any bounds, any reciprocal.  Tests node `cur` (already in `node`)
 * for the lanes whose cursor is there, moves the lane cursors, picks the next
 * node for the wave and fetches it.  Returns whether any lane entered. */
template <int COUNT>
SOLR_DEV bool stepGeneral(const Scene &S, const WalkRay &r, bool fastBoxes, float farDistance, int &cursor, int &cur,
                          Row2 &node, int &nbPrimitives, bool &entered, Counters &cnt)
{
    const int nbBoxes = S.nbBoxes;
    const Row2 ahead = boxNode(S, (cur + 1 < nbBoxes) ? cur + 1 : cur);
    const float4 lo = nodeLo(node);
    const float4 hi = nodeHi(node);
    nbPrimitives = uniform(nodeCount(node));
    const int skip = uniform(nodeSkip(node));
    const bool here = (cursor == cur);
    countAdd<COUNT>(cnt.wNodes, 1);
    if (here)
        countAdd<COUNT>(cnt.boxes, 1);
    /* evaluated by every lane: pure arithmetic, no exec-mask bookkeeping */
    bool inBox;
    if (fastBoxes)
        inBox = boxIntersectionFast(lo, hi, r, 0.f, farDistance);
    else
        inBox = boxIntersectionExact(lo, hi, r, 0.f, farDistance);
    entered = here & inBox;
    cursor = entered ? cur + 1 : (here ? cur + skip : cursor);
    const bool anyEntered = ballot(entered) != 0ull;
    const int next = nextNode(S, cur, skip, anyEntered, cursor);
    node = (next == cur + 1) ? ahead : boxNode(S, next < nbBoxes ? next : cur);
    cur = next;
    return anyEntered;
}

/* The tidy walk: skip pointers nested, every node ordered and finite, every
 * lane's origin and reciprocals finite (the conditions of boxIntersectionFast).
 * Hand-scheduled because this loop is where a frame spends its time: it runs
 * from node `cur` to the next leaf that some lane enters and costs 20 vector +
 * 18 scalar instructions per node, about half of what hipcc makes of
 * stepGeneral:
 *   - the node record is one s_load_dwordx8; the record of cur+1 is requested
 *     into the other register bank while cur is tested (two copies of the body
 *     with the banks swapped, so nothing is ever moved), and it is the next node
 *     whenever a lane enters cur or cur is a leaf;
 *   - the slab products are three v_pk_add_f32 + three v_pk_mul_f32 on the
 *     (x,y) (z,z) (x,y) pairs of the record - same IEEE operations as the
 *     scalar form, two per issue slot;
 *   - lane selection is a chain of v_cmpx: cursor == cur narrows exec to the
 *     lanes at this node (they get cur+skip), the three slab conditions narrow it
 *     to the lanes that enter (they get cur+1), exec is the `entered` mask - no
 *     mask algebra, no v_cndmask;
 *   - the wave's next node is one s_cselect on "mask != 0".
 * Hazards (gfx950): every SGPR a vector instruction reads here is written by the
 * scalar unit; v_cmpx results are only read back through s_mov from exec.
 * Fixed registers: s[64:94], v[58:63] (sub-registers of pairs cannot be named
 * through operands).  All scalar loads are drained before the statement ends.
 *
 * Returns the leaf (>= 0) with `entered` lanes and its primitive count, cur
 * already advanced past it, or -1 when the list is exhausted. */
typedef float f2v __attribute__((ext_vector_type(2)));
struct PackedRay
{
    f2v oxy, ozz, ixy, izz;
};
SOLR_DEV PackedRay packRay(const WalkRay &r)
{
    PackedRay p;
    p.oxy = (f2v){r.o.x, r.o.y};
    p.ozz = (f2v){r.o.z, r.o.z};
    p.ixy = (f2v){r.inv.x, r.inv.y};
    p.izz = (f2v){r.inv.z, r.inv.z};
    return p;
}

/* Control flow of one half.  The common case - no lane enters the node - FALLS THROUGH: half A runs into half
 * B, half B branches back to A, so a run of nodes nobody enters costs one taken branch per two nodes (a taken
 * branch empties the wave's instruction buffer; the first form of this loop took two per node).  The record
 * of cur + 1 is requested into the other bank before the test; the list is followed by a pad record so that
 * the request of the last node's successor stays inside the arena.
 *   LW<half>   test the node in this half's bank
 *   LE<half>   some lane entered: a leaf with primitives leaves the loop (LL), an inner node goes on at cur + 1
 *   LR<half>   the wave skipped to a node that is not cur + 1: fetch it into this half's bank, test it here */
#define SOLR_WALK2_HALF(SELF, OTHER, LOXY, ZZ, HIXY, NB, SKIP, SELF_REGS, OTHER_REGS, TAIL)                            \
    "LW" SELF "_%=:\n"                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n"                                                                                           \
    "s_add_i32 s85, %[cur], 1\n"                                                                                       \
    "s_lshl_b32 s84, s85, 5\n"                                                                                         \
    "s_load_dwordx8 " OTHER_REGS ", %[base], s84\n"                                                                    \
    "s_add_i32 s86, %[cur], " SKIP "\n"                                                                                \
    "v_pk_add_f32 v[58:59], " LOXY ", %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"                                            \
    "v_pk_add_f32 v[60:61], " HIXY ", %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"                                            \
    "v_pk_add_f32 v[62:63], " ZZ ", %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n"                                              \
    "v_pk_mul_f32 v[58:59], %[ixy], v[58:59]\n"                                                                    \
    "v_pk_mul_f32 v[60:61], %[ixy], v[60:61]\n"                                                                    \
    "v_pk_mul_f32 v[62:63], %[izz], v[62:63]\n"                                                                    \
    "v_min_f32 %[t], v58, v60\n"                                                                                     \
    "v_max_f32 v58, v58, v60\n"                                                                                     \
    "v_min_f32 v60, v59, v61\n"                                                                                     \
    "v_max_f32 v59, v59, v61\n"                                                                                     \
    "v_min_f32 v61, v62, v63\n"                                                                                     \
    "v_max_f32 v62, v62, v63\n"                                                                                     \
    "v_max3_f32 %[t], %[t], v60, v61\n"                                                                              \
    "v_min3_f32 v58, v58, v59, v62\n"                                                                              \
    "v_cmpx_eq_u32_e32 vcc, %[cur], %[cursor]\n"                                                                       \
    "v_mov_b32 %[cursor], s86\n"                                                                                       \
    "v_cmpx_le_f32_e32 vcc, %[t], v58\n"                                                                              \
    "v_cmpx_lt_f32_e32 vcc, %[t], %[far]\n"                                                                            \
    "v_cmpx_lt_f32_e32 vcc, 0, v58\n"                                                                                 \
    "v_mov_b32 %[cursor], s85\n"                                                                                       \
    "s_cbranch_execnz LE" SELF "_%=\n"                                                                                 \
    "s_mov_b64 exec, s[80:81]\n"                                                                                       \
    "s_mov_b32 %[cur], s86\n"                                                                                          \
    "s_cmp_ge_i32 s86, %[n]\n"                                                                                         \
    "s_cbranch_scc1 LD_%=\n"                                                                                           \
    "s_cmp_lg_u32 s86, s85\n"                                                                                          \
    "s_cbranch_scc1 LR" SELF "_%=\n"                                                                                   \
    TAIL

#define SOLR_WALK2_SIDE(SELF, OTHER, NB, SELF_REGS)                                                                     \
    "LE" SELF "_%=:\n"                                                                                                 \
    "s_mov_b64 s[82:83], exec\n"                                                                                       \
    "s_mov_b64 exec, s[80:81]\n"                                                                                       \
    "s_cmp_gt_i32 " NB ", 0\n"                                                                                         \
    "s_cbranch_scc1 LL" SELF "_%=\n"                                                                                   \
    "s_mov_b32 %[cur], s85\n"                                                                                          \
    "s_cmp_ge_i32 s85, %[n]\n"                                                                                         \
    "s_cbranch_scc1 LD_%=\n"                                                                                           \
    "s_branch LW" OTHER "_%=\n"                                                                                        \
    "LL" SELF "_%=:\n"                                                                                                 \
    "s_mov_b32 %[leaf], %[cur]\n"                                                                                      \
    "s_mov_b32 %[nb], " NB "\n"                                                                                        \
    "s_mov_b32 %[cur], s85\n"                                                                                          \
    "s_branch LX_%=\n"                                                                                                 \
    "LR" SELF "_%=:\n"                                                                                                 \
    "s_lshl_b32 s84, s86, 5\n"                                                                                         \
    "s_load_dwordx8 " SELF_REGS ", %[base], s84\n"                                                                     \
    "s_branch LW" SELF "_%=\n"

SOLR_DEV int advanceTidyShallow(const Scene &S, const PackedRay &p, float farDistance, int &cursor, int &cur,
                         int &nbPrimitives, bool &entered)
{
    const unsigned long base = (unsigned long)S.geo + ((unsigned long)S.offBoxes << 4);
    int leaf, nb, flag;
    float t;
    asm volatile("s_mov_b64 s[80:81], exec\n"
                 "s_lshl_b32 s84, %[cur], 5\n"
                 "s_load_dwordx8 s[64:71], %[base], s84\n"
                 SOLR_WALK2_HALF("A", "B", "s[64:65]", "s[66:67]", "s[68:69]", "s70", "s71", "s[64:71]", "s[72:79]", "")
                 SOLR_WALK2_HALF("B", "A", "s[72:73]", "s[74:75]", "s[76:77]", "s78", "s79", "s[72:79]", "s[64:71]",
                                "s_branch LWA_%=\n")
                 SOLR_WALK2_SIDE("A", "B", "s70", "s[64:71]")
                 SOLR_WALK2_SIDE("B", "A", "s78", "s[72:79]")
                 "LD_%=:\n"
                 "s_mov_b64 s[82:83], 0\n"
                 "s_mov_b32 %[leaf], -1\n"
                 "s_mov_b32 %[nb], 0\n"
                 "LX_%=:\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 "v_cndmask_b32_e64 %[flag], 0, 1, s[82:83]\n"
                 : [cursor] "+v"(cursor), [cur] "+s"(cur), [leaf] "=&s"(leaf), [nb] "=&s"(nb), [flag] "=&v"(flag),
                   [t] "=&v"(t)
                 : [oxy] "v"(p.oxy), [ozz] "v"(p.ozz), [ixy] "v"(p.ixy), [izz] "v"(p.izz), [far] "v"(farDistance),
                   [base] "s"(base), [n] "s"(S.nbBoxes)
                 : "vcc", "scc", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75",
                   "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "v58", "v59",
                   "v60", "v61", "v62", "v63");
    /* (readfirstlane: the compiler takes inline-asm results for divergent; where two such statements meet it would
     * otherwise have to move a "vector" value into the scalar registers the next statement asks for) */
    nbPrimitives = uniform(nb);
    cur = uniform(cur);
    entered = flag != 0;
    return uniform(leaf);
}

/* Control flow.  Three register banks take turns: while the node in one bank is tested, BOTH nodes the wave can
 * go to next are on their way into the other two - cur + 1 (some lane entered an inner node) and cur + skip
 * (nobody entered, or a leaf: skip = 1) - so that a skip over a subtree, which in a deep tree lands on a
 * record far away in the list and out of the scalar cache, has had a node test's time to arrive instead of
 * being waited for from scratch (the two-bank form requested only cur + 1; on the 100k-triangle mesh and the
 * molecule waves sat on s_waitcnt for 45 % of their cycles).  The copies are laid out so that the common case
 * - nobody enters, go to cur + skip - falls through: A -> C -> B -> (branch) A.  The list is followed by a pad
 * record: cur + 1 and cur + skip are at most `n`.
 *   LW<x>   test the node in bank x
 *   LE<x>   some lane entered: a leaf with primitives leaves the loop (LL), an inner node goes on at cur + 1 */
/* ORDER: how the six slab products become the entry and exit parameters.  SOLR_ORDER_ANY orders the two products of every
 * axis by min / max (any record, any ray: eight instructions).  Where the record already holds, per axis, the bound a ray
 * of the wave reaches FIRST in its first slot (a copy of an order-free list sorted for its octant, scene_layout.h) and
 * every lane's direction has that octant's signs, the products are ordered as they come - SOLR_ORDER_SORTED: two
 * instructions, the same values into the same max3 / min3, bit for bit; for rays of the OPPOSITE octant (the shadow
 * walks take the lamp's side first) every axis is the other way round - SOLR_ORDER_REVERSED. */
#define SOLR_ORDER_ANY                                                                                                 \
    "v_min_f32 %[t], v58, v60\n"                                                                                     \
    "v_max_f32 v58, v58, v60\n"                                                                                     \
    "v_min_f32 v60, v59, v61\n"                                                                                     \
    "v_max_f32 v59, v59, v61\n"                                                                                     \
    "v_min_f32 v61, v62, v63\n"                                                                                     \
    "v_max_f32 v62, v62, v63\n"                                                                                     \
    "v_max3_f32 %[t], %[t], v60, v61\n"                                                                              \
    "v_min3_f32 v58, v58, v59, v62\n"
#define SOLR_ORDER_SORTED                                                                                              \
    "v_max3_f32 %[t], v58, v59, v62\n"                                                                              \
    "v_min3_f32 v58, v60, v61, v63\n"
#define SOLR_ORDER_REVERSED                                                                                            \
    "v_max3_f32 %[t], v60, v61, v63\n"                                                                              \
    "v_min3_f32 v58, v58, v59, v62\n"
/* NEXT: how the two successors' records are asked for.  SOLR_NEXT_BY_NUMBER: cursors and skip words count nodes (every
 * list as the builders leave it), the byte offset is a shift away.  SOLR_NEXT_BY_BYTES: in the copies with sorted bounds
 * the skip word holds BYTES (32 x the nodes) and so do the wave's and the lanes' cursors for the length of such a walk:
 * two scalar instructions per node less (11 instead of 13 on a scalar unit that four SIMDs share). */
#define SOLR_NEXT_BY_NUMBER(SKIP, NEXT1_REGS, NEXTS_REGS)                                                              \
    "s_add_i32 s93, %[cur], 1\n"                                                                                       \
    "s_add_i32 s94, %[cur], " SKIP "\n"                                                                                \
    "s_lshl_b32 s92, s93, 5\n"                                                                                         \
    "s_load_dwordx8 " NEXT1_REGS ", %[base], s92\n"                                                                    \
    "s_lshl_b32 s92, s94, 5\n"                                                                                         \
    "s_load_dwordx8 " NEXTS_REGS ", %[base], s92\n"
#define SOLR_NEXT_BY_BYTES(SKIP, NEXT1_REGS, NEXTS_REGS)                                                               \
    "s_add_i32 s93, %[cur], 32\n"                                                                                      \
    "s_add_i32 s94, %[cur], " SKIP "\n"                                                                                \
    "s_load_dwordx8 " NEXT1_REGS ", %[base], s93\n"                                                                    \
    "s_load_dwordx8 " NEXTS_REGS ", %[base], s94\n"
#define SOLR_WALK_BANK(SELF, NEXT1, NEXTS, LOXY, ZZ, HIXY, NB, SKIP, NEXT1_REGS, NEXTS_REGS, TAIL, ORDER, NEXT)       \
    "LW" SELF "_%=:\n"                                                                                                 \
    "s_waitcnt lgkmcnt(0)\n"                                                                                           \
    NEXT(SKIP, NEXT1_REGS, NEXTS_REGS)                                                                                 \
    "v_pk_add_f32 v[58:59], " LOXY ", %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"                                            \
    "v_pk_add_f32 v[60:61], " HIXY ", %[oxy] neg_lo:[0,1] neg_hi:[0,1]\n"                                            \
    "v_pk_add_f32 v[62:63], " ZZ ", %[ozz] neg_lo:[0,1] neg_hi:[0,1]\n"                                              \
    "v_pk_mul_f32 v[58:59], %[ixy], v[58:59]\n"                                                                    \
    "v_pk_mul_f32 v[60:61], %[ixy], v[60:61]\n"                                                                    \
    "v_pk_mul_f32 v[62:63], %[izz], v[62:63]\n"                                                                    \
    ORDER                                                                                                              \
    "v_cmpx_eq_u32_e32 vcc, %[cur], %[cursor]\n"                                                                       \
    "v_mov_b32 %[cursor], s94\n"                                                                                       \
    "v_cmpx_le_f32_e32 vcc, %[t], v58\n"                                                                              \
    "v_cmpx_lt_f32_e32 vcc, %[t], %[far]\n"                                                                            \
    "v_cmpx_lt_f32_e32 vcc, 0, v58\n"                                                                                 \
    "v_mov_b32 %[cursor], s93\n"                                                                                       \
    "s_cbranch_execnz LE" SELF "_%=\n"                                                                                 \
    "s_mov_b64 exec, s[88:89]\n"                                                                                       \
    "s_mov_b32 %[cur], s94\n"                                                                                          \
    "s_cmp_ge_i32 s94, %[n]\n"                                                                                         \
    "s_cbranch_scc1 LD_%=\n"                                                                                           \
    TAIL

#define SOLR_WALK_SIDE(SELF, NEXT1, NB)                                                                                \
    "LE" SELF "_%=:\n"                                                                                                 \
    "s_mov_b64 s[90:91], exec\n"                                                                                       \
    "s_mov_b64 exec, s[88:89]\n"                                                                                       \
    "s_cmp_gt_i32 " NB ", 0\n"                                                                                         \
    "s_cbranch_scc1 LL" SELF "_%=\n"                                                                                   \
    "s_mov_b32 %[cur], s93\n"                                                                                          \
    "s_cmp_ge_i32 s93, %[n]\n"                                                                                         \
    "s_cbranch_scc1 LD_%=\n"                                                                                           \
    "s_branch LW" NEXT1 "_%=\n"                                                                                        \
    "LL" SELF "_%=:\n"                                                                                                 \
    "s_mov_b32 %[leaf], %[cur]\n"                                                                                      \
    "s_mov_b32 %[nb], " NB "\n"                                                                                        \
    "s_mov_b32 %[cur], s93\n"                                                                                          \
    "s_branch LX_%=\n"

#define SOLR_ADVANCE_TIDY_DEEP(NAME, ORDER, NEXT, ENTRY, S92)                                                          \
    SOLR_DEV int NAME(const Scene &S, const PackedRay &p, float farDistance, int &cursor, int &cur, int &nbPrimitives,  \
                      bool &entered)                                                                                   \
    {                                                                                                                  \
        /* (through readfirstlane, which folds away where the compiler knows the address to be uniform: the statement     \
         * asks for it in scalar registers) */                                                                         \
        const unsigned long at = (unsigned long)S.geo + ((unsigned long)S.offBoxes << 4);                              \
        const unsigned long base = ((unsigned long)(unsigned)uniform((int)(at >> 32)) << 32) |                         \
                                   (unsigned long)(unsigned)uniform((int)at);                                          \
        int leaf, nb, flag;                                                                                            \
        float t;                                                                                                       \
        /* banks: A = s[64:71], B = s[72:79], C = s[80:87]; a node's successors go: A -> (cur+1: B, cur+skip: C),     \
         * C -> (A, B), B -> (C, A) */                                                                                 \
        asm volatile("s_mov_b64 s[88:89], exec\n"                                                                     \
                     ENTRY                                                                                             \
                     SOLR_WALK_BANK("A", "B", "C", "s[64:65]", "s[66:67]", "s[68:69]", "s70", "s71", "s[72:79]",     \
                                    "s[80:87]", "", ORDER, NEXT)                                                       \
                     SOLR_WALK_BANK("C", "A", "B", "s[80:81]", "s[82:83]", "s[84:85]", "s86", "s87", "s[64:71]",     \
                                    "s[72:79]", "", ORDER, NEXT)                                                       \
                     SOLR_WALK_BANK("B", "C", "A", "s[72:73]", "s[74:75]", "s[76:77]", "s78", "s79", "s[80:87]",     \
                                    "s[64:71]", "s_branch LWA_%=\n", ORDER, NEXT)                                     \
                     SOLR_WALK_SIDE("A", "B", "s70")                                                                   \
                     SOLR_WALK_SIDE("C", "A", "s86")                                                                   \
                     SOLR_WALK_SIDE("B", "C", "s78")                                                                   \
                     "LD_%=:\n"                                                                                       \
                     "s_mov_b64 s[90:91], 0\n"                                                                        \
                     "s_mov_b32 %[leaf], -1\n"                                                                        \
                     "s_mov_b32 %[nb], 0\n"                                                                           \
                     "LX_%=:\n"                                                                                       \
                     "s_waitcnt lgkmcnt(0)\n"                                                                         \
                     "v_cndmask_b32_e64 %[flag], 0, 1, s[90:91]\n"                                                    \
                     : [cursor] "+v"(cursor), [cur] "+s"(cur), [leaf] "=&s"(leaf), [nb] "=&s"(nb), [flag] "=&v"(flag), \
                       [t] "=&v"(t)                                                                                    \
                     : [oxy] "v"(p.oxy), [ozz] "v"(p.ozz), [ixy] "v"(p.ixy), [izz] "v"(p.izz), [far] "v"(farDistance), \
                       [base] "s"(base), [n] "s"(S.nbBoxes)                                                            \
                     : "vcc", "scc", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74",      \
                       "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87",      \
                       "s88", "s89", "s90", "s91", S92 "s93", "s94", "v58", "v59", "v60", "v61", "v62", "v63");        \
        /* (readfirstlane: the compiler takes inline-asm results for divergent; where two such statements meet it    \
         * would otherwise have to move a "vector" value into the scalar registers the next statement asks for) */    \
        nbPrimitives = uniform(nb);                                                                                    \
        cur = uniform(cur);                                                                                            \
        entered = flag != 0;                                                                                           \
        return uniform(leaf);                                                                                          \
    }
#define SOLR_CLOBBER_S92 "s92",
#define SOLR_CLOBBER_NO_S92
#define SOLR_ENTRY_BY_NUMBER "s_lshl_b32 s92, %[cur], 5\n" "s_load_dwordx8 s[64:71], %[base], s92\n"
#define SOLR_ENTRY_BY_BYTES "s_load_dwordx8 s[64:71], %[base], %[cur]\n"
SOLR_ADVANCE_TIDY_DEEP(advanceTidyDeep, SOLR_ORDER_ANY, SOLR_NEXT_BY_NUMBER, SOLR_ENTRY_BY_NUMBER, SOLR_CLOBBER_S92)
/* the same loop over a copy of an order-free list whose bounds are sorted for the octant of every ray of the wave, and
 * for the opposite octant (SOLR_ORDER_* above): 14 vector instructions per node instead of 20, 11 scalar ones instead of
 * 13.  cursor, cur, S.nbBoxes and the leaf they return are in BYTES (32 x the node's number, SOLR_NEXT_BY_BYTES). */
SOLR_ADVANCE_TIDY_DEEP(advanceTidyDeepSorted, SOLR_ORDER_SORTED, SOLR_NEXT_BY_BYTES, SOLR_ENTRY_BY_BYTES, SOLR_CLOBBER_NO_S92)
SOLR_ADVANCE_TIDY_DEEP(advanceTidyDeepReversed, SOLR_ORDER_REVERSED, SOLR_NEXT_BY_BYTES, SOLR_ENTRY_BY_BYTES, SOLR_CLOBBER_NO_S92)

/* Which loop: the three-bank form pays one more request and two more scalar instructions per node, and seven more
 * reserved scalar registers, for having the skip target on its way.  That wins where skips are frequent and
 * land far away - deep trees over megabytes of nodes (mesh 0.594 -> 0.557 ms, molecule 0.951 -> 0.921) - and
 * loses where the list is a short run of leaves that lives in the scalar cache (Cornell's 34 nodes: 0.343 ->
 * 0.355 ms; with both loops in one kernel 0.357: it is the register reservation that costs).  So it is a
 * compile-time property of the instantiation (F_DEEP) and the host launches the one that fits the list. */
template <int FEAT>
SOLR_DEV int advanceTidy(const Scene &S, const PackedRay &p, float farDistance, int &cursor, int &cur,
                         int &nbPrimitives, bool &entered)
{
    if (FEAT & F_DEEP)
        return advanceTidyDeep(S, p, farDistance, cursor, cur, nbPrimitives, entered);
    return advanceTidyShallow(S, p, farDistance, cursor, cur, nbPrimitives, entered);
}

/* the conditions of boxIntersectionFast that depend on the ray */
SOLR_DEV bool finiteRay(const WalkRay &r)
{
    const float big = 3.0e38f;
    return fabsf(r.inv.x) < big && fabsf(r.inv.y) < big && fabsf(r.inv.z) < big && fabsf(r.o.x) < big &&
           fabsf(r.o.y) < big && fabsf(r.o.z) < big;
}

/* GI:667-772, wave-synchronous.  `active` lanes trace origin -> target. */
/* |direction| >= 2 (and not so long that the margins of the order-free cut-off would underflow): see closestHitWalk */
SOLR_DEV bool longRay(v3 d)
{
    const float dd = dot(d, d);
    return dd >= 4.f && dd <= 1.0e24f;
}

/* TIGHT LEAVES.  The reference's builder gives a plane the box p0 +- size in all three axes (GPUKernel.cpp:762-830) - a
 * wall of the Cornell room, size (w, h, d), gets half the room - so every ray of that frame entered all six walls'
 * leaves and made six plane tests (two divisions each) to find the one wall it can reach: 19 leaf entries per pixel,
 * 15 of them walls.  A test that misses has no effect on a walk (what the order-free cut-off already relies on), so a
 * leaf need only be entered by the rays that can HIT one of its primitives.  Behind the walk-order list and the
 * order-free lists lies a copy of their node rows (solr_scene.hip tightenList) in which a leaf that holds nothing but
 * plain axis planes (KIND_PLANE_*: the test is `(front || rear) && |I.u - p0.u| < size.u && |I.v - p0.v| < size.v`,
 * GI:424-567) is the planes' rectangle, a margin m thick and m wider, cut with the reference's box; inner nodes are the
 * unions of their leaves.  A ray that hits such a plane crosses the rectangle's plane at t* > 0 inside the rectangle,
 * so all three slab intervals of the thin box contain t* - up to rounding, which the margin covers: every quantity of
 * the plane test and of the slab test is bounded by M = |o| + |p0| + |size| (the hit point lies inside the rectangle),
 * each operation loses at most 2^-24 of that, a handful of operations - against m = 2^-10 of the scene's extent E with
 * |o| <= viewDistance <= 64 E demanded below (M <= 66 E: a slack of some 2^8).  Demanded of the ray, because the
 * argument needs them: no zero direction component (the reference gives a zero component the reciprocal 1, GI:39-41,
 * and its slab test on ITS boxes then decides by values that mean nothing - reproducible only with its boxes);
 * |direction| >= 2, so that the reference's cut-off - slab parameter against closest DISTANCE - cannot hide a box the
 * hit lies in (for the bounce rays, |direction| = 0.95, which box is entered FIRST decides: the thin copy is not for
 * them); the origin within viewDistance of zero.  The host offers the copy (S.tightLists) only with extended geometry
 * (without it a plane record is tested as a triangle) and viewDistance <= 64 E.  Same frames bit for bit
 * (solr_hip_set_variant(8) walks the reference's boxes; tests/test_gpu_parity.py), 0.2865 -> see DESIGN.md section 5. */
SOLR_DEV bool tightRay(const WalkRay &r, const SceneInfo &si)
{
    return longRay(r.d) && r.d.x != 0.f && r.d.y != 0.f && r.d.z != 0.f && fabsf(r.o.x) <= si.viewDistance &&
           fabsf(r.o.y) <= si.viewDistance && fabsf(r.o.z) <= si.viewDistance;
}

template <int COUNT, int FEAT>
SOLR_DEV bool closestHitWalk(const Scene &S, const SceneInfo &si, bool active, v3 origin, v3 target, int iteration,
                             int currentMaterialId, int &closestPrimitive, v3 &closestIntersection,
                             v3 &closestNormal, v3 &closestAreas, v3 &colorBox, Counters &cnt)
{
    bool intersections = false;
    if (ballot(active) == 0ull)
        return false;
    SOLR_T(const unsigned long long tw0 = SOLR_NOW();)
    float minDistance = (iteration < 2) ? si.viewDistance : si.viewDistance / (iteration + 1);
    const WalkRay r = makeWalkRay(origin, target - origin);
    if (active)
        countAdd<COUNT>(cnt.closest, 1);
    countAdd<COUNT>(cnt.wClosest, 1);

    /* sign-free slab test when it is provably identical (see boxIntersectionFast) */
    const bool fastBoxes = S.orderedBoxes && (ballot(active && !finiteRay(r)) == 0ull);
    const bool showBoxes = (FEAT & F_FULL) && si.renderBoxes != 0;
    const bool tidy = COUNT != 1 && fastBoxes && S.nested && !showBoxes;
    /* The order-free list (solr_scene.hip, buildFreeOrderLists): the same leaves under a hierarchy of our own, in an
     * order of our own.  The reference's result does not depend on the order in which the leaves are visited when
     * (a) ties go to the smaller flattened index - the primitive the reference visits first - and (b) the cut-off
     * `slab parameter < closest distance so far` never hides a nearer hit: the parameter is in units of the
     * un-normalised direction, so for |direction| >= 2 it is at most half the Euclidean distance to anything
     * inside the box, and every primitive of such a scene lies inside its leaf's box (checked at upload).  Primary
     * rays qualify (|direction| is thousands); bounce rays are unit vectors and walk the reference's order. */
    const bool freeOrder =
        tidy && S.nbBoxesFree > 0 && ballot(active && !longRay(r.d)) == 0ull;
    /* the thin copy of the list this walk takes (tightRay above) */
    constexpr bool thinLeaves = (FEAT & F_PLANE) != 0; /* (thin copies exist for leaves of plain axis planes only) */
    const bool tight = thinLeaves && tidy && S.tightLists && ballot(active && !tightRay(r, si)) == 0ull;
    /* ... and the SHORT rays (the bounce rays, |direction| = L = 1 - rayEpsilon) in the reference's order: for them the
     * reference's cut-off - slab parameter t against closest DISTANCE - decides by WHEN a box is entered, and a thin
     * leaf is entered later than the reference's fat one.  The reference tests the primitives of leaf l iff its own box
     * passes [0, closest so far); so does this walk, literally: it runs over the thin copy with the cut-off widened to
     * far' = 1.002 closest / min(L, 1) + slack (a hit at distance D < closest has t* = D / L < far': its thin leaf is
     * entered), and at every leaf it enters it makes the reference's own test on the reference's own box with the
     * reference's cut-off (`fatCheck` below).  What it skips is a leaf the reference enters and the thin test does not:
     * no ray-rectangle crossing at all, or one at t* >= far' - a distance of at least 1.001 closest, which the
     * reference rejects (GI:751) - and a test that misses leaves no trace.  Leaves that are no thinner than the
     * reference's pass the thin test whenever they pass the reference's (far' >= far).  Same order, same tests that
     * count, same closest-so-far after every leaf: by induction the same result. */
    const float ddShort = dot(r.d, r.d);
    const bool shortTightLane = ddShort >= 0.25f && ddShort < 4.f && r.d.x != 0.f && r.d.y != 0.f && r.d.z != 0.f &&
                                fabsf(r.o.x) <= si.viewDistance && fabsf(r.o.y) <= si.viewDistance &&
                                fabsf(r.o.z) <= si.viewDistance;
    const bool tightShort = thinLeaves && tidy && S.tightLists && !tight && !freeOrder && !(COUNT != 1 && (FEAT & F_DEEP) && (FEAT & F_TRI)) &&
                            ballot(active && !shortTightLane) == 0ull;
    const float shortFarScale = 1.002f / fminf(sqrtf(ddShort) * (1.f - 1.0e-5f), 1.f);
    const float shortFarOffset = 2.0e-4f * (fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z));
    /* Short rays (the bounce rays: |direction| = 1 - rayEpsilon, CudaRayTracer.cu:322-323) in the order-free lists,
     * CHECKED.  The reference culls a box when its slab parameter t = (entry distance) / |direction| reaches the closest
     * DISTANCE so far.  For |direction| = L <= 1 that hides boxes whose entry lies between L x and 1 x the closest
     * distance - so which primitive wins CAN depend on the order of the leaves, but only like this: the true nearest
     * hit P (distance D, its leaf entered at e <= D) is passed over iff a farther hit Q was found before it with
     * D_Q <= e / L <= D / L (the same holds for every inner box around P's leaf, entered earlier still).  So the walk
     * runs order-free, sees every hit up to 1.001 / L times the closest so far (the cut-off below is widened by that
     * much), keeps the second smallest distance it met, and afterwards the lanes whose best hit has such a RIVAL - a
     * second hit within D / L, rounding included - are walked again in the reference's order.  For every other lane
     * nothing the reference can have found earlier hides P's boxes, and no hit is nearer: the reference returns P
     * (equal distances: the smaller flattened index, as above).  The same band sits under the initial bound B: a lone
     * hit at D < L x B has e < L x B and is visited; at D >= L x B it may or may not be - such hits are not accepted
     * here and send the lane to the second walk too (it finds a hit there if it found one here: a hit outside the band
     * passes the reference's cut-off against the initial bound, so nothing needs restoring).  For 1 < L < 2 the
     * reference culls less than the distance would and the margins are those of rounding alone.  Only in the
     * long-list triangle instantiations (the mesh: the bounce walks are 45 % of its longest tiles); compiled into
     * every kernel it costs the Cornell box 3 % and the molecule 0.7 % (short lists, few bounce rays, the bookkeeping
     * in every accept). */
    constexpr bool CHECKED_BUILD = COUNT != 1 && (FEAT & F_DEEP) != 0 && (FEAT & F_TRI) != 0;
    const float dd = dot(r.d, r.d);
    /* (S.shortRayLists: the host's choice, solr_scene.hip shortRayListsChoice - the repeated lanes lengthen a frame's longest tiles, so a frame
     * that is as long as its longest tile keeps the reference's order for such rays; either way the same bits.  Rays
     * within rounding of length 1 need no more than the rounding margins and always qualify.) */
    const float shortest = S.shortRayLists ? 0.25f : 0.9998f;
    const bool unitRays = CHECKED_BUILD && tidy && S.nbBoxesFree > 0 && !freeOrder &&
                          ballot(active && !(dd >= shortest && dd < 4.f)) == 0ull;
    const float initialDistance = minDistance;
    const float slack = 1.0e-4f * (fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z));
    /* min(L, 1), a little under: the band may be wider than it need be, never narrower */
    const float shortBy = unitRays ? fminf(sqrtf(dd) * (1.f - 1.0e-5f), 1.f) : 1.f;
    const float rivalScale = 1.001f / shortBy;
    const float bandStart = initialDistance * shortBy * (1.f - 2.0e-3f) - 2.f * slack;
    float second = INFINITY;
    bool bandHit = false;
    bool lanesNow = active;
    for (int attempt = 0; attempt < 2; ++attempt)
    {
        const bool checked = unitRays && attempt == 0;
        const bool freeList = freeOrder || checked;
        SOLR_T(const unsigned long long tAttempt = SOLR_NOW();)
        Scene W = S;
        int octant = 0;
        int order = 0; /* the form of the node loop: 0 any record and ray, 1 sorted bounds (SOLR_ORDER_SORTED) */
        if (freeList)
        {
            /* eight flattenings of the same hierarchy, the near child first for a direction of that sign octant: the
             * wave takes the octant of its first active lane (any list gives the same result) */
            const int signs = (r.d.x < 0.f ? 1 : 0) | (r.d.y < 0.f ? 2 : 0) | (r.d.z < 0.f ? 4 : 0);
            const int lane = (int)__builtin_ctzll(ballot(lanesNow));
            octant = __builtin_amdgcn_readlane(signs, lane);
            W.offBoxes = S.offBoxesFree + 2u * (unsigned)(octant * S.nbBoxesFree);
            W.offLeaf = S.offLeafFree + 4u * (unsigned)(octant * S.nbBoxesFree);
            W.nbBoxes = S.nbBoxesFree;
            /* every ray of the wave points into the list's octant (a tile of primary rays, but for the tiles the
             * camera's axes run through): the copy with sorted bounds, the node loop without its six min / max */
            if ((FEAT & F_DEEP) && S.sortedLists && !tight && ballot(lanesNow && signs != octant) == 0ull)
            {
                order = 1;
                W.offBoxes += 32u * (unsigned)S.nbBoxesFree + 4u;
                W.nbBoxes = S.nbBoxesFree << 5; /* (this walk's cursors count bytes: SOLR_NEXT_BY_BYTES) */
            }
        }
        if (tight) /* the same nodes, leaf records and start indices: only the bounds differ */
            W.offBoxes += freeList ? 16u * (unsigned)S.nbBoxesFree + 2u : 2u * (unsigned)S.nbBoxes + 2u;
        const bool fatCheck = tightShort && !freeList;
        if (fatCheck)
            W.offBoxes += 2u * (unsigned)S.nbBoxes + 2u;
        /* The reference's cut-off never culls for such rays (a slab parameter of order 1 against a distance of
         * thousands).  The order-free walk may cull by the TRUE distance: a box whose entry point lies farther than the
         * closest hit so far holds nothing that could replace it, not even on a tie.  The margins cover the rounding
         * of both sides: 2e-4 of the distance for the computed hit distance and the products, 1e-4 of the origin's
         * coordinates for the cancellation in (bound - origin) - a thousand times the half-ulp that subtraction can
         * lose.  With the near child first this is what ends a walk early. */
        const float invLength = freeList ? 1.f / length(r.d) : 1.f;
        const float farScale = freeList ? (checked ? rivalScale : 1.0002f) * invLength : 1.f;
        const float farOffset = freeList ? slack * invLength : 0.f;
        int tieIndex = -1; /* the primitive that holds minDistance */
        auto closer = [&](float distance, int pi) {
            bool better = distance < minDistance || (freeOrder && distance == minDistance && pi < tieIndex);
            if (checked)
            {
                const bool inBand = distance >= bandStart;
                bandHit = bandHit || inBand;
                better = better && !inBand;
                second = fminf(second, better ? minDistance : distance);
            }
            return better;
        };
        const PackedRay pr = packRay(r);
        const int nbBoxes = W.nbBoxes;
        int cursor = lanesNow ? 0 : SOLR_CURSOR_DONE;
        int cur = 0;
        Row2 node;
        node.a = node.b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!tidy)
            node = boxNode(S, 0);
        while (cur < nbBoxes)
        {
            int leaf = cur, nbPrimitives;
            bool entered;
            SOLR_T(unsigned long long ta = SOLR_NOW();)
            if (tidy)
            {
                const float far = freeList ? minDistance * farScale + farOffset
                                           : (fatCheck ? minDistance * shortFarScale + shortFarOffset : minDistance);
                if ((FEAT & F_DEEP) && order == 1)
                {
                    leaf = advanceTidyDeepSorted(W, pr, far, cursor, cur, nbPrimitives, entered);
                    leaf = leaf < 0 ? leaf : leaf >> 5;
                }
                else
                    leaf = advanceTidy<FEAT>(W, pr, far, cursor, cur, nbPrimitives, entered);
                SOLR_T(const unsigned long long tb = SOLR_NOW(); cnt.tNode += tb - ta; ++cnt.nAdvance; ta = tb;
                       if (attempt == 0) ++cnt.nAdvanceFirst; else ++cnt.nAdvanceAgain;)
                if (leaf < 0)
                    break;
                if (fatCheck)
                {
                    /* the reference's own entry test of this leaf: its box, its cut-off (see tightShort) */
                    const Row2 fat = boxNode(S, uniform(leaf));
                    entered = entered && boxIntersectionFast(nodeLo(fat), nodeHi(fat), r, 0.f, minDistance);
                    if (ballot(entered) == 0ull)
                        continue;
                }
            }
            else
            {
                if (!stepGeneral<COUNT>(S, r, fastBoxes, minDistance, cursor, cur, node, nbPrimitives, entered, cnt))
                    continue;
                if (showBoxes)
                {
                    if (entered)
                    {
                        const int start = boxStart(S, leaf);
                        const float4 c = loadMaterialHot(S, (int)((unsigned)start % (unsigned)NB_MAX_MATERIALS)).color;
                        colorBox.x += c.x / 200.f;
                        colorBox.y += c.y / 200.f;
                        colorBox.z += c.z / 200.f;
                    }
                    continue;
                }
                if (nbPrimitives <= 0)
                    continue;
            }
            /* (readfirstlane: the compiler takes an inline-asm result for divergent and would fetch the record per
             * lane) */
            const Row4 L = leafRecord(W, uniform(leaf));
            const int start = uniform(asint(L.d.w));
            for (int k = 0; k < nbPrimitives; ++k)
            {
                const PrimRec rec = leafPrimitive<FEAT>(S, si, L, start, k);
                const int pi = rec.pi;
                const Row2 &head = rec.head;
                const int tag = uniform(asint(head.a.w));
                const int materialId = uniform(asint(head.b.w));
                const int kind = primKind<FEAT>(si, tag);
                /* Short paths.  A primitive with a kind has a FAST0 material (GI:704-705: every lane that entered the
                 * leaf tests it) and its type and material facts are settled: the general tests with constants. */
                if ((FEAT & F_SPHERE) && kind == KIND_SPHERE)
                {
                    countAdd<COUNT>(cnt.wPrims, 1);
                    if (entered)
                        countAdd<COUNT>(cnt.prims, 1);
                    Hit h;
                    bool back;
                    const bool i = sphereHit(si, V4(head.a), head.b.x, r, entered, h.intersection, back);
                    if (ballot(i) == 0ull)
                        continue; /* nobody hit it: no distance to take (a square root) */
                    const float distance = length(h.intersection - r.o);
                    const bool keep = i && distance > si.geometryEpsilon && closer(distance, pi);
                    if (ballot(keep) != 0ull)
                    {
                        if (keep)
                        {
                            sphereNormal<false>(si, V4(head.a), V4(head.b), false, false, back, r, h);
                            minDistance = distance;
                            tieIndex = pi;
                            closestPrimitive = pi;
                            closestIntersection = h.intersection;
                            closestNormal = h.normal;
                            closestAreas = V(0.f, 0.f, 0.f);
                            intersections = true;
                        }
                    }
                    continue;
                }
                if ((FEAT & F_PLANE) && kind >= KIND_PLANE_XY && kind <= KIND_PLANE_XZ)
                {
                    countAdd<COUNT>(cnt.wPrims, 1);
                    if (entered)
                    {
                        countAdd<COUNT>(cnt.prims, 1);
                        Hit h;
                        h.intersection = V(0.f, 0.f, 0.f);
                        h.normal = V(0.f, 0.f, 0.f);
                        h.shadowIntensity = 0.f;
                        const bool i = planePlain(S, si, kind, V4(head.a), V4(head.b), recPlaneNormal(S, rec),
                                                  recPlaneAverage(S, rec), r, h);
                        float distance = 0.f;
                        if (i)
                            distance = length(h.intersection - r.o); /* (skipped by the wave when nobody hit) */
                        if (i && distance > si.geometryEpsilon && closer(distance, pi))
                        {
                            minDistance = distance;
                            tieIndex = pi;
                            closestPrimitive = pi;
                            closestIntersection = h.intersection;
                            closestNormal = h.normal;
                            closestAreas = V(0.f, 0.f, 0.f);
                            intersections = true;
                        }
                    }
                    continue;
                }
                if ((FEAT & F_TRI) && kind == KIND_TRIANGLE)
                {
                    /* as the general triangle branch below */
                    countAdd<COUNT>(cnt.wPrims, 1);
                    if (entered)
                        countAdd<COUNT>(cnt.prims, 1);
                    Hit h;
                    h.intersection = V(0.f, 0.f, 0.f);
                    bool i = false;
                    const v3 p0 = V4(head.a);
                    const v3 p1 = recP1(S, rec);
                    const v3 p2 = recP2(S, rec);
                    if (entered)
                        i = triangleHit(si, p0, p1, p2, r, h.intersection);
                    if (ballot(i) == 0ull)
                        continue;
                    const float distance = length(h.intersection - r.o);
                    const bool keep = i && distance > si.geometryEpsilon && closer(distance, pi);
                    if (ballot(keep) != 0ull)
                    {
                        const v3 n0 = V4(primRow(S, pi, ROW_N0));
                        const v3 n1 = V4(primRow(S, pi, ROW_N1));
                        const v3 n2 = V4(primRow(S, pi, ROW_N2));
                        if (keep)
                        {
                            triangleNormal(p0, p1, p2, n0, n1, n2, r, h);
                            minDistance = distance;
                            tieIndex = pi;
                            closestPrimitive = pi;
                            closestIntersection = h.intersection;
                            closestNormal = h.normal;
                            closestAreas = h.areas;
                            intersections = true;
                        }
                    }
                    continue;
                }
                if ((FEAT & F_CYL) && kind == KIND_CYLINDER)
                {
                    countAdd<COUNT>(cnt.wPrims, 1);
                    if (entered)
                    {
                        countAdd<COUNT>(cnt.prims, 1);
                        Hit h;
                        h.intersection = V(0.f, 0.f, 0.f);
                        h.normal = V(0.f, 0.f, 0.f);
                        h.shadowIntensity = 0.f;
                        const bool i = cylinderIntersection(si, V4(head.a), recP1(S, rec), recP2(S, rec),
                                                            V4(primRow(S, pi, ROW_N1)), V4(head.b), r, h);
                        float distance = 0.f;
                        if (i)
                            distance = length(h.intersection - r.o); /* (skipped by the wave when nobody hit) */
                        if (i && distance > si.geometryEpsilon && closer(distance, pi))
                        {
                            minDistance = distance;
                            tieIndex = pi;
                            closestPrimitive = pi;
                            closestIntersection = h.intersection;
                            closestNormal = h.normal;
                            closestAreas = V(0.f, 0.f, 0.f);
                            intersections = true;
                        }
                    }
                    continue;
                }
                /* GI:704-705 */
                const bool lanes = entered && ((tag & PRIM_FAST0) != 0 ||
                                               ((tag & PRIM_FAST1) != 0 && currentMaterialId != materialId));
                if (ballot(lanes) == 0ull)
                    continue;
                countAdd<COUNT>(cnt.wPrims, 1);
                if (lanes)
                    countAdd<COUNT>(cnt.prims, 1);
                const int type = tag & PRIM_TYPE_MASK;
                if ((FEAT & (F_SPHERE | F_PROC)) && extendedGeometry<FEAT>(si) &&
                    (type == ptSphere || type == ptEnvironment))
                {
                    /* spheres: decide on the intersection point, pay for the normal only when
                     * the hit becomes the closest one (GI:749-760 uses nothing else before) */
                    Hit h;
                    bool back;
                    const bool i = sphereHit(si, V4(head.a), head.b.x, r, lanes, h.intersection, back);
                    float distance = 0.f;
                    if (i)
                        distance = length(h.intersection - r.o);
                    const bool keep = i && distance > si.geometryEpsilon && closer(distance, pi);
                    if (ballot(keep) != 0ull)
                    {
                        if (keep)
                        {
                            sphereNormal<(FEAT & F_PROC) != 0>(si, V4(head.a), V4(head.b), (tag & PRIM_PROCEDURAL) != 0,
                                                               false, back, r, h);
                            minDistance = distance;
                            tieIndex = pi;
                            closestPrimitive = pi;
                            closestIntersection = h.intersection;
                            closestNormal = h.normal;
                            closestAreas = V(0.f, 0.f, 0.f);
                            intersections = true;
                        }
                    }
                }
                else if ((FEAT & F_TRI) && (type == ptTriangle || !extendedGeometry<FEAT>(si)))
                {
                    /* triangles (every primitive, without extended geometry, GI:743-747): areas and the
                     * interpolated normal only for hits that become the closest one */
                    Hit h;
                    h.intersection = V(0.f, 0.f, 0.f);
                    bool i = false;
                    const v3 p0 = V4(head.a);
                    const v3 p1 = recP1(S, rec);
                    const v3 p2 = recP2(S, rec);
                    if (lanes)
                        i = triangleHit(si, p0, p1, p2, r, h.intersection);
                    float distance = 0.f;
                    if (i)
                        distance = length(h.intersection - r.o);
                    const bool keep = i && distance > si.geometryEpsilon && closer(distance, pi);
                    if (ballot(keep) != 0ull)
                    {
                        const v3 n0 = V4(primRow(S, pi, ROW_N0));
                        const v3 n1 = V4(primRow(S, pi, ROW_N1));
                        const v3 n2 = V4(primRow(S, pi, ROW_N2));
                        if (keep)
                        {
                            triangleNormal(p0, p1, p2, n0, n1, n2, r, h);
                            minDistance = distance;
                            tieIndex = pi;
                            closestPrimitive = pi;
                            closestIntersection = h.intersection;
                            closestNormal = h.normal;
                            closestAreas = h.areas;
                            intersections = true;
                        }
                    }
                }
                else if (lanes)
                {
                    Hit h;
                    h.intersection = V(0.f, 0.f, 0.f);
                    h.normal = V(0.f, 0.f, 0.f);
                    h.areas = V(0.f, 0.f, 0.f);
                    h.shadowIntensity = 0.f;
                    const bool i = testPrimitive<false, FEAT>(S, si, rec, tag, r, h);
                    float distance = 0.f;
                    if (i)
                        distance = length(h.intersection - r.o);
                    if (i && distance > si.geometryEpsilon && closer(distance, pi))
                    {
                        minDistance = distance;
                        tieIndex = pi;
                        closestPrimitive = pi;
                        closestIntersection = h.intersection;
                        closestNormal = h.normal;
                        closestAreas = h.areas;
                        intersections = true;
                    }
                }
            }
            SOLR_T(cnt.tLeaf += SOLR_NOW() - ta; ++cnt.nLeaf;)
        }
        if (COUNT == 2) /* this attempt, for the walk's own ceiling: the cut-off it ended with */
            recordWalk(cnt, tidy ? WALK_CLOSEST : WALK_GENERAL, freeList, octant, lanesNow, r,
                       freeList ? minDistance * farScale + farOffset
                                : (fatCheck ? minDistance * shortFarScale + shortFarOffset : minDistance),
                       SOLR_CURSOR_DONE, tight || fatCheck, order);
        SOLR_T(if (attempt == 1) cnt.tAgain += SOLR_NOW() - tAttempt;)
        if (!checked)
            break;
        SOLR_T(++cnt.nChecked;)
        /* the lanes whose result the order could have decided: once more, in the reference's order */
        const bool again = lanesNow && (intersections ? !(second > minDistance * rivalScale + slack) : bandHit);
        if (ballot(again) == 0ull)
            break;
        SOLR_T(++cnt.nAgain; cnt.nAgainLanes += (unsigned)__builtin_popcountll(ballot(again));)
        lanesNow = again;
        minDistance = again ? initialDistance : minDistance;
        intersections = again ? false : intersections;
    }
    SOLR_T(cnt.tClosest += SOLR_NOW() - tw0; if (iteration == 0) cnt.tClosestPrimary += SOLR_NOW() - tw0;)
    return intersections;
}

/* GI:798-908, wave-synchronous.  objectId is the flattened index of the
 * shaded primitive, compared with Primitive.index like the reference does. */
template <int COUNT, int FEAT>
SOLR_DEV float shadowWalk(const Scene &S, const SceneInfo &si, bool active, v3 lampCenter, v3 origin, int lightId,
                          int iteration, v3 &color, int objectId, Counters &cnt)
{
    float result = 0.f;
    color = V(0.f, 0.f, 0.f);
    if (ballot(active) == 0ull)
        return 0.f;
    SOLR_T(const unsigned long long tw0 = SOLR_NOW();)
    WalkRay r = makeWalkRay(origin, lampCenter - origin);
    r.o = origin + r.dn * si.rayEpsilon; /* GI:810-811 */
    const float minDistance = (iteration < 2) ? si.viewDistance : si.viewDistance / (iteration + 1);
    const float lengthOL = length(r.d);
    if (active)
        countAdd<COUNT>(cnt.shadow, 1);
    countAdd<COUNT>(cnt.wShadow, 1);

    const bool fastBoxes = S.orderedBoxes && (ballot(active && !finiteRay(r)) == 0ull);
    const bool tidy = COUNT != 1 && fastBoxes && S.nested;
    /* Order-free shadows.  Where no primitive of the scene is transparent (or a textured plane) the first occluder
     * between the point and the lamp saturates the shadow - result = 0 + 1 x shadowIntensity, exactly - and the
     * lane is done: which occluder that was, and in which order the leaves were visited, cannot be seen in the
     * result.  Such walks take an order-free list (closestHitWalk above; the ray reaches from the point to the
     * lamp, thousands of units long) and leave out every box that begins beyond the lamp: an occluder must be
     * hit before it (l < lengthOL, i.e. slab parameter < 1), which the reference's cut-off - the parameter
     * against the view distance, hence the condition on it - never uses.  The list of the OPPOSITE octant, the
     * lamp's side first: for an any-hit query that measured best (mesh -5 %, molecule -2 %; the near side first
     * +10 % on the molecule: the point's own neighbourhood is where the boxes are entered and the tests miss). */
    const bool freeOrder = tidy && S.nbBoxesFree > 0 && S.opaqueShadows &&
                           ballot(active && !(longRay(r.d) && minDistance >= 2.f)) == 0ull;
    Scene W = S;
    float farFree = 0.f;
    int octant = 0;
    bool reversed = false;
    if (freeOrder)
    {
        const int signs = ((r.d.x < 0.f ? 1 : 0) | (r.d.y < 0.f ? 2 : 0) | (r.d.z < 0.f ? 4 : 0)) ^ 7;
        const int lane = (int)__builtin_ctzll(ballot(active));
        octant = __builtin_amdgcn_readlane(signs, lane);
        /* every ray of the wave points into the octant OPPOSITE the list's (the points of a tile towards one lamp): the
         * copy with sorted bounds read the other way round (SOLR_ORDER_REVERSED) */
        reversed = (FEAT & F_DEEP) && S.sortedLists && ballot(active && signs != octant) == 0ull;
        W.offBoxes = S.offBoxesFree + 2u * (unsigned)(octant * S.nbBoxesFree);
        W.offLeaf = S.offLeafFree + 4u * (unsigned)(octant * S.nbBoxesFree);
        W.nbBoxes = S.nbBoxesFree;
        farFree = 1.0002f + 1.0e-4f * (fabsf(r.o.x) + fabsf(r.o.y) + fabsf(r.o.z)) / lengthOL;
    }
    /* the thin copy of that list (tightRay: a shadow ray reaches from the point to the lamp, thousands of units) */
    const bool tight = (FEAT & F_PLANE) && tidy && S.tightLists && ballot(active && !tightRay(r, si)) == 0ull;
    if (tight)
        W.offBoxes += freeOrder ? 16u * (unsigned)S.nbBoxesFree + 2u : 2u * (unsigned)S.nbBoxes + 2u;
    reversed = reversed && !tight;
    if (reversed)
    {
        W.offBoxes += 32u * (unsigned)S.nbBoxesFree + 4u;
        W.nbBoxes = S.nbBoxesFree << 5; /* (this walk's cursors count bytes: SOLR_NEXT_BY_BYTES) */
    }
    const int nbBoxes = W.nbBoxes;
    const PackedRay pr = packRay(r);
    int cursor = (active && result < si.shadowIntensity) ? 0 : SOLR_CURSOR_DONE;
    const bool walked = cursor != SOLR_CURSOR_DONE;
    int visit = 0, doneAfter = SOLR_CURSOR_DONE; /* (COUNT == 2: the leaf visit after which the lane had its shadow) */
    int cur = 0;
    Row2 node;
    node.a = node.b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!tidy)
        node = boxNode(S, 0);
    while (cur < nbBoxes)
    {
        if (ballot(cursor != SOLR_CURSOR_DONE) == 0ull)
            break;
        int leaf = cur, nbPrimitives;
        bool entered;
        SOLR_T(unsigned long long ta = SOLR_NOW();)
        if (tidy)
        {
            if ((FEAT & F_DEEP) && reversed)
            {
                leaf = advanceTidyDeepReversed(W, pr, farFree, cursor, cur, nbPrimitives, entered);
                leaf = leaf < 0 ? leaf : leaf >> 5;
            }
            else
                leaf = advanceTidy<FEAT>(W, pr, freeOrder ? farFree : minDistance, cursor, cur, nbPrimitives, entered);
            SOLR_T(const unsigned long long tb = SOLR_NOW(); cnt.tNode += tb - ta; ++cnt.nAdvance; ta = tb;)
            if (leaf < 0)
                break;
        }
        else if (!stepGeneral<COUNT>(S, r, fastBoxes, minDistance, cursor, cur, node, nbPrimitives, entered, cnt) ||
                 nbPrimitives <= 0)
            continue;
        /* (readfirstlane: the compiler takes an inline-asm result for divergent and would fetch the record per lane) */
        const Row4 L = leafRecord(W, uniform(leaf));
        const int start = uniform(asint(L.d.w));
        for (int k = 0; k < nbPrimitives; ++k)
        {
            const PrimRec rec = leafPrimitive<FEAT>(S, si, L, start, k);
            const int pi = rec.pi;
            const Row2 &head = rec.head;
            const int tag = uniform(asint(head.a.w));
            const int index = uniform(recIndex(S, rec));
            /* GI:829-830 */
            const bool lanes = entered && result < si.shadowIntensity && index != lightId && index != objectId &&
                               (tag & PRIM_FAST0) != 0;
            if (ballot(lanes) == 0ull)
                continue;
            countAdd<COUNT>(cnt.wPrims, 1);
            if (lanes)
                countAdd<COUNT>(cnt.prims, 1);
            const int type = tag & PRIM_TYPE_MASK;
            Hit h;
            h.intersection = V(0.f, 0.f, 0.f);
            h.normal = V(0.f, 0.f, 0.f);
            h.areas = V(0.f, 0.f, 0.f);
            h.shadowIntensity = 0.f;
            bool hit = false;
            const int kind = primKind<FEAT>(si, tag);
            if ((FEAT & F_SPHERE) && kind == KIND_SPHERE)
            {
                /* plain spheres: intensity 1 and no normal when opaque (GI:281, 880) */
                bool back;
                hit = sphereHit(si, V4(head.a), head.b.x, r, lanes, h.intersection, back);
                h.shadowIntensity = 1.f;
                if (tag & PRIM_TRANSPARENT)
                {
                    if (hit)
                        sphereNormal<false>(si, V4(head.a), V4(head.b), false, true, back, r, h);
                }
            }
            else if ((FEAT & F_TRI) && kind == KIND_TRIANGLE && !si.doubleSidedTriangles)
            {
                /* as the general triangle branch below */
                const v3 p0 = V4(head.a);
                const v3 p1 = recP1(S, rec);
                const v3 p2 = recP2(S, rec);
                if (lanes)
                    hit = triangleHit(si, p0, p1, p2, r, h.intersection);
                h.shadowIntensity = 1.f;
                if (tag & PRIM_TRANSPARENT)
                {
                    const v3 n0 = V4(primRow(S, pi, ROW_N0));
                    const v3 n1 = V4(primRow(S, pi, ROW_N1));
                    const v3 n2 = V4(primRow(S, pi, ROW_N2));
                    if (hit)
                        triangleNormal(p0, p1, p2, n0, n1, n2, r, h);
                }
            }
            else if ((FEAT & F_CYL) && kind == KIND_CYLINDER)
            {
                if (lanes)
                    hit = cylinderIntersection(si, V4(head.a), recP1(S, rec), recP2(S, rec), V4(primRow(S, pi, ROW_N1)),
                                               V4(head.b), r, h);
            }
            else if ((FEAT & F_PLANE) && kind >= KIND_PLANE_XY && kind <= KIND_PLANE_XZ)
            {
                if (lanes)
                    hit = planePlain(S, si, kind, V4(head.a), V4(head.b), recPlaneNormal(S, rec), recPlaneAverage(S, rec),
                                     r, h);
            }
            else if ((FEAT & (F_SPHERE | F_PROC)) && extendedGeometry<FEAT>(si) && type == ptSphere)
            {
                /* opaque spheres shadow with intensity 1 and need no normal (GI:281, 880) */
                bool back;
                hit = sphereHit(si, V4(head.a), head.b.x, r, lanes, h.intersection, back);
                h.shadowIntensity = 1.f;
                if ((tag & PRIM_TRANSPARENT) || ((FEAT & F_PROC) && (tag & PRIM_PROCEDURAL)))
                {
                    if (hit)
                        sphereNormal<(FEAT & F_PROC) != 0>(si, V4(head.a), V4(head.b), (tag & PRIM_PROCEDURAL) != 0,
                                                           (tag & PRIM_TRANSPARENT) != 0, back, r, h);
                }
            }
            else if ((FEAT & F_TRI) && !si.doubleSidedTriangles && (type == ptTriangle || !extendedGeometry<FEAT>(si)))
            {
                /* an opaque triangle shadows with intensity 1 whatever its normal (GI:880); with
                 * double-sided triangles the normal decides (GI:643-647) and the full test is used */
                const v3 p0 = V4(head.a);
                const v3 p1 = recP1(S, rec);
                const v3 p2 = recP2(S, rec);
                if (lanes)
                    hit = triangleHit(si, p0, p1, p2, r, h.intersection);
                h.shadowIntensity = 1.f;
                if (tag & PRIM_TRANSPARENT)
                {
                    const v3 n0 = V4(primRow(S, pi, ROW_N0));
                    const v3 n1 = V4(primRow(S, pi, ROW_N1));
                    const v3 n2 = V4(primRow(S, pi, ROW_N2));
                    if (hit)
                        triangleNormal(p0, p1, p2, n0, n1, n2, r, h);
                }
            }
            else if (lanes)
                hit = testPrimitive<true, FEAT>(S, si, rec, tag, r, h);
            if (hit)
            {
                const float l = length(h.intersection - r.o);
                if (l > si.geometryEpsilon && l < lengthOL)
                {
                    float ratio = h.shadowIntensity * si.shadowIntensity;
                    if (tag & PRIM_TRANSPARENT)
                    {
                        /* coloured shadow through a transparent primitive, GI:880-892 */
                        const MaterialHot mh = loadMaterialHot(S, uniform(asint(head.b.w)));
                        v3 O_L = r.dn;
                        float a = fabsf(dot(O_L, h.normal));
                        float rr = (mh.transparency == 0.f) ? 1.f : (1.f - mh.transparency);
                        ratio *= rr * a;
                        color.x += ratio * (0.3f - 0.3f * mh.color.x);
                        color.y += ratio * (0.3f - 0.3f * mh.color.y);
                        color.z += ratio * (0.3f - 0.3f * mh.color.z);
                    }
                    result += ratio;
                }
            }
        }
        /* the reference re-tests `result < shadowIntensity` before every node */
        if (COUNT == 2)
            ++visit;
        if (cursor != SOLR_CURSOR_DONE && !(result < si.shadowIntensity))
        {
            cursor = SOLR_CURSOR_DONE;
            if (COUNT == 2)
                doneAfter = visit;
        }
        SOLR_T(cnt.tLeaf += SOLR_NOW() - ta; ++cnt.nLeaf;)
    }
    if (COUNT == 2)
        recordWalk(cnt, tidy ? WALK_SHADOW : WALK_GENERAL, freeOrder, octant, walked, r, freeOrder ? farFree : minDistance,
                   doneAfter, tight, reversed ? 2 : 0);
    result = fmaxf(0.f, fminf(result, si.shadowIntensity));
    SOLR_T(cnt.tShadow += SOLR_NOW() - tw0;)
    return result;
}

/* ---------------------------------------------------------------------- */
/* Shading                                                                 */
/* ---------------------------------------------------------------------- */

SOLR_DEV float rnd(const Scene &S, long i)
{
    /* out-of-range reads (reference UB) return 0 */
    return (i >= 0 && i < S.nbRandoms) ? S.randoms[i] : 0.f;
}

/* GS:36-124 for the lane's own primitive (per-lane gathers) */
template <int FEAT>
SOLR_DEV float4 intersectionShader(const Scene &S, const SceneInfo &si, int pi, int type, int materialId,
                                   const MaterialHot &mh, v3 intersection, v3 areas, const TexOut &o)
{
    float4 c = mh.color;
    c.w = 0.f;
    const bool textured = (FEAT & F_TEX) && mh.ids.x != TEXTURE_NONE;
    if (si.extendedGeometry)
    {
        switch (type)
        {
        case ptCone:
        case ptCylinder:
        case ptEnvironment:
        case ptSphere:
        case ptEllipsoid:
            if (textured)
            {
                const v3 p0 = V4(primRow(S, pi, ROW_P0_TYPE));
                const float vt1x = primRow(S, pi, ROW_N2).w;
                const float vt1y = primRow(S, pi, ROW_UV).x;
                c = sphereUVMapping(p0, vt1x, vt1y, mh.color, loadMaterialCold(S, materialId), S.textures, intersection, o);
            }
            break;
        case ptCheckboard:
        {
            const v3 p0 = V4(primRow(S, pi, ROW_P0_TYPE));
            const v3 size = V4(primRow(S, pi, ROW_SIZE_MAT));
            if (textured)
                c = cubeMapping(si, type, p0, size, mh.color, loadMaterialCold(S, materialId), S.textures, intersection, o);
            else
            {
                int x = (int)(si.viewDistance + ((intersection.x - p0.x) / size.x));
                int z = (int)(si.viewDistance + ((intersection.z - p0.z) / size.x));
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
        }
        case ptXYPlane:
        case ptYZPlane:
        case ptXZPlane:
        case ptCamera:
            if (textured)
            {
                const v3 p0 = V4(primRow(S, pi, ROW_P0_TYPE));
                const v3 size = V4(primRow(S, pi, ROW_SIZE_MAT));
                c = cubeMapping(si, type, p0, size, mh.color, loadMaterialCold(S, materialId), S.textures, intersection, o);
            }
            break;
        case ptTriangle:
            if (textured)
            {
                const float2 vt0 = make_float2(primRow(S, pi, ROW_N0).w, primRow(S, pi, ROW_N1).w);
                const float4 T = primRow(S, pi, ROW_UV);
                const float2 vt1 = make_float2(primRow(S, pi, ROW_N2).w, T.x);
                const float2 vt2 = make_float2(T.y, T.z);
                c = triangleUVMapping(si, vt0, vt1, vt2, mh.attributes.y, mh.color, loadMaterialCold(S, materialId),
                                      S.textures, areas, o);
            }
            break;
        default:
            break;
        }
    }
    else if (textured)
    {
        const float2 vt0 = make_float2(primRow(S, pi, ROW_N0).w, primRow(S, pi, ROW_N1).w);
        const float4 T = primRow(S, pi, ROW_UV);
        const float2 vt1 = make_float2(primRow(S, pi, ROW_N2).w, T.x);
        const float2 vt2 = make_float2(T.y, T.z);
        c = triangleUVMapping(si, vt0, vt1, vt2, mh.attributes.y, mh.color, loadMaterialCold(S, materialId), S.textures,
                              areas, o);
    }
    return c;
}

/* GI:916-1080.  Every lane of the wave calls this together; `active` marks
 * the lanes that actually shade.  The light loop is wave-uniform, the shadow
 * walk inside it is wave-synchronous. */
template <int COUNT, int FEAT>
SOLR_DEV v3 primitiveShader(const Scene &S, bool active, int index, const SceneInfo &si, v3 origin, v3 &normal,
                            int objectId, v3 intersection, v3 areas, v3 &closestColor, int iteration,
                            float &shadowIntensity, v3 &totalBlinn, float4 &attributes, Counters &cnt)
{
    SOLR_T(const unsigned long long tShade0 = SOLR_NOW();)
    const int pi = active ? objectId : 0;
    const int type = asint(primRow(S, pi, ROW_P0_TYPE).w) & PRIM_TYPE_MASK; /* row 0 carries type + material facts */
    const int materialId = asint(primRow(S, pi, ROW_SIZE_MAT).w);
    const int primIndex = asint(primRow(S, pi, ROW_P1_INDEX).w);
    const MaterialHot mh = loadMaterialHot(S, materialId);
    v3 lampsColor = V(0.f, 0.f, 0.f);
    v3 intersectionColor = V(0.f, 0.f, 0.f);
    float4 specular = make_float4(mh.specular.x, mh.specular.y, mh.specular.z, 0.f);
    float ambientOcclusion = 0.f;
    bool shade = active;

    if (active)
    {
        shadowIntensity = 0.f;
        v3 bumpNormal = V(0.f, 0.f, 0.f);
        TexOut o = {&bumpNormal, &specular, &attributes, &ambientOcclusion};
        float4 ic = intersectionShader<FEAT>(S, si, pi, type, materialId, mh, intersection, areas, o);
        intersectionColor = V(ic.x, ic.y, ic.z);
        normal = normal + bumpNormal;
        normal = normalize(normal);
        if (mh.attributes.z == 1)
            shade = false; /* wireframe: returns the texel, GI:947-951 */
    }
    const bool wire = active && !shade;

    if (si.graphicsLevel > glNoShading)
    {
        if (shade)
        {
            closestColor.x *= mh.innerIllumination.x;
            closestColor.y *= mh.innerIllumination.x;
            closestColor.z *= mh.innerIllumination.x;
        }
        for (int cpt = 0; cpt < S.nbLights; ++cpt)
        {
            const int cptLamp =
                (si.pathTracingIteration >= NB_MAX_ITERATIONS) ? (si.pathTracingIteration % S.nbLights) : 0;
            const LightPlane li = loadLight(S, cptLamp);
            const int lightPrimitiveId = asint(li.location.w);
            const int lightMaterialId = li.materialId;
            /* materials[MATERIAL_NONE] is read out of bounds by the reference */
            const MaterialHot m = loadMaterialHot(S, lightMaterialId < 0 ? 0 : lightMaterialId);
            const bool lit = shade && (lightPrimitiveId != primIndex);

            v3 center = V4(li.location);
            const int t = (index + si.timestamp) % (MAX_BITMAP_SIZE - 3);
            if (si.pathTracingIteration >= NB_MAX_ITERATIONS)
            {
                float a = m.innerIllumination.y * 10.f * si.pathTracingIteration / si.maxPathTracingIterations;
                center.x += rnd(S, t) * a;
                center.y += rnd(S, t + 1) * a;
                center.z += rnd(S, t + 2) * a;
            }
            v3 lightRay = center - intersection;
            const float lightRayLength = length(lightRay);
            const bool inRange = lit && (lightRayLength < m.innerIllumination.z);
            v3 shadowColor = V(0.f, 0.f, 0.f);
            float lambert = 0.f;
            if (inRange)
            {
                lightRay = normalize(lightRay);
                lambert = mh.innerIllumination.x + dot(normal, lightRay);
            }
            const bool wantShadow = inRange && lambert > 0.f && si.graphicsLevel > 3 && iteration < 4 &&
                                    mh.innerIllumination.x == 0.f;
            {
                v3 sc;
                float s = shadowWalk<COUNT, FEAT>(S, si, wantShadow, center, intersection, lightPrimitiveId, iteration, sc,
                                            objectId, cnt);
                if (wantShadow)
                {
                    shadowIntensity = s;
                    shadowColor = sc;
                }
            }
            if (inRange) /* graphicsLevel > glNoShading holds here, GI:1000 */
            {
                float photonEnergy = sqrt_ieee(lightRayLength / m.innerIllumination.z);
                photonEnergy = (photonEnergy > 1.f) ? 1.f : photonEnergy;
                photonEnergy = (photonEnergy < 0.f) ? 0.f : photonEnergy;
                lambert *= (lambert < 0.f) ? -mh.transparency : 1.f;
                if (lightMaterialId != MATERIAL_NONE)
                    lambert *= m.innerIllumination.x;
                else
                    lambert *= li.color.w;
                if (mh.innerIllumination.w != 0.f)
                    lambert *= (1.f + rnd(S, t) * mh.innerIllumination.w * 100.f);
                lambert *= (1.f - shadowIntensity);
                lambert += si.backgroundColor.w;
                lambert *= (1.f - photonEnergy);
                lampsColor.x += lambert * li.color.x - shadowColor.x;
                lampsColor.y += lambert * li.color.y - shadowColor.y;
                lampsColor.z += lambert * li.color.z - shadowColor.z;
                if (si.graphicsLevel > 1 && shadowIntensity < si.shadowIntensity)
                {
                    v3 viewRay = normalize(intersection - origin);
                    v3 blinnDir = lightRay - viewRay;
                    float temp = sqrt_ieee(dot(blinnDir, blinnDir));
                    if (temp != 0.f)
                    {
                        blinnDir = blinnDir * (1.f / temp);
                        float blinnTerm = dot(blinnDir, normal);
                        blinnTerm = (blinnTerm < 0.f) ? 0.f : blinnTerm;
                        blinnTerm = specular.x * pow_f(blinnTerm, specular.y);
                        blinnTerm *= (1.f - photonEnergy);
                        totalBlinn.x += li.color.x * li.color.w * blinnTerm;
                        totalBlinn.y += li.color.y * li.color.w * blinnTerm;
                        totalBlinn.z += li.color.z * li.color.w * blinnTerm;
                    }
                }
            }
            if (shade)
            {
                closestColor.x += intersectionColor.x * lampsColor.x;
                closestColor.y += intersectionColor.y * lampsColor.y;
                closestColor.z += intersectionColor.z * lampsColor.z;
                if ((FEAT & F_TEX) && mh.ids.y != TEXTURE_NONE) /* ambient-occlusion map, GI:1063 */
                {
                    closestColor.x *= ambientOcclusion;
                    closestColor.y *= ambientOcclusion;
                    closestColor.z *= ambientOcclusion;
                }
                saturate3(closestColor);
                saturate3(totalBlinn);
            }
        }
    }
    else if (shade)
        closestColor = intersectionColor;

    SOLR_T(cnt.tShade += SOLR_NOW() - tShade0;)
    return wire ? intersectionColor : closestColor;
}

/* GI:87-151 (per lane) */
template <int FEAT>
SOLR_DEV v3 skyboxMapping(const Scene &S, const SceneInfo &si, v3 origin, v3 target)
{
    const MaterialHot mh = loadMaterialHot(S, si.skyboxMaterialId);
    v3 result = V(mh.color.x, mh.color.y, mh.color.z);
    v3 dir = normalize(target - origin);
    float a = 2.f * dot(dir, dir);
    float b = 2.f * dot(origin, dir);
    float c = dot(origin, origin) - (float)(si.skyboxRadius * si.skyboxRadius);
    float d = b * b - 2.f * a * c;
    if (d <= 0.f || a == 0.f)
        return result;
    float r = sqrt_ieee(d);
    float t1 = (-b - r) / a;
    float t2 = (-b + r) / a;
    if (t1 <= si.geometryEpsilon && t2 <= si.geometryEpsilon)
        return result;
    float t = 0.f;
    if (t1 <= si.geometryEpsilon)
        t = t2;
    else if (t2 <= si.geometryEpsilon)
        t = t1;
    else
        t = (t1 < t2) ? t1 : t2;
    if (t < si.geometryEpsilon)
        return result;
    if (!(FEAT & F_TEX))
        return result; /* the host selects a F_TEX instantiation whenever the skybox material has a texture */
    /* GI:87-151 fetches a texel whatever the material: without a diffuse texture the "computed texture"
     * mapping (40000 x 40000) indexes gigabytes past the atlas - a fault here.  Such a skybox shows the
     * material's colour, wherever the test falls in the reference's order of operations: nothing that
     * follows has a side effect */
    if (mh.ids.x < 0)
        return result;
    v3 I = normalize(origin + dir * t);
    float U = ((atan2_f(I.x, I.z) / SOLR_PI) + 1.f) * .5f;
    float Vv = (asin_f(I.y) / SOLR_PI) + .5f;
    const MaterialCold mc = loadMaterialCold(S, si.skyboxMaterialId);
    int u = (int)(mc.textureMapping.x * U);
    int v = (int)(mc.textureMapping.y * Vv);
    if (mc.textureMapping.x != 0)
        u %= mc.textureMapping.x;
    if (mc.textureMapping.y != 0)
        v %= mc.textureMapping.y;
    if (u >= 0 && u < mc.textureMapping.x && v >= 0 && v < mc.textureMapping.y)
    {
        int A = (v * mc.textureMapping.x + u) * mc.textureMapping.w;
        int B = mc.textureMapping.x * mc.textureMapping.y * mc.textureMapping.w;
        int idx = A % B;
        int i = mc.textureOffset.x + idx;
        result.x = S.textures[i] / 256.f;
        result.y = S.textures[i + 1] / 256.f;
        result.z = S.textures[i + 2] / 256.f;
    }
    return result;
}

/* per-lane column of the LDS colour stack: slot s, component c */
/* Per-lane record kept in LDS behind the colour stack: path state that is touched once per
 * bounce (or once per pixel) but would otherwise sit in VGPRs through both walks of every
 * bounce.  At four waves per SIMD the kernel has 128 VGPRs; what does not fit goes to scratch,
 * and scratch of 4096 resident waves does not fit the L2 either (measured: 0.3 GB read + 0.8 GB
 * written per 1080p Cornell frame against 0.1 GB of framebuffer) - LDS is the cheaper home. */
enum ColdField
{
    C_RRO = 0,        /* deferred reflection ray: origin, target (CRT:262-269) */
    C_RRD = 3,
    C_RRATIO = 6,
    C_RRAYS = 7,      /* int */
    C_RBLINN = 8,     /* recursiveBlinn */
    C_LATEST = 11,    /* latestIntersection */
    C_RAYLENGTH = 14,
    C_REFRACTION = 15, /* initialRefraction */
    C_LASTIT = 16,    /* int */
    C_ID_X = 17,      /* primitiveXYId.x .z .w (int) */
    C_ID_Z = 18,
    C_ID_W = 19,
    C_DOF = 20,
    C_ROO = 21,       /* the bounce loop's current ray: origin, target */
    C_ROD = 24,
    COLD_FIELDS = 27
};

struct ColorStack
{
    float *base;  /* &lds[lane] */
    int stride;   /* floats between consecutive (slot, component) cells = block size */
    int cold;     /* first cell of the cold record = 4 * stack slots */
    /* F_STACK instantiations: slots ldsSlots, ldsSlots + 1 ... of this lane's pixel, one float4 each, `deepStride`
     * float4s apart (a plane of the strip per slot: the 8 pixels of a tile row are 128 contiguous bytes) */
    float4 *deep;
    long deepStride;
    int ldsSlots;
    SOLR_DEV float &coldf(int field) const { return base[(cold + field) * stride]; }
    SOLR_DEV int &coldi(int field) const { return *(int *)&base[(cold + field) * stride]; }
    SOLR_DEV float &at(int slot, int c) const { return base[(slot * 4 + c) * stride]; }
    SOLR_DEV void set(int slot, v3 v) const
    {
        at(slot, 0) = v.x;
        at(slot, 1) = v.y;
        at(slot, 2) = v.z;
    }
    SOLR_DEV v3 get(int slot) const { return V(at(slot, 0), at(slot, 1), at(slot, 2)); }
    /* the same three with a slot that may lie beyond the LDS ones (SP: an F_STACK instantiation; otherwise the code
     * above, unchanged).  A lane only ever reads back what it wrote itself: program order is all that is needed. */
    template <bool SP>
    SOLR_DEV void put(int slot, v3 v, float k) const
    {
        if (SP && slot >= ldsSlots)
            deep[(long)(slot - ldsSlots) * deepStride] = make_float4(v.x, v.y, v.z, k);
        else
        {
            set(slot, v);
            at(slot, 3) = k;
        }
    }
    template <bool SP>
    SOLR_DEV float4 take(int slot) const
    {
        if (SP && slot >= ldsSlots)
            return deep[(long)(slot - ldsSlots) * deepStride];
        return make_float4(at(slot, 0), at(slot, 1), at(slot, 2), at(slot, 3));
    }
    template <bool SP>
    SOLR_DEV void add(int slot, v3 d) const
    {
        if (SP && slot >= ldsSlots)
        {
            float4 &cell = deep[(long)(slot - ldsSlots) * deepStride];
            const float4 was = cell;
            cell = make_float4(was.x + d.x, was.y + d.y, was.z + d.z, was.w);
        }
        else
        {
            at(slot, 0) += d.x;
            at(slot, 1) += d.y;
            at(slot, 2) += d.z;
        }
    }
};

/* three consecutive cold fields used like a v3 */
struct V3Ref
{
    float &x, &y, &z;
    SOLR_DEV V3Ref(const ColorStack &cs, int field) : x(cs.coldf(field)), y(cs.coldf(field + 1)), z(cs.coldf(field + 2)) {}
    SOLR_DEV operator v3() const { return V(x, y, z); }
    SOLR_DEV const V3Ref &operator=(const v3 &v) const
    {
        x = v.x;
        y = v.y;
        z = v.z;
        return *this;
    }
};

/* CRT:69-408.  Called by the whole wave; `active` lanes own a pixel.
 *
 * The reference traces up to three kinds of rays per pixel: the bounce loop
 * (CRT:125-294), one deferred reflection of the first transparent hit
 * (CRT:296-315) and one global-illumination ray (CRT:317-378).  Here they are
 * phases 0, 1 and 2 of one wave-uniform loop with a single walk and a single
 * shader call site, which keeps the instruction footprint and the live
 * register set of the kernel small. */
template <int COUNT, int FEAT>
SOLR_DEV v3 launchRayTracing(const Scene &S, bool active, int index, v3 rayO, v3 rayD, const SceneInfo &si,
                             float &depthOfField, int4 &primitiveXYId, const ColorStack &cs, Counters &cnt)
{
    SOLR_T(const unsigned long long tTrace0 = SOLR_NOW();)
    v3 intersectionColor = V(0.f, 0.f, 0.f);
    v3 closestIntersection = V(0.f, 0.f, 0.f);
    v3 normal = V(0.f, 0.f, 0.f);
    int closestPrimitive = -1;
    bool carryon = true;
    const V3Ref roO(cs, C_ROO), roD(cs, C_ROD);
    roO = rayO;
    roD = rayD;
    float &initialRefraction = cs.coldf(C_REFRACTION);
    initialRefraction = 1.f;
    int iteration = 0;     /* loop trip counter of phase 0, identical in every running lane */
    int &lastIteration = cs.coldi(C_LASTIT); /* per lane: trips it took part in = the reference's final `iteration` */
    lastIteration = 0;
    int &idX = cs.coldi(C_ID_X), &idZ = cs.coldi(C_ID_Z), &idW = cs.coldi(C_ID_W); /* primitiveXYId.x .z .w */
    idX = -1;
    idZ = 0;
    idW = 0;
    int currentMaterialId = -2;

    int currentMaxIteration =
        (si.graphicsLevel < glReflectionsAndRefractions) ? 1 : si.nbRayIterations + si.pathTracingIteration;
    currentMaxIteration = (currentMaxIteration > NB_MAX_ITERATIONS) ? NB_MAX_ITERATIONS : currentMaxIteration;

    /* slots 0 .. currentMaxIteration-1 are the ones ever written or read (the reference's array
     * has one more, CRT:92).  (The slots beyond the LDS ones of an F_STACK frame need no zero: every slot below a lane's
     * `lastIteration` is written by the trip that made it - hit or background - before the deferred reflection adds to
     * it and the blend reads it; only slot 0 can be read unwritten, by a lane that never ran.) */
    constexpr bool SP = (FEAT & F_STACK) != 0;
    int zeroSlots = currentMaxIteration > 1 ? currentMaxIteration : 1;
    if (SP && zeroSlots > cs.ldsSlots)
        zeroSlots = cs.ldsSlots;
    for (int s = 0; s < zeroSlots; ++s)
    {
        cs.at(s, 0) = 0.f;
        cs.at(s, 1) = 0.f;
        cs.at(s, 2) = 0.f;
        cs.at(s, 3) = 0.f;
    }

    const V3Ref recursiveBlinn(cs, C_RBLINN);
    recursiveBlinn = V(0.f, 0.f, 0.f);
    float shadowIntensity = 0.f;
    v3 closestColor = V(0.f, 0.f, 0.f);
    v3 colorBox = V(0.f, 0.f, 0.f);
    const V3Ref latestIntersection(cs, C_LATEST);
    latestIntersection = rayO;
    float &rayLength = cs.coldf(C_RAYLENGTH);
    rayLength = 0.f;
    float &dofCold = cs.coldf(C_DOF);
    dofCold = si.viewDistance;

    int &reflectedRays = cs.coldi(C_RRAYS);
    reflectedRays = -1;
    const V3Ref rrO(cs, C_RRO), rrD(cs, C_RRD);
    rrO = V(0.f, 0.f, 0.f);
    rrD = V(0.f, 0.f, 0.f);
    float &reflectedRatio = cs.coldf(C_RRATIO);
    reflectedRatio = 0.f;

    /* FULL = false is the lean instantiation the host selects when neither
     * global illumination nor the box-debug view is requested */
    const bool giEnabled = (FEAT & F_FULL) && (si.advancedIllumination == aiBasic || si.advancedIllumination == aiFull);
    const bool giPass = giEnabled && si.pathTracingIteration >= NB_MAX_ITERATIONS;
    v3 ptO = V(0.f, 0.f, 0.f), ptD = V(0.f, 0.f, 0.f);
    float pathTracingRatio = 0.f;
    v3 pathTracingColor = V(0.f, 0.f, 0.f);
    bool useGlobalIllumination = false;
    bool test = true;

    v3 rBlinn = V(0.f, 0.f, 0.f);
    bool running = active;
    int phase = 0;

    while (true)
    {
        /* ---- which ray does each lane trace in this trip? ---- */
        bool want;
        v3 tO, tD;
        int tIter, tMat;
        if (phase == 0)
        {
            running = running && (iteration < currentMaxIteration) && (rayLength < si.viewDistance) && carryon;
            if (ballot(running) == 0ull)
            {
                phase = 1;
                continue;
            }
            want = running;
            tO = roO;
            tD = roD;
            tIter = iteration;
            tMat = currentMaterialId;
        }
        else if (phase == 1)
        {
            want = active && si.graphicsLevel >= glReflectionsAndRefractions && reflectedRays != -1;
            tO = rrO;
            tD = rrD;
            tIter = reflectedRays;
            tMat = currentMaterialId;
        }
        else
        {
            want = active && useGlobalIllumination && si.advancedIllumination == aiFull;
            tO = ptO;
            tD = ptD;
            tIter = 30; /* only consider close geometry (max distance / 31), CRT:326 */
            tMat = MATERIAL_NONE;
        }

        v3 areas = V(0.f, 0.f, 0.f);
        const bool hit = closestHitWalk<COUNT, FEAT>(S, si, want, tO, tD, tIter, tMat, closestPrimitive,
                                               closestIntersection, normal, areas, colorBox, cnt);
        const bool hitLane = want && hit;

        /* material of the hit (per-lane gather) */
        const int cp = hitLane ? closestPrimitive : 0;
        const int cpMaterial = asint(primRow(S, cp, ROW_SIZE_MAT).w);
        const MaterialHot cm = loadMaterialHot(S, cpMaterial < 0 ? 0 : cpMaterial);
        float4 attributes = make_float4(0.f, 0.f, 0.f, 0.f);
        bool shadeLane = hitLane;
        int shadeIteration = tIter;

        if (phase == 0)
        {
            if (running)
                carryon = hit;
            attributes = make_float4(cm.reflection, cm.transparency, cm.refraction, cm.opacity);
            if (hitLane)
            {
                currentMaterialId = cpMaterial;
                if (iteration == 0)
                {
                    cs.set(0, V(0.f, 0.f, 0.f));
                    cs.at(0, 3) = 1.f;
                    latestIntersection = closestIntersection;
                    dofCold = length(closestIntersection - tO); /* tO is the primary origin in trip 0 */
                    if (giEnabled && cm.innerIllumination.x == 0.f)
                    {
                        int t = (index + si.pathTracingIteration * 100 + si.timestamp) % (MAX_BITMAP_SIZE - 3);
                        ptO = closestIntersection + normal * si.rayEpsilon;
                        ptD.x = normal.x + 100.f * rnd(S, t);
                        ptD.y = normal.y + 100.f * rnd(S, t + 1);
                        ptD.z = normal.z + 100.f * rnd(S, t + 2);
                        float cos_theta = dot(normalize(ptD), normal);
                        if (cos_theta < 0.f)
                            ptD = vneg(ptD);
                        ptD = ptD + closestIntersection;
                        pathTracingRatio = (1.f - attributes.y) * fabsf(cos_theta);
                        useGlobalIllumination = true;
                    }
                    idX = asint(primRow(S, cp, ROW_P1_INDEX).w);
                }
            }
        }
        else if (phase == 1)
        {
            attributes.x = cm.reflection; /* the only field the reference sets, CRT:305-306 */
        }
        else
        {
            shadeLane = false;
            if (hitLane)
            {
                if (cpMaterial != MATERIAL_NONE)
                {
                    if (cm.innerIllumination.x == 0.f)
                    {
                        cs.at(0, 0) = cm.color.x * cm.innerIllumination.x * pathTracingRatio;
                        cs.at(0, 1) = cm.color.y * cm.innerIllumination.x * pathTracingRatio;
                        cs.at(0, 2) = cm.color.z * cm.innerIllumination.x * pathTracingRatio;
                        test = false;
                    }
                    else
                    {
                        cs.at(0, 0) = cm.color.x * pathTracingRatio;
                        cs.at(0, 1) = cm.color.y * pathTracingRatio;
                        cs.at(0, 2) = cm.color.z * pathTracingRatio;
                    }
                }
                if (test)
                {
                    pathTracingRatio *= STANDARD_LUNINANCE_STRENGTH;
                    if (cm.innerIllumination.x == 0.f)
                    {
                        cs.at(0, 0) -= si.shadowIntensity;
                        cs.at(0, 1) -= si.shadowIntensity;
                        cs.at(0, 2) -= si.shadowIntensity;
                    }
                    else
                        shadeLane = true;
                }
            }
            shadeIteration = lastIteration;
        }

        const v3 shaded = primitiveShader<COUNT, FEAT>(S, shadeLane, index, si, tO, normal, closestPrimitive,
                                                 closestIntersection, areas, closestColor, shadeIteration,
                                                 shadowIntensity, rBlinn, attributes, cnt);

        if (phase == 0)
        {
            if (hitLane)
            {
                v3 colorIt = shaded;
                float contribution;
                v3 reflectedTarget = V(0.f, 0.f, 0.f);
                idZ = (int)((float)idZ + cm.innerIllumination.x * 256);

                float segmentLength = length(closestIntersection - latestIntersection);
                latestIntersection = closestIntersection;

                float transparency = attributes.y;
                float a = 0.f;
                if (attributes.y != 0.f)
                {
                    float refraction = attributes.z;
                    if (initialRefraction == refraction)
                    {
                        refraction = 1.f;
                        float len = segmentLength * (attributes.w * (1.f - transparency));
                        rayLength += len;
                        rayLength = (rayLength > si.viewDistance) ? si.viewDistance : rayLength;
                        a = (rayLength / si.viewDistance);
                        colorIt.x -= a;
                        colorIt.y -= a;
                        colorIt.z -= a;
                    }
                    v3 O_E = normalize(closestIntersection - roO);
                    reflectedTarget = vectorRefraction(O_E, refraction, normal, initialRefraction);
                    contribution = transparency - a;
                    initialRefraction = refraction;
                    if (reflectedRays == -1 && attributes.x != 0.f)
                    {
                        v3 rd = vectorReflection(O_E, normal);
                        rrO = closestIntersection + rd * si.rayEpsilon;
                        rrD = closestIntersection + rd;
                        reflectedRatio = attributes.x;
                        reflectedRays = iteration;
                    }
                }
                else if (attributes.x != 0.f)
                {
                    v3 O_E = normalize(closestIntersection - roO);
                    reflectedTarget = vectorReflection(O_E, normal);
                    contribution = attributes.x;
                }
                else
                {
                    carryon = false;
                    contribution = 1.f;
                    /* the reference keeps the previous bounce's reflectedTarget
                     * here; the ray it builds from it is never traced */
                }
                cs.template put<SP>(iteration, colorIt, contribution);

                rBlinn.x /= (float)(iteration + 1);
                rBlinn.y /= (float)(iteration + 1);
                rBlinn.z /= (float)(iteration + 1);
                recursiveBlinn.x = (rBlinn.x > recursiveBlinn.x) ? rBlinn.x : recursiveBlinn.x;
                recursiveBlinn.y = (rBlinn.y > recursiveBlinn.y) ? rBlinn.y : recursiveBlinn.y;
                recursiveBlinn.z = (rBlinn.z > recursiveBlinn.z) ? rBlinn.z : recursiveBlinn.z;

                roO = closestIntersection + reflectedTarget * si.rayEpsilon;
                roD = closestIntersection + reflectedTarget;

                if (si.pathTracingIteration != 0 && cm.color.w != 0.f)
                {
                    float ratio = cm.color.w;
                    ratio *= (attributes.y == 0.f) ? 1000.f : 1.f;
                    int rindex = (index + si.timestamp) % (MAX_BITMAP_SIZE - 3);
                    roD.x += rnd(S, rindex) * ratio;
                    roD.y += rnd(S, rindex + 1) * ratio;
                    roD.z += rnd(S, rindex + 2) * ratio;
                }
            }
            else if (running)
            {
                v3 c;
                if (si.skyboxMaterialId != MATERIAL_NONE)
                {
                    c = skyboxMapping<FEAT>(S, si, roO, roD);
                    float rad = c.x + c.y + c.z;
                    idZ = (int)((float)idZ + ((rad > 2.5f) ? rad * 256.f : 0.f));
                }
                else if (si.gradientBackground)
                {
                    v3 up = V(0.f, 1.f, 0.f);
                    v3 dir = normalize(roD - roO);
                    float angle = 0.5f - dot(up, dir);
                    angle = (angle > 1.f) ? 1.f : angle;
                    c.x = (1.f - angle) * si.backgroundColor.x;
                    c.y = (1.f - angle) * si.backgroundColor.y;
                    c.z = (1.f - angle) * si.backgroundColor.z;
                }
                else
                    c = V(si.backgroundColor.x, si.backgroundColor.y, si.backgroundColor.z);
                cs.template put<SP>(iteration, c, 1.f);
            }
            if (running)
                lastIteration = iteration + 1;
            iteration++;
        }
        else if (phase == 1)
        {
            if (hitLane)
            {
                cs.template add<SP>(reflectedRays, V(shaded.x * reflectedRatio, shaded.y * reflectedRatio, shaded.z * reflectedRatio));
                idW = (int)(shadowIntensity * 255);
            }
            if (!giPass)
                break;
            phase = 2;
        }
        else
        {
            if (shadeLane)
                pathTracingColor = shaded;
            if (active && !hitLane && si.skyboxMaterialId != MATERIAL_NONE)
            {
                pathTracingColor = skyboxMapping<FEAT>(S, si, ptO, ptD);
                pathTracingRatio *= SKYBOX_LUNINANCE_STRENGTH;
            }
            if (active && test)
            {
                cs.at(0, 0) += pathTracingColor.x * pathTracingRatio;
                cs.at(0, 1) += pathTracingColor.y * pathTracingRatio;
                cs.at(0, 2) += pathTracingColor.z * pathTracingRatio;
            }
            break;
        }
    }

    int iterationsOut = 0;
    if (active)
    {
        const int iterations = lastIteration;
        iterationsOut = iterations;
        if (test)
        {
            /* back-to-front blend, CRT:382-386.  (The lane's column of the stack from the base as it is held here,
             * through a value the compiler cannot trace back to the lane number: it would otherwise keep threadIdx.x
             * alive through the whole trace - in scratch - just to form these addresses again.) */
            ColorStack bs = cs;
            asm volatile("" : "+v"(bs.base));
            const float4 last = bs.template take<SP>(iterations >= 1 ? iterations - 1 : 0);
            v3 next = V(last.x, last.y, last.z);
            for (int i = iterations - 2; i >= 0; --i)
            {
                const float4 cell = bs.template take<SP>(i);
                v3 ci = V(cell.x, cell.y, cell.z);
                float k = cell.w;
                next.x = ci.x * (1.f - k) + next.x * k;
                next.y = ci.y * (1.f - k) + next.y * k;
                next.z = ci.z * (1.f - k) + next.z * k;
            }
            intersectionColor = next;
            intersectionColor.x += recursiveBlinn.x;
            intersectionColor.y += recursiveBlinn.y;
            intersectionColor.z += recursiveBlinn.z;
        }
        else
            intersectionColor = cs.get(0);

        float D1 = si.viewDistance * 0.95f;
        const float dofNow = dofCold;
        if (si.atmosphericEffect == aeFog && dofNow > D1)
        {
            float D2 = si.viewDistance * 0.05f;
            float a = dofNow - D1;
            float b = 1.f - (a / D2);
            intersectionColor.x = intersectionColor.x * b + si.backgroundColor.x * (1.f - b);
            intersectionColor.y = intersectionColor.y * b + si.backgroundColor.y * (1.f - b);
            intersectionColor.z = intersectionColor.z * b + si.backgroundColor.z * (1.f - b);
        }
        intersectionColor.x -= colorBox.x;
        intersectionColor.y -= colorBox.y;
        intersectionColor.z -= colorBox.z;
    }
    depthOfField = dofCold;
    /* all four words are written, whether the lane owns a pixel or not (the caller stores the ids of active lanes only):
     * nothing of the record the caller came in with has to stay alive - in a register, in practice in scratch - through
     * the whole trace */
    primitiveXYId.x = idX;
    primitiveXYId.y = iterationsOut;
    primitiveXYId.z = idZ;
    primitiveXYId.w = idW;
    SOLR_T(cnt.tTrace += SOLR_NOW() - tTrace0;)
    return intersectionColor;
}

/* CRT:50-67 launchVolumeRendering = GI:1088-1265 intersectionsWithPrimitives, the trace of k_volumeRenderer
 * (cameraType ctVolumeRendering): every primitive the ray meets beyond ppi.param1 is shaded as a first hit
 * (primitiveShader, iteration 0, its own shadow rays) and kept, nearest first, in ten layers, which are then
 * composited along the ray in 500 steps of viewDistance / 500 with weight 1 / param2.  Wave-synchronous like the
 * other walks: the wave follows the reference's list (the exact one: the host selects it for this camera, a tie
 * between two layers goes to the primitive the reference meets first), every lane that entered a leaf tests its
 * primitives, and the lanes that hit one shade it together.  The layers are the lane's colour-stack slots in LDS
 * (eleven: GI:1198-1200 shifts the tenth into the element behind the array; the host sizes the stack for it).
 * VOLUME_RENDERING_NORMALS is not defined in the reference (GI:22, Consts.h:57), `normalize(color)` at GI:1254
 * discards its result. */
template <int COUNT, int FEAT>
SOLR_DEV v3 launchVolumeRendering(const Scene &S, bool active, int index, v3 rayO, v3 rayD, const SceneInfo &si,
                                  const PostProcessingInfo &ppi, int4 &primitiveXYId, const ColorStack &cs, Counters &cnt)
{
    constexpr int MAXDEPTH = 10;
    primitiveXYId.x = -1;
    primitiveXYId.y = 1;
    primitiveXYId.z = 0;
    for (int k = 0; k <= MAXDEPTH; ++k)
    {
        cs.at(k, 0) = 0.f;
        cs.at(k, 1) = 0.f;
        cs.at(k, 2) = 0.f;
        cs.at(k, 3) = si.viewDistance;
    }
    if (ballot(active) == 0ull)
        return V(0.f, 0.f, 0.f);
    const WalkRay r = makeWalkRay(rayO, rayD - rayO);
    if (active)
        countAdd<COUNT>(cnt.closest, 1);
    countAdd<COUNT>(cnt.wClosest, 1);
    const bool fastBoxes = S.orderedBoxes && (ballot(active && !finiteRay(r)) == 0ull);
    int nbIntersections = 0;
    int cursor = active ? 0 : SOLR_CURSOR_DONE;
    int cur = 0;
    const int nbBoxes = S.nbBoxes;
    Row2 node = boxNode(S, 0);
    while (cur < nbBoxes)
    {
        const int leaf = cur;
        int nbPrimitives;
        bool entered;
        if (!stepGeneral<COUNT>(S, r, fastBoxes, si.viewDistance, cursor, cur, node, nbPrimitives, entered, cnt))
            continue;
        if (nbPrimitives <= 0)
            continue;
        const Row4 L = leafRecord(S, uniform(leaf));
        const int start = uniform(asint(L.d.w));
        for (int k = 0; k < nbPrimitives; ++k)
        {
            const PrimRec rec = leafPrimitive<FEAT>(S, si, L, start, k);
            const int pi = rec.pi;
            const int tag = uniform(asint(rec.head.a.w));
            const int materialId = uniform(asint(rec.head.b.w));
            countAdd<COUNT>(cnt.wPrims, 1);
            Hit h;
            h.intersection = V(0.f, 0.f, 0.f);
            h.normal = V(0.f, 0.f, 0.f);
            h.areas = V(0.f, 0.f, 0.f);
            h.shadowIntensity = 0.f;
            bool i = false;
            if (entered)
            {
                countAdd<COUNT>(cnt.prims, 1);
                i = testPrimitive<false, FEAT>(S, si, rec, tag, r, h);
            }
            float dist = 0.f;
            if (i)
                dist = length(h.intersection - r.o);
            const bool keep = i && dist > ppi.param1;
            if (ballot(keep) == 0ull)
                continue;
            const MaterialHot mh = loadMaterialHot(S, materialId);
            v3 color = V(mh.color.x, mh.color.y, mh.color.z);
            v3 normal = h.normal;
            if (si.graphicsLevel != glNoShading)
            {
                float4 attributes = make_float4(mh.reflection, mh.transparency, mh.refraction, mh.opacity);
                v3 rBlinn = V(0.f, 0.f, 0.f);
                v3 closestColor = V(mh.color.x, mh.color.y, mh.color.z);
                float shadowIntensity = 0.f;
                color = primitiveShader<COUNT, FEAT>(S, keep, index, si, r.o, normal, pi, h.intersection, h.areas, closestColor,
                                                     0, shadowIntensity, rBlinn, attributes, cnt);
            }
            if (keep)
            {
                ++nbIntersections;
                for (int layer = 0; layer < MAXDEPTH; ++layer)
                    if (dist < cs.at(layer, 3))
                    {
                        const float a = dot(r.dn, normal);
                        for (int j = MAXDEPTH - 1; j >= layer; --j)
                            for (int c = 0; c < 4; ++c)
                                cs.at(j + 1, c) = cs.at(j, c);
                        cs.at(layer, 0) = color.x * fabsf(a);
                        cs.at(layer, 1) = color.y * fabsf(a);
                        cs.at(layer, 2) = color.z * fabsf(a);
                        cs.at(layer, 3) = dist;
                        break;
                    }
            }
        }
    }
    v3 color = V(cs.at(0, 0) * si.backgroundColor.w, cs.at(0, 1) * si.backgroundColor.w, cs.at(0, 2) * si.backgroundColor.w);
    if (nbIntersections > 0)
    {
        float D = cs.at(0, 3);
        const int precision = 500;
        const float step = si.viewDistance / (float)precision;
        const float alpha = 1.f / ppi.param2;
        int c = 0;
        for (int k = 0; k < precision && c < MAXDEPTH - 1; ++k)
        {
            if (D > cs.at(c, 3))
            {
                color.x += cs.at(c, 0) * alpha;
                color.y += cs.at(c, 1) * alpha;
                color.z += cs.at(c, 2) * alpha;
            }
            D += step;
            if (D >= cs.at(c + 1, 3))
                ++c;
        }
    }
    return color;
}

/* GS:132-165 */
/* DEVICE_SCOPE: the three bytes go out with device scope - written through to memory, where the copy engine reads a band
 * of the image while other tiles still render (renderer.h, ImageStreaming); ftRGB only (the host streams no other) */
template <bool DEVICE_SCOPE = false>
SOLR_DEV void makeColor(const SceneInfo &si, v3 color, unsigned char *__restrict__ bitmap, int index)
{
    color.x = (color.x > 1.f) ? 1.f : color.x;
    color.y = (color.y > 1.f) ? 1.f : color.y;
    color.z = (color.z > 1.f) ? 1.f : color.z;
    color.x = (color.x < 0.f) ? 0.f : color.x;
    color.y = (color.y < 0.f) ? 0.f : color.y;
    color.z = (color.z < 0.f) ? 0.f : color.z;
    if (DEVICE_SCOPE)
    {
        unsigned char *at = bitmap + (size_t)index * SOLR_COLOR_DEPTH;
        __hip_atomic_store(at, (unsigned char)(color.x * 255.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(at + 1, (unsigned char)(color.y * 255.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(at + 2, (unsigned char)(color.z * 255.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (si.frameBufferType == ftBGR)
    {
        int y = index / si.size.y;
        int x = index % si.size.x;
        int i = (y + 1) * si.size.y - x - 1;
        i *= SOLR_COLOR_DEPTH;
        bitmap[i] = (unsigned char)(color.z * 255.f);
        bitmap[i + 1] = (unsigned char)(color.y * 255.f);
        bitmap[i + 2] = (unsigned char)(color.x * 255.f);
    }
    else
    {
        int i = index * SOLR_COLOR_DEPTH;
        bitmap[i] = (unsigned char)(color.x * 255.f);
        bitmap[i + 1] = (unsigned char)(color.y * 255.f);
        bitmap[i + 2] = (unsigned char)(color.z * 255.f);
    }
}
} // namespace solrdev

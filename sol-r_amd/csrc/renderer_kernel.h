/*
 * renderer_kernel.h - k_standardRenderer and k_walkBound, the templates (included by the files under csrc/rows, which instantiate
 * them; the host side only sees renderer.h).
 *
 *   k_standardRenderer   one wave = one 8x8 pixel tile; primary-ray setup (CudaRayTracer.cu:437-563), the bounce loop,
 *                        and - when no neighbourhood post-process is requested - the float->RGB8 conversion of
 *                        k_default fused in (CudaRayTracer.cu:1057-1073, GeometryShaders.cuh:132-165)
 *   k_walkBound          the walks a recorded frame made, replayed with nothing but the node loop
 */
#ifndef SOLR_RENDERER_KERNEL_H
#define SOLR_RENDERER_KERNEL_H

#include "renderer.h"

using namespace solrdev;

/* VOLUME: the instantiations of the volume camera (ctVolumeRendering) - kernels of their own, so that its trace (a
 * second inlined shader) is not carried, in registers and spills, by every frame of the all-features kernels */
template <int COUNT, int FEAT, bool VOLUME = false>
#ifndef SOLR_GENERIC_WAVES
#define SOLR_GENERIC_WAVES 0 /* experiments: waves per SIMD the instantiations with the texture tier are compiled for (0: as the
                             * others; 3 - 168 registers, 1-51 spills instead of 51-176 - measured within 4 % either way) */
#endif
#ifndef SOLR_WAVES_PER_EU
#define SOLR_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(WAVE, (SOLR_GENERIC_WAVES && (FEAT & F_TEX)) ? SOLR_GENERIC_WAVES : SOLR_WAVES_PER_EU) void k_standardRenderer(const SceneArgs SA, const FrameArgs F,
                                                           PixelRecord *__restrict__ pp,
                                                           int4 *__restrict__ ids, unsigned char *__restrict__ bitmap,
                                                           unsigned long long *__restrict__ counters)
{
    extern __shared__ float ldsStack[];
    const Scene S = makeScene(SA);
    const SceneInfo &si = F.si;
    const int lane = threadIdx.x;
    /* the order is a permutation by construction (k_orderTiles); the clamp keeps a damaged one inside the frame */
    const unsigned entry = F.tileOrder ? F.tileOrder[blockIdx.x] : blockIdx.x;
    if (entry == ORDER_NOTHING)
        return;
    const int tile = (int)min(entry & ORDER_TILE_MASK, (unsigned)F.nbTiles - 1u);
    const int part = (int)((entry >> ORDER_PART_SHIFT) & 31u); /* 0: the whole tile, 1 ... SPLIT_PARTS: one of its square parts */
    unsigned long long clock0 = 0ull;
    if (F.tileClock || F.tileCost)
        clock0 = __builtin_amdgcn_s_memrealtime();
    /* (a division by a number only the host knows is two dozen instructions each time; the frame's reciprocal is one
     * multiplication) */
    const int ty = F.tileMagic ? (int)(__umulhi((unsigned)tile, F.tileMagic) >> F.tileShift) : tile;
    const int tx = tile - ty * F.tilesX;
    const int x = tx * TILE_W + (lane & (TILE_W - 1));
    const int yLocal = ty * TILE_H + (lane >> SOLR_TILE_W_LOG2);
    const int W = si.size.x;
    /* a quadrant wave: only the lanes of its quadrant take part; every lane's path is its own (the walks are
     * wave-synchronous, not wave-dependent), so the pixels come out the same whichever wave renders them */
    const bool mine = part == 0 || ((((lane & (TILE_W - 1)) >> (SOLR_TILE_W_LOG2 - SOLR_SPLIT_LOG2)) |
                                     ((lane >> SOLR_TILE_W_LOG2) >> (6 - SOLR_TILE_W_LOG2 - SOLR_SPLIT_LOG2)) << SOLR_SPLIT_LOG2)) == part - 1;
#ifdef SOLR_PRIORITY_FOR_SPLIT_TILES
    /* experiment (profiles/r4/wave_priority.txt): the waves of a split tile - the frame's critical path - ask the SIMD's
     * arbiter for priority over the waves they share it with */
    if (part != 0)
        __builtin_amdgcn_s_setprio(SOLR_PRIORITY_FOR_SPLIT_TILES);
#endif
#ifdef SOLR_PRIORITY_FOR_FIRST_WAVES
    /* experiment: the cost-ordered launch starts the most expensive tiles first; they also get priority */
    if (F.tileOrder && blockIdx.x < SOLR_PRIORITY_FOR_FIRST_WAVES)
        __builtin_amdgcn_s_setprio(2);
#endif
    const bool inside = mine && (x < W) && (yLocal < F.nbRows);
    const int index0 = inside ? yLocal * W + x : 0;
    const int yGlobal = F.firstRow + yLocal;
    const int gindex = yGlobal * W + x; /* global pixel index: random-buffer addressing */

    /* ctVolumeRendering is dispatched to k_volumeRenderer (CRT:1777-1806, 592-713): the frame of the standard
     * renderer around another trace (rt_device.h launchVolumeRendering), with the rotated-grid offset on every pass
     * (CRT:670-671) and ids whose fourth component is left as it was (CRT:59-61) */
    constexpr bool volume = VOLUME; /* (the host launches these instantiations for that camera and no other) */
    int4 id = make_int4(0, 0, 0, 0);
    bool active = inside;
    const bool refinementPass = si.pathTracingIteration > 0 && si.pathTracingIteration <= NB_MAX_ITERATIONS;
    if (inside && (refinementPass || volume))
    {
        /* progressive refinement: skip pixels whose previous pass ended early (CRT:454-458) */
        id = ids[index0];
        if (refinementPass && si.pathTracingIteration > id.y && id.w == 0)
            active = false;
    }

    ColorStack cs;
    cs.base = ldsStack + lane;
    cs.stride = WAVE;
    cs.cold = 4 * F.stackSlots;
    cs.ldsSlots = F.stackSlots;
    cs.deep = (FEAT & F_STACK) ? F.deepStack + index0 : nullptr;
    cs.deepStride = F.deepStride;

    Counters cnt = {};
    if (COUNT == 2) /* a frame whose walks are recorded (rt_device.h recordWalk): `counters` is the record buffer */
        cnt.record = (char *)counters + (size_t)blockIdx.x * SOLR_WALK_SLOT_BYTES;
    SOLR_T(const unsigned long long tKernel0 = SOLR_NOW();)

    v3 rayO = V(F.ox, F.oy, F.oz);
    v3 rayD = V(F.dx, F.dy, F.dz);
    v3 rotationCenter = V(0.f, 0.f, 0.f);
    if (si.cameraType == ctVR)
        rotationCenter = rayO;
    /* the five-ray camera keeps the primary ray and a colour sum alive across the whole path trace:
     * only the instantiations with F_FULL carry it (the host selects one of them for that camera) */
    const bool antialiasingActivated = (FEAT & F_FULL) && (si.cameraType == ctAntialiazed);

    /* first-hit distance of the previous pass, only needed for the natural
     * depth-of-field jitter of the accumulation passes (CRT:470-479) */
    if (F.ppi.type != ppe_depthOfField && si.pathTracingIteration >= NB_MAX_ITERATIONS)
    {
        const float previousDepth = active ? pp[index0].colorInfo.w : 0.f;
        float a = (F.ppi.param1 / 20000.f);
        long rindex = (long)gindex + si.timestamp % (MAX_BITMAP_SIZE - 2);
        rayO.x += rnd(S, rindex) * previousDepth * a;
        rayO.y += rnd(S, rindex + 1) * previousDepth * a;
    }

    float dof = 0.f;
    if (si.cameraType == ctOrthographic)
    {
        rayD.x = rayO.z * 0.001f * (float)(x - (si.size.x / 2));
        rayD.y = -rayO.z * 0.001f * (float)(yGlobal - (si.size.y / 2));
        rayO.x = rayD.x;
        rayO.y = rayD.y;
    }
    else
    {
        rayD.x = rayD.x - F.stepx * (float)(x - (si.size.x / 2));
        rayD.y = rayD.y + F.stepy * (float)(yGlobal - (si.size.y / 2));
    }
    /* (the camera position is the same for every pixel here, and so is its rotation: taking it from the host
     * instead - three values in scalar registers for as long as the first trace lasts - measured 5 % slower: the
     * scalar registers are the scarce ones, profiles/r3/occupancy_experiments.txt) */
    rayO = vectorRotation(rayO, rotationCenter, F.trig);
    rayD = vectorRotation(rayD, rotationCenter, F.trig);

    v3 color = V(0.f, 0.f, 0.f);
    if (!antialiasingActivated && (volume || si.pathTracingIteration >= NB_MAX_ITERATIONS))
    {
        /* rotated-grid jitter of the accumulation passes, CRT:515-522 */
        const int k = si.pathTracingIteration % 4;
        rayD.x += (k == 0) ? 3.f : (k == 1) ? 5.f : (k == 2) ? -3.f : -5.f;
        rayD.y += (k == 0) ? 5.f : (k == 1) ? -3.f : (k == 2) ? -5.f : 3.f;
    }
    /* ctAntialiazed: four rotated-grid rays with cumulative origin offsets
     * (+3,+5) (+8,+2) (+5,-3) (0,0), then the centre ray, which therefore
     * starts from the same origin as the fourth (CRT:504-514, 523-525);
     * otherwise the centre ray only.  The offsets are re-applied from the
     * unjittered origin with the reference's order of additions. */
    /* ctAnaglyph (k_anaglyphRenderer, CRT:840-950): one trace per eye from origin.x -+ eyeSeparation, no
     * jitter; like the five-ray camera it lives in the F_FULL instantiations only */
    const bool anaglyph = (FEAT & F_FULL) && (si.cameraType == ctAnaglyph);
    /* ctPanoramic (k_fishEyeRenderer, CRT:741-815): the look-at point turns about the eye by
     * angles.y + 2 pi x / W; plain store like the anaglyph camera */
    const bool fishEye = (FEAT & F_FULL) && (si.cameraType == ctPanoramic);
    /* ctVR is dispatched to k_3DVisionRenderer (CRT:1737-1755, 953-1043): side-by-side views of the two
     * eyes, their distance scaled by look-at depth / depth of the focus pixel (F.focusDepth: the value
     * before this frame, see the oracle's visionRendererPixel for the race it stands for) */
    const bool vision = (FEAT & F_FULL) && (si.cameraType == ctVR);
    const bool plainStore = anaglyph || fishEye || vision;
    v3 leftEye = V(0.f, 0.f, 0.f);
    const int nbRays = antialiasingActivated ? 5 : (anaglyph ? 2 : 1);
#pragma unroll 1
    for (int I = 0; I < nbRays; ++I)
    {
        v3 rO = rayO;
        v3 rD = rayD;
        if (anaglyph)
        {
            const float stepx = F.stepx, stepy = F.stepy;
            rO = V((I == 0) ? F.ox - si.eyeSeparation : F.ox + si.eyeSeparation, F.oy, F.oz);
            rD = V(F.dx - stepx * (float)(x - (si.size.x / 2)), F.dy + stepy * (float)(yGlobal - (si.size.y / 2)), F.dz);
            rO = vectorRotation(rO, V(0.f, 0.f, 0.f), F.trig);
            rD = vectorRotation(rD, V(0.f, 0.f, 0.f), F.trig);
        }
        if (vision)
        {
            const float focus = fabsf(F.focusDepth - F.oz);
            const float eyeSeparation = si.eyeSeparation * (F.dz / focus);
            const int halfWidth = si.size.x / 2;
            const float stepx = F.stepx, stepy = F.stepy;
            const v3 eye = V(F.ox, F.oy, F.oz);
            if (x < halfWidth)
            {
                rO = V(F.ox + eyeSeparation, F.oy, F.oz);
                rD.x = F.dx - stepx * (float)(x - (si.size.x / 2) + halfWidth / 2) + si.eyeSeparation;
            }
            else
            {
                rO = V(F.ox - eyeSeparation, F.oy, F.oz);
                rD.x = F.dx - stepx * (float)(x - (si.size.x / 2) - halfWidth / 2) - si.eyeSeparation;
            }
            rD.y = F.dy + stepy * (float)(yGlobal - (si.size.y / 2));
            rD.z = F.dz;
            rO = vectorRotation(rO, eye, F.trig);
            rD = vectorRotation(rD, eye, F.trig);
            dof = F.ppi.param1;
        }
        if (fishEye)
        {
            rO = V(F.ox, F.oy, F.oz);
            rD = V(F.dx, F.dy, F.dz);
            if (si.pathTracingIteration >= NB_MAX_ITERATIONS)
            {
                const int rindex = (gindex + si.timestamp) % (MAX_BITMAP_SIZE - 3);
                const float a = (float)si.pathTracingIteration / (float)si.maxPathTracingIterations;
                const float depth = active ? pp[index0].colorInfo.w : 0.f;
                rD.x += rnd(S, rindex) * depth * F.ppi.param2 * a;
                rD.y += rnd(S, rindex + 1) * depth * F.ppi.param2 * a;
                rD.z += rnd(S, rindex + 2) * depth * F.ppi.param2 * a;
            }
            rD.y = rD.y + F.stepy * (float)(yGlobal - (si.size.y / 2));
            const float stepx = 2.f * 3.14159265358979323846f / (float)si.size.x;
            const float turn = F.ay + stepx * (float)x;
            Trig t;
            t.cx = 1.f; /* cos(0), sin(0): exact in every libm */
            t.sx = 0.f;
            t.cy = cos_f(turn);
            t.sy = sin_f(turn);
            t.cz = 1.f;
            t.sz = 0.f;
            rD = vectorRotation(rD, rO, t);
        }
        if (antialiasingActivated)
        {
            rO.x += 3.f;
            rO.y += 5.f;
            if (I >= 1)
            {
                rO.x += 5.f;
                rO.y += -3.f;
            }
            if (I >= 2)
            {
                rO.x += -3.f;
                rO.y += -5.f;
            }
            if (I >= 3)
            {
                rO.x += -5.f;
                rO.y += 3.f;
            }
        }
        v3 c;
        if constexpr (volume)
        {
            c = launchVolumeRendering<COUNT, FEAT>(S, active, gindex, rO, rD, si, F.ppi, id, cs, cnt);
            c = V(0.f + c.x, 0.f + c.y, 0.f + c.z);
        }
        else
            c = launchRayTracing<COUNT, FEAT>(S, active, gindex, rO, rD, si, dof, id, cs, cnt);
        if (anaglyph)
        {
            if (I == 0)
                leftEye = c;
            else
                color = V((leftEye.x * 0.299f + leftEye.y * 0.587f + leftEye.z * 0.114f) + 0.f, 0.f + c.y, 0.f + c.z);
        }
        else
            color = color + c;
    }

    SOLR_T(const unsigned long long tEpilogue0 = SOLR_NOW();)
    /* The pixel's coordinates once more, from the wave's tile number and the lane's position in the wave,
     * through values the compiler cannot identify with the ones above: what the prologue computed would
     * otherwise stay alive - in vector registers, in practice in scratch - through the whole path trace
     * just to address the stores below. */
    int tileAgain = tile, partAgain = part, laneAgain = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+s"(tileAgain), "+s"(partAgain), "+v"(laneAgain));
    const int tyAgain = F.tileMagic ? (int)(__umulhi((unsigned)tileAgain, F.tileMagic) >> F.tileShift) : tileAgain;
    const int xAgain = (tileAgain - tyAgain * F.tilesX) * TILE_W + (laneAgain & (TILE_W - 1));
    const int yAgain = tyAgain * TILE_H + (laneAgain >> SOLR_TILE_W_LOG2);
    const int index = ((xAgain < si.size.x) && (yAgain < F.nbRows)) ? yAgain * si.size.x + xAgain : 0;
    const int gindexAgain = (F.firstRow + yAgain) * si.size.x + xAgain;

    if ((!plainStore || vision) && si.advancedIllumination == aiRandomIllumination)
    {
        int rindex = (gindexAgain + si.timestamp) % MAX_BITMAP_SIZE;
        float rv = rnd(S, rindex);
        color.x += si.backgroundColor.x * rv * 5.f;
        color.y += si.backgroundColor.y * rv * 5.f;
        color.z += si.backgroundColor.z * rv * 5.f;
    }
    if (antialiasingActivated)
    {
        color.x /= 5.f;
        color.y /= 5.f;
        color.z /= 5.f;
    }

    if (active)
    {
        float4 ppColor = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 ppScene = make_float4(0.f, 0.f, 0.f, 0.f);
        if (si.pathTracingIteration > 0)
        {
            ppColor = pp[index].colorInfo;
            ppScene = pp[index].sceneInfo;
        }
        if (si.pathTracingIteration == 0)
            ppColor.w = dof;
        if (plainStore) /* plain store / accumulate, the last-sample record is not touched (CRT:801-812, 936-947) */
        {
            if (si.pathTracingIteration <= NB_MAX_ITERATIONS)
            {
                ppColor.x = color.x;
                ppColor.y = color.y;
                ppColor.z = color.z;
            }
            else
            {
                ppColor.x += color.x;
                ppColor.y += color.y;
                ppColor.z += color.z;
            }
        }
        else if (si.pathTracingIteration <= NB_MAX_ITERATIONS)
        {
            ppColor.x = color.x;
            ppColor.y = color.y;
            ppColor.z = color.z;
            ppScene.x = color.x;
            ppScene.y = color.y;
            ppScene.z = color.z;
        }
        else
        {
            ppScene.x = (id.z > 0) ? fmaxf(ppScene.x, color.x) : color.x;
            ppScene.y = (id.z > 0) ? fmaxf(ppScene.y, color.y) : color.y;
            ppScene.z = (id.z > 0) ? fmaxf(ppScene.z, color.z) : color.z;
            ppColor.x += ppScene.x;
            ppColor.y += ppScene.y;
            ppColor.z += ppScene.z;
        }
        pp[index].colorInfo = ppColor;
        pp[index].sceneInfo = ppScene;
        if ((FEAT & F_STREAM) && (F.fuseDefault & 4)) /* (ImageStreaming with the ids: written through to memory, where the copy engine reads the band) */
        {
            typedef int FourInts __attribute__((ext_vector_type(4)));
            const FourInts four = {id.x, id.y, id.z, id.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ids + index), "v"(four) : "memory");
        }
        else
            ids[index] = id;

        if (F.fuseDefault)
        {
            v3 c = V(ppColor.x, ppColor.y, ppColor.z);
            if (si.pathTracingIteration > NB_MAX_ITERATIONS)
            {
                float d = (float)(si.pathTracingIteration - NB_MAX_ITERATIONS + 1);
                c.x /= d;
                c.y /= d;
                c.z /= d;
            }
            if ((FEAT & F_STREAM) && (F.fuseDefault & 2))
                makeColor<true>(si, c, bitmap, index);
            else
                makeColor(si, c, bitmap, index);
        }
    }
    if ((FEAT & F_STREAM) && (F.fuseDefault & 2)) /* ImageStreaming (renderer.h): this tile's bytes are out; is its row, is its band? */
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned units = partAgain ? 1u : (unsigned)SPLIT_PARTS;
        unsigned before = 0u;
        if (laneAgain == 0)
            before = __hip_atomic_fetch_add(F.rowDone + 64 * tyAgain, units, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        before = (unsigned)__builtin_amdgcn_readfirstlane((int)before);
        if (before + units == F.streamSerial * (unsigned)(F.tilesX * SPLIT_PARTS) && laneAgain == 0)
        {
            const StreamPlan *plan = F.streamPlan;
            int band = 0;
            while (band + 1 < plan->bands && tyAgain >= plan->firstRow[band + 1])
                ++band;
            const int rows = plan->firstRow[band + 1] - plan->firstRow[band];
            const unsigned rowsBefore = __hip_atomic_fetch_add(plan->bandDone + 64 * band, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rowsBefore + 1u == F.streamSerial * (unsigned)rows)
                __hip_atomic_store(plan->hostWord + band, F.streamSerial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

    if (F.tileClock && laneAgain == 0)
    {
        F.tileClock[2 * tileAgain] = clock0;
        F.tileClock[2 * tileAgain + 1] = __builtin_amdgcn_s_memrealtime();
    }
    if (F.tileCost && laneAgain == 0) /* what this tile cost, for the launch order of the next frames */
    {
        /* a quadrant takes about 0.8 of what its whole tile takes (profiles/r1: DESIGN.md section 5): reported
         * as twice its own time, a split tile stays among the expensive ones and stays split (four waves write
         * the same word; any of them will do) */
        const unsigned cost = (unsigned)(__builtin_amdgcn_s_memrealtime() - clock0);
        F.tileCost[tileAgain] = partAgain ? (unsigned)(SOLR_SPLIT_LOG2 + 1) * cost : cost;
    }
#ifdef SOLR_TIMING
    if (COUNT == 0 && laneAgain == 0 && counters)
    {
        /* one record per workgroup, summed by the host (atomics on one address would serialise the frame) */
        unsigned long long *slot = counters + 16 + 16ull * blockIdx.x;
        slot[0] += SOLR_NOW() - tKernel0;
        slot[8] += cnt.tShade;
        slot[9] += cnt.tTrace;
        slot[10] += SOLR_NOW() - tEpilogue0;
        slot[1] += cnt.tClosest;
        slot[2] += cnt.tShadow;
        slot[3] += cnt.tNode;
        slot[4] += cnt.tLeaf;
        slot[5] += (unsigned long long)cnt.nAdvance;
        slot[6] += (unsigned long long)cnt.nLeaf;
        slot[7] += 1ull;
        slot[11] += cnt.tClosestPrimary;
        slot[12] += cnt.tAgain;
        slot[13] += ((unsigned long long)cnt.nAgain << 32) | cnt.nAgainLanes;
        slot[14] += ((unsigned long long)cnt.nAdvanceFirst << 32) | cnt.nAdvanceAgain;
        slot[15] += (unsigned long long)cnt.nChecked;
    }
#endif
    if (COUNT == 2 && lane == 0)
        ((int4 *)cnt.record)[0] = make_int4((int)cnt.ordinal, 0, 0, 0);
    if (COUNT == 1)
    {
        unsigned int vals[4] = {cnt.closest, cnt.shadow, cnt.boxes, cnt.prims};
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            unsigned int v = vals[k];
            for (int off = 32; off > 0; off >>= 1)
                v += __shfl_xor(v, off, 64);
            if (lane == 0)
                atomicAdd(&counters[k], (unsigned long long)v);
        }
        if (lane == 0)
        {
            atomicAdd(&counters[4], (unsigned long long)cnt.wNodes);
            atomicAdd(&counters[5], (unsigned long long)cnt.wPrims);
            atomicAdd(&counters[6], (unsigned long long)cnt.wClosest);
            atomicAdd(&counters[7], (unsigned long long)cnt.wShadow);
        }
    }
}

/* The walk's own ceiling (rt_device.h, WalkRecord): the walks a recorded frame made, replayed with nothing but the node
 * loop.  Same grid as the recorded launch - workgroup b replays what workgroup b recorded - same dynamic LDS (so that as
 * many waves share a CU as in the renderer), same FEAT (the two-bank or the three-bank loop).  `visits` keeps the loop
 * observable: leaf entries per lane. */
template <int FEAT>
__global__ __launch_bounds__(WAVE, SOLR_WAVES_PER_EU) void k_walkBound(const SceneArgs SA, const char *__restrict__ records,
                                                                       unsigned *__restrict__ visits, unsigned *__restrict__ skipped)
{
    extern __shared__ float ldsStack[];
    const Scene S = makeScene(SA);
    const int lane = threadIdx.x;
    const char *slot = records + (size_t)blockIdx.x * SOLR_WALK_SLOT_BYTES;
    const int4 *head = (const int4 *)slot;
    const float4 *rays = (const float4 *)(slot + 16 * (SOLR_WALK_SLOTS + 1));
    const int recorded = __builtin_amdgcn_readfirstlane(head[0].x);
    const int walks = recorded < SOLR_WALK_SLOTS ? recorded : SOLR_WALK_SLOTS;
    unsigned entries = 0;
    int left_out = recorded - walks;
    for (int j = 0; j < walks; ++j)
    {
        const int4 h = head[1 + j];
        const int kind = __builtin_amdgcn_readfirstlane(h.x);
        if (kind == WALK_GENERAL)
        {
            ++left_out;
            continue;
        }
        const float4 a = rays[((size_t)j * 64 + lane) * 2], b = rays[((size_t)j * 64 + lane) * 2 + 1];
        const int doneAfter = __float_as_int(b.w);
        const bool took_part = doneAfter >= 0;
        /* (a lane that took no part gets a harmless ray: its cursor is done from the start) */
        const WalkRay r = makeWalkRay(V(a.x, a.y, a.z), took_part ? V(b.x, b.y, b.z) : V(1.f, 1.f, 1.f));
        Scene W = S;
        if (__builtin_amdgcn_readfirstlane(h.y))
        {
            const int octant = __builtin_amdgcn_readfirstlane(h.z);
            W.offBoxes = S.offBoxesFree + 2u * (unsigned)(octant * S.nbBoxesFree);
            W.offLeaf = S.offLeafFree + 4u * (unsigned)(octant * S.nbBoxesFree);
            W.nbBoxes = S.nbBoxesFree;
        }
        if (__builtin_amdgcn_readfirstlane(h.w) & 1) /* the walk took the thin copy of its list (rt_device.h tightRay) */
            W.offBoxes += __builtin_amdgcn_readfirstlane(h.y) ? 16u * (unsigned)S.nbBoxesFree + 2u : 2u * (unsigned)S.nbBoxes + 2u;
        /* the form of the node loop the recorded walk took (rt_device.h walkOrder: bits 1-2 of the record's fourth word) */
        const int order = (FEAT & F_DEEP) ? (__builtin_amdgcn_readfirstlane(h.w) >> 1) & 3 : 0;
        if (order)
        {
            W.offBoxes += 32u * (unsigned)S.nbBoxesFree + 4u;
            W.nbBoxes = S.nbBoxesFree << 5; /* (such a walk's cursors count bytes: rt_device.h SOLR_NEXT_BY_BYTES) */
        }
        const PackedRay pr = packRay(r);
        const float cutOff = a.w;
        int cursor = took_part ? 0 : SOLR_CURSOR_DONE;
        int cur = 0, visit = 0;
        while (cur < W.nbBoxes)
        {
            if (ballot(cursor != SOLR_CURSOR_DONE) == 0ull)
                break;
            int nbPrimitives;
            bool entered;
            int leaf;
            if ((FEAT & F_DEEP) && order == 1)
                leaf = advanceTidyDeepSorted(W, pr, cutOff, cursor, cur, nbPrimitives, entered);
            else if ((FEAT & F_DEEP) && order == 2)
                leaf = advanceTidyDeepReversed(W, pr, cutOff, cursor, cur, nbPrimitives, entered);
            else
                leaf = advanceTidy<FEAT>(W, pr, cutOff, cursor, cur, nbPrimitives, entered);
            if (leaf < 0)
                break;
            ++visit;
            entries += entered ? 1u : 0u;
            if (doneAfter == visit)
                cursor = SOLR_CURSOR_DONE;
        }
    }
    visits[(size_t)blockIdx.x * WAVE + lane] = entries;
    if (lane == 0 && left_out > 0)
        atomicAdd(skipped, (unsigned)left_out);
}

#endif

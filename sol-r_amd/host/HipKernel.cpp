/*
 * HipKernel.cpp - see HipKernel.h.  Frame protocol after the reference's
 * CudaKernel.cpp:116-166 (device lifetime), :174-302 (render_begin) and
 * :304-312 (render_end, minus the OpenGL blit).
 */
#include "HipKernel.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>
#include <iostream>

#include "../../include/solr_hip.h"

namespace solr
{
HipKernel::HipKernel()
    : GPUKernel()
    , m_sharedMemSize(0)
    , m_deviceInitialized(false)
{
    /* reference default 12x12 (CudaKernel.cpp:85-88); informational here */
    m_blockSize = make_vec4i(8, 8, 1, 0);
    m_occupancyParameters = make_vec2i(1, 1);
    m_gpuDescription = "HIP device";
}

HipKernel::~HipKernel()
{
    releaseDevice();
}

void HipKernel::initBuffers()
{
    GPUKernel::initBuffers();
    queryDevice();
    initializeDevice();
}

void HipKernel::cleanup()
{
    GPUKernel::cleanup();
    releaseDevice();
}

void HipKernel::queryDevice()
{
    int n = solr_hip_device_count();
    m_gpuDescription = n > 0 ? "AMD Instinct (HIP), " + std::to_string(n) + " device(s) visible" : "no HIP device";
}

void HipKernel::initializeDevice()
{
    if (m_deviceInitialized)
        releaseDevice();
    initialize_scene(m_occupancyParameters, m_sceneInfo, NB_MAX_PRIMITIVES, NB_MAX_LAMPS, NB_MAX_MATERIALS);
    reshape_scene(m_occupancyParameters, m_sceneInfo);
    m_deviceInitialized = true;
}

void HipKernel::releaseDevice()
{
    syncHost(); /* rotations that only the device has seen would be lost with it */
    fetchPrimitiveIds(); /* and so would the last frame's ids */
    fetchBitmap();       /* ... and its image, where only the caller of SolR_RunKernel has it so far */
    flushFrames();
    if (m_bitmapView && !m_bitmap.empty())
    {
        /* the engine's page-locked images go with it.  They are as large as the frame rendered last; m_bitmap only
         * ever grows (GPUKernel::render_begin) and may be larger - after a reshape to a smaller size */
        const size_t frame = (size_t)m_sceneInfo.size.x * (size_t)m_sceneInfo.size.y * (size_t)SOLR_COLOR_DEPTH;
        memcpy(m_bitmap.data(), m_bitmapView, frame < m_bitmap.size() ? frame : m_bitmap.size());
    }
    m_bitmapView = nullptr;
    if (m_deviceInitialized)
        finalize_scene(m_occupancyParameters);
    m_deviceInitialized = false;
}

void HipKernel::reshape()
{
    flushFrames();
    m_bitmapView = nullptr; /* the engine's images are re-made for the new size */
    GPUKernel::reshape();
    m_idsOnDevice = false; /* the buffers are re-made for the new size */
    m_bitmapOnDevice = false;
    if (m_deviceInitialized)
        reshape_scene(m_occupancyParameters, m_sceneInfo);
}

int HipKernel::lastError(std::string *message)
{
    char buf[512];
    buf[0] = 0;
    int code = solr_hip_last_error(buf, sizeof(buf));
    if (message)
        *message = buf;
    return code;
}

void HipKernel::render_begin(const float timer)
{
    /* SOLR_HIP_DEBUG_TIMING: the host side of a frame, step by step */
    static const bool timing = getenv("SOLR_HIP_DEBUG_TIMING") != nullptr;
    auto last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing)
            return;
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        if (ms >= 0.05)
            fprintf(stderr, "solr host: %-30s %8.2f ms\n", what, ms);
        last = now;
    };
    GPUKernel::render_begin(timer);
    mark("render_begin: base class");
    if (m_deterministicSeed < 0)
    {
        /* one process of several (solr_hip_comm_init): the timestamp the base class has just drawn from rand() has to
         * be every rank's - drawn from a generator that all of them seeded alike instead */
        const unsigned shared = solr_hip_comm_shared_seed();
        if (shared != m_sharedSeed)
        {
            m_sharedSeed = shared;
            m_sharedState = shared;
        }
        if (shared != 0)
        {
            m_sharedState = m_sharedState * 1664525u + 1013904223u;
            m_sceneInfo.timestamp = (int)((m_sharedState >> 8) % 10000u);
        }
    }
    if (m_refresh)
    {
        if (!m_primitivesTransfered)
            syncHost(); /* somebody asked for a fresh upload while rotations were pending on the device */
        Frame &f = frameAsIs();
        int nbBoxes = f.nbActiveBoxes;
        int nbPrimitives = f.nbActivePrimitives;
        int nbLamps = f.nbActiveLamps;
        int nbMaterials = m_nbActiveMaterials + 1;

        if (!m_primitivesTransfered)
        {
            h2d_scene(m_occupancyParameters, m_hBoundingBoxes.data(), nbBoxes, m_hPrimitives.data(), nbPrimitives,
                      m_hLamps.data(), nbLamps);
            mark("render_begin: h2d_scene");
            h2d_lightInformation(m_occupancyParameters, m_lightInformation.data(), m_lightInformationSize);
            solr_hip_set_movable(m_hMovable.data(), (int)m_hMovable.size());
            m_primitivesTransfered = true;
            m_hostTouched = false; /* device and host hold the same scene from here on */
            mark("render_begin: lights, movable flags");
        }
        if (!m_randomsTransfered)
        {
            if (m_hRandoms.size() > (size_t)MAX_BITMAP_SIZE) /* a frame beyond the reference's 1920 x 1080 */
                solr_hip_h2d_randoms_sized(m_hRandoms.data(), (long)m_hRandoms.size());
            else
                h2d_randoms(m_occupancyParameters, m_hRandoms.data());
            m_randomsTransfered = true;
            mark("render_begin: randoms");
        }
        if (!m_materialsTransfered)
        {
            realignTexturesAndMaterials();
            h2d_materials(m_occupancyParameters, m_hMaterials.data(), nbMaterials);
            m_materialsTransfered = true;
            mark("render_begin: materials");
        }
        if (!m_texturesTransfered)
        {
            h2d_textures(m_occupancyParameters, NB_MAX_TEXTURES, m_hTextures);
            m_texturesTransfered = true;
            mark("render_begin: textures");
        }

        vec4i objects = make_vec4i(nbBoxes, nbPrimitives, nbLamps, m_lightInformationSize);
        SceneInfo sceneInfo = m_sceneInfo;
        /* draft-mode overrides, CudaKernel.cpp:287-291 */
        if (m_sceneInfo.draftMode && m_sceneInfo.pathTracingIteration == 0)
            sceneInfo.graphicsLevel = glNoShading;
        if (m_sceneInfo.draftMode && m_sceneInfo.pathTracingIteration == m_sceneInfo.maxPathTracingIterations)
            sceneInfo.cameraType = ctAntialiazed;

        /* one frame at a time: the frame's waves say when a band of rows is complete, and render_end sends every band
         * off to the host - m_bitmap or SolR_RunKernel's array - while the rows below still render (include/solr_hip.h,
         * solr_hip_stream_next_image) */
        const bool stream = m_flights == 1 && solr_hip_stream_next_image(1) == 1;
        cudaRender(m_occupancyParameters, m_blockSize, sceneInfo, objects, m_postProcessingInfo, m_viewPos, m_viewDir,
                   m_angles);
        mark("render_begin: cudaRender");
        m_streamed = stream && solr_hip_stream_next_image(-1) == 1;
        if (m_flights > 1)
        {
            /* frames in flight: the image starts for the host behind the kernel, on the engine's copy stream */
            const int ticket = solr_hip_d2h_image_async();
            if (ticket >= 0)
                m_tickets.push_back(ticket);
        }
    }
    m_refresh = (m_sceneInfo.pathTracingIteration < m_sceneInfo.maxPathTracingIterations);
}

bool HipKernel::deviceRotatePrimitives(const vec3f &center, const vec3f &cosA, const vec3f &sinA)
{
    if (!m_deviceInitialized)
        return false;
    const float c[3] = {center.x, center.y, center.z};
    const float co[3] = {cosA.x, cosA.y, cosA.z};
    const float si[3] = {sinA.x, sinA.y, sinA.z};
    return solr_hip_rotate_primitives(c, co, si, m_sceneInfo.viewDistance) == 1;
}

int HipKernel::deviceBuildTree(const std::vector<Primitive> &primitives, const std::vector<unsigned char> &emissive,
                               const vec3f &minPos, const vec3f &maxPos, float viewDistance,
                               std::vector<BoundingBox> &boxes, std::vector<int> &order, int &nbLamps)
{
    if (solr_hip_device_count() < 1)
        return -2;
    const int n = (int)primitives.size();
    /* every level has at most as many boxes as the one below, the top one box more: (depth + 1) n + 1 at most.
     * Uninitialised memory for that worst case (a value-initialised vector of it is 58 MB of zeros for 100 k
     * primitives, twice the time of the build itself); the nodes that exist are copied out */
    const size_t capacity = (size_t)n * 12 + 16;
    std::unique_ptr<BoundingBox[]> raw(new BoundingBox[capacity]);
    order.resize((size_t)n);
    const float mn[3] = {minPos.x, minPos.y, minPos.z}, mx[3] = {maxPos.x, maxPos.y, maxPos.z};
    int nbBoxes = 0;
    const int depth = solr_hip_build_tree(primitives.data(), emissive.data(), n, mn, mx, viewDistance, raw.get(), (int)capacity,
                                          order.data(), &nbBoxes, &nbLamps);
    if (depth < 1)
    {
        boxes.clear();
        order.clear();
        return depth;
    }
    boxes.assign(raw.get(), raw.get() + nbBoxes);
    return depth;
}

bool HipKernel::primitivesFromDevice(Frame &f)
{
    if (!m_deviceInitialized)
        return false;
    /* 8 rows of 4 floats per flattened primitive, sol-r_amd/csrc/scene_layout.h: p0 | size | p1 | p2 | n0 | n1 | n2 | uv */
    const int rows = solr_hip_read_primitives(nullptr, 0);
    if (rows <= 0 || (size_t)rows != 8 * m_hPrimitives.size())
        return false; /* not the scene the flattened arrays describe */
    std::vector<float> data((size_t)rows * 4);
    if (solr_hip_read_primitives(data.data(), rows) != rows)
        return false;
    /* nothing is written before everything is known to be writable: a half-updated store could not fall back
     * to the replay */
    std::vector<CPUPrimitive *> target(m_hPrimitives.size(), nullptr);
    for (size_t i = 0; i < m_hPrimitives.size(); ++i)
        if (m_hMovable[i] && !(target[i] = f.primitives.lookup((unsigned int)m_hPrimitives[i].index)))
            return false;
    for (size_t i = 0; i < m_hPrimitives.size(); ++i)
    {
        CPUPrimitive *p = target[i];
        if (!p)
            continue;
        const float *r = &data[i * 32];
        p->p0 = make_vec3f(r[0], r[1], r[2]);
        p->p1 = make_vec3f(r[8], r[9], r[10]);
        p->p2 = make_vec3f(r[12], r[13], r[14]);
        p->n0 = make_vec3f(r[16], r[17], r[18]);
        p->n1 = make_vec3f(r[20], r[21], r[22]);
        p->n2 = make_vec3f(r[24], r[25], r[26]);
    }
    return true;
}

void HipKernel::render_end()
{
    /* the image now, the ids when somebody asks (fetchPrimitiveIds): they stay valid on the device until
     * the next frame is rendered into the same buffers */
    if (m_flights > 1)
    {
        /* frames in flight: the oldest read-backs are delivered until fewer than m_flights are under way - the
         * frame render_begin has just launched keeps rendering.  (The ids on the device - getPrimitiveAt - are the
         * NEWEST frame's, up to m_flights - 1 frames ahead of the image on show.)
         * While nothing has been delivered since the pipeline was switched on (or the frame re-shaped) the oldest
         * frame is waited for at once: every render_end that returns hands the caller a rendered image, never the
         * zeros or the stale frame from before the switch; the lag builds up over the following calls. */
        while ((int)m_tickets.size() >= m_flights || (m_bitmapView == nullptr && !m_tickets.empty()))
        {
            deliver(m_tickets.front());
            m_tickets.pop_front();
        }
        m_idsOnDevice = true;
        m_bitmapOnDevice = false;
        return;
    }
    m_bitmapView = nullptr;
    /* (the frame's bands leave for m_bitmap as they are complete) */
    if (!(m_streamed && solr_hip_d2h_streamed_image(m_bitmap.data()) == 1))
        d2h_bitmap(m_occupancyParameters, m_sceneInfo, m_bitmap.data(), nullptr);
    m_streamed = false;
    m_idsOnDevice = true;
    m_bitmapOnDevice = false;
}

/* SolR_RunKernel's frame (SolRStub.cpp:154-164): the reference waits for the frame, copies it to m_bitmap and copies
 * m_bitmap to the caller - 6 MB through the host's caches a second time, 0.2 ms of a 0.6 ms frame.  Here the device's
 * image goes straight into the caller's array and m_bitmap is brought up to date when somebody asks for it (getBitmap
 * -> fetchBitmap): 0.39 ms, what render_begin + render_end take (profiles/r6/api_frame_bands.txt). */
void HipKernel::render_end(BitmapBuffer *image)
{
    if (m_flights > 1 || !image || !m_deviceInitialized)
    {
        GPUKernel::render_end(image);
        return;
    }
    m_bitmapView = nullptr;
    /* (the frame's bands leave for the caller's array as they are complete) */
    if (!(m_streamed && solr_hip_d2h_streamed_image(image) == 1))
        d2h_bitmap(m_occupancyParameters, m_sceneInfo, image, nullptr);
    m_streamed = false;
    m_idsOnDevice = true;
    m_bitmapOnDevice = true;
}

void HipKernel::fetchBitmap()
{
    if (!m_bitmapOnDevice || !m_deviceInitialized || m_bitmap.empty())
        return;
    m_bitmapOnDevice = false;
    d2h_bitmap(m_occupancyParameters, m_sceneInfo, m_bitmap.data(), nullptr);
}

void HipKernel::deliver(int ticket)
{
    const BitmapBuffer *image = solr_hip_image_wait(ticket);
    if (image)
        m_bitmapView = const_cast<BitmapBuffer *>(image);
}

void HipKernel::setFramesInFlight(int n)
{
    flushFrames();
    m_flights = n < 1 ? 1 : (n > 4 ? 4 : n);
    /* the engine's own buffer sets: one while the host lags a single frame, two beyond - a third render stream
     * ends up sharing a hardware queue with the copy stream (sol-r_amd/csrc/solr_image_ring.hip, solr_hip_d2h_image_async);
     * the depth beyond that is host lag, which the second RGB image of every set absorbs */
    solr_hip_set_frames_in_flight(m_flights <= 2 ? 1 : 2);
    if (m_flights == 1)
        m_bitmapView = nullptr;
}

/* occupancyParameters.x: how many devices of this process the frame is shared out over (the reference's
 * CudaKernel.cpp:90 - a member nobody can set there; include/solr_hip.h initialize_scene).  Takes the device down
 * and up again; everything is uploaded anew with the next frame. */
void HipKernel::setGpuCount(int n)
{
    n = n < 1 ? 1 : n;
    if (n == m_occupancyParameters.x)
        return;
    const bool up = m_deviceInitialized;
    if (up)
        releaseDevice();
    m_occupancyParameters.x = n;
    if (up)
    {
        initializeDevice();
        setFramesInFlight(m_flights);
        m_primitivesTransfered = m_materialsTransfered = m_texturesTransfered = m_randomsTransfered = false;
        m_refresh = true;
    }
}

int HipKernel::getGpuCount() const
{
    return m_deviceInitialized ? solr_hip_gpu_count() : m_occupancyParameters.x;
}

void HipKernel::flushFrames()
{
    while (!m_tickets.empty())
    {
        deliver(m_tickets.front());
        m_tickets.pop_front();
    }
}

void HipKernel::fetchPrimitiveIds()
{
    if (!m_idsOnDevice || !m_deviceInitialized)
        return;
    d2h_bitmap(m_occupancyParameters, m_sceneInfo, nullptr, m_hPrimitivesXYIds.data());
    m_idsOnDevice = false;
}

void HostOnlyKernel::render_begin(const float timer)
{
    GPUKernel::render_begin(timer); /* host-side part of the frame protocol (timestamp, random buffer) */
    m_failed = true;
    std::cerr << "HostOnlyKernel::render_begin: this engine has no device; rendering requires the HIP engine"
              << std::endl;
}

void HostOnlyKernel::render_end()
{
    m_failed = true;
}

int HostOnlyKernel::lastError(std::string *message)
{
    if (m_failed && message)
        *message = "host-only engine cannot render";
    return m_failed ? -1 : 0;
}

void ReplayKernel::render_begin(const float timer)
{
    HostOnlyKernel::render_begin(timer);
    if (!m_primitivesTransfered)
        syncHost();
    m_primitivesTransfered = true; /* the upload a real engine does here */
    m_hostTouched = false;
}

bool ReplayKernel::deviceRotatePrimitives(const vec3f &, const vec3f &, const vec3f &)
{
    ++m_claimed;
    return true;
}

/* reference: GPUKernel.cpp:115-128 (compile-time there, run-time here) */
static std::string gEngineName = "hip";

void SingletonKernel::selectEngine(const char *name)
{
    destroy();
    gEngineName = name ? name : "hip";
}

void SingletonKernel::destroy()
{
    delete m_kernel;
    m_kernel = nullptr;
}

GPUKernel *SingletonKernel::kernel()
{
    if (!m_kernel)
    {
        if (gEngineName == "host-only")
            m_kernel = new HostOnlyKernel();
        else if (gEngineName == "host-replay")
            m_kernel = new ReplayKernel();
        else
            m_kernel = new HipKernel();
    }
    return m_kernel;
}
}

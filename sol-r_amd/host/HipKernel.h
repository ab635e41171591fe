/*
 * HipKernel.h - the engine class that drives the MI355X device layer.
 *
 * Takes the place of the reference's solr::CudaKernel
 * (reference: solr/engines/cuda/CudaKernel.h:27-72, CudaKernel.cpp:81-389):
 * a GPUKernel subclass whose render_begin performs the dirty-flag driven
 * uploads and launches the frame through the ten extern "C" entry points of
 * include/solr_hip.h, and whose render_end brings the bitmap and primitive
 * ids back.  No OpenGL calls (the reference blits in render_end,
 * CudaKernel.cpp:313-388; see INTEGRATION.md for where that block goes).
 */
#pragma once

#include <deque>

#include "GPUKernel.h"

namespace solr
{
class HipKernel : public GPUKernel
{
public:
    HipKernel();
    ~HipKernel();

    void initBuffers() override;
    void cleanup() override;
    void reshape() override;
    void queryDevice() override;
    std::string getGPUDescription() override { return m_gpuDescription; }

    /* reference: CudaKernel.h:44-47 */
    void initializeDevice();
    void releaseDevice();
    void resetBoxesAndPrimitives() {}

    /* reference: CudaKernel.cpp:174-302 / 304-389 */
    void render_begin(const float timer) override;
    void render_end() override;
    void render_end(BitmapBuffer *image) override;
    int lastError(std::string *message = nullptr) override;
    void setFramesInFlight(int n) override;
    int getFramesInFlight() const override { return m_flights; }
    void flushFrames() override;
    void setGpuCount(int n) override;
    int getGpuCount() const override;

    /* reference: CudaKernel.h:59-65; accepted and forwarded, the wave64 tile
     * shape is the engine's choice */
    void setBlockSize(int x, int y, int z)
    {
        m_blockSize.x = x;
        m_blockSize.y = y;
        m_blockSize.z = z;
    }
    void setSharedMemSize(int sharedMemSize) { m_sharedMemSize = sharedMemSize; }

protected:
    /* rotatePrimitives on the resident scene, solr_hip_rotate_primitives (include/solr_hip.h) */
    bool deviceRotatePrimitives(const vec3f &center, const vec3f &cosA, const vec3f &sinA) override;
    void fetchPrimitiveIds() override;
    void fetchBitmap() override;
    bool primitivesFromDevice(Frame &f) override;
    int deviceBuildTree(const std::vector<Primitive> &primitives, const std::vector<unsigned char> &emissive,
                        const vec3f &minPos, const vec3f &maxPos, float viewDistance, std::vector<BoundingBox> &boxes,
                        std::vector<int> &order, int &nbLamps) override;

private:
    vec4i m_blockSize;
    int m_sharedMemSize;
    bool m_deviceInitialized;
    bool m_idsOnDevice = false;
    bool m_streamed = false;       /* the frame render_begin launched counts its tiles: render_end takes its image band by band */
    bool m_bitmapOnDevice = false; /* render_end(image) delivered to the caller's array: m_bitmap follows when asked */
    unsigned m_sharedSeed = 0, m_sharedState = 0; /* solr_hip_comm_shared_seed and the generator it seeds (render_begin) */
    int m_flights = 1;            /* frames in flight through render_begin / render_end (setFramesInFlight) */
    std::deque<int> m_tickets;    /* read-backs under way, oldest first (solr_hip_d2h_image_async) */
    void deliver(int ticket);
};

/* Scene store without a device: everything up to compactBoxes works, any
 * attempt to render fails loudly.  Exists so that the host logic can be
 * tested on machines without a GPU; it is NOT a CPU rendering fallback. */
class HostOnlyKernel : public GPUKernel
{
public:
    void render_begin(const float timer) override;
    void render_end() override;
    int lastError(std::string *message = nullptr) override;

protected:
    bool m_failed = false;
};

/* Test double for the animated-scene protocol (GPUKernel::rotatePrimitives / syncHost): a host-only
 * store that claims, like an engine with a resident scene would, to have uploaded the scene at the first
 * frame and to apply every rotation "over there" - so that the lazy replay of the pending rotations can be
 * checked against the eager host route without a GPU.  Renders nothing. */
class ReplayKernel : public HostOnlyKernel
{
public:
    void render_begin(const float timer) override;
    int nbClaimedRotations() const { return m_claimed; }

protected:
    bool deviceRotatePrimitives(const vec3f &, const vec3f &, const vec3f &) override;

private:
    int m_claimed = 0;
};
}

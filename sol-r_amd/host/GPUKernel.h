/*
 * GPUKernel.h - host-side scene store and box-tree builder in front of the
 * rendering engine.
 *
 * Mirrors the public interface of the reference's solr::GPUKernel
 * (reference: solr/engines/GPUKernel.h:79-444) for everything that feeds the
 * per-pixel rendering path: primitives, materials, textures, camera, scene
 * and post-processing settings, the box-grid build (compactBoxes) and the
 * render_begin / render_end / getBitmap frame protocol.  Method names,
 * argument order and the dirty-flag protocol (GPUKernel.h:371-374,387) are the
 * reference's, so code written against the reference class reads the same
 * here.  Out of scope (and absent): file loaders, fake-GL vertex assembly,
 * Kinect/Oculus hooks, JPEG screenshots.
 *
 * The flattened arrays this class produces (BoundingBox[], Primitive[],
 * Lamp[], LightInformation[], Material[]) are bit-for-bit what the reference
 * builder would hand to its device layer for the same calls: same grid hash,
 * same std::map key order, same depth-first flattening with subtree sizes as
 * skip pointers (reference: GPUKernel.cpp:741-1281).  Storage differs (vectors
 * sized to the scene instead of NB_MAX_* arrays).
 */
#pragma once

#include <algorithm>
#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/solr_types.h"

namespace solr
{
/* reference: GPUKernel.h:46-65 */
struct CPUPrimitive
{
    bool belongsToModel;
    bool movable;
    vec3f p0, p1, p2;
    vec3f n0, n1, n2;
    vec3f size;
    int type;
    int materialId;
    vec2f vt0, vt1, vt2;
    vec3f speed0, speed1, speed2;
};

/* reference: GPUKernel.h:67-73 */
struct CPUBoundingBox
{
    vec3f parameters[2];
    vec3f center;
    std::vector<long> primitives; /* primitive ids (level 0) or child box keys (level > 0) */
    /* level > 0: where each child box sits in the level below (OrderedMap::at), filled next to `primitives`
     * by processOutterBoxes so that the refit and the flattening need not hash 500 k keys per pass.  Only
     * trusted while it has one entry per key: the light cell also lists light ids (GPUKernel.cpp:1189-1215)
     * and goes the hashed way. */
    std::vector<unsigned int> childAt;
    long indexForNextBox;
};

/* Ordered associative container with the slice of std::map's interface the builder uses (find,
 * operator[], insert, ascending iteration, size, clear) and the same observable behaviour, built for
 * the builder's access pattern: hundreds of thousands of inserts and look-ups per level, then ONE
 * ordered sweep.  Values live in a vector in insertion order, look-ups go through a hash index, and
 * the ascending key order is produced by one sort when iteration starts (and kept until the next
 * insertion).  The reference uses std::map<unsigned, CPUBoundingBox> per level
 * (GPUKernel.h:52-53 there); on 100k primitives the red-black trees made compactBoxes(true) 1.3 s. */
/* key -> position, open addressing with linear probing (the node-based std::unordered_map spent most of a
 * rebuild allocating: nine levels, 900 k insertions for 100 k primitives) */
class FlatIndex
{
public:
    static constexpr unsigned int NONE = 0xffffffffu;
    void clear()
    {
        m_keys.clear();
        m_positions.clear();
        m_count = 0;
        m_shift = 32;
    }
    void reserve(size_t n)
    {
        size_t wanted = 16;
        while (wanted < 2 * n)
            wanted *= 2;
        if (wanted > m_positions.size())
            rehash(wanted);
    }
    unsigned int find(unsigned int key) const
    {
        if (m_positions.empty())
            return NONE;
        const size_t mask = m_positions.size() - 1;
        for (size_t slot = home(key);; slot = (slot + 1) & mask)
        {
            if (m_positions[slot] == NONE)
                return NONE;
            if (m_keys[slot] == key)
                return m_positions[slot];
        }
    }
    /* the key must not be present */
    void insert(unsigned int key, unsigned int position)
    {
        if (2 * (m_count + 1) > m_positions.size())
            rehash(m_positions.empty() ? 16 : 2 * m_positions.size());
        place(key, position);
        ++m_count;
    }

private:
    size_t home(unsigned int key) const { return (size_t)((key * 0x9E3779B1u) >> m_shift); }
    void place(unsigned int key, unsigned int position)
    {
        const size_t mask = m_positions.size() - 1;
        size_t slot = home(key);
        while (m_positions[slot] != NONE)
            slot = (slot + 1) & mask;
        m_keys[slot] = key;
        m_positions[slot] = position;
    }
    void rehash(size_t capacity)
    {
        std::vector<unsigned int> keys, positions;
        keys.swap(m_keys);
        positions.swap(m_positions);
        m_keys.assign(capacity, 0u);
        m_positions.assign(capacity, NONE);
        m_shift = 32;
        for (size_t c = capacity; c > 1; c /= 2)
            --m_shift;
        for (size_t i = 0; i < positions.size(); ++i)
            if (positions[i] != NONE)
                place(keys[i], positions[i]);
    }
    std::vector<unsigned int> m_keys, m_positions;
    size_t m_count = 0;
    unsigned int m_shift = 32;
};

template <typename V>
class OrderedMap
{
public:
    typedef std::pair<unsigned int, V> value_type;
    class iterator
    {
    public:
        iterator() : m_owner(nullptr), m_rank(0) {}
        iterator(OrderedMap *owner, size_t rank) : m_owner(owner), m_rank(rank) {}
        value_type &operator*() const { return m_owner->m_items[m_owner->m_order[m_rank]]; }
        value_type *operator->() const { return &m_owner->m_items[m_owner->m_order[m_rank]]; }
        /* where the element lives: stable until clear(), see OrderedMap::at */
        unsigned int position() const { return m_owner->m_order[m_rank]; }
        iterator &operator++()
        {
            ++m_rank;
            return *this;
        }
        bool operator==(const iterator &o) const { return m_rank == o.m_rank; }
        bool operator!=(const iterator &o) const { return m_rank != o.m_rank; }

    private:
        OrderedMap *m_owner;
        size_t m_rank;
    };
    typedef iterator const_iterator;

    size_t size() const { return m_items.size(); }
    bool empty() const { return m_items.empty(); }
    void clear()
    {
        m_items.clear();
        m_order.clear();
        m_index.clear();
        m_sorted = true;
    }
    iterator begin() const
    {
        sortOrder();
        return iterator(const_cast<OrderedMap *>(this), 0);
    }
    iterator end() const { return iterator(const_cast<OrderedMap *>(this), m_items.size()); }
    /* like std::map::find, but the iterator is only good for comparison with end() and dereference */
    iterator find(unsigned int key) const
    {
        if (m_index.find(key) == FlatIndex::NONE)
            return end();
        sortOrder();
        /* rank of the key in ascending order */
        size_t lo = 0, hi = m_order.size();
        while (lo < hi)
        {
            const size_t mid = (lo + hi) / 2;
            if (m_items[m_order[mid]].first < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        return iterator(const_cast<OrderedMap *>(this), lo);
    }
    bool contains(unsigned int key) const { return m_index.find(key) != FlatIndex::NONE; }
    /* the element at a position an iterator reported: no hashing (positions survive insertions) */
    V &at(unsigned int position) { return m_items[position].second; }
    /* ... provided it still is the element with that key (a level may have been cleared and refilled since) */
    V *atIfKey(unsigned int position, unsigned int key)
    {
        return (position < m_items.size() && m_items[position].first == key) ? &m_items[position].second : nullptr;
    }
    V *lookup(unsigned int key)
    {
        if (key < m_items.size() && m_items[key].first == key) /* dense ids: position == key */
            return &m_items[key].second;
        const unsigned int at = m_index.find(key);
        return at == FlatIndex::NONE ? nullptr : &m_items[at].second;
    }
    V &operator[](unsigned int key)
    {
        if (key < m_items.size() && m_items[key].first == key)
            return m_items[key].second;
        const unsigned int at = m_index.find(key);
        if (at != FlatIndex::NONE)
            return m_items[at].second;
        return append(key, V());
    }
    std::pair<iterator, bool> insert(const value_type &kv)
    {
        if (m_index.find(kv.first) != FlatIndex::NONE)
            return std::make_pair(end(), false);
        append(kv.first, kv.second);
        return std::make_pair(end(), true);
    }
    void reserve(size_t n)
    {
        m_items.reserve(n);
        m_index.reserve(n);
    }

private:
    V &append(unsigned int key, const V &value)
    {
        if (m_sorted && !m_items.empty() && key < m_items[m_order.empty() ? m_items.size() - 1 : m_order.back()].first)
            m_sorted = false;
        m_index.insert(key, (unsigned int)m_items.size());
        m_items.push_back(value_type(key, value));
        if (m_sorted)
            m_order.push_back((unsigned int)m_items.size() - 1);
        return m_items.back().second;
    }
    void sortOrder() const
    {
        if (m_sorted && m_order.size() == m_items.size())
            return;
        m_order.resize(m_items.size());
        for (size_t i = 0; i < m_order.size(); ++i)
            m_order[i] = (unsigned int)i;
        std::sort(m_order.begin(), m_order.end(),
                  [this](unsigned int a, unsigned int b) { return m_items[a].first < m_items[b].first; });
        m_sorted = true;
    }
    std::vector<value_type> m_items;
    mutable std::vector<unsigned int> m_order; /* item indices by ascending key, valid when m_sorted */
    FlatIndex m_index;
    mutable bool m_sorted = true;
};

typedef OrderedMap<CPUBoundingBox> BoxContainer;
typedef OrderedMap<CPUPrimitive> PrimitiveContainer;

inline vec2i make_vec2i(int x = 0, int y = 0) { vec2i v; v.x = x; v.y = y; return v; }
inline vec4i make_vec4i(int x = 0, int y = 0, int z = 0, int w = 0) { vec4i v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
inline vec2f make_vec2f(float x = 0.f, float y = 0.f) { vec2f v; v.x = x; v.y = y; return v; }
inline vec3f make_vec3f(float x = 0.f, float y = 0.f, float z = 0.f) { vec3f v; v.x = x; v.y = y; v.z = z; return v; }
inline vec4f make_vec4f(float x = 0.f, float y = 0.f, float z = 0.f, float w = 0.f) { vec4f v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }

class GPUKernel
{
public:
    GPUKernel();
    virtual ~GPUKernel();

    /* reference: GPUKernel.h:85-87 */
    virtual void initBuffers();
    virtual void cleanup();
    virtual void reshape();

    /* reference: GPUKernel.h:90-96,289 (engine selection hooks) */
    virtual void setPlatformId(const int) {}
    virtual void setDeviceId(const int) {}
    virtual void setKernelFilename(const std::string &) {}
    virtual void queryDevice() {}
    virtual void recompileKernels() {}
    virtual std::string getGPUDescription() { return m_gpuDescription; }

    /* ---------- Rendering (GPUKernel.h:100-102) ---------- */
    virtual void render_begin(const float timer);
    virtual void render_end() = 0;
    /* the image of the frame render_end delivered last (with frames in flight - setFramesInFlight - that is the
     * engine's page-locked copy of it, not m_bitmap: nothing is copied twice) */
    BitmapBuffer *getBitmap()
    {
        fetchBitmap();
        return m_bitmapView ? m_bitmapView : (m_bitmap.empty() ? nullptr : m_bitmap.data());
    }
    /* render_end with the frame's image delivered into the CALLER'S array instead of m_bitmap (what SolR_RunKernel is
     * for, SolRStub.cpp:154-164: the reference copies m_bitmap to the caller afterwards).  An engine with a device
     * copies from the device straight into `image` and leaves m_bitmap to the next getBitmap() (fetchBitmap); the
     * default is the reference's two steps. */
    virtual void render_end(BitmapBuffer *image);
    /* Extension (no reference equivalent; the reference's render_end waits for the frame and then copies it,
     * CudaKernel.cpp:304-312).  n > 1: render_begin also starts the read-back of its frame's image, and render_end
     * delivers the image of the frame n - 1 calls back - the one whose copy has had n - 1 frames' time to land -
     * while the newer ones render; flushFrames() waits for all of them and delivers the newest.  1 (default): the
     * reference's protocol.  Engines without a device ignore it. */
    virtual void setFramesInFlight(int n) { (void)n; }
    virtual int getFramesInFlight() const { return 1; }
    virtual void flushFrames() {}
    /* devices of this process the frame is shared out over (m_occupancyParameters.x, which the reference leaves at 1
     * with no way to set it: CudaKernel.cpp:90); engines without a device keep 1 */
    virtual void setGpuCount(int n) { (void)n; }
    virtual int getGpuCount() const { return 1; }
    /* 0 = ok, otherwise the engine's pending error (no reference equivalent:
     * the reference exits the process on a device error) */
    virtual int lastError(std::string *message = nullptr) { (void)message; return 0; }

protected:
    BitmapBuffer *m_bitmapView = nullptr; /* see getBitmap */

public:

    /* ---------- Primitives (GPUKernel.h:108-149) ---------- */
    int addPrimitive(PrimitiveType type, bool belongsToModel = false);
    void setPrimitive(const int &index, float x0, float y0, float z0, float w, float h, float d, int materialId);
    void setPrimitive(const int &index, float x0, float y0, float z0, float x1, float y1, float z1, float w, float h,
                      float d, int materialId);
    void setPrimitive(const int &index, float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2,
                      float z2, float w, float h, float d, int materialId);
    unsigned int getPrimitiveAt(int x, int y);
    void setPrimitiveIsMovable(const int &index, bool movable);
    void setPrimitiveBellongsToModel(const int &index, bool bellongsToModel);
    void setPrimitiveMaterial(unsigned int index, int materialId);
    int getPrimitiveMaterial(unsigned int index);
    vec4f getPrimitiveCenter(unsigned int index);
    void setPrimitiveCenter(unsigned int index, const vec3f &center);
    void setPrimitiveTextureCoordinates(const unsigned int index, const vec2f &vt0, const vec2f &vt1,
                                        const vec2f &vt2);
    void setPrimitiveNormals(unsigned int index, vec3f n0, vec3f n1, vec3f n2);
    CPUPrimitive *getPrimitive(const unsigned int index);
    int getLight(int index);

    /* rotation of the whole scene about a centre (GPUKernel.h:122-125) */
    void rotatePrimitives(const vec3f &rotationCenter, const vec4f &angles);
    void translatePrimitives(const vec3f &translation);
    void scalePrimitives(float scale, unsigned int from, unsigned int to);

    /* ---------- Complex objects (GPUKernel.h:162-164) ---------- */
    int addCube(float x, float y, float z, float radius, int materialId);
    int addRectangle(float x, float y, float z, float w, float h, float d, int materialId);

    /* ---------- Materials (GPUKernel.h:168-190) ---------- */
    int addMaterial();
    void setMaterial(unsigned int index, const Material &material);
    void setMaterial(unsigned int index, float r, float g, float b, float noise, float reflection, float refraction,
                     bool procedural, bool wireframe, int wireframeWidth, float transparency, float opacity,
                     int diffuseTextureId, int normalTextureId, int bumpTextureId, int specularTextureId,
                     int reflectionTextureId, int transparentTextureId, int ambientOcclusionTextureId, float specValue,
                     float specPower, float specCoef, float innerIllumination, float illuminationDiffusion,
                     float illuminationPropagation, bool fastTransparency);
    void setMaterialColor(unsigned int index, float r, float g, float b);
    Material *getMaterial(const int index);

    /* ---------- Camera (GPUKernel.h:194) ---------- */
    void setCamera(const vec3f &eye, const vec3f &dir, const vec4f &angles);

    /* ---------- Textures (GPUKernel.h:198-205) ---------- */
    void setTexture(const int index, const TextureInfo &textureInfo);
    void getTexture(const int index, TextureInfo &textureInfo);
    void setTexturesTransfered(const bool transfered) { m_texturesTransfered = transfered; }
    void realignTexturesAndMaterials();
    void processTextureOffsets();
    TextureInfo &getTextureInformation(const int index);

    /* ---------- Scene (GPUKernel.h:208-221) ---------- */
    void setSceneInfo(int width, int height, float transparentColor, int graphicsLevel, float viewDistance,
                      float shadowIntensity, int nbRayIterations, vec4f backgroundColor, int cameraType,
                      float eyeSeparation, bool renderBoxes, int pathTracingIteration, int maxPathTracingIterations,
                      FrameBufferType frameBufferType, int timestamp, int atmosphericEffect, int skyboxSize,
                      int skyboxMaterialId);
    void setSceneInfo(const SceneInfo &sceneInfo);
    SceneInfo &getSceneInfo();
    void setPostProcessingInfo(PostProcessingType type, float param1, float param2, int param3);
    void setPostProcessingInfo(const PostProcessingInfo &postProcessingInfo);
    PostProcessingInfo &getPostProcessingInfo() { return m_postProcessingInfo; }

    /* ---------- Counters / frames (GPUKernel.h:267-298) ---------- */
    unsigned int getNbActiveBoxes();
    unsigned int getNbActivePrimitives();
    unsigned int getNbActiveLamps();
    unsigned int getNbActiveMaterials();
    unsigned int getNbActiveTextures();
    void resetFrame();
    void resetAll();
    void setNbFrames(const int nbFrames) { m_nbFrames = nbFrames; }
    void setFrame(const int frame)
    {
        if ((unsigned int)frame != m_frame)
            syncHost();
        m_frame = frame;
    }
    int getNbFrames() { return m_nbFrames; }
    int getFrame() { return m_frame; }
    /* key-frame animation (GPUKernel.h:275-298 of the reference) */
    void nextFrame();
    void previousFrame();
    void morphPrimitives();
    void resetAddingIndex() { m_addingIndex = 0; }
    void doneWithAdding(const bool &done) { m_doneWithAdding = done; }
    void getPrimitiveOtherCenter(unsigned int index, vec3f &center);
    int getCurrentMaterial() { return m_currentMaterial; }
    void setCurrentMaterial(const int currentMaterial) { m_currentMaterial = currentMaterial; }
    void setMaterialTextureId(unsigned int textureId);

    /* ---------- Box tree (GPUKernel.h:304-309) ---------- */
    int compactBoxes(bool reconstructBoxes);
    void streamDataToGPU();
    void resetBoxes(bool resetPrimitives);
    void setPrimitivesTransfered(const bool value) { m_primitivesTransfered = value; }

    vec3f &getViewPos() { return m_viewPos; }
    vec3f &getViewDir() { return m_viewDir; }
    vec4f &getViewAngles() { return m_angles; }

    /* ---------- Extensions (no reference equivalent) ---------- */
    /* The reference draws sceneInfo.timestamp and the random buffer from
     * rand()/time(0) in render_begin (GPUKernel.cpp:2719-2726).  With a
     * non-negative seed the timestamp given in SceneInfo is kept and the
     * random buffer is filled once from a 32-bit LCG with the reference's
     * distribution 5e-6 * (k % 2000 - 1000); -1 restores rand(). */
    void setDeterministic(long seed);
    /* flattened arrays of the current frame, as handed to the device layer */
    const std::vector<BoundingBox> &hostBoxes() { syncHost(); return m_hBoundingBoxes; }
    const std::vector<Primitive> &hostPrimitives() { syncHost(); return m_hPrimitives; }
    const std::vector<Lamp> &hostLamps() { syncHost(); return m_hLamps; }
    const std::vector<LightInformation> &hostLights() { syncHost(); return m_lightInformation; }
    /* per flattened primitive: would rotatePrimitives move it (level-0 box, movable, not the camera) */
    const std::vector<unsigned char> &hostMovable() { syncHost(); return m_hMovable; }
    /* Animated scenes: rotatePrimitives + compactBoxes(false) per frame (MoleculeScene.cpp:75-81).  An
     * engine that keeps the scene resident may apply the rotation there (deviceRotatePrimitives); the
     * host copy then lags behind by the rotations listed in m_pendingRotations and is caught up - the same
     * arithmetic, replayed in order - by syncHost() before anything reads or changes it. */
    void syncHost();
    size_t nbPendingRotations() const { return m_pendingRotations.size() + m_unrecordedRotations; }
    int lightInformationSize() const { return m_lightInformationSize; }
    const Material *hostMaterials() const { return m_hMaterials.data(); }
    const std::vector<RandomBuffer> &hostRandoms();
    size_t randomsNeeded() const;
    void setHostBuildOnly(bool on) { m_hostBuildOnly = on; }
    const std::vector<BitmapBuffer> &hostTextureAtlas();
    PrimitiveXYIdBuffer *hostPrimitiveIds() { fetchPrimitiveIds(); return m_hPrimitivesXYIds.data(); }
    unsigned int treeDepth() const { return m_treeDepth; }

protected:
    struct Frame
    {
        BoxContainer boundingBoxes[BOUNDING_BOXES_TREE_DEPTH];
        PrimitiveContainer primitives;
        int nbActiveBoxes = 0;
        int nbActivePrimitives = 0;
        int nbActiveLamps = 0;
        vec3f minPos = {0.f, 0.f, 0.f};
        vec3f maxPos = {0.f, 0.f, 0.f};
        bool levelsBuilt = true; /* false: the flattened arrays came from the device build, the level maps are empty */
    };
    /* every access to the scene store: catches the host copy up and ends the device-side fast path until
     * the next upload (the caller may be about to change what the device holds) */
    Frame &frame()
    {
        m_hostTouched = true;
        /* a tree that came from the device build left the per-level maps empty: they are made now, from the
         * primitives as they still are, before whoever asks can change one (the cells a primitive sits in are
         * those of the last full build, GPUKernel.cpp:1378-1460 refits them, it does not re-hash) */
        if (!m_frames[m_frame].levelsBuilt)
            buildLevelsOnHost();
        if ((!m_pendingRotations.empty() || m_unrecordedRotations) && !m_buildingLevels)
            syncHost();
        return m_frames[m_frame];
    }
    /* a look that changes nothing (getPrimitiveCenter, getLight ...): up to date, fast path kept */
    const CPUPrimitive *peekPrimitive(unsigned int index)
    {
        syncHost();
        return m_frames[m_frame].primitives.lookup(index);
    }
    /* counters and the frame protocol: reads that stay valid while rotations are pending */
    Frame &frameAsIs() { return m_frames[m_frame]; }
    /* engine hook: the per-pixel primitive ids of the last frame are wanted on the host (getPrimitiveAt).
     * The reference copies all of them back after every frame (CudaKernel.cpp:304-312 via d2h_bitmap) -
     * 16 bytes per pixel, five times the image - although only picking ever looks at them; an engine may
     * leave them on the device until then */
    virtual void fetchPrimitiveIds() {}
    /* m_bitmap brought up to date with the frame rendered last, where render_end(image) left it behind */
    virtual void fetchBitmap() {}
    /* engine hook: the primitives as the resident scene holds them now, written into the store (the same
     * bits a replay of the pending rotations produces - tests/test_animation_gpu.py - at a cost that does not
     * grow with their number); false = not available, replay */
    virtual bool primitivesFromDevice(Frame &) { return false; }
    /* engine hook: GPUKernel::compactBoxes(true) on the device (include/solr_hip.h, solr_hip_build_tree): the
     * flattened node list, the order in which the primitives are streamed and the number of lamps, from the
     * frame's primitives in index order.  Returns the tree depth, or a negative value when the engine has no
     * such thing or leaves this scene to the host builder */
    virtual int deviceBuildTree(const std::vector<Primitive> &, const std::vector<unsigned char> &, const vec3f &,
                                const vec3f &, float, std::vector<BoundingBox> &, std::vector<int> &, int &)
    {
        return -2;
    }
    /* the per-level maps of the host builder, built when something needs them after a device-side build
     * (host-side rotations and translations, compactBoxes(false)) */
    void ensureLevels();
    bool buildTreeOnDevice();
    void buildLevelsOnHost();
    /* engine hook: apply the rotation to the resident scene; false = not done, nothing changed */
    virtual bool deviceRotatePrimitives(const vec3f &, const vec3f &, const vec3f &) { return false; }
    void rotatePrimitivesOnly(Frame &f, const vec3f &rotationCenter, const vec3f &cosA, const vec3f &sinA);
    void refitBoxes(Frame &f);

    /* box-tree build (GPUKernel.h:318-324) */
    int processBoxes(const int boxSize, bool simulate);
    int processOutterBoxes(const int boxSize, const int boundingBoxesDepth);
    bool updateBoundingBox(CPUBoundingBox &box);
    bool updateOutterBoundingBox(CPUBoundingBox &box, const int depth);
    void resetBox(CPUBoundingBox &box, bool resetPrimitives);
    void recursiveDataStreamToGPU(const int depth, CPUBoundingBox &parent);
    void appendPrimitive(long id, bool inLevel0Box);
    void fillRandoms();

    float vectorLength(const vec3f &v);
    void normalizeVector(vec3f &v);
    vec3f crossProduct(const vec3f &b, const vec3f &c);
    void rotateVector(vec3f &v, const vec3f &rotationCenter, const vec3f &cosAngles, const vec3f &sinAngles);

protected:
    /* flattened, device-bound arrays (reference: GPUKernel.h:328-339) */
    std::vector<BoundingBox> m_hBoundingBoxes;
    std::vector<Primitive> m_hPrimitives;
    std::vector<Lamp> m_hLamps;
    std::vector<unsigned char> m_hMovable;
    std::vector<Material> m_hMaterials;
    TextureInfo m_hTextures[NB_MAX_TEXTURES];
    std::vector<BitmapBuffer> m_textureAtlas;
    std::vector<RandomBuffer> m_hRandoms;
    bool m_randomsFilled = false;
    bool m_buildingLevels = false; /* the lazy build reads the store as it is: pending rotations stay pending */
    bool m_hostBuildOnly = false; /* SolRx_HostBuild(1) / SOLR_HOST_BUILD=1: never ask the engine for the tree */
    std::vector<PrimitiveXYIdBuffer> m_hPrimitivesXYIds;
    std::vector<LightInformation> m_lightInformation;
    std::vector<BitmapBuffer> m_bitmap;

    std::map<unsigned int, Frame> m_frames;
    int m_nbActiveMaterials;
    int m_nbActiveTextures;
    int m_lightInformationSize;
    size_t m_maxPrimitivesPerBox;
    bool m_doneWithAdding;
    int m_currentMaterial = 0;
    int m_addingIndex;

    vec3f m_viewPos;
    vec3f m_viewDir;
    vec4f m_angles;

    unsigned int m_frame;
    unsigned int m_nbFrames;
    unsigned int m_treeDepth;

    bool m_primitivesTransfered;
    bool m_materialsTransfered;
    bool m_texturesTransfered;
    bool m_randomsTransfered;
    bool m_hostTouched = true; /* scene store accessed since the last upload */
    struct PendingRotation
    {
        vec3f center, cosA, sinA;
    };
    std::vector<PendingRotation> m_pendingRotations;
    /* an animation that runs for hours: past this many the rotations are only counted - the store then
     * catches up from the device's primitives, which it does for more than a handful anyway */
    static constexpr size_t MAX_RECORDED_ROTATIONS = 100000;
    size_t m_unrecordedRotations = 0;

    SceneInfo m_sceneInfo;
    PostProcessingInfo m_postProcessingInfo;
    bool m_refresh;
    std::string m_gpuDescription;
    vec2i m_occupancyParameters;
    bool m_buffersInitialized;
    long m_deterministicSeed;
};

/* reference: GPUKernel.h:446-455 */
class SingletonKernel
{
public:
    static GPUKernel *kernel();
    /* extension: "hip" (default) or "host-only" (scene store without a
     * device, render_begin fails loudly; used by the CPU test-suite) */
    static void selectEngine(const char *name);
    static void destroy();

private:
    SingletonKernel();
    static GPUKernel *m_kernel;
};
}

/*
 * solr_hip.h - C-ABI drop-in boundary of the MI355X (gfx950) rendering engine.
 *
 * Part 1 re-declares, name for name and argument for argument, the ten
 * extern "C" entry points through which the reference's host engine class
 * drives its device code (reference: solr/engines/cuda/CudaRayTracer.h:25-67;
 * the only caller is solr/engines/cuda/CudaKernel.cpp:138,144,160,201,204,210,
 * 219,228,293,307).  A host built against the reference headers links against
 * libsolr_hip.so instead of the CUDA object and keeps working: same symbols,
 * same by-value SysV argument passing, same ownership (host arrays are
 * borrowed for the duration of the call, device buffers are owned here).
 *
 * Part 2 adds what a one-process-per-GPU deployment needs and the reference
 * never had: device/stream selection, the framebuffer row strip this process
 * renders, device-pointer access for the RCCL gather, pointer-argument twins
 * of the by-value calls for FFI callers (ctypes/cgo cannot align a by-value
 * struct to 16 bytes), timing and ray counting for bench.py, and an error
 * query (the reference exits the process on a CUDA error,
 * helper_cuda.h:749-763; this library records the error, makes every later
 * call a no-op and lets the host decide - set SOLR_HIP_FATAL=1 to get the
 * reference's exit(EXIT_FAILURE) behaviour back).
 *
 * No torch / HIP types appear in any signature: plain pointers and sizes.
 */
#ifndef SOLR_HIP_H
#define SOLR_HIP_H

#include "solr_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================= */
/* Part 1 - the reference boundary                                          */
/* ======================================================================= */

/* CudaRayTracer.h:25 / CudaRayTracer.cu:1408-1491.  occupancyParameters.x is the reference's in-process GPU
 * count and is honoured as the reference honours it (CudaRayTracer.cu:1413-1424, 1536-1555, 1647-1672,
 * 1694-1815): that many devices of THIS process (clamped, with a notice, to the devices there are), the scene
 * replicated by every h2d_* call, the frame cut into equal row strips - device d renders strip d - and d2h_bitmap
 * copying every device's strip to its place in the host arrays.  Values below 1 read as 1 here and as "what
 * initialize_scene was given" in the other nine calls; any other value there is an error.  occupancyParameters.y
 * (streams per device) is accepted and not used.  Several devices in one process and the one-process-per-GPU
 * model of part 2 (solr_hip_set_strip, solr_hip_comm_*) refuse each other; neighbourhood post-processing stays
 * inside a device's strip in this mode, as in the reference.  The nb* capacities are hints: device arrays are
 * sized to the scene actually uploaded, not to NB_MAX_*. */
void initialize_scene(vec2i occupancyParameters, SceneInfo sceneInfo, int nbPrimitives, int nbLamps,
                      int nbMaterials);

/* CudaRayTracer.h:34 / CudaRayTracer.cu:1499-1532 */
void finalize_scene(vec2i occupancyParameters);

/* CudaRayTracer.h:41 / CudaRayTracer.cu:1360-1400: (re)allocates the
 * per-pixel buffers (float framebuffer, RGB bitmap, primitive ids, randoms) */
void reshape_scene(vec2i occupancyParameters, SceneInfo sceneInfo);

/* CudaRayTracer.h:43 / CudaRayTracer.cu:1540-1555: uploads the flattened box
 * tree, the primitives (in box order) and the lamp index list; the AoS host
 * records are re-packed into the device plane layout here */
void h2d_scene(vec2i occupancyParameters, BoundingBox *boundingBoxes, int nbActiveBoxes, Primitive *primitives,
               int nbPrimitives, Lamp *lamps, int nbLamps);

/* CudaRayTracer.h:46 / CudaRayTracer.cu:1557-1565 */
void h2d_materials(vec2i occupancyParameters, Material *materials, int nbActiveMaterials);

/* CudaRayTracer.h:48 / CudaRayTracer.cu:1567-1576: always MAX_BITMAP_SIZE floats */
void h2d_randoms(vec2i occupancyParameters, float *randoms);

/* CudaRayTracer.h:50 / CudaRayTracer.cu:1578-1613: packs every non-null
 * texture at its TextureInfo.offset into one byte atlas */
void h2d_textures(vec2i occupancyParameters, int activeTextures, TextureInfo *textureInfos);

/* CudaRayTracer.h:52 / CudaRayTracer.cu:1615-1625 */
void h2d_lightInformation(vec2i occupancyParameters, LightInformation *lightInformation, int lightInformationSize);

/* CudaRayTracer.h:55 / CudaRayTracer.cu:1647-1672: waits for the frame and
 * copies the RGB bitmap (W*H*3 bytes) and the primitive-id buffer (W*H*16
 * bytes) of this process's strip to the strip's position in the host arrays.
 * Extension: either pointer may be NULL and is then skipped - the ids, five times the image, can be left
 * on the device until picking needs them and fetched with a second call (they stay valid until the next
 * frame is rendered into the same buffers; HipKernel::render_end / fetchPrimitiveIds do that) */
void d2h_bitmap(vec2i occupancyParameters, SceneInfo sceneInfo, BitmapBuffer *bitmap,
                PrimitiveXYIdBuffer *primitivesXYIds);

/* CudaRayTracer.h:58 / CudaRayTracer.cu:1680-1908: launches the renderer and
 * the post-processing stage; asynchronous, ordered on the engine's stream.
 * objects = {nbBoxes, nbPrimitives, nbLamps, lightInformationSize}.
 * blockSize is accepted for signature compatibility; the wave64 tile shape
 * is chosen by the engine. */
void cudaRender(vec2i occupancyParameters, vec4i blockSize, SceneInfo sceneInfo, vec4i objects,
                PostProcessingInfo postProcessingInfo, vec3f origin, vec3f direction, vec4f angles);

/* ======================================================================= */
/* Part 2 - extensions                                                      */
/* ======================================================================= */

/* Diagnostics: the walk's own ceiling ("achieved Mrays/s vs a measured empty-traversal upper bound").
 * Renders one frame whose walks are recorded - which list, and per lane the ray, the cut-off, the leaf visit after which
 * a shadow lane was done - and replays those walks `repeats` times with NOTHING BUT THE NODE LOOP: the same waves with
 * the same 64 rays together, the same lists, no leaf record, no primitive test, no shading, no camera, no frame buffer,
 * at the renderer's occupancy.  Shadow walks replay node for node; a closest-hit walk runs with its final cut-off in
 * place from the first node (the fewest nodes a walk that finds that hit can visit).  ms[1] / ms[2]: mean / fastest
 * replay in milliseconds (HIP events, one launch at a time); rays of the frame / that time is the ceiling.  stats: walks
 * recorded (per wave), walks left out of the replay, leaf entries of the replay (per lane), workgroups.  One GPU, engine
 * 0, scenes of untextured spheres / planes / triangles / cylinders (the lean kernels).  0, or -1 with the error set. */
int solr_hip_walk_bound(const SceneInfo *sceneInfo, const vec4i *objects, const PostProcessingInfo *postProcessingInfo,
                        const float origin[3], const float direction[3], const float angles[4], int repeats, double ms[3],
                        unsigned long long stats[4]);
/* Which list the walks of the frame recorded by the last solr_hip_walk_bound (called with stats != NULL) took, per wave:
 * closest-hit walks in the reference's order / on an order-free list, shadow walks likewise, walks not through the node
 * loop, walks beyond a workgroup's record slots (not classified).  A checked walk that repeats lanes in the reference's
 * order counts once in each. */
void solr_hip_walk_bound_lists(unsigned long long out[6]);
/* The record behind solr_hip_walk_bound in the caller's hands (tools/ray_regroup.py: what would a frame's rays sorted by
 * where they go, or a walk kernel of its own at twice the occupancy, buy the node loop?).  keep(1): the next
 * solr_hip_walk_bound leaves its records on the device.  info: {workgroups recorded, bytes per workgroup slot, dynamic LDS
 * of the recorded launch, walk slots per workgroup} (layout: sol-r_amd/csrc/rt_device.h, "the walk's own ceiling").
 * copy: the first `grid` slots to the host (toDevice == 0) or back.  replay: those slots with nothing but the node loop,
 * `ldsBytes` of dynamic LDS a wave (< 0: the recorded launch's, i.e. the renderer's occupancy; 0: as many waves as the
 * replay kernel's 64 registers allow), ms / stats as solr_hip_walk_bound.  release: the buffers given back. */
void solr_hip_walk_records_keep(int keep);
int solr_hip_walk_records_info(unsigned long long info[4]);
int solr_hip_walk_records_copy(void *host, unsigned grid, int toDevice);
int solr_hip_walk_replay(unsigned grid, long ldsBytes, int repeats, double ms[3], unsigned long long stats[4]);
void solr_hip_walk_records_release(void);

/* 0 when no error is pending; otherwise the HIP error code (or -1 for an
 * argument/state error) and, if buf != NULL, its text. Does not clear. */
int solr_hip_last_error(char *buf, int len);
void solr_hip_clear_error(void);

/* number of visible GPUs (0 when none / no driver); never sets the error */
int solr_hip_device_count(void);
/* in-process devices: what the pointer form solr_hip_initialize passes as occupancyParameters.x (default 1), and
 * how many engines are rendering since the last initialize_scene */
void solr_hip_set_gpu_count(int n);
int solr_hip_gpu_count(void);
/* device this process renders on (call before initialize_scene; default 0) */
void solr_hip_set_device(int device);
int solr_hip_get_device(void);
/* stream (a hipStream_t passed as void*) every copy and launch is issued on;
 * NULL = a stream owned by the engine */
void solr_hip_set_stream(void *stream);
void solr_hip_synchronize(void);

/* Row strip rendered by this process: rows [firstRow, firstRow + nbRows) of
 * the full sceneInfo.size image.  nbRows < 0 restores the full frame; nbRows
 * == 0 is an EMPTY strip (a process left without rows when there are more
 * processes than rows to share out): cudaRender and d2h_bitmap then do
 * nothing.  The device buffers hold only the strip (row 0 of the buffer =
 * firstRow); d2h_bitmap places it at its position in a full-size host image. */
void solr_hip_set_strip(int firstRow, int nbRows);
/* the strip in force (after solr_hip_balance_strips: the one it cut for this rank) */
void solr_hip_get_strip(int *firstRow, int *nbRows);

/* Multi-GPU without any framework: one process per GPU, the frame split into row strips as the reference
 * splits it over the devices of its one process (CudaRayTracer.cu:1694-1696, 1709-1815), the strips sent
 * to one root with RCCL (xGMI) where the reference copies them through the host (d2h_bitmap :1647-1672).
 *   solr_hip_strip_rows       the rows of rank r of n (contiguous strips, the last absorbs the remainder)
 *   solr_hip_comm_unique_id   rank 0: 128 bytes to hand to every other rank (ncclGetUniqueId)
 *   solr_hip_comm_init        every rank, after initialize_scene and the uploads of its first frame: joins the
 *                             communicator (ncclCommInitRank) and takes over rank 0's random buffer (below)
 *   solr_hip_comm_ranks       ranks of the communicator as RCCL reports them (ncclCommCount); 0 without one
 *   solr_hip_gather_strips    after cudaRender, every rank, every frame: this rank's strip -> root, enqueued on
 *                             the stream that rendered the frame (one grouped ncclSend / ncclRecv per peer);
 *                             returns at once
 *   solr_hip_gathered_frame   root: device pointer of the assembled height x width x 3 frame
 *   solr_hip_d2h_gathered     root: waits for the gather and copies the assembled frame to the host
 *   solr_hip_gather_ids       every rank, when picking asks (GPUKernel::getPrimitiveAt, GPUKernel.cpp:729-739): the
 *                             PrimitiveXYIdBuffer strips of the frame rendered last -> root (the reference copies them
 *                             after every frame, CudaRayTracer.cu:1664-1670: 16 bytes per pixel, five times the image)
 *   solr_hip_d2h_gathered_ids root: waits for that gather and copies the height x width records to the host
 *   solr_hip_comm_finalize    leaves the communicator
 *   solr_hip_comm_set_per_flight / solr_hip_comm_count   one communicator for everything (default) or one per frame in
 *                             flight: RCCL orders the operations of one communicator across streams
 *   solr_hip_image_share, solr_hip_d2h_image_async, solr_hip_image_wait   the delivered frame: every rank's strip over
 *                             its own PCIe link into one page-locked host image the ranks' processes share
 *   solr_hip_d2h_gathered_async   the other route: rank 0 copies the assembled frame, pipelined
 * All return 0, or -1 with solr_hip_last_error set.  RCCL is loaded when the first of them is called
 * (SOLR_HIP_RCCL_LIBRARY names another build of it).
 *
 * What the ranks owe each other.  They run the same host program: the same sequence of
 * these calls, of cudaRender and of h2d_randoms on every rank, with the same SceneInfo (timestamp and
 * pathTracingIteration included) and PostProcessingInfo.  Given that, no rank can leave another waiting: a rank in
 * an error state, or whose strip is not the one the strip table gives it, still takes part in every collective -
 * it contributes zeros, keeps its error and returns -1 - and the blocking collectives fail on all ranks together.
 * The random buffer (ambient-occlusion taps, depth of field, jitter of accumulation passes) is rank 0's on every
 * rank: solr_hip_comm_init and every h2d_randoms / solr_hip_h2d_randoms_sized after it end with a broadcast, since
 * hosts seed theirs from the clock unless told otherwise (GPUKernel.cpp:89) and strips rendered from different
 * buffers do not assemble to the frame one GPU renders.
 *
 * Cost-balanced strips (an extension; the reference's split is equal, CudaRayTracer.cu:1694-1696): equal strips
 * share out rows, not work - the frame is as slow as its slowest rank.
 *   solr_hip_strip_row_costs  what each row of this process's strip cost in the last frame (tile durations)
 *   solr_hip_balanced_strips  pure arithmetic: contiguous strips of equal cost from the rows' costs of the whole
 *                             frame, boundaries on multiples of `align` rows (8 = a tile)
 *   solr_hip_set_strip_table  the strips of all ranks, for the gather and the halo exchange (world = 0: forget)
 *   solr_hip_balance_strips   all of it with one ncclAllReduce: every rank, between frames; sets this process's
 *                             strip as well.  With the ambient-occlusion post-process the strips are cut on
 *                             multiples of the taps' reach (a multiple of 8 rows), so that the halo exchange, which
 *                             trades rows with the next rank only, still covers every tap */
void solr_hip_strip_rows(int rank, int world, int height, int *firstRow, int *nbRows, int *rowsPerRank);
int solr_hip_comm_unique_id(void *id128);
int solr_hip_comm_init(int rank, int world, const void *id128);
int solr_hip_comm_ranks(void);
/* One communicator per frame in flight (call before solr_hip_comm_init, every rank alike; SOLR_HIP_COMM_PER_FLIGHT=0/1
 * in the environment overrides).  RCCL orders the operations of one communicator whatever streams they are on, so
 * with frames in flight the gather of frame n + 1 waits for the gather of frame n; communicators split off the
 * first one (ncclCommSplit) do not order against each other.  Default 0 (one for everything) until an N > 1 run
 * has measured both; solr_hip_comm_count: how many this process holds (0, 1, or one per possible flight). */
void solr_hip_comm_set_per_flight(int on);
int solr_hip_comm_count(void);
/* a number every rank holds alike (rank 0's draw at solr_hip_comm_init; 0 without a communicator of several ranks):
 * the seed for whatever the hosts draw per frame - GPUKernel::render_begin's timestamp (GPUKernel.cpp:2712-2727) - so
 * that every rank renders the same frame without a word per frame between them */
unsigned solr_hip_comm_shared_seed(void);
int solr_hip_gather_strips(int root);
void *solr_hip_gathered_frame(void);
int solr_hip_d2h_gathered(BitmapBuffer *hostBitmap);
/* root: the assembled frame of the gather issued last -> a page-locked host image on the copy stream, behind that
 * gather; returns a ticket for solr_hip_image_wait at once (the next gather into the same flight's frame waits for
 * the copy).  Other ranks: -2, nothing to deliver (no error).  The delivered frame of an N-GPU job, pipelined. */
int solr_hip_d2h_gathered_async(void);
int solr_hip_gather_ids(int root);
int solr_hip_d2h_gathered_ids(PrimitiveXYIdBuffer *hostIds);
void solr_hip_comm_finalize(void);
int solr_hip_strip_row_costs(float *rowCost, int height);
int solr_hip_balanced_strips(const float *rowCost, int height, int world, int align, int *firstRows, int *nbRows);
int solr_hip_set_strip_table(const int *firstRows, const int *nbRows, int world, int height);
int solr_hip_balance_strips(void);

/* Neighbourhood post-processing on a strip.  The ambient-occlusion kernel (CudaRayTracer.cu:1128-1181) compares a
 * pixel's depth with 256 taps up to 16 * param2 * max|random| / 10 pixels away: rows of the ranks above and below.
 * The reference's split post-processes each device's strip on its own, so its frames have seams there (SURVEY.md
 * section 8e asks for a halo).  Here, with a communicator (solr_hip_comm_init) and the strips of
 * solr_hip_strip_rows, cudaRender trades the depths of the boundary rows with the neighbouring ranks over RCCL,
 * on the frame's stream, between the renderer and the post-processing kernel: the assembled frame is the one a
 * single GPU renders.  How many rows are traded is agreed over the communicator (the maximum of the ranks' own
 * figures; one all-reduce whenever the random buffer was uploaded or param2 changed), so neighbours always post
 * transfers of the same size.  A host that moves the rows itself hands them over with solr_hip_set_depth_halo:
 * nbAbove rows of `width` floats (PostProcessingBuffer.colorInfo.w) just above its strip, nbBelow just below;
 * they are used by the frames that follow until (NULL, 0, NULL, 0) ends it.  The other neighbourhood kernels
 * (depth of field, radiosity: random gathers over the frame) stay inside the strip as in the reference. */
void solr_hip_set_depth_halo(const float *above, int nbAbove, const float *below, int nbBelow);

/* GPUKernel::compactBoxes(true) on the device (sol-r_amd/csrc/solr_tree.hip): the reference's box grid -
 * processBoxes, processOutterBoxes, streamDataToGPU, GPUKernel.cpp:917-1281 - built from the scene's
 * primitives and flattened, bit for bit the tree the host builder makes.
 *   primitives   the scene's primitives in index order as Primitive records (what setPrimitive stored)
 *   emissive     one byte per primitive: 1 if its material's innerIllumination.x != 0
 *   minPos, maxPos, viewDistance   the scene extent the host keeps (GPUKernel.cpp:671-678) and SceneInfo's
 *   boxes        out: the flattened node list (capacity boxCapacity); *nbBoxes receives its length
 *   order        out: nbPrimitives ints: order[k] = index of the primitive streamed k-th; the first *nbLamps
 *                are the lamps (the host fills its Primitive records, lamp and light lists from it)
 * Returns the tree depth (>= 1); -2 when the scene is one of the few cases left to the host builder (no
 * emissive primitive, key collisions with the lamp box, more than NB_MAX_BOXES nodes: solr_tree.hip lists
 * them); -1 on an error, text in solr_hip_build_tree_message(). */
int solr_hip_build_tree(const Primitive *primitives, const unsigned char *emissive, int nbPrimitives,
                        const float minPos[3], const float maxPos[3], float viewDistance, BoundingBox *boxes,
                        int boxCapacity, int *order, int *nbBoxes, int *nbLamps);
const char *solr_hip_build_tree_message(void);

/* Device pointers of the current per-pixel buffers (strip-sized), for
 * collectives issued by the launcher (RCCL gather of the RGB strip). */
void *solr_hip_device_bitmap(void);
void *solr_hip_device_primitive_ids(void);
void *solr_hip_device_postprocessing(void);
/* Render into caller-owned device memory instead (e.g. a torch tensor that
 * is the send buffer of the gather); NULL restores the engine's own buffer. */
void solr_hip_bind_device_bitmap(void *deviceBitmap);

/* h2d_randoms for frames beyond the reference's 1920 x 1080 limit: `count` >= MAX_BITMAP_SIZE values, of
 * which the renderer's random-index expressions (CudaRayTracer.cu:475: pixel index + timestamp % (N - 2))
 * can reach width * height + 10 000.  Reads past `count` give 0. */
void solr_hip_h2d_randoms_sized(const float *randoms, long count);

/* Pipelined read-back of the image: after cudaRender, solr_hip_d2h_image_async enqueues the copy of the RGB image
 * of the frame rendered last (this process's strip at its place in a full-size image) into a page-locked host image
 * owned by the engine, on a copy stream behind that frame's kernel, and returns a ticket (>= 0) at once; with
 * solr_hip_set_frames_in_flight(2 ... 4) the next frames render while it lands.  solr_hip_image_wait(ticket) waits
 * for that copy alone and returns the host image, valid until five more tickets have been handed out (a ring of six).
 * A ticket carries its generation: waiting on one whose image has since been handed out again - or re-allocated for
 * a larger frame - is an error (NULL, solr_hip_last_error), never another frame's image.  With several in-process
 * devices every device copies its strip into the same image and the wait is for all of them.  d2h_bitmap
 * (CudaRayTracer.cu:1647-1672: wait for the frame, then copy, nothing rendering meanwhile) keeps working next to it;
 * the primitive ids are still fetched with d2h_bitmap when picking asks.  HipKernel::setFramesInFlight builds the
 * reference's render_begin / render_end protocol on this. */
int solr_hip_d2h_image_async(void);
const BitmapBuffer *solr_hip_image_wait(int ticket);
/* where the copy of solr_hip_d2h_image_async is enqueued: 0 (default) a copy stream of its own behind the frame's
 * kernel - best for whole frames with two buffer sets; 1 the frame's own stream (it delays that stream's next frame
 * only) - best for a rank's strip of an N-GPU frame with three buffer sets (profiles/r4/readback_routes.txt) */
void solr_hip_set_copy_route(int onTheFramesOwnStream);
/* One frame at a time (the reference's render_begin ... render_end, CudaKernel.cpp:174-312): kernel, then 6 MB over PCIe,
 * nothing rendering meanwhile.  solr_hip_stream_next_image(1) before a cudaRender makes that frame's waves say - in words
 * of page-locked host memory - when a band of tile rows is complete (five bands; sol-r_amd/csrc/renderer.h,
 * ImageStreaming); solr_hip_d2h_streamed_image(image) behind it copies every band to its rows of `image` (host memory of
 * any kind) as soon as its word has come - or the kernel has ended, whichever is first - and returns when the last has
 * landed: 1 done, 0 the frame rendered last was not such a frame and nothing was copied (d2h_bitmap then), -1 error.
 * The image leaves while the rows below still render: 0.30 instead of 0.39 ms per Cornell frame at 1080p
 * (profiles/r6/api_frame_cornell.txt).  Same bytes.  Applies to whole frames of one device whose RGB image the renderer
 * itself writes (no neighbourhood post-process, ftRGB), one frame in flight, no tile that the cost-ordered launch would
 * render as four quadrant waves (such a frame keeps that order); any other frame is rendered and read back as before.
 * solr_hip_stream_next_image returns 1 when frames can be streamed, 0 when not (SOLR_HIP_NO_IMAGE_STREAMING=1); on = -1:
 * was the frame rendered last such a frame (1 / 0); on = -2: how many images have left in bands since initialize_scene.
 * HipKernel's render_begin / render_end and SolR_RunKernel do this when they run one frame at a time: an unchanged host
 * gets it. */
int solr_hip_stream_next_image(int on);
int solr_hip_d2h_streamed_image(BitmapBuffer *image);
/* the same with the primitive ids, as d2h_bitmap hands both over (CudaRayTracer.cu:1647-1672: 16 bytes per pixel, five times
 * the image, every frame): solr_hip_stream_next_image(2) before the cudaRender, then this in place of d2h_bitmap.  The 39 MB of
 * a 1080p frame take PCIe 0.76 ms whatever the kernel does; in bands they start 0.17 ms earlier (profiles/r6). */
int solr_hip_d2h_streamed(BitmapBuffer *image, PrimitiveXYIdBuffer *primitivesXYIds);
/* One host image for all ranks of a multi-process job: the ring of page-locked images becomes a POSIX shared-memory
 * segment `name` ("/something"; rank 0 creates it, the others open it), registered with the HIP runtime in every
 * process.  Every rank's solr_hip_d2h_image_async then copies its strip, over its own PCIe link, to its rows of the
 * same image - what the reference's d2h_bitmap does with the devices of its one process (CudaRayTracer.cu:1647-1672)
 * - and solr_hip_image_wait on rank 0 returns when every rank's strip of that frame has landed (the others return
 * when theirs has); the image stays valid until rank 0's NEXT solr_hip_image_wait (only then may a rank that runs
 * ahead overwrite its rows).  The ranks run the same program: the same sequence of tickets.  After reshape_scene (the frame
 * size is the segment's); undone by finalize_scene.  0, or -1 with the error set. */
int solr_hip_image_share(const char *name, int rank, int world);
/* after a barrier that follows every rank's solr_hip_image_share: the root removes the segment's NAME (the mappings
 * stay), so that a job that dies later leaves nothing in /dev/shm; a no-op elsewhere */
void solr_hip_image_share_sealed(void);
void solr_hip_image_unshare(void);

/* Float framebuffer of the strip back to the host (parity tests) */
void solr_hip_d2h_postprocessing(PostProcessingBuffer *hostBuffer);
/* Host float framebuffer / primitive ids into the strip (accumulation tests) */
void solr_hip_h2d_postprocessing(const PostProcessingBuffer *hostBuffer, const PrimitiveXYIdBuffer *ids);

/* Pointer-argument twins of the by-value entry points */
void solr_hip_initialize(const SceneInfo *sceneInfo);
void solr_hip_reshape(const SceneInfo *sceneInfo);
void solr_hip_render(const SceneInfo *sceneInfo, const vec4i *objects, const PostProcessingInfo *postProcessingInfo,
                     const float origin[3], const float direction[3], const float angles[4]);
void solr_hip_d2h(const SceneInfo *sceneInfo, BitmapBuffer *bitmap, PrimitiveXYIdBuffer *primitivesXYIds);

/* Kernel timing: enable = n > 0 makes every n-th cudaRender/solr_hip_render bracket its
 * launch with HIP events on the engine's stream (1: every launch; 0: off).  solr_hip_kernel_time
 * synchronises, returns the summed milliseconds of the renderer kernel and
 * writes the number of timed launches; reset != 0 clears both afterwards. */
void solr_hip_enable_timing(int enable);
double solr_hip_kernel_time(int *nbLaunches, int reset);
/* the same launches one by one: each one's kernel duration and the time from the end of the timed launch before it
 * to its own end (-1 for the first); call before the resetting solr_hip_kernel_time.  Returns the samples written. */
int solr_hip_timing_samples(float *kernelMs, float *intervalMs, int capacity);

/* Frames in flight.  n = 2..4 (whole frames gain nothing beyond 3, a 1/8 strip of a multi-GPU frame up to 4): consecutive first-pass frames (pathTracingIteration == 0) rotate over n
 * streams and n sets of per-pixel buffers owned by the engine, so that the tail of a frame - a few long
 * waves on an otherwise idle chip - overlaps the start of the next one (the 100k-triangle frame:
 * 0.76 -> 0.60 ms with two).  Refinement and accumulation passes stay on the set of the pass before them.
 * d2h_bitmap and the device-pointer accessors refer to the frame rendered last; every upload waits for
 * both streams.  Ignored (one frame in flight) while the engine runs on a caller's stream.  Default 1,
 * the reference's behaviour. */
void solr_hip_set_frames_in_flight(int n);
int solr_hip_get_frames_in_flight(void);
/* For callers that chain their own work (a collective, a copy) behind a frame without blocking the host:
 * the HIP stream of buffer set 0 / 1, and the set the next first-pass frame will be issued on. */
void *solr_hip_flight_stream(int flight);
int solr_hip_next_flight(void);
/* n streams of the caller for the buffer sets (e.g. streams of a framework's pool, which the framework
 * has already spread over the hardware queues); they stay the caller's. */
void solr_hip_set_flight_streams(void *const *streams, int n);

/* Cost-ordered launch.  Every wave records what its 8x8 tile cost; when recent frames of the same
 * geometry had a heavy tail (the most expensive tile > 2 x the mean) the following frames are launched
 * most-expensive-first (a one-workgroup sorting kernel every 64th frame).  Changes the order of work
 * only.  mode 0: off, 1: automatic (default), 2: always.  solr_hip_tile_scheduling_active() tells
 * whether the last render was launched in cost order. */
void solr_hip_set_tile_scheduling(int mode);
int solr_hip_tile_scheduling_active(void);
int solr_hip_split_tiles(void); /* tiles the current order renders as four quadrant waves each */

/* Load-balance diagnostics: when enabled every render records, per 8x8 tile
 * (one wavefront), the 100 MHz timestamps at which its wave started and ended.
 * solr_hip_tile_clocks synchronises and copies {start, end} pairs of the last
 * render in launch order (tile = ty * tilesX + tx); returns the number of tiles. */
void solr_hip_enable_tile_clocks(int enable);
int solr_hip_tile_clocks(unsigned long long *clocks, int capacityTiles);

/* Renders the frame once with the counting variant of the kernel and returns
 * the number of box-tree traversals: counts[0] = closest-hit walks
 * (intersectionWithPrimitives calls), counts[1] = shadow walks
 * (processShadows calls), counts[2] = box nodes visited (summed over lanes),
 * counts[3] = primitive tests (summed over lanes); counts[4..7] are the same
 * four quantities counted once per WAVE (nodes the wave stepped through,
 * primitive tests it issued, closest-hit and shadow walks it ran), which
 * gives the SIMD efficiency of the wave-synchronous walk.  Output buffers
 * are written exactly as by the normal variant. */
void solr_hip_render_counting(const SceneInfo *sceneInfo, const vec4i *objects,
                              const PostProcessingInfo *postProcessingInfo, const float origin[3],
                              const float direction[3], const float angles[4], unsigned long long counts[8]);

/* A/B measurements: 0 = automatic; 3 = walk the node list exactly as uploaded (no collapsed chains, no
 * grouping nodes); 4 = always the all-features kernel; 5 = no grouping nodes (takes effect at the next
 * h2d_scene); 6 = no order-free lists (every walk in the reference's order); 7 = the whole colour stack in LDS
 * however deep a frame may bounce (as before round 5: nine waves per CU for ten bounces; otherwise three slots in LDS,
 * the deeper ones in HBM); 8 = every walk on the reference's own leaf boxes (no thin copies of the leaves that hold
 * plain axis planes, rt_device.h tightRay); 9 = k_ambientOcclusion with a fixed stride of tiles per workgroup instead of
 * the frame's heavy tiles first (solr_post.hip); 10 = the colour-stack slots an F_STACK frame keeps in HBM filled with
 * NaNs before every launch (they are never zeroed: no slot is read before the frame has written it); 12 = no walk takes
 * the copies of the order-free lists with sorted bounds (the node loop without its min / max, rt_device.h
 * SOLR_ORDER_SORTED); 13 = a streamed frame (solr_hip_stream_next_image) whose waves do not write the bands' words: the
 * host goes by the end of the kernel (solr_hip_d2h_streamed_image); 14 = the tile counters of streamed frames are zeroed
 * every third frame, as they are when a count nears 2^32.  Every setting renders the same frame. */
void solr_hip_set_variant(int variant);
/* Bounce rays (|direction| = 1 - rayEpsilon) of the long-list triangle kernels on the order-free lists, checked: lanes
 * whose hit has a rival the reference's cut-off could have preferred are walked again in the reference's order
 * (rt_device.h closestHitWalk).  -1 (default): on with frames in flight (solr_hip_set_frames_in_flight >= 2: the saved
 * work shows), off one frame at a time (the repeated lanes lengthen the frame's longest tiles); 0 / 1: off / on.
 * Every setting renders the same frame.  solr_hip_short_ray_lists: what the next frame will do (engine 0). */
void solr_hip_set_short_ray_lists(int mode);
int solr_hip_short_ray_lists(void);
/* nodes per order-free list of the resident scene when long rays' closest-hit walks use them, else 0 */
int solr_hip_order_free_nodes(void);
/* 1 if the shadow walks take them as well (nothing in the scene is transparent or a textured plane), else 0 */
int solr_hip_order_free_shadows(void);
int solr_hip_get_variant(void);

/* Animated scenes.  The reference re-runs GPUKernel::rotatePrimitives (GPUKernel.cpp:1378-1460: rotate
 * the primitives of the level-0 boxes, refit every level) and compactBoxes(false) (:1151-1281: flatten
 * again) on the host and uploads the whole scene for every frame of a rotating model
 * (apps/scenes/science/MoleculeScene.cpp:75-81).  The flattened tree keeps its shape under that, so the
 * engine can do the same arithmetic on the resident scene:
 *   solr_hip_set_movable        after h2d_scene: per flattened primitive, 1 if rotatePrimitives moves it
 *                               (in a level-0 box, movable, not the camera primitive);
 *   solr_hip_rotate_primitives  rotation centre, cos and sin of angles.x/.y/.z as the host computes them
 *                               (cosf / sinf), sceneInfo.viewDistance (the seed of the outer boxes).
 *                               Returns 1 when the resident scene now holds, bit for bit, what the host
 *                               rotation + a fresh h2d_scene would have produced for the reference's node
 *                               list and primitives (the engine's own walk-order list is refitted the same way;
 *                               the reference's list itself only when a frame or a read-back needs it),
 *                               0 when it cannot serve the request (no flags, a tree the engine did not
 *                               validate as nested, viewDistance > 1e6): nothing changed, take the host
 *                               route;
 *   solr_hip_device_rotations   how many requests were served since initialize_scene;
 *   solr_hip_read_nodes / _primitives   the resident records, for tests (float4 rows: 2 per node in the
 *                               layout of sol-r_amd/csrc/scene_layout.h, 8 per primitive); they return the
 *                               number of rows, with rows == NULL only that.  `exact`: 1 the reference's
 *                               list, 0 the walk-order list, 2 ... 9 the order-free list of octant exact - 2. */
void solr_hip_set_movable(const unsigned char *flags, int nbPrimitives);
int solr_hip_rotate_primitives(const float center[3], const float cosAngles[3], const float sinAngles[3],
                               float viewDistance);
int solr_hip_device_rotations(void);
int solr_hip_read_nodes(int exact, float *rows, int capacityRows);
int solr_hip_read_primitives(float *rows, int capacityRows);

/* Bytes of HBM this engine currently holds for {scene planes, materials,
 * textures, per-pixel buffers}; for DESIGN.md's layout table and tests. */
void solr_hip_memory_usage(unsigned long long bytes[4]);

#ifdef __cplusplus
}
#endif
#endif /* SOLR_HIP_H */

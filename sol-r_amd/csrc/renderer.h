/*
 * renderer.h - what the host side of the engine (solr_launch.hip) and the translation units that hold the renderer's
 * instantiations (the files under csrc/rows) share: the frame's arguments, the kernel's signature, and one look-up per row of
 * renderImpl's table.  The kernel template itself is renderer_kernel.h, included by the row files only - so that an
 * experiment on one instantiation rebuilds one object (make -j: eight objects side by side instead of one 70-second
 * translation unit).  gfx950 only.
 */
#ifndef SOLR_RENDERER_H
#define SOLR_RENDERER_H

#include <hip/hip_runtime.h>

#include "../../include/solr_hip.h"
#include "rt_device.h"

using namespace solrdev;

struct FrameArgs
{
    SceneInfo si;
    PostProcessingInfo ppi;
    float ox, oy, oz;     /* camera position */
    float dx, dy, dz;     /* camera look-at */
    float ax, ay, az, aw; /* camera angles, w = field scale */
    Trig trig;            /* cos/sin of the angles, evaluated on the host */
    float stepx, stepy;   /* the pixel pitch of the perspective cameras (CRT:490-492): the same three binary32 operations, once */
    int firstRow;         /* first image row of this process's strip */
    int nbRows;           /* rows in the strip */
    int tilesX;
    unsigned tileMagic;   /* tile / tilesX = (tile * tileMagic) >> (32 + tileShift) for every tile of the frame (checked on the host) */
    int tileShift;
    int fuseDefault;      /* bit 0: write the RGB bitmap from the renderer; bit 1: ImageStreaming, the frame counts its tiles (rowDone,
                           * streamPlan, streamSerial below); bit 2: ... and its primitive ids leave with the image.  One word that
                           * the epilogue reads anyway: a frame that is not streamed loads nothing more than it did */
    int stackSlots;       /* colour-stack slots per lane in LDS */
    float4 *deepStack;    /* F_STACK instantiations: the slots beyond them, [slot - stackSlots][pixel of the strip] */
    long deepStride;      /* float4s between two slots of a pixel = pixels of the strip */
    float focusDepth;     /* ctVR: depth of the focus pixel before this frame (k_3DVisionRenderer) */
    unsigned long long *tileClock; /* diagnostics: {start, end} of every tile in 100 MHz ticks, or null */
    /* cost-ordered launch (see TileScheduling below); all null when off */
    unsigned *tileCost;        /* out: duration of every tile of this frame, 100 MHz ticks */
    const unsigned *tileOrder; /* in: workgroup -> order entry (see ORDER_* below), most expensive tiles first */
    int nbTiles;               /* tiles of the frame; the ordered launch has 3 * SPLIT_TILES_MAX workgroups more */
    /* ImageStreaming (see StreamPlan below); read only when fuseDefault says so */
    unsigned *rowDone;                    /* one word per tile row, 64 words apart: units of it rendered, over all streamed frames */
    const struct StreamPlan *streamPlan;  /* bands of tile rows and the words the host watches */
    unsigned streamSerial;                /* this frame is the n-th streamed frame since the counters were zeroed */
};

/* ImageStreaming.  A host that takes one frame at a time waits for the kernel and then for 6 MB over PCIe (0.26 + 0.13 ms
 * for the Cornell box at 1080p), and nothing renders meanwhile.  The image of such a frame leaves in bands of tile rows
 * WHILE the kernel renders the rows below: every wave, its pixels stored with device scope and the stores waited for,
 * counts itself into its tile row (4 units a whole tile, 1 a quadrant wave of a split tile); the wave that completes a
 * row counts the row into its band, and the wave that completes a band stores the frame's serial into the band's word
 * in page-locked host memory.  The host, which would be waiting for the kernel anyway, watches the words and sends a
 * band's copy off when its word has come - or when the kernel has ended, whichever is first: everything the kernel
 * wrote is in memory then, so the scheme cannot lose a byte or wait for ever whatever becomes of a word
 * (solr_image_ring.hip, solr_hip_d2h_streamed_image).  No launch boundary, no second kernel, no queue that waits on
 * another: what was tried instead - a resident copier kernel, the last wave of a row copying it, the command processor
 * waiting on the words (hipStreamWaitValue32: 0.334 ms per Cornell frame against 0.318, and a deadlock under a tool that
 * runs one dispatch at a time over all queues) - is in tools/stream_probe.hip / profiles/r6/stream_probe.txt.  Counters
 * only ever grow; the host zeroes them when the frame geometry changes or the count nears 2^32. */
#define SOLR_STREAM_BANDS_MAX 8
struct StreamPlan
{
    unsigned *bandDone; /* one word per band, 64 words apart: rows of it complete, over all streamed frames */
    unsigned *hostWord; /* one word per band in page-locked host memory: the serial of the newest frame whose band is complete */
    int bands;
    int firstRow[SOLR_STREAM_BANDS_MAX + 1]; /* band b is the tile rows firstRow[b] ... firstRow[b + 1] - 1; firstRow[bands] = all of them */
};

/* the same cuts in tiles, for the sort that launches a streamed frame band after band (k_orderTiles); bands = 0: by cost alone */
struct BandCuts
{
    int bands;
    int heavyShare; /* the heaviest n / heavyShare tiles go first, wherever they lie (k_orderTiles) */
    int firstTile[SOLR_STREAM_BANDS_MAX + 1];
};

/* An entry of the launch order: the tile in bits 0-27, and in bits 28-30 which part of it this wave renders -
 * 0 the whole 8 x 8 tile, 1-4 one of its 4 x 4 quadrants (the few most expensive tiles are rendered by four
 * waves, see k_orderTiles).  ORDER_NOTHING pads the list to its fixed length. */
#define ORDER_TILE_MASK 0x03ffffffu
#define ORDER_PART_SHIFT 26
/* a split tile is rendered by (2^SOLR_SPLIT_LOG2)^2 waves: 1 = four 4 x 4 quadrants, 2 = sixteen 2 x 2 blocks (experiments) */
#ifndef SOLR_SPLIT_LOG2
#define SOLR_SPLIT_LOG2 1
#endif
#define SPLIT_PARTS (1 << (2 * SOLR_SPLIT_LOG2))
#define ORDER_NOTHING 0xffffffffu
#define SOLR_TIMING_SLOTS (160000ul) /* timing build: workgroups of the largest frame it is used on (3840 x 2160 + split tiles) */
#define SPLIT_TILES_MAX 256

/* a wave's 64 pixels: TILE_W x TILE_H (8 x 8; -DSOLR_TILE_W_LOG2=4 is 16 x 4 ... an experiment) */
#ifndef SOLR_TILE_W_LOG2
#define SOLR_TILE_W_LOG2 3
#endif
#define TILE_W (1 << SOLR_TILE_W_LOG2)
#define TILE_H (64 >> SOLR_TILE_W_LOG2)
#define TILE 8 /* rows the strips of a multi-process job are aligned to */
#define WAVE 64

/* device view of PostProcessingBuffer (same 32-byte layout, HIP vector types) */
struct PixelRecord
{
    float4 colorInfo;
    float4 sceneInfo;
};
static_assert(sizeof(PixelRecord) == sizeof(PostProcessingBuffer), "PixelRecord layout");


/* depths of the rows next to a strip that belong to the ranks above and below (multi-GPU frames: §6 of
 * DESIGN.md): `above` holds the nbAbove rows just above the strip, `below` the nbBelow rows just below it */
struct DepthHalo
{
    const float *above, *below;
    int nbAbove, nbBelow;
};

/* the renderer kernel and the replay of its walks (renderer_kernel.h), as the host launches them */
typedef void (*RendererFn)(const SceneArgs, const FrameArgs, PixelRecord *, int4 *, unsigned char *, unsigned long long *);
typedef void (*WalkBoundFn)(const SceneArgs, const char *, unsigned *, unsigned *);

namespace solrrows
{
/* k_standardRenderer<count, features, volume>, or null when no row file instantiates it.  count: 0 a frame, 1 the ray
 * census, 2 a frame that records its walks (the four lean rows only).  features: enum Feature of rt_device.h, with
 * F_DEEP where the three-bank node loop is wanted. */
RendererFn renderer(int count, int features, bool volume);
/* k_walkBound<features> of lean row `row` of renderImpl's table (0 ... 3), or null */
WalkBoundFn walkBound(int row, int features);
/* one look-up per row file (each returns null for what it does not hold) */
RendererFn spherePlane(int count, int features);
RendererFn sphereTriangle(int count, int features);
RendererFn sphereCylinder(int count, int features);
RendererFn untexturedMix(int count, int features);
RendererFn textured(int count, int features);
RendererFn specialCameras(int count, int features);
RendererFn everything(int count, int features, bool volume);
WalkBoundFn walkBoundRow0(int features);
WalkBoundFn walkBoundRow1(int features);
WalkBoundFn walkBoundRow2(int features);
WalkBoundFn walkBoundRow3(int features);
} // namespace solrrows

#endif

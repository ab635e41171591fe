/* solr_lists.hip: the order-free node lists built on the device (see that file).  Internal to the library. */
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

/* the lists left where they were made: three hipMalloc'ed buffers the caller frees (rows: 16 float4 per node of a
 * list, start / origin: 8 ints) */
struct SolrDeviceLists
{
    float4 *rows = nullptr;
    int *start = nullptr;
    int *origin = nullptr;
};

/* rows / start / origin: a nested node list (2 float4 rows per node as in scene_layout.h, the leaf start indices,
 * for every node the node of the reference's list it is) - in host memory; or, with origin == nullptr, rows and start
 * in DEVICE memory (the arena's exact list) and every node its own origin.  threshold: pruneInnerNodes' (1 test).  Output as
 * buildFreeOrderLists': 8 lists of the returned length, one after the other - in the three vectors, or, when `stay` is
 * given, in device buffers handed over through it (the vectors are then left alone: 38 MB that need not cross the bus
 * twice for a 100k-primitive scene).  Returns the length of a list, or -1 when the scene is left to the host builder. */
int solrBuildOrderFreeListsOnDevice(const float4 *rows, const int *start, const int *origin, int n, double threshold,
                                    std::vector<float4> &outRows, std::vector<int> &outStart, std::vector<int> &outOrigin,
                                    int *nbPruned, hipStream_t stream, SolrDeviceLists *stay = nullptr);

/* pruneInnerNodes' decisions (solr_scene.hip) for a nested node list: keep[i] = 0 for the inner nodes that are left
 * out.  Returns how many, or -1 when left to the host. */
int solrPruneDecisionsOnDevice(const float4 *rows, int n, double threshold, std::vector<char> &keep, hipStream_t stream);

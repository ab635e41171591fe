/* solr_lists.hip: the order-free node lists built on the device (see that file).  Internal to the library. */
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

/* rows / start / origin: a nested node list (2 float4 rows per node as in scene_layout.h, the leaf start indices,
 * for every node the node of the reference's list it is).  threshold: pruneInnerNodes' (1 test).  Output as
 * buildFreeOrderLists': 8 lists of the returned length, one after the other.  Returns the length of a list, or -1
 * when the scene is left to the host builder. */
int solrBuildOrderFreeListsOnDevice(const float4 *rows, const int *start, const int *origin, int n, double threshold,
                                    std::vector<float4> &outRows, std::vector<int> &outStart, std::vector<int> &outOrigin,
                                    int *nbPruned, hipStream_t stream);

/* pruneInnerNodes' decisions (solr_hip.hip) for a nested node list: keep[i] = 0 for the inner nodes that are left
 * out.  Returns how many, or -1 when left to the host. */
int solrPruneDecisionsOnDevice(const float4 *rows, int n, double threshold, std::vector<char> &keep, hipStream_t stream);

/* Scratch memory of the device-side builders (solr_tree.hip, solr_lists.hip).  Internal to the library.
 *
 * A build makes about a hundred short-lived arrays; a hipMalloc / hipFree pair for each (every hipFree waits for the
 * device) was a third of the time of a 100k-primitive build.  The builders take them from one allocation instead:
 * bump-allocated, given back all at once when the build ends, kept for the next build and sized by what the largest
 * build so far asked for.  What does not fit falls back to hipMalloc (correct, slower) and the pool grows next time. */
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>

struct SolrScratchPool
{
    std::mutex busy; /* one build at a time */
    char *base = nullptr;
    size_t capacity = 0, used = 0, asked = 0, wanted = 0;
    int device = -1;

    /* start of a build on the current device: everything handed out before is void */
    void begin(size_t estimate)
    {
        int now = -1;
        (void)hipGetDevice(&now);
        const size_t want = estimate > wanted ? estimate : wanted;
        if (base && (now != device || capacity < want))
            release();
        if (!base && want)
        {
            if (hipMalloc((void **)&base, want) == hipSuccess)
                capacity = want, device = now;
            else
                base = nullptr, capacity = 0, (void)hipGetLastError();
        }
        used = asked = 0;
    }
    /* 256-byte aligned, nullptr when the pool is full (the caller allocates for itself then) */
    void *take(size_t bytes)
    {
        const size_t rounded = (bytes + 255) & ~(size_t)255;
        asked += rounded;
        if (!base || used + rounded > capacity)
            return nullptr;
        void *p = base + used;
        used += rounded;
        return p;
    }
    void end()
    {
        if (asked > wanted)
            wanted = asked + asked / 8;
    }
    void release()
    {
        if (base)
            (void)hipFree(base);
        base = nullptr;
        capacity = used = 0;
        device = -1;
    }
};

/* the library's one pool (solr_tree.hip) */
SolrScratchPool &solrScratchPool();

/* Host side of the same story: the host images of a 100k-primitive scene are a dozen vectors of 1-13 MB that live for
 * one upload.  glibc serves such sizes with mmap and gives them back with munmap, so every upload touches ~100 MB of
 * fresh pages - 8 of 41 ms for the molecule scene on the measured box (page faults of a virtual machine are dear).
 * Called once per process by the first engine entry point: large blocks come from the heap and stay in it when freed
 * (up to 512 MB are kept).  SOLR_HIP_MALLOC_DEFAULTS=1 leaves the allocator alone. */
void solrTuneHostAllocator();

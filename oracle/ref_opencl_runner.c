/*
 * ref_opencl_runner.c - TEST INFRASTRUCTURE ONLY (see oracle/solr_oracle.h).
 *
 * Runs the REFERENCE'S OWN per-pixel renderer on the GPU: its OpenCL engine's
 * kernels k_standardRenderer and k_default (solr/engines/opencl/RayTracer.cl:2448,
 * 2926), compiled offline for gfx950 by oracle/Makefile (target `ref`) from the
 * source where it lies under /root/reference into oracle/_ref/RayTracer_gfx950.co.
 * Nothing of the reference is copied here: this file is only the host side that
 * the reference's OpenCLKernel::render_begin plays (OpenCLKernel.cpp:793-815, the
 * twenty kernel arguments in that order, then k_default :884-889), written against
 * the OpenCL C API, plus the conversion of this repository's flat scene arrays
 * (CUDA-flavour records, include/solr_types.h) into the records the .cl file
 * declares (RayTracer.cl:117-289: float4-based BoundingBox 48 B, Primitive 160 B,
 * LightInformation 48 B; SceneInfo, Material and PostProcessingInfo have the same
 * layout in both flavours).
 *
 * Purpose: pin oracle/solr_oracle.c against outputs of the reference itself
 * (tests/test_reference_opencl.py).  The OpenCL engine is an older sibling of the
 * CUDA engine the oracle restates: float4 arithmetic, and a primary-ray jitter it
 * applies on every pass (RayTracer.cl:2526-2527) where the CUDA engine only does
 * so on accumulation passes (CudaRayTracer.cu:515-522); the agreement is therefore
 * checked at image level with stated tolerances, not bit for bit.
 */
#define CL_TARGET_OPENCL_VERSION 120
#include <CL/cl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/solr_types.h"

typedef struct
{
    cl_float4 parameters[2];
    cl_int nbPrimitives;
    cl_int startIndex;
    cl_int2 indexForNextBox;
} ClBoundingBox;

typedef struct
{
    cl_float4 p0, p1, p2, n0, n1, n2, size;
    cl_int type, index, materialId;
    cl_int pad0;
    cl_float2 vt0, vt1, vt2;
    cl_int pad1[2];
} ClPrimitive;

typedef struct
{
    cl_int primitiveId, materialId;
    cl_int pad[2];
    cl_float4 location, color;
} ClLightInformation;

_Static_assert(sizeof(ClBoundingBox) == 48, "ClBoundingBox");
_Static_assert(sizeof(ClPrimitive) == 160, "ClPrimitive");
_Static_assert(sizeof(ClLightInformation) == 48, "ClLightInformation");

static cl_float4 f4(vec3f v)
{
    cl_float4 r;
    r.s[0] = v.x;
    r.s[1] = v.y;
    r.s[2] = v.z;
    r.s[3] = 0.f;
    return r;
}

static cl_mem makeBuffer(cl_context ctx, cl_mem_flags flags, size_t bytes, const void *src, cl_int *err)
{
    return clCreateBuffer(ctx, flags | (src ? CL_MEM_COPY_HOST_PTR : 0), bytes, (void *)src, err);
}

#define FAIL(code, ...)                                                                                                \
    do                                                                                                                 \
    {                                                                                                                  \
        if (log && logCapacity > 0)                                                                                    \
            snprintf(log, (size_t)logCapacity, __VA_ARGS__);                                                           \
        status = (code);                                                                                               \
        goto done;                                                                                                     \
    } while (0)
#define CHECK(call, what)                                                                                              \
    do                                                                                                                 \
    {                                                                                                                  \
        cl_int e_ = (call);                                                                                            \
        if (e_ != CL_SUCCESS)                                                                                          \
            FAIL(e_ ? e_ : -1, "%s failed: %d", what, (int)e_);                                                        \
    } while (0)

int solr_ref_opencl_render(const char *codeObjectPath, const BoundingBox *boxes, int nbBoxes, const Primitive *prims,
                           int nbPrims, const LightInformation *lights, int nbLights, int nbLamps,
                           const Material *materials, int nbMaterials, const float *randoms, int nbRandoms,
                           const SceneInfo *sceneInfo, const PostProcessingInfo *ppInfo, const float eye[3],
                           const float direction[3], const float angles[4], float *ppOut, int *idsOut,
                           unsigned char *rgbOut, int repeats, double *rendererMs, char *log, int logCapacity)
{
    int status = 0;
    cl_platform_id platforms[8];
    cl_uint nbPlatforms = 0;
    cl_device_id device = NULL;
    cl_context ctx = NULL;
    cl_command_queue queue = NULL;
    cl_program program = NULL;
    cl_kernel kRender = NULL, kDefault = NULL;
    cl_mem dBoxes = NULL, dPrims = NULL, dLights = NULL, dMaterials = NULL, dTextures = NULL, dRandoms = NULL,
           dPp = NULL, dIds = NULL, dBitmap = NULL;
    unsigned char *binary = NULL;
    ClBoundingBox *cb = NULL;
    ClPrimitive *cp = NULL;
    ClLightInformation *cl = NULL;
    cl_int err = 0;
    const int W = sceneInfo->size.x, H = sceneInfo->size.y;

    if (W <= 0 || H <= 0 || (W % 8) || (H % 8))
        FAIL(-2, "image size must be a positive multiple of 8");

    FILE *f = fopen(codeObjectPath, "rb");
    if (!f)
        FAIL(-3, "cannot open %s (run `make -C oracle ref` where /root/reference exists)", codeObjectPath);
    fseek(f, 0, SEEK_END);
    size_t binarySize = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    binary = (unsigned char *)malloc(binarySize);
    if (!binary || fread(binary, 1, binarySize, f) != binarySize)
    {
        fclose(f);
        FAIL(-3, "cannot read %s", codeObjectPath);
    }
    fclose(f);

    CHECK(clGetPlatformIDs(8, platforms, &nbPlatforms), "clGetPlatformIDs");
    for (cl_uint p = 0; p < nbPlatforms && !device; ++p)
    {
        cl_uint n = 0;
        if (clGetDeviceIDs(platforms[p], CL_DEVICE_TYPE_GPU, 1, &device, &n) != CL_SUCCESS || n == 0)
            device = NULL;
    }
    if (!device)
        FAIL(-4, "no OpenCL GPU device");
    ctx = clCreateContext(NULL, 1, &device, NULL, NULL, &err);
    CHECK(err, "clCreateContext");
    queue = clCreateCommandQueue(ctx, device, CL_QUEUE_PROFILING_ENABLE, &err);
    CHECK(err, "clCreateCommandQueue");
    {
        const unsigned char *bins[1] = {binary};
        cl_int binStatus = 0;
        program = clCreateProgramWithBinary(ctx, 1, &device, &binarySize, bins, &binStatus, &err);
        CHECK(err, "clCreateProgramWithBinary");
        err = clBuildProgram(program, 1, &device, "", NULL, NULL);
        if (err != CL_SUCCESS)
        {
            char buildLog[2048] = "";
            clGetProgramBuildInfo(program, device, CL_PROGRAM_BUILD_LOG, sizeof(buildLog) - 1, buildLog, NULL);
            FAIL(err, "clBuildProgram failed: %d: %.1800s", (int)err, buildLog);
        }
    }
    kRender = clCreateKernel(program, "k_standardRenderer", &err);
    CHECK(err, "clCreateKernel(k_standardRenderer)");
    kDefault = clCreateKernel(program, "k_default", &err);
    CHECK(err, "clCreateKernel(k_default)");

    /* records in the OpenCL engine's layout */
    cb = (ClBoundingBox *)calloc((size_t)(nbBoxes > 0 ? nbBoxes : 1), sizeof(ClBoundingBox));
    cp = (ClPrimitive *)calloc((size_t)(nbPrims > 0 ? nbPrims : 1), sizeof(ClPrimitive));
    cl = (ClLightInformation *)calloc((size_t)(nbLights > 0 ? nbLights : 1), sizeof(ClLightInformation));
    if (!cb || !cp || !cl)
        FAIL(-5, "out of memory");
    for (int i = 0; i < nbBoxes; ++i)
    {
        cb[i].parameters[0] = f4(boxes[i].parameters[0]);
        cb[i].parameters[1] = f4(boxes[i].parameters[1]);
        cb[i].nbPrimitives = boxes[i].nbPrimitives;
        cb[i].startIndex = boxes[i].startIndex;
        cb[i].indexForNextBox.s[0] = boxes[i].indexForNextBox.x;
        cb[i].indexForNextBox.s[1] = boxes[i].indexForNextBox.y;
    }
    for (int i = 0; i < nbPrims; ++i)
    {
        const Primitive *p = &prims[i];
        cp[i].p0 = f4(p->p0);
        cp[i].p1 = f4(p->p1);
        cp[i].p2 = f4(p->p2);
        cp[i].n0 = f4(p->n0);
        cp[i].n1 = f4(p->n1);
        cp[i].n2 = f4(p->n2);
        cp[i].size = f4(p->size);
        cp[i].type = p->type;
        cp[i].index = p->index;
        cp[i].materialId = p->materialId;
        cp[i].vt0.s[0] = p->vt0.x;
        cp[i].vt0.s[1] = p->vt0.y;
        cp[i].vt1.s[0] = p->vt1.x;
        cp[i].vt1.s[1] = p->vt1.y;
        cp[i].vt2.s[0] = p->vt2.x;
        cp[i].vt2.s[1] = p->vt2.y;
    }
    for (int i = 0; i < nbLights; ++i)
    {
        cl[i].primitiveId = lights[i].primitiveId;
        cl[i].materialId = lights[i].materialId;
        cl[i].location = f4(lights[i].location);
        cl[i].color.s[0] = lights[i].color.x;
        cl[i].color.s[1] = lights[i].color.y;
        cl[i].color.s[2] = lights[i].color.z;
        cl[i].color.s[3] = lights[i].color.w;
    }

#define BUF(var, flags, bytes, src)                                                                                    \
    var = makeBuffer(ctx, (flags), (bytes), (const void *)(src), &err);                                                \
    CHECK(err, "clCreateBuffer(" #var ")")
    const size_t pixels = (size_t)W * (size_t)H;
    static const unsigned char noTexture[16] = {0};
    BUF(dBoxes, CL_MEM_READ_ONLY, sizeof(ClBoundingBox) * (size_t)(nbBoxes > 0 ? nbBoxes : 1), cb);
    BUF(dPrims, CL_MEM_READ_ONLY, sizeof(ClPrimitive) * (size_t)(nbPrims > 0 ? nbPrims : 1), cp);
    BUF(dLights, CL_MEM_READ_ONLY, sizeof(ClLightInformation) * (size_t)(nbLights > 0 ? nbLights : 1), cl);
    BUF(dMaterials, CL_MEM_READ_ONLY, sizeof(Material) * (size_t)nbMaterials, materials);
    BUF(dTextures, CL_MEM_READ_ONLY, sizeof(noTexture), noTexture);
    BUF(dRandoms, CL_MEM_READ_ONLY, sizeof(float) * (size_t)nbRandoms, randoms);
    BUF(dPp, CL_MEM_READ_WRITE, 32 * pixels, NULL);
    BUF(dIds, CL_MEM_READ_WRITE, 16 * pixels, NULL);
    BUF(dBitmap, CL_MEM_READ_WRITE, 3 * pixels, NULL);
    {
        /* the frame buffers start zeroed, like a fresh OpenCLKernel */
        void *zero = calloc(pixels, 32);
        if (!zero)
            FAIL(-5, "out of memory");
        err = clEnqueueWriteBuffer(queue, dPp, CL_TRUE, 0, 32 * pixels, zero, 0, NULL, NULL);
        if (err == CL_SUCCESS)
            err = clEnqueueWriteBuffer(queue, dIds, CL_TRUE, 0, 16 * pixels, zero, 0, NULL, NULL);
        free(zero);
        CHECK(err, "clEnqueueWriteBuffer(zero)");
    }

    {
        cl_int2 occupancy;
        occupancy.s[0] = 1;
        occupancy.s[1] = 1;
        cl_int zero = 0, nb = nbBoxes, np = nbPrims, nl = nbLights, nlamps = nbLamps;
        cl_float4 vPos, vDir, vAng;
        vPos.s[0] = eye[0], vPos.s[1] = eye[1], vPos.s[2] = eye[2], vPos.s[3] = 0.f;
        vDir.s[0] = direction[0], vDir.s[1] = direction[1], vDir.s[2] = direction[2], vDir.s[3] = 0.f;
        vAng.s[0] = angles[0], vAng.s[1] = angles[1], vAng.s[2] = angles[2], vAng.s[3] = angles[3];
        int a = 0;
#define ARG(k, v) CHECK(clSetKernelArg(k, a++, sizeof(v), &(v)), "clSetKernelArg")
        /* OpenCLKernel.cpp:793-813 */
        ARG(kRender, occupancy);
        ARG(kRender, zero);
        ARG(kRender, zero);
        ARG(kRender, dBoxes);
        ARG(kRender, nb);
        ARG(kRender, dPrims);
        ARG(kRender, np);
        ARG(kRender, dLights);
        ARG(kRender, nl);
        ARG(kRender, nlamps);
        ARG(kRender, dMaterials);
        ARG(kRender, dTextures);
        ARG(kRender, dRandoms);
        ARG(kRender, vPos);
        ARG(kRender, vDir);
        ARG(kRender, vAng);
        CHECK(clSetKernelArg(kRender, a++, sizeof(SceneInfo), sceneInfo), "clSetKernelArg(sceneInfo)");
        CHECK(clSetKernelArg(kRender, a++, sizeof(PostProcessingInfo), ppInfo), "clSetKernelArg(ppInfo)");
        ARG(kRender, dPp);
        ARG(kRender, dIds);
        const size_t global[2] = {(size_t)W, (size_t)H}, local[2] = {8, 8};
        /* repeats > 1: the same pass again (pass 0 overwrites its outputs), timed with OpenCL events */
        double ms = 0.0;
        for (int rep = 0; rep < (repeats > 1 ? repeats : 1); ++rep)
        {
            cl_event ev = NULL;
            CHECK(clEnqueueNDRangeKernel(queue, kRender, 2, NULL, global, local, 0, NULL, &ev),
                  "clEnqueueNDRangeKernel(k_standardRenderer)");
            CHECK(clWaitForEvents(1, &ev), "clWaitForEvents");
            cl_ulong t0 = 0, t1 = 0;
            clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_START, sizeof(t0), &t0, NULL);
            clGetEventProfilingInfo(ev, CL_PROFILING_COMMAND_END, sizeof(t1), &t1, NULL);
            clReleaseEvent(ev);
            if (rep > 0 || repeats <= 1)
                ms += (double)(t1 - t0) * 1e-6;
        }
        if (rendererMs)
            *rendererMs = ms / (repeats > 1 ? repeats - 1 : 1);
        /* OpenCLKernel.cpp:884-889 */
        a = 0;
        ARG(kDefault, occupancy);
        CHECK(clSetKernelArg(kDefault, a++, sizeof(SceneInfo), sceneInfo), "clSetKernelArg(sceneInfo)");
        ARG(kDefault, dPp);
        ARG(kDefault, dBitmap);
        CHECK(clEnqueueNDRangeKernel(queue, kDefault, 2, NULL, global, local, 0, NULL, NULL),
              "clEnqueueNDRangeKernel(k_default)");
    }
    CHECK(clFinish(queue), "clFinish");
    if (ppOut)
        CHECK(clEnqueueReadBuffer(queue, dPp, CL_TRUE, 0, 32 * pixels, ppOut, 0, NULL, NULL), "read pp");
    if (idsOut)
        CHECK(clEnqueueReadBuffer(queue, dIds, CL_TRUE, 0, 16 * pixels, idsOut, 0, NULL, NULL), "read ids");
    if (rgbOut)
        CHECK(clEnqueueReadBuffer(queue, dBitmap, CL_TRUE, 0, 3 * pixels, rgbOut, 0, NULL, NULL), "read bitmap");

done:
    if (dBoxes) clReleaseMemObject(dBoxes);
    if (dPrims) clReleaseMemObject(dPrims);
    if (dLights) clReleaseMemObject(dLights);
    if (dMaterials) clReleaseMemObject(dMaterials);
    if (dTextures) clReleaseMemObject(dTextures);
    if (dRandoms) clReleaseMemObject(dRandoms);
    if (dPp) clReleaseMemObject(dPp);
    if (dIds) clReleaseMemObject(dIds);
    if (dBitmap) clReleaseMemObject(dBitmap);
    if (kRender) clReleaseKernel(kRender);
    if (kDefault) clReleaseKernel(kDefault);
    if (program) clReleaseProgram(program);
    if (queue) clReleaseCommandQueue(queue);
    if (ctx) clReleaseContext(ctx);
    free(binary);
    free(cb);
    free(cp);
    free(cl);
    return status;
}

/* ---- generic launcher for the probe kernels of oracle/ref_probes.cl (and for the reference's own
 * post-processing kernels): one kernel of a code object over a 1-D or 2-D range with a list of
 * arguments - a value passed as it is, or a buffer that is uploaded before and/or read back after. */
typedef struct
{
    int kind;     /* 0 value, 1 buffer in, 2 buffer out (starts zeroed), 3 buffer in/out */
    void *data;   /* value bytes or host buffer */
    size_t bytes;
} RefArg;

int solr_ref_opencl_run(const char *codeObjectPath, const char *kernelName, int nbArgs, RefArg *args, int dims,
                        const size_t *global, const size_t *local, char *log, int logCapacity)
{
    int status = 0;
    cl_platform_id platforms[8];
    cl_uint nbPlatforms = 0;
    cl_device_id device = NULL;
    cl_context ctx = NULL;
    cl_command_queue queue = NULL;
    cl_program program = NULL;
    cl_kernel kernel = NULL;
    cl_mem *buffers = NULL;
    unsigned char *binary = NULL;
    cl_int err = 0;

    FILE *f = fopen(codeObjectPath, "rb");
    if (!f)
        FAIL(-3, "cannot open %s (run `make -C oracle ref` where /root/reference exists)", codeObjectPath);
    fseek(f, 0, SEEK_END);
    size_t binarySize = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    binary = (unsigned char *)malloc(binarySize);
    if (!binary || fread(binary, 1, binarySize, f) != binarySize)
    {
        fclose(f);
        FAIL(-3, "cannot read %s", codeObjectPath);
    }
    fclose(f);
    buffers = (cl_mem *)calloc((size_t)(nbArgs > 0 ? nbArgs : 1), sizeof(cl_mem));
    if (!buffers)
        FAIL(-5, "out of memory");

    CHECK(clGetPlatformIDs(8, platforms, &nbPlatforms), "clGetPlatformIDs");
    for (cl_uint p = 0; p < nbPlatforms && !device; ++p)
    {
        cl_uint n = 0;
        if (clGetDeviceIDs(platforms[p], CL_DEVICE_TYPE_GPU, 1, &device, &n) != CL_SUCCESS || n == 0)
            device = NULL;
    }
    if (!device)
        FAIL(-4, "no OpenCL GPU device");
    ctx = clCreateContext(NULL, 1, &device, NULL, NULL, &err);
    CHECK(err, "clCreateContext");
    queue = clCreateCommandQueue(ctx, device, 0, &err);
    CHECK(err, "clCreateCommandQueue");
    {
        const unsigned char *bins[1] = {binary};
        cl_int binStatus = 0;
        program = clCreateProgramWithBinary(ctx, 1, &device, &binarySize, bins, &binStatus, &err);
        CHECK(err, "clCreateProgramWithBinary");
        err = clBuildProgram(program, 1, &device, "", NULL, NULL);
        if (err != CL_SUCCESS)
        {
            char buildLog[2048] = "";
            clGetProgramBuildInfo(program, device, CL_PROGRAM_BUILD_LOG, sizeof(buildLog) - 1, buildLog, NULL);
            FAIL(err, "clBuildProgram failed: %d: %.1800s", (int)err, buildLog);
        }
    }
    kernel = clCreateKernel(program, kernelName, &err);
    if (err != CL_SUCCESS)
        FAIL(err, "clCreateKernel(%s) failed: %d", kernelName, (int)err);
    for (int a = 0; a < nbArgs; ++a)
    {
        if (args[a].kind == 0)
        {
            err = clSetKernelArg(kernel, (cl_uint)a, args[a].bytes, args[a].data);
            if (err != CL_SUCCESS)
                FAIL(err, "clSetKernelArg(%s, %d, value of %zu bytes) failed: %d", kernelName, a, args[a].bytes, (int)err);
            continue;
        }
        const size_t bytes = args[a].bytes > 0 ? args[a].bytes : 16;
        buffers[a] = clCreateBuffer(ctx, CL_MEM_READ_WRITE, bytes, NULL, &err);
        if (err != CL_SUCCESS)
            FAIL(err, "clCreateBuffer(arg %d, %zu bytes) failed: %d", a, bytes, (int)err);
        if (args[a].kind == 2 || args[a].bytes == 0)
        {
            void *zero = calloc(bytes, 1);
            if (!zero)
                FAIL(-5, "out of memory");
            err = clEnqueueWriteBuffer(queue, buffers[a], CL_TRUE, 0, bytes, zero, 0, NULL, NULL);
            free(zero);
        }
        else
            err = clEnqueueWriteBuffer(queue, buffers[a], CL_TRUE, 0, args[a].bytes, args[a].data, 0, NULL, NULL);
        if (err != CL_SUCCESS)
            FAIL(err, "clEnqueueWriteBuffer(arg %d) failed: %d", a, (int)err);
        err = clSetKernelArg(kernel, (cl_uint)a, sizeof(cl_mem), &buffers[a]);
        if (err != CL_SUCCESS)
            FAIL(err, "clSetKernelArg(%s, %d, buffer) failed: %d", kernelName, a, (int)err);
    }
    err = clEnqueueNDRangeKernel(queue, kernel, (cl_uint)dims, NULL, global, local, 0, NULL, NULL);
    if (err != CL_SUCCESS)
        FAIL(err, "clEnqueueNDRangeKernel(%s) failed: %d", kernelName, (int)err);
    CHECK(clFinish(queue), "clFinish");
    for (int a = 0; a < nbArgs; ++a)
        if ((args[a].kind == 2 || args[a].kind == 3) && args[a].bytes > 0)
        {
            err = clEnqueueReadBuffer(queue, buffers[a], CL_TRUE, 0, args[a].bytes, args[a].data, 0, NULL, NULL);
            if (err != CL_SUCCESS)
                FAIL(err, "clEnqueueReadBuffer(arg %d) failed: %d", a, (int)err);
        }

done:
    if (buffers)
        for (int a = 0; a < nbArgs; ++a)
            if (buffers[a])
                clReleaseMemObject(buffers[a]);
    free(buffers);
    if (kernel) clReleaseKernel(kernel);
    if (program) clReleaseProgram(program);
    if (queue) clReleaseCommandQueue(queue);
    if (ctx) clReleaseContext(ctx);
    free(binary);
    return status;
}

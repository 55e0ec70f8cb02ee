/*
 * vali_hip.h -- C ABI of libvali_hip.so, the MI355X (gfx950) surface-processing
 * kernel library behind the python_vali Surface/Task API.
 *
 * This header is the drop-in boundary of the hot path.  In the reference the
 * same boundary is the table of dlsym'd NVIDIA NPP entry points
 * (reference: src/TC/inc/LibNpp.hpp:35-198, loader src/TC/src/LibNpp.cpp:20-51)
 * plus the two first-party launchers UD_NV12 / UD_NV12_HBD
 * (reference: src/TC/inc/ResizeUtils.hpp:30-50) and the CUDA driver calls of
 * src/TC/inc/LibCuda.hpp.  Conventions are the NPP ones:
 *   - raw device pointers + byte pitches, sizes in pixels, no allocation and no
 *     ownership transfer inside an operator;
 *   - every operator is asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, a negative VALI_ERR_* otherwise; the text of the last
 *     failure on the calling thread is vali_last_error().
 * No torch / pybind11 / C++ types appear in any signature.
 */
#ifndef VALI_HIP_H
#define VALI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VALI_API __attribute__((visibility("default")))

/* ---- status codes ------------------------------------------------------- */
#define VALI_OK 0
#define VALI_ERR_INVALID_ARG (-1) /* null pointer, bad size, bad enum value      */
#define VALI_ERR_UNSUPPORTED (-2) /* format pair / parameter combination absent  */
#define VALI_ERR_RUNTIME (-3)     /* a HIP runtime call failed                   */
#define VALI_ERR_NO_DEVICE (-4)   /* no usable GPU                               */

/* ---- pixel formats: numbering identical to the reference's Pixel_Format
 *      (reference: src/TC/inc/MemoryInterfaces.hpp:29-46) ------------------- */
enum vali_pixel_format {
  VALI_FMT_UNDEFINED = 0,
  VALI_FMT_Y = 1,
  VALI_FMT_RGB = 2,
  VALI_FMT_NV12 = 3,
  VALI_FMT_YUV420 = 4,
  VALI_FMT_RGB_PLANAR = 5,
  VALI_FMT_BGR = 6,
  VALI_FMT_YUV444 = 7,
  VALI_FMT_RGB_32F = 8,
  VALI_FMT_RGB_32F_PLANAR = 9,
  VALI_FMT_YUV422 = 10,
  VALI_FMT_P10 = 11,
  VALI_FMT_P12 = 12,
  VALI_FMT_YUV444_10BIT = 13,
  VALI_FMT_YUV420_10BIT = 14,
  VALI_FMT_GRAY12 = 15
};

typedef void* vali_stream_t; /* hipStream_t */
typedef void* vali_event_t;  /* hipEvent_t  */
typedef void* vali_graph_t;  /* hipGraphExec_t */

/*
 * A borrowed view of one surface: what the reference hands to NPP as
 * pSrc[]/nSrcStep/oSizeROI (e.g. TaskConvertSurface.cpp:120-136).
 * plane[c] is the device pointer of COMPONENT c, i.e. Surface::PixelPtr(c):
 *   NV12/P10/P12   plane[0]=Y, plane[1]=interleaved UV (= Y + height*pitch)
 *   YUV420/422/444 plane[0..2] = Y,U,V allocations
 *   RGB/BGR/RGB_32F plane[0]    = packed pixels
 *   RGB_PLANAR/RGB_32F_PLANAR plane[c] = base + c*height*pitch
 *   Y              plane[0]
 * width/height are in pixels of the full-resolution (luma) grid.
 * Every plane must be smaller than 4 GiB (the gather kernels form row offsets in 32 bits);
 * the single-surface entry points check it (VALI_ERR_INVALID_ARG), batch callers guarantee it.
 */
typedef struct vali_surface {
  void* plane[3];
  int32_t pitch[3]; /* bytes */
  int32_t width;
  int32_t height;
  int32_t format; /* enum vali_pixel_format */
} vali_surface;

/*
 * YUV -> RGB colour matrix.  One instance per NPP colour variant the reference
 * selects between in nv12_rgb / yuv420_rgb / yuv444_rgb
 * (reference: src/TC/src/TaskConvertSurface.cpp:128-149, 271-292, 360-383).
 *   Yf = cy * (Y - y0);  Uc = U - 128;  Vc = V - 128
 *   R = Yf + crv*Vc ;  G = Yf + (cgu*Uc + cgv*Vc) ;  B = Yf + cbu*Uc
 * exact operation order and rounding: oracle/vali_oracle.c (the specification).
 * This 32-byte block is what the multi-GPU pipeline broadcasts over RCCL.
 */
typedef struct vali_csc {
  float y0, cy, crv, cgu, cgv, cbu;
  float reserved[2];
} vali_csc;

/* ---- runtime: replaces the LibCuda dlsym table
 *      (reference: src/TC/inc/LibCuda.hpp, src/TC/src/CudaUtils.cpp) --------- */

VALI_API const char* vali_last_error(void);
VALI_API const char* vali_version(void);

/* cuDeviceGetCount (CudaUtils.cpp:185-205) */
VALI_API int vali_device_count(int* count);
/* Sizes of surfaces with subsampled chroma -- ONE rule for every entry point below: 4:2:0 formats (NV12, P10, P12,
 * YUV420, YUV420_10BIT) need an even width AND height, YUV422 an even width (their chroma planes hold W/2 x H/2 resp.
 * W/2 x H samples; the reference allocates with integer division and would read past them).  VALI_ERR_INVALID_ARG
 * otherwise; python_vali reports (False, TaskExecInfo.INVALID_INPUT) at Run time -- Surface.Make itself, like the
 * reference's, allocates any size. */

/* cuCtxPushCurrent of the reference's CudaCtxPush (CudaUtils.hpp:29-47), without the pop: makes `device` the calling
 * thread's current device.  Every entry point that takes a stream runs on the STREAM's device; the null stream has
 * none and means "the legacy default stream of the thread's current device" -- a host that is handed stream 0 together
 * with a GPU index (python_vali's Task(gpu_id, stream) constructors) calls this first. */
VALI_API int vali_device_set(int device);
/* the calling thread's current device (cuCtxGetCurrent): what a caller saves before vali_device_set and puts back after
 * the call -- the pop of the reference's CudaCtxPush (CudaUtils.hpp:29-47) */
VALI_API int vali_device_get(int* device);
/* device of a device pointer: GetDeviceIdByDptr (CudaUtils.cpp:150-163) */
VALI_API int vali_ptr_device(const void* dptr, int* device);

/* one non-blocking stream per call: cuStreamCreate(CU_STREAM_NON_BLOCKING)
 * (CudaUtils.cpp:222-238) */
VALI_API int vali_stream_create(int device, vali_stream_t* stream);
/* waits for what the stream has been given (hipStreamSynchronize), drops the library's per-stream state (completion word,
 * tap-table reader lists), then destroys it */
VALI_API int vali_stream_destroy(int device, vali_stream_t stream);
VALI_API int vali_stream_sync(int device, vali_stream_t stream);
/* the same guarantee -- everything issued on `stream` so far has finished -- through a host-visible completion word the
 * stream writes (hipStreamWriteValue32 into pinned memory) and the caller spins on: 3 us less per blocking call than
 * hipStreamSynchronize around a small kernel; falls back to it after ~150 us.  The blocking Run forms of python_vali
 * (Run = RunAsync + event record + host wait, PySurfaceConverter.cpp:76-86) end here. */
VALI_API int vali_stream_wait(int device, vali_stream_t stream);

/* CudaStreamEvent (CudaUtils.cpp:35-68); timing enabled so bench.py can use them */
VALI_API int vali_event_create(int device, vali_event_t* event);
VALI_API int vali_event_destroy(int device, vali_event_t event);
VALI_API int vali_event_record(int device, vali_event_t event, vali_stream_t stream);
VALI_API int vali_event_sync(int device, vali_event_t event);
/* cuEventQuery: *done = 1 when the event has happened, 0 while work in front of it is still running (no wait) */
VALI_API int vali_event_query(int device, vali_event_t event, int* done);
/* cuStreamWaitEvent: work issued on `stream` after this call starts only when `event` has happened (no host wait).  The
 * ingest ring of the batched pipeline orders its copy stream and its compute stream with it. */
VALI_API int vali_stream_wait_event(int device, vali_stream_t stream, vali_event_t event);
VALI_API int vali_event_elapsed_ms(vali_event_t start, vali_event_t stop, float* ms);

/*
 * Stream capture (no reference counterpart: the reference issues every NPP call eagerly).
 * Per-frame chains of small launches (BASELINE config 2: 1080p, batch 1) are bound by the
 * ~5 us host cost of each launch, not by the GPU.  Between capture_begin and capture_end every
 * asynchronous vali_* call on `stream` is RECORDED instead of executed (device pointers and
 * sizes are frozen into the graph); vali_graph_launch replays the whole chain with one
 * submission.  Capture is thread-local; do not synchronise or allocate while capturing.
 */
VALI_API int vali_graph_capture_begin(int device, vali_stream_t stream);
VALI_API int vali_graph_capture_end(int device, vali_stream_t stream, vali_graph_t* graph);
VALI_API int vali_graph_launch(int device, vali_graph_t graph, vali_stream_t stream);
VALI_API int vali_graph_destroy(int device, vali_graph_t graph);

/* cuMemAllocPitch / cuMemAlloc / cuMemFree (SurfacePlane.cpp:186-213).
 * Pitch policy: width_bytes rounded up to 256 B, so every row starts on two
 * 128-byte lines and 16-byte vector access is always legal. */
VALI_API int vali_mem_alloc_pitch(int device, size_t width_bytes, size_t height,
                                  void** dptr, size_t* pitch);
VALI_API int vali_mem_alloc(int device, size_t bytes, void** dptr);
VALI_API int vali_mem_free(int device, void* dptr);
/* page-locked host memory (cuMemAllocHost): the staging buffers of asynchronous host <-> device copies; pageable memory
 * makes vali_memcpy2d_async synchronous and halves its rate */
VALI_API int vali_host_alloc(int device, size_t bytes, void** hptr);
VALI_API int vali_host_free(int device, void* hptr);
/* free / total bytes of the device (cuMemGetInfo): bench.py's pre-flight check */
VALI_API int vali_mem_info(int device, size_t* free_bytes, size_t* total_bytes);
/* PCI address "dddd:bb:dd.f" of the device (cuDeviceGetPCIBusId), for NUMA placement of its host thread */
VALI_API int vali_device_pci_bus_id(int device, char* out, int len);

/* cuMemcpy2DAsync (TaskCudaUploadFrame.cpp:54-72, TaskCudaDownloadSurface.cpp:54-72,
 * MemoryInterfaces.cpp:413-431).  kind: 0 = host->device, 1 = device->host,
 * 2 = device->device. */
VALI_API int vali_memcpy2d_async(int device, void* dst, size_t dst_pitch,
                                 const void* src, size_t src_pitch,
                                 size_t width_bytes, size_t height, int kind,
                                 vali_stream_t stream);
VALI_API int vali_memset2d_async(int device, void* dst, size_t dst_pitch, int value,
                                 size_t width_bytes, size_t height,
                                 vali_stream_t stream);

/* ---- colour conversion: replaces the NPP nppicc entry points ---------------- */

/*
 * NV12 -> RGB / BGR (packed u8) or RGB_PLANAR (u8); any other dst->format returns
 * VALI_ERR_UNSUPPORTED.  (Float outputs -- the fused form of the reference's
 * NV12->RGB->RGB_32F(->PLANAR) chain -- are vali_nv12_preproc below.)
 * Replaces nppiNV12ToRGB_709HDTV_8u_P2C3R_Ctx, nppiNV12ToRGB_709CSC_8u_P2C3R_Ctx,
 * nppiNV12ToRGB_8u_P2C3R_Ctx and their BGR twins
 * (reference call sites: TaskConvertSurface.cpp:61-156; LibNpp.hpp table).
 * dst->format selects the output layout; src->format must be VALI_FMT_NV12.
 */
VALI_API int vali_nv12_to_rgb(const vali_surface* src, const vali_surface* dst,
                              const vali_csc* csc, vali_stream_t stream);

/*
 * Batched form: ONE launch over n independent frames of identical geometry.
 * d_src / d_dst are DEVICE arrays of n vali_surface descriptors (upload them
 * once with vali_memcpy2d_async or hipMemcpy); width/height/dst_format restate
 * the common geometry for grid sizing.  Semantics per frame are exactly those
 * of vali_nv12_to_rgb.  Precedent for the list-in / one-sync idiom:
 * reference src/python_vali/src/PyNvJpegEncoder.cpp:31-81.
 */
VALI_API int vali_nv12_to_rgb_batch(const vali_surface* d_src,
                                    const vali_surface* d_dst, int n, int width,
                                    int height, int dst_format, const vali_csc* csc,
                                    vali_stream_t stream);

/*
 * Parameters of the generic converter.  yuv2rgb is used by YUV -> RGB pairs; rgb2yuv by
 * RGB -> YUV / grey pairs: row c = (kR, kG, kB, offset) of output channel c (Y, U/Cb, V/Cr),
 *   out_c = kB*B + (kG*G + (kR*R + offset))   evaluated with fused multiply-adds.
 */
typedef struct vali_cvt_params {
  vali_csc yuv2rgb;
  float rgb2yuv[3][4];
} vali_cvt_params;

/*
 * Every other 8-bit pair of ConvertSurface::GetSupportedConversions()
 * (reference: src/TC/src/TaskConvertSurface.cpp:966-994) plus the three element-type
 * conversions, selected by (src->format, dst->format):
 *   NV12<->YUV420, NV12->Y, Y->YUV444, RGB<->RGB_PLANAR, RGB<->BGR   byte permutations
 *   YUV420->RGB/BGR, YUV444->RGB/BGR, NV12->RGB/BGR/RGB_PLANAR       params->yuv2rgb
 *   RGB/BGR/RGB_PLANAR->YUV444, RGB->YUV420, RGB->Y                  params->rgb2yuv
 *   P10/P12->NV12 (round(v/256), saturated), RGB->RGB_32F (v/255), RGB_32F->RGB_32F_PLANAR
 * replacing the NPP calls listed at the top of vali_amd/csrc/cvt_generic.hip.
 * 4:2:0 formats need even width and height.  Unsupported pair -> VALI_ERR_UNSUPPORTED.
 */
VALI_API int vali_convert(const vali_surface* src, const vali_surface* dst,
                          const vali_cvt_params* params, vali_stream_t stream);
VALI_API int vali_convert_batch(const vali_surface* d_src, const vali_surface* d_dst, int n,
                                int src_format, int dst_format, int width, int height,
                                const vali_cvt_params* params, vali_stream_t stream);

/* ---- fused inference pre-processing ------------------------------- */

/*
 * NV12 -> (bilinear resize to the dst size) -> RGB u8 -> float -> normalised, one launch.
 * No reference kernel: replaces the CHAIN the reference's samples run
 * (tests/test_TorchSegmentation.py:176-240): PySurfaceConverter NV12->RGB, RGB->RGB_32F
 * (nppiScale_8u32f_C3R, v/255), RGB_32F->RGB_32F_PLANAR, torch.divide(x, 255.0),
 * torchvision Normalize -- optionally behind a PySurfaceResizer.  Defined as that chain step
 * by step and bit-identical to running it with vali_resize + vali_nv12_to_rgb + vali_convert:
 *   q_c   = u8 RGB of vali_nv12_to_rgb(csc) on the (resized) NV12
 *   out_c = ((q_c / 255.0f) / div - mean[c]) / std_[c]          c = R, G, B ; IEEE float32
 * div = 1, mean = 0, std_ = 1 gives plain NV12 -> RGB_32F[_PLANAR].
 * dst->format: VALI_FMT_RGB_32F_PLANAR or VALI_FMT_RGB_32F; all four sizes even.
 * 8-bit destinations (VALI_FMT_RGB, VALI_FMT_BGR, VALI_FMT_RGB_PLANAR) stop after q_c: the
 * fused form of PySurfaceResizer -> PySurfaceConverter; div / mean / std_ are ignored.
 */
typedef struct vali_preproc_params {
  vali_csc csc;
  float div;
  float mean[3];
  float std_[3];
  float reserved;
} vali_preproc_params;

VALI_API int vali_nv12_preproc(const vali_surface* src, const vali_surface* dst,
                               const vali_preproc_params* params, vali_stream_t stream);
VALI_API int vali_nv12_preproc_batch(const vali_surface* d_src, const vali_surface* d_dst, int n,
                                     int src_width, int src_height, int dst_width, int dst_height,
                                     int dst_format, const vali_preproc_params* params,
                                     vali_stream_t stream);

/* ---- UD: chroma upsample + resize (+ YUV->RGB) in one pass ---------------------- */

/*
 * Replaces the reference's first-party launchers UD_NV12 / UD_NV12_HBD
 * (reference: src/TC/inc/ResizeUtils.hpp:30-50, src/TC/src/ResizeUtils.cu:98-176; callers
 * src/TC/src/UDSurface.cpp:84-115).  src->format: NV12 or P10; dst->format one of
 * YUV444, RGB, RGB_PLANAR, RGB_32F, RGB_32F_PLANAR (NV12) / YUV444_10BIT, RGB_32F,
 * RGB_32F_PLANAR (P10) -- the semi-planar rows of UDSurface::SupportedConversions()
 * (UDSurface.cpp:117-133); anything else returns VALI_ERR_UNSUPPORTED.
 * As in the reference the dst size alone defines the scale (no size validation).
 */
VALI_API int vali_ud_nv12(const vali_surface* src, const vali_surface* dst,
                          vali_stream_t stream);
/* (the frames of a batch share one geometry: src_width x src_height -> dst_width x dst_height) */
VALI_API int vali_ud_nv12_batch(const vali_surface* d_src, const vali_surface* d_dst, int n,
                                int src_format, int src_width, int src_height, int dst_width, int dst_height,
                                int dst_format, vali_stream_t stream);

/*
 * UD with the result written rotated by `quarter_turns` x 90 degrees (1..3; 0 = vali_ud_nv12):
 * the fused form of the chain PySurfaceUD -> PySurfaceRotator (BASELINE config 4), bit-identical
 * to vali_ud_nv12 followed by vali_rotate with the canonical quarter-turn shifts
 * (PySurfaceRotator.cpp:47-73), without the intermediate surface.  NV12 -> RGB only.
 * `dst` is the ROTATED surface: for odd quarter_turns the UD output is dst->height x dst->width.
 */
VALI_API int vali_ud_nv12_rot(const vali_surface* src, const vali_surface* dst, int quarter_turns,
                              vali_stream_t stream);
VALI_API int vali_ud_nv12_rot_batch(const vali_surface* d_src, const vali_surface* d_dst, int n,
                                    int src_format, int src_width, int src_height, int dst_width, int dst_height,
                                    int dst_format, int quarter_turns, vali_stream_t stream);

/* ---- resize: replaces nppiResize_{8u,32f}_C{1,3}R_Ctx --------------------------------- */

/* Values follow NppiInterpolationMode (NPPI_INTER_LINEAR = 2 is not used by the reference;
 * NPPI_INTER_LANCZOS = 16 is what every nppiResize call site passes; NPPI_INTER_CUBIC = 4 is the
 * "bicubic" BASELINE.json's north_star names).  All modes sample on the grid
 * src = dst * (src_size / dst_size), no half-pixel shift -- pinned by the reference fixture
 * test_small.nv12 (tests/test_oracle_resize.py). */
enum vali_interpolation {
  VALI_INTERP_LINEAR = 1,   /* bilinear: BASELINE.json config 3 */
  VALI_INTERP_CUBIC = 4,    /* 4x4 Keys / Catmull-Rom cubic convolution (a = -1/2) */
  VALI_INTERP_LANCZOS = 16  /* 6x6 interpolating Lanczos-3 (TaskResizeSurface.cpp:67,116,224,273): the reference's
                               only filter and PySurfaceResizer's default; taps + grid pinned against NPP output at a
                               non-integer ratio (tests/test_oracle_reference_pins.py) */
};

/*
 * Resize every plane of a surface in one launch (src->format == dst->format).
 * Replaces ResizeSurface and its five per-format implementations
 * (reference: src/TC/src/TaskResizeSurface.cpp:34-286; NV12 via the NV12->YUV420->resize->
 * YUV420->NV12 round trip :132-188).  Formats: Y, NV12, P10, P12, YUV420(_10BIT), YUV422,
 * YUV444(_10BIT), RGB, BGR, RGB_32F, RGB_PLANAR, RGB_32F_PLANAR.  The reference's resizer is
 * Lanczos-only (NPPI_INTER_LANCZOS, :67); bilinear is BASELINE.json's definition.
 *
 * Device memory the library owns itself: the Lanczos / bicubic forms keep per-geometry TAP TABLES (32 B per destination
 * column / row).  All tables of a device live in ONE 32 MiB arena that the first Lanczos / bicubic call of the process on
 * that device reserves (the only hipMalloc the operators ever make: once per device and process); every later call --
 * new geometries included -- allocates nothing and is asynchronous on `stream`.  Least recently used tables are evicted
 * when the arena or VALI_TUNE_TAP_MAX_TABLES is exhausted; during a graph capture, and for axes longer than 32 768 samples,
 * no table is used (the kernels compute their taps: same result).  vali_stream_destroy makes the library forget the stream.
 */
VALI_API int vali_resize(const vali_surface* src, const vali_surface* dst, int interpolation,
                         vali_stream_t stream);
VALI_API int vali_resize_batch(const vali_surface* d_src, const vali_surface* d_dst, int n,
                               int format, int src_width, int src_height, int dst_width,
                               int dst_height, int interpolation, vali_stream_t stream);

/*
 * UDPlanar: every plane of a planar 4:2:0 surface resized to the size of the matching plane of a 4:4:4
 * surface (chroma 2x up + the common scale), all three planes in ONE launch.  Replaces UDPlanar
 * (reference: src/TC/src/UDSurface.cpp:33-93, three nppiResize calls) for the planar rows of
 * UDSurface::SupportedConversions() (:117-133): YUV420 -> YUV444, YUV420_10BIT -> YUV444_10BIT; any other
 * pair returns VALI_ERR_UNSUPPORTED.  The reference passes NPPI_INTER_LANCZOS (:45,72): pass
 * VALI_INTERP_LANCZOS for its result; the other modes are accepted too.  As in the reference the
 * destination size alone defines the scale.
 */
VALI_API int vali_ud_planar(const vali_surface* src, const vali_surface* dst, int interpolation,
                            vali_stream_t stream);
VALI_API int vali_ud_planar_batch(const vali_surface* d_src, const vali_surface* d_dst, int n,
                                  int src_format, int dst_format, int src_width, int src_height,
                                  int dst_width, int dst_height, int interpolation, vali_stream_t stream);

/* ---- rotation: replaces nppiRotate_{8u,16u,32f}_{C1,C3}R_Ctx ------------------------ */

/*
 * Rotate every plane of a surface by `angle` degrees and shift (NPP model: x' = x cos a +
 * y sin a + shift_x, y' = -x sin a + y cos a + shift_y; inverse-mapped, bilinear; destination
 * pixels whose source point is outside the source plane are left untouched).
 * Reference call sites: Rot_8U_C1 ... Rot_32F_C3 via RotPlanar / RotPacked,
 * src/TC/src/RotateSurface.cpp:22-159 (one NPP call per plane; here one launch per surface
 * or per batch).  Formats: Y, RGB, BGR, RGB_32F, YUV420(_10BIT), YUV422, YUV444(_10BIT).
 * per_plane_shifts != 0 (angle must be a multiple of 90): ignore shift_x/shift_y and give
 * each plane the shifts PySurfaceRotator derives from ITS size
 * (src/python_vali/src/PySurfaceRotator.cpp:47-73) -- the exact quarter-turn permutation,
 * done as an LDS-tiled transpose for 90 / 270 degrees.
 */
VALI_API int vali_rotate(const vali_surface* src, const vali_surface* dst, double angle,
                         double shift_x, double shift_y, int per_plane_shifts,
                         vali_stream_t stream);
VALI_API int vali_rotate_batch(const vali_surface* d_src, const vali_surface* d_dst, int n,
                               int format, int src_width, int src_height, int dst_width,
                               int dst_height, double angle, double shift_x, double shift_y,
                               int per_plane_shifts, vali_stream_t stream);
/* cos/sin of the angle as the kernels use them (exact 0/+-1 for multiples of 90 degrees) */
VALI_API int vali_rotate_coeffs(double angle_deg, float* c, float* s);

/* ---- tuning switches and tracing -------------------------------------------------------------
 *
 * Every alternative kernel form the library keeps for A/B measurements and for path-coverage tests is
 * selected through this ONE table (no getenv anywhere else in the library).  A switch never changes a
 * result -- every form is bit-identical (tests/test_gpu_tuning.py runs each operator under every value) --
 * only which kernel produces it.  Values are process-wide and may be changed at any time between calls.
 * Initial values: the default below, or the environment variable of the same name (VALI_<KEY>) read once,
 * when the library is first used.
 */
enum vali_tuning_key {
  VALI_TUNE_NV12_ROWPAIRS = 0,        /* row pairs stacked in one workgroup of the streaming converters; 0 = auto */
  VALI_TUNE_WAVES_PER_CU = 1,         /* residency cap of the streaming converters; 0 = auto                      */
  VALI_TUNE_NV12_DIRECT_STORE = 2,    /* 1: NV12->RGB stores 48 B per lane instead of going through the LDS strip */
  VALI_TUNE_RESIZE_FORCE_GATHER = 3,  /* 1: every resize geometry through the direct-gather form                  */
  VALI_TUNE_RESIZE_POINT = 4,         /* 0: arithmetic form at integer scale factors; 2: staged point form only (default 1) */
  VALI_TUNE_UD_FORCE_GATHER = 5,      /* 1: every UD geometry through the direct-gather form, no exact-ratio kernels */
  VALI_TUNE_UD_DOWN2 = 6,             /* 0: general UD kernel also at the exact 2:1 / 1:1 width ratios; 1 (default):
                                         exact-ratio kernels for output widths that are multiples of 8; 2: for all */
  VALI_TUNE_UD_OCC5 = 7,              /* no effect since round 2 (selected a 96-register instantiation of the staged UD
                                         kernel, removed: it spilled); the key keeps its number                       */
  VALI_TUNE_ROTATE_NO_TILE = 8,       /* 1: quarter / half turns through the bilinear kernel; 2: column-major tile walk;
                                         3: fused UD + quarter turn on 128-row tiles (default 64)                  */
  VALI_TUNE_ROCTX = 9,                /* 1: a roctx range around every operator entry point (see below)            */
  VALI_TUNE_RESIZE_NO_SEPARABLE = 10, /* Lanczos / bicubic rows per wave: 0 by launch size, 1: few, 2: fewest (and the slot walk
                                         where the 3:2-both-ways form has a static one), 3: most (growing planes: 32-row waves), 4: growing planes on
                                         64-row waves whatever the launch size; 11 .. 42: an explicit count (rows per slot / row
                                         pairs per wave: measurements only)                                          */
  VALI_TUNE_ROWS_PER_WAVE = 11,       /* UD, bilinear / point resize, fused pre-processing: dst rows (row pairs) a wave
                                         walks: 0 by launch size (8 for batches, 4 or 2 for small launches), 2 / 4 / 8  */
  VALI_TUNE_BLOCKING_WAIT = 12,       /* vali_stream_wait: 0 completion word + spin (default), 1 hipStreamSynchronize */
  VALI_TUNE_RESIZE_ROWS = 13,         /* Lanczos / bicubic of planes that grow: 1 (default) filtered rows in registers
                                         (resize_rows.hip) where the geometry fits, exact 3:2 enlargements through their static
                                         form; 2: the same without the 3:2 form; 3: without the register form of round 4 (8-bit planes
                                         that grow along x too: row pass from registers, no LDS stage); 0: round 2's LDS-ring kernel  */
  VALI_TUNE_RESIZE_COLS = 14,         /* Lanczos / bicubic of planes that shrink, general ratios: 0 (default) a workgroup = one wave that
                                         walks the source rows + one that runs the pass along the rows (round 5), taps from per-geometry
                                         tables (tap_table.hip); 2: the same, taps computed in the kernel; 1: both passes in every wave
                                         (round 4's form)                                                             */
  VALI_TUNE_ROTATE_AFFINE = 15,       /* rotation by an angle that is no canonical quarter / half turn: 0 (default) a workgroup stages the source
                                         box of its destination tile in LDS (round 6), tile shape by launch size; 1: per-pixel gathers from global
                                         memory (round 2's form, also taken for planes narrower than a staged row); tile shapes: 2 = 32 x 64,
                                         3 = 64 x 32, 4 = 64 x 64, 5 = 64 x 128 (one-channel 8-bit planes), 6 = 32 x 32   */
  VALI_TUNE_TAP_MAX_TABLES = 16,      /* tap tables (Lanczos / bicubic axes) kept per device before the least recently used is evicted; default 1024, at least 8
                                         (they share one 32 MiB arena per device, reserved by the first Lanczos / bicubic call on it)        */
  VALI_TUNE_TAP_FALLBACKS = 17,       /* COUNTER (read with vali_tuning_get, reset by setting 0): resize calls that got no tap table outside a
                                         graph capture and computed their taps in the kernel (axis > 32768 samples, nothing evictable, HIP error) */
  VALI_TUNE_TAP_EVICTIONS = 18,       /* COUNTER: tap tables evicted so far                                              */
  VALI_TUNE_COUNT = 19
};
VALI_API int vali_tuning_set(int key, int value);
VALI_API int vali_tuning_get(int key, int* value);

/*
 * Tracing: with VALI_TUNE_ROCTX = 1 (or VALI_ROCTX=1 in the environment) every operator entry point of this
 * header pushes a roctx range named after itself for the duration of the call -- the equivalent of the
 * reference's NvtxMark around every converter / resizer (src/TC/inc/Tasks.hpp:32-59, e.g.
 * TaskConvertSurface.cpp:112).  `rocprofv3 --marker-trace --kernel-trace` then shows which launch belongs to
 * which call.  The roctx library (librocprofiler-sdk-roctx / libroctx64) is dlopen'ed on first use; if it is
 * absent tracing stays off and vali_tuning_set(VALI_TUNE_ROCTX, 1) returns VALI_ERR_UNSUPPORTED.
 */

/* ---- diagnostics (used by tests only) --------------------------------------- */

/* out[i] = float->u8 quantiser of the colour kernels applied to in[i]. */
VALI_API int vali_debug_quantize_u8(const float* d_in, uint8_t* d_out, int n,
                                    vali_stream_t stream);
/* same with the instruction-independent formulation (rint, clamp, convert). */
VALI_API int vali_debug_quantize_u8_portable(const float* d_in, uint8_t* d_out, int n,
                                             vali_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VALI_HIP_H */

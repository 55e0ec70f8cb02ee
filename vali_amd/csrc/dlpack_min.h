/*
 * Minimal DLPack (v0.x "legacy" ABI) structure definitions, written from the
 * public DLPack specification.  Only what Surface/SurfacePlane exchange needs:
 * DLManagedTensor inside a PyCapsule named "dltensor".
 * The reference hard-codes kDLCUDA (src/TC/src/SurfacePlane.cpp:255); PyTorch-ROCm
 * exchanges HIP memory as kDLROCM = 10, which is what this build exports/accepts.
 */
#ifndef VALI_DLPACK_MIN_H
#define VALI_DLPACK_MIN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  kDLCPU = 1,
  kDLCUDA = 2,
  kDLCUDAHost = 3,
  kDLOpenCL = 4,
  kDLVulkan = 7,
  kDLMetal = 8,
  kDLVPI = 9,
  kDLROCM = 10,
  kDLROCMHost = 11,
  kDLExtDev = 12,
  kDLCUDAManaged = 13,
} DLDeviceType;

typedef struct {
  int32_t device_type; /* DLDeviceType */
  int32_t device_id;
} DLDevice;

typedef enum {
  kDLInt = 0U,
  kDLUInt = 1U,
  kDLFloat = 2U,
  kDLOpaqueHandle = 3U,
  kDLBfloat = 4U,
  kDLComplex = 5U,
  kDLBool = 6U,
} DLDataTypeCode;

typedef struct {
  uint8_t code;
  uint8_t bits;
  uint16_t lanes;
} DLDataType;

typedef struct {
  void* data;
  DLDevice device;
  int32_t ndim;
  DLDataType dtype;
  int64_t* shape;
  int64_t* strides; /* in elements; NULL = compact row-major */
  uint64_t byte_offset;
} DLTensor;

typedef struct DLManagedTensor {
  DLTensor dl_tensor;
  void* manager_ctx;
  void (*deleter)(struct DLManagedTensor* self);
} DLManagedTensor;

#ifdef __cplusplus
}
#endif
#endif

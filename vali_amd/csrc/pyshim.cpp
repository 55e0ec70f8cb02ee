// _vali_shim: thin pybind11 binding of the C ABI in include/vali_hip.h.
//
// Nothing here computes anything: every function forwards plain integers
// (device pointers, stream handles) and small PODs to libvali_hip.so, the same
// way the reference's pybind11 layer forwards to its Task objects
// (reference: src/python_vali/src/*.cpp).  The host logic of the path (format
// dispatch, validation, error codes) lives in Python (vali_amd/*.py).
// The only non-forwarding code is DLPack capsule plumbing, which needs C.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "dlpack_min.h"
#include "vali_hip.h"

namespace py = pybind11;

namespace {

void check(int rc, const char* what) {
  if (rc == VALI_OK)
    return;
  std::string msg = std::string(what) + ": " + vali_last_error();
  if (rc == VALI_ERR_INVALID_ARG || rc == VALI_ERR_UNSUPPORTED)
    throw py::value_error(msg);
  throw std::runtime_error(msg);
}

inline void* P(uintptr_t v) { return reinterpret_cast<void*>(v); }

struct SurfaceDesc {
  vali_surface s;
};

struct Csc {
  vali_csc k;
};

struct CvtParams {
  vali_cvt_params p;
};

struct PreprocParams {
  vali_preproc_params p;
};

// ---- DLPack ------------------------------------------------------------------

struct ExportCtx {
  PyObject* owner; // keeps the producing Surface / SurfacePlane alive
  int64_t dims[8];
};

void export_deleter(DLManagedTensor* self) {
  if (!self)
    return;
  auto* ctx = static_cast<ExportCtx*>(self->manager_ctx);
  if (ctx) {
    if (ctx->owner) {
      py::gil_scoped_acquire gil;
      Py_DECREF(ctx->owner);
    }
    delete ctx;
  }
  delete self;
}

void capsule_destructor(PyObject* cap) {
  // Only un-consumed capsules still own the tensor.
  if (!PyCapsule_IsValid(cap, "dltensor"))
    return;
  auto* m = static_cast<DLManagedTensor*>(PyCapsule_GetPointer(cap, "dltensor"));
  if (m && m->deleter)
    m->deleter(m);
}

py::object dlpack_export(uintptr_t ptr, const std::vector<int64_t>& shape,
                         const std::vector<int64_t>& strides, int code, int bits,
                         int device_type, int device_id, py::object owner) {
  const size_t nd = shape.size();
  if (nd == 0 || nd > 4 || strides.size() != nd)
    throw py::value_error("dlpack_export: bad shape/strides");
  auto* ctx = new ExportCtx();
  auto* m = new DLManagedTensor();
  std::memset(m, 0, sizeof(*m));
  for (size_t i = 0; i < nd; ++i) {
    ctx->dims[i] = shape[i];
    ctx->dims[4 + i] = strides[i];
  }
  ctx->owner = owner.is_none() ? nullptr : owner.inc_ref().ptr();
  m->dl_tensor.data = P(ptr);
  m->dl_tensor.device.device_type = device_type;
  m->dl_tensor.device.device_id = device_id;
  m->dl_tensor.ndim = (int32_t)nd;
  m->dl_tensor.dtype.code = (uint8_t)code;
  m->dl_tensor.dtype.bits = (uint8_t)bits;
  m->dl_tensor.dtype.lanes = 1;
  m->dl_tensor.shape = ctx->dims;
  m->dl_tensor.strides = ctx->dims + 4;
  m->dl_tensor.byte_offset = 0;
  m->manager_ctx = ctx;
  m->deleter = export_deleter;
  PyObject* cap = PyCapsule_New(m, "dltensor", capsule_destructor);
  if (!cap) {
    export_deleter(m);
    throw py::error_already_set();
  }
  return py::reinterpret_steal<py::object>(cap);
}

// Owns an imported DLManagedTensor; the producer's deleter runs when the last
// Surface borrowing that memory goes away.
struct DLPackHolder {
  DLManagedTensor* m = nullptr;
  ~DLPackHolder() {
    if (m && m->deleter)
      m->deleter(m);
  }
};

py::tuple dlpack_import(py::object capsule) {
  PyObject* cap = capsule.ptr();
  if (!PyCapsule_CheckExact(cap))
    throw std::runtime_error("Empty capsule.");
  if (!PyCapsule_IsValid(cap, "dltensor"))
    throw std::runtime_error("Capsule doesn't contain dltensor.");
  auto* m = static_cast<DLManagedTensor*>(PyCapsule_GetPointer(cap, "dltensor"));
  if (!m)
    throw std::runtime_error("Capsule doesn't contain dltensor.");
  const DLTensor& t = m->dl_tensor;
  py::dict info;
  std::vector<int64_t> shape(t.shape, t.shape + t.ndim), strides;
  if (t.strides) {
    strides.assign(t.strides, t.strides + t.ndim);
  } else { // compact row-major
    strides.resize(t.ndim);
    int64_t acc = 1;
    for (int i = t.ndim - 1; i >= 0; --i) {
      strides[i] = acc;
      acc *= t.shape[i];
    }
  }
  info["ptr"] = (uintptr_t)t.data + t.byte_offset;
  info["shape"] = shape;
  info["strides"] = strides;
  info["code"] = (int)t.dtype.code;
  info["bits"] = (int)t.dtype.bits;
  info["lanes"] = (int)t.dtype.lanes;
  info["device_type"] = (int)t.device.device_type;
  info["device_id"] = (int)t.device.device_id;
  // consume: standard DLPack hand-over
  PyCapsule_SetName(cap, "used_dltensor");
  PyCapsule_SetDestructor(cap, nullptr);
  auto holder = std::make_unique<DLPackHolder>();
  holder->m = m;
  return py::make_tuple(info, py::cast(std::move(holder)));
}

} // namespace

PYBIND11_MODULE(_vali_shim, m) {
  m.doc() = "1:1 binding of libvali_hip.so (include/vali_hip.h)";

  m.attr("OK") = VALI_OK;
  m.attr("ERR_INVALID_ARG") = VALI_ERR_INVALID_ARG;
  m.attr("ERR_UNSUPPORTED") = VALI_ERR_UNSUPPORTED;
  m.attr("ERR_RUNTIME") = VALI_ERR_RUNTIME;
  m.attr("ERR_NO_DEVICE") = VALI_ERR_NO_DEVICE;
  m.attr("SURFACE_DESC_SIZE") = sizeof(vali_surface);

  py::class_<SurfaceDesc>(m, "SurfaceDesc")
      .def(py::init([](std::vector<uintptr_t> planes, std::vector<int> pitches, int width,
                       int height, int format) {
             SurfaceDesc d;
             std::memset(&d.s, 0, sizeof(d.s));
             for (size_t i = 0; i < planes.size() && i < 3; ++i)
               d.s.plane[i] = P(planes[i]);
             for (size_t i = 0; i < pitches.size() && i < 3; ++i)
               d.s.pitch[i] = pitches[i];
             d.s.width = width;
             d.s.height = height;
             d.s.format = format;
             return d;
           }),
           py::arg("planes"), py::arg("pitches"), py::arg("width"), py::arg("height"),
           py::arg("format"))
      .def("tobytes",
           [](const SurfaceDesc& d) { return py::bytes((const char*)&d.s, sizeof(d.s)); })
      .def_property_readonly("width", [](const SurfaceDesc& d) { return d.s.width; })
      .def_property_readonly("height", [](const SurfaceDesc& d) { return d.s.height; })
      .def_property_readonly("format", [](const SurfaceDesc& d) { return d.s.format; });

  py::class_<Csc>(m, "Csc")
      .def(py::init([](float y0, float cy, float crv, float cgu, float cgv, float cbu) {
             Csc c;
             std::memset(&c.k, 0, sizeof(c.k));
             c.k.y0 = y0; c.k.cy = cy; c.k.crv = crv; c.k.cgu = cgu; c.k.cgv = cgv; c.k.cbu = cbu;
             return c;
           }))
      .def("astuple", [](const Csc& c) {
        return py::make_tuple(c.k.y0, c.k.cy, c.k.crv, c.k.cgu, c.k.cgv, c.k.cbu);
      });

  py::class_<DLPackHolder>(m, "DLPackHolder");

  m.def("last_error", []() { return std::string(vali_last_error()); });
  // tuning switches / tracing (include/vali_hip.h: vali_tuning_key)
  for (auto kv : {std::pair<const char*, int>{"TUNE_NV12_ROWPAIRS", VALI_TUNE_NV12_ROWPAIRS},
                  {"TUNE_WAVES_PER_CU", VALI_TUNE_WAVES_PER_CU},
                  {"TUNE_NV12_DIRECT_STORE", VALI_TUNE_NV12_DIRECT_STORE},
                  {"TUNE_RESIZE_FORCE_GATHER", VALI_TUNE_RESIZE_FORCE_GATHER},
                  {"TUNE_RESIZE_POINT", VALI_TUNE_RESIZE_POINT},
                  {"TUNE_UD_FORCE_GATHER", VALI_TUNE_UD_FORCE_GATHER},
                  {"TUNE_UD_DOWN2", VALI_TUNE_UD_DOWN2},
                  {"TUNE_UD_OCC5", VALI_TUNE_UD_OCC5},
                  {"TUNE_ROTATE_NO_TILE", VALI_TUNE_ROTATE_NO_TILE},
                  {"TUNE_ROCTX", VALI_TUNE_ROCTX},
                  {"TUNE_RESIZE_NO_SEPARABLE", VALI_TUNE_RESIZE_NO_SEPARABLE},
                  {"TUNE_ROWS_PER_WAVE", VALI_TUNE_ROWS_PER_WAVE},
                  {"TUNE_BLOCKING_WAIT", VALI_TUNE_BLOCKING_WAIT},
                  {"TUNE_RESIZE_ROWS", VALI_TUNE_RESIZE_ROWS},
                  {"TUNE_RESIZE_COLS", VALI_TUNE_RESIZE_COLS},
                  {"TUNE_ROTATE_AFFINE", VALI_TUNE_ROTATE_AFFINE},
                  {"TUNE_TAP_MAX_TABLES", VALI_TUNE_TAP_MAX_TABLES},
                  {"TUNE_TAP_FALLBACKS", VALI_TUNE_TAP_FALLBACKS},
                  {"TUNE_TAP_EVICTIONS", VALI_TUNE_TAP_EVICTIONS},
                  {"TUNE_COUNT", VALI_TUNE_COUNT}})
    m.attr(kv.first) = kv.second;
  m.def("tuning_set", [](int key, int value) { return vali_tuning_set(key, value); });
  m.def("tuning_get", [](int key) {
    int v = 0;
    check(vali_tuning_get(key, &v), "vali_tuning_get");
    return v;
  });
  m.def("version", []() { return std::string(vali_version()); });
  m.def("device_count", []() {
    int n = 0;
    (void)vali_device_count(&n); // "no device" is a count of 0, not an exception
    return n;
  });
  m.def("device_set", [](int device) { check(vali_device_set(device), "device_set"); });
  m.def("device_get", []() {
    int d = -1;
    check(vali_device_get(&d), "device_get");
    return d;
  });
  m.def("ptr_device", [](uintptr_t p) {
    int d = -1;
    check(vali_ptr_device(P(p), &d), "vali_ptr_device");
    return d;
  });

  m.def("stream_create", [](int device) {
    vali_stream_t s = nullptr;
    check(vali_stream_create(device, &s), "vali_stream_create");
    return (uintptr_t)s;
  });
  m.def("stream_destroy", [](int device, uintptr_t s) {
    check(vali_stream_destroy(device, P(s)), "vali_stream_destroy");
  });
  m.def("stream_wait",
        [](int device, uintptr_t s) { check(vali_stream_wait(device, P(s)), "vali_stream_wait"); },
        py::call_guard<py::gil_scoped_release>());
  m.def("stream_sync",
        [](int device, uintptr_t s) { check(vali_stream_sync(device, P(s)), "vali_stream_sync"); },
        py::call_guard<py::gil_scoped_release>());

  m.def("event_create", [](int device) {
    vali_event_t e = nullptr;
    check(vali_event_create(device, &e), "vali_event_create");
    return (uintptr_t)e;
  });
  m.def("event_destroy", [](int device, uintptr_t e) {
    check(vali_event_destroy(device, P(e)), "vali_event_destroy");
  });
  m.def("event_record", [](int device, uintptr_t e, uintptr_t s) {
    check(vali_event_record(device, P(e), P(s)), "vali_event_record");
  });
  m.def("event_query", [](int device, uintptr_t e) {
    int done = 0;
    check(vali_event_query(device, P(e), &done), "vali_event_query");
    return done != 0;
  });
  m.def("stream_wait_event", [](int device, uintptr_t s, uintptr_t e) {
    check(vali_stream_wait_event(device, P(s), P(e)), "vali_stream_wait_event");
  });
  m.def("host_alloc", [](int device, size_t bytes) {
    void* p = nullptr;
    check(vali_host_alloc(device, bytes, &p), "vali_host_alloc");
    return (uintptr_t)p;
  });
  m.def("host_free", [](int device, uintptr_t p) { check(vali_host_free(device, P(p)), "vali_host_free"); });
  m.def("mem_info", [](int device) {
    size_t f = 0, t = 0;
    check(vali_mem_info(device, &f, &t), "vali_mem_info");
    return py::make_tuple(f, t);
  });
  m.def("device_pci_bus_id", [](int device) {
    char buf[32] = {0};
    check(vali_device_pci_bus_id(device, buf, (int)sizeof buf), "vali_device_pci_bus_id");
    return std::string(buf);
  });
  m.def("event_sync",
        [](int device, uintptr_t e) { check(vali_event_sync(device, P(e)), "vali_event_sync"); },
        py::call_guard<py::gil_scoped_release>());
  m.def("event_elapsed_ms", [](uintptr_t a, uintptr_t b) {
    float ms = 0.f;
    check(vali_event_elapsed_ms(P(a), P(b), &ms), "vali_event_elapsed_ms");
    return ms;
  });

  m.def("graph_capture_begin",
        [](int device, uintptr_t s) { check(vali_graph_capture_begin(device, P(s)), "vali_graph_capture_begin"); });
  m.def("graph_capture_end", [](int device, uintptr_t s) {
    vali_graph_t g = nullptr;
    check(vali_graph_capture_end(device, P(s), &g), "vali_graph_capture_end");
    return (uintptr_t)g;
  });
  m.def("graph_launch",
        [](int device, uintptr_t g, uintptr_t s) { check(vali_graph_launch(device, P(g), P(s)), "vali_graph_launch"); },
        py::call_guard<py::gil_scoped_release>());
  m.def("graph_destroy", [](int device, uintptr_t g) { return vali_graph_destroy(device, P(g)); });

  m.def("mem_alloc_pitch", [](int device, size_t width_bytes, size_t height) {
    void* p = nullptr;
    size_t pitch = 0;
    check(vali_mem_alloc_pitch(device, width_bytes, height, &p, &pitch), "vali_mem_alloc_pitch");
    return py::make_tuple((uintptr_t)p, pitch);
  });
  m.def("mem_alloc", [](int device, size_t bytes) {
    void* p = nullptr;
    check(vali_mem_alloc(device, bytes, &p), "vali_mem_alloc");
    return (uintptr_t)p;
  });
  m.def("mem_free", [](int device, uintptr_t p) {
    // destructor path: never throw
    return vali_mem_free(device, P(p));
  });

  m.def("memcpy2d_async",
        [](int device, uintptr_t dst, size_t dpitch, uintptr_t src, size_t spitch, size_t wbytes,
           size_t height, int kind, uintptr_t stream) {
          check(vali_memcpy2d_async(device, P(dst), dpitch, P(src), spitch, wbytes, height, kind,
                                    P(stream)),
                "vali_memcpy2d_async");
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("memset2d_async",
        [](int device, uintptr_t dst, size_t dpitch, int value, size_t wbytes, size_t height,
           uintptr_t stream) {
          check(vali_memset2d_async(device, P(dst), dpitch, value, wbytes, height, P(stream)),
                "vali_memset2d_async");
        });

  // address of a Python buffer (numpy array, bytes, bytearray) for H2D / D2H copies
  m.def("buffer_info", [](py::buffer b, bool writable) {
    py::buffer_info info = b.request(writable);
    bool contiguous = true;
    py::ssize_t expect = info.itemsize;
    for (int i = (int)info.ndim - 1; i >= 0; --i) {
      if (info.shape[i] != 1 && info.strides[i] != expect)
        contiguous = false;
      expect *= info.shape[i];
    }
    return py::make_tuple((uintptr_t)info.ptr, (size_t)(info.size * info.itemsize), contiguous);
  });

  // upload a list of descriptors into a fresh device array (blocking)
  m.def("descs_upload", [](int device, const std::vector<SurfaceDesc>& descs, uintptr_t stream) {
    if (descs.empty())
      throw py::value_error("descs_upload: empty list");
    std::vector<vali_surface> host(descs.size());
    for (size_t i = 0; i < descs.size(); ++i)
      host[i] = descs[i].s;
    const size_t bytes = host.size() * sizeof(vali_surface);
    void* d = nullptr;
    check(vali_mem_alloc(device, bytes, &d), "vali_mem_alloc");
    int rc = vali_memcpy2d_async(device, d, bytes, host.data(), bytes, bytes, 1, 0, P(stream));
    if (rc == VALI_OK)
      rc = vali_stream_sync(device, P(stream));
    if (rc != VALI_OK) {
      vali_mem_free(device, d);
      check(rc, "descs_upload");
    }
    return (uintptr_t)d;
  });

  // ---- operators: return the C status code; Python maps it to TaskExecInfo ----
  m.def("nv12_to_rgb",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, const Csc& csc, uintptr_t stream) {
          return vali_nv12_to_rgb(&src.s, &dst.s, &csc.k, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("nv12_to_rgb_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int width, int height, int dst_format,
           const Csc& csc, uintptr_t stream) {
          return vali_nv12_to_rgb_batch((const vali_surface*)P(d_src),
                                        (const vali_surface*)P(d_dst), n, width, height,
                                        dst_format, &csc.k, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());

  py::class_<CvtParams>(m, "CvtParams")
      .def(py::init([](const Csc* csc, const std::vector<std::vector<float>>& rgb2yuv) {
             CvtParams p;
             std::memset(&p.p, 0, sizeof(p.p));
             if (csc)
               p.p.yuv2rgb = csc->k;
             for (size_t i = 0; i < rgb2yuv.size() && i < 3; ++i)
               for (size_t j = 0; j < rgb2yuv[i].size() && j < 4; ++j)
                 p.p.rgb2yuv[i][j] = rgb2yuv[i][j];
             return p;
           }),
           py::arg("csc") = nullptr, py::arg("rgb2yuv") = std::vector<std::vector<float>>());
  m.def("convert",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, const CvtParams& p, uintptr_t stream) {
          return vali_convert(&src.s, &dst.s, &p.p, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("convert_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int src_format, int dst_format, int width,
           int height, const CvtParams& p, uintptr_t stream) {
          return vali_convert_batch((const vali_surface*)P(d_src), (const vali_surface*)P(d_dst), n,
                                    src_format, dst_format, width, height, &p.p, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());

  py::class_<PreprocParams>(m, "PreprocParams")
      .def(py::init([](const Csc& csc, float div, const std::vector<float>& mean,
                       const std::vector<float>& std_) {
        PreprocParams p;
        std::memset(&p.p, 0, sizeof(p.p));
        p.p.csc = csc.k;
        p.p.div = div;
        for (size_t i = 0; i < 3; ++i) {
          p.p.mean[i] = i < mean.size() ? mean[i] : 0.0f;
          p.p.std_[i] = i < std_.size() ? std_[i] : 1.0f;
        }
        return p;
      }));
  m.def("nv12_preproc",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, const PreprocParams& p, uintptr_t stream) {
          return vali_nv12_preproc(&src.s, &dst.s, &p.p, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("nv12_preproc_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int src_width, int src_height, int dst_width,
           int dst_height, int dst_format, const PreprocParams& p, uintptr_t stream) {
          return vali_nv12_preproc_batch((const vali_surface*)P(d_src), (const vali_surface*)P(d_dst), n,
                                         src_width, src_height, dst_width, dst_height, dst_format, &p.p,
                                         P(stream));
        },
        py::call_guard<py::gil_scoped_release>());

  m.def("ud_nv12",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, uintptr_t stream) {
          return vali_ud_nv12(&src.s, &dst.s, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("ud_nv12_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int src_format, int src_w, int src_h, int dst_w,
           int dst_h, int dst_format, uintptr_t stream) {
          return vali_ud_nv12_batch((const vali_surface*)P(d_src), (const vali_surface*)P(d_dst),
                                    n, src_format, src_w, src_h, dst_w, dst_h, dst_format, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());

  m.def("ud_nv12_rot",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, int quarter_turns, uintptr_t stream) {
          return vali_ud_nv12_rot(&src.s, &dst.s, quarter_turns, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("ud_nv12_rot_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int src_format, int src_w, int src_h, int dst_w,
           int dst_h, int dst_format, int quarter_turns, uintptr_t stream) {
          return vali_ud_nv12_rot_batch((const vali_surface*)P(d_src), (const vali_surface*)P(d_dst),
                                        n, src_format, src_w, src_h, dst_w, dst_h, dst_format,
                                        quarter_turns, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());

  m.def("resize",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, int interp, uintptr_t stream) {
          return vali_resize(&src.s, &dst.s, interp, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("resize_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int format, int src_w, int src_h, int dst_w,
           int dst_h, int interp, uintptr_t stream) {
          return vali_resize_batch((const vali_surface*)P(d_src), (const vali_surface*)P(d_dst), n,
                                   format, src_w, src_h, dst_w, dst_h, interp, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("ud_planar",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, int interp, uintptr_t stream) {
          return vali_ud_planar(&src.s, &dst.s, interp, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("ud_planar_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int src_format, int dst_format, int src_w, int src_h,
           int dst_w, int dst_h, int interp, uintptr_t stream) {
          return vali_ud_planar_batch((const vali_surface*)P(d_src), (const vali_surface*)P(d_dst), n,
                                      src_format, dst_format, src_w, src_h, dst_w, dst_h, interp, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.attr("INTERP_LINEAR") = (int)VALI_INTERP_LINEAR;
  m.attr("INTERP_CUBIC") = (int)VALI_INTERP_CUBIC;
  m.attr("INTERP_LANCZOS") = (int)VALI_INTERP_LANCZOS;

  m.def("rotate",
        [](const SurfaceDesc& src, const SurfaceDesc& dst, double angle, double shx, double shy,
           int per_plane, uintptr_t stream) {
          return vali_rotate(&src.s, &dst.s, angle, shx, shy, per_plane, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("rotate_batch",
        [](uintptr_t d_src, uintptr_t d_dst, int n, int format, int sw, int sh, int dw, int dh,
           double angle, double shx, double shy, int per_plane, uintptr_t stream) {
          return vali_rotate_batch((const vali_surface*)P(d_src), (const vali_surface*)P(d_dst), n,
                                   format, sw, sh, dw, dh, angle, shx, shy, per_plane, P(stream));
        },
        py::call_guard<py::gil_scoped_release>());
  m.def("rotate_coeffs", [](double angle) {
    float c = 0, s = 0;
    check(vali_rotate_coeffs(angle, &c, &s), "vali_rotate_coeffs");
    return py::make_tuple(c, s);
  });

  m.def("debug_quantize_u8", [](uintptr_t in, uintptr_t out, int n, uintptr_t stream) {
    return vali_debug_quantize_u8((const float*)P(in), (uint8_t*)P(out), n, P(stream));
  });
  m.def("debug_quantize_u8_portable", [](uintptr_t in, uintptr_t out, int n, uintptr_t stream) {
    return vali_debug_quantize_u8_portable((const float*)P(in), (uint8_t*)P(out), n, P(stream));
  });

  m.def("dlpack_export", &dlpack_export, py::arg("ptr"), py::arg("shape"), py::arg("strides"),
        py::arg("code"), py::arg("bits"), py::arg("device_type"), py::arg("device_id"),
        py::arg("owner"));
  m.def("dlpack_import", &dlpack_import);
}

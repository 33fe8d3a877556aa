// torch_radon_cuda -- compiled pybind11 module with the surface of the reference's Radon backend
// (LoopDetection/torch-radon/src/pytorch.cpp:175-260: forward + the configuration / cache classes the Python package
// torch_radon constructs), bound to libmrslam_hip.so through the C ABI.
//
// With this module on the path the reference's OWN Python package (torch_radon/radon.py, volumes.py, projection.py,
// differentiable_functions.py) runs unmodified: ParallelBeam.forward -> RadonForward.apply -> cuda_backend.forward(x,
// angles, tex_cache, vol_cfg, proj_cfg, exec_cfg) lands here.  The role of the reference's TextureCache (CUDA arrays +
// texture objects, src/texture.cu) is taken by a cache of mrs_radon_plan objects keyed by geometry: gfx950 has no
// texture sampler, the image is staged in the LDS by the kernel itself.
//
// Tensors are handled through their Python interface (data_ptr / shape / device), so the module needs pybind11 only --
// no libtorch headers, no hipcc: g++ + -lmrslam_hip.  Everything MR_SLAM never calls (back-projection, noise, FFT
// helpers, fan / cone beam, 3-D volumes, half precision) raises NotImplementedError when used.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "mrslam_hip.h"

namespace py = pybind11;

namespace {

struct VolumeCfg {   // include/parameter_classes.h:9-35
    int depth, height, width;
    float dz, dy, dx, sz, sy, sx;
    bool is_3d;
    VolumeCfg(int d, int h, int w, float dz_, float dy_, float dx_, float sz_, float sy_, float sx_, bool ddd)
        : depth(d), height(h), width(w), dz(dz_), dy(dy_), dx(dx_), sz(sz_), sy(sy_), sx(sx_), is_3d(ddd) {}
};

struct ProjectionCfg {   // include/parameter_classes.h:37-66
    int det_count_u; float det_spacing_u; int det_count_v; float det_spacing_v;
    int n_angles = 0;
    float s_dist, d_dist, pitch, initial_z;
    int projection_type;
    ProjectionCfg(int dc_u, float ds_u, int dc_v = 0, float ds_v = 1.0f, float sd = 0.0f, float dd = 0.0f, float pi = 0.0f,
                  float iz = 0.0f, int pt = 0)
        : det_count_u(dc_u), det_spacing_u(ds_u), det_count_v(dc_v), det_spacing_v(ds_v), s_dist(sd), d_dist(dd), pitch(pi),
          initial_z(iz), projection_type(pt) {}
    bool is_2d() const { return projection_type == 0 || projection_type == 1; }
    ProjectionCfg copy() const { return *this; }
};

struct ExecCfg {
    int bx, by, bz, channels;
    ExecCfg(int x, int y, int z, int ch) : bx(x), by(y), bz(z), channels(ch) {}
};

void check(int st, const char* what)
{
    if (st != MRS_OK) throw std::runtime_error(std::string(what) + ": " + mrs_status_str(st) + ": " + mrs_last_error());
}

// one context per device, created on first use and kept for the life of the process
mrs_ctx* context(int device)
{
    static std::mutex mu;
    static std::map<int, mrs_ctx*> ctxs;
    std::lock_guard<std::mutex> lock(mu);
    auto it = ctxs.find(device);
    if (it != ctxs.end()) return it->second;
    mrs_ctx* c = nullptr;
    check(mrs_ctx_create(device, &c), "mrs_ctx_create");
    ctxs[device] = c;
    return c;
}

// stands where the reference's TextureCache stands: per-geometry resources reused across calls
struct TextureCache {
    using Key = std::tuple<int, int, int, int, float, std::vector<float>>;   // device, H, W, det, spacing, angles
    size_t capacity;
    std::map<Key, mrs_radon_plan*> plans;
    std::mutex mu;
    explicit TextureCache(size_t n) : capacity(n ? n : 1) {}
    ~TextureCache() { free(); }
    void free()
    {
        std::lock_guard<std::mutex> lock(mu);
        for (auto& kv : plans) mrs_radon_plan_destroy(kv.second);
        plans.clear();
    }
    mrs_radon_plan* plan(int device, int H, int W, int det, float spacing, const std::vector<float>& angles)
    {
        std::lock_guard<std::mutex> lock(mu);
        Key k(device, H, W, det, spacing, angles);
        auto it = plans.find(k);
        if (it != plans.end()) return it->second;
        if (plans.size() >= capacity) {   // the reference's cache evicts too (src/cache.h); plans are cheap to rebuild
            mrs_radon_plan_destroy(plans.begin()->second);
            plans.erase(plans.begin());
        }
        mrs_radon_plan* p = nullptr;
        check(mrs_radon_plan_create(context(device), angles.data(), (int)angles.size(), det, spacing, H, W, &p), "mrs_radon_plan_create");
        plans[k] = p;
        return p;
    }
};

struct FFTCache {
    explicit FFTCache(size_t) {}
    void free() {}
};

struct RadonNoiseGenerator {
    explicit RadonNoiseGenerator(unsigned) {}
    void set_seed(unsigned) {}
    void free() {}
};

[[noreturn]] void not_on_path(const char* what)
{
    PyErr_SetString(PyExc_NotImplementedError,
                    (std::string("torch_radon_cuda.") + what + " is not on MR_SLAM's loop-closure path (only the parallel-beam forward "
                     "projection is provided by the MI355X backend)").c_str());
    throw py::error_already_set();
}

// src/pytorch.cpp:42-81 radon_forward(x, angles, tex_cache, vol_cfg, proj_cfg, exec_cfg)
py::object radon_forward(py::object x, py::object angles, TextureCache& cache, const VolumeCfg& vol, const ProjectionCfg& proj,
                         const ExecCfg&)
{
    py::module_ torch = py::module_::import("torch");
    // CHECK_INPUT (pytorch.cpp:16-20): CUDA tensor, contiguous
    if (!x.attr("is_cuda").cast<bool>()) throw std::runtime_error("x must be a CUDA tensor");
    if (!x.attr("is_contiguous")().cast<bool>()) throw std::runtime_error("x must be contiguous");
    if (!angles.attr("is_cuda").cast<bool>()) throw std::runtime_error("angles must be a CUDA tensor");
    if (vol.is_3d || !proj.is_2d() || proj.projection_type != 0) not_on_path("forward for fan / cone beam or 3-D volumes");
    if (!x.attr("dtype").equal(torch.attr("float32"))) not_on_path("forward in half precision");
    if (vol.dx != 0.0f || vol.dy != 0.0f || vol.sx != 1.0f || vol.sy != 1.0f) not_on_path("forward with a shifted / scaled volume");
    const auto shape = x.attr("shape").cast<std::vector<int64_t>>();
    if (shape.size() != 3) throw std::runtime_error("x must be [batch, height, width]");
    const int batch = (int)shape[0], H = (int)shape[1], W = (int)shape[2];
    const std::vector<float> ang = angles.attr("detach")().attr("cpu")().attr("tolist")().cast<std::vector<float>>();
    const int device = x.attr("device").attr("index").is_none() ? 0 : x.attr("device").attr("index").cast<int>();
    mrs_radon_plan* plan = cache.plan(device, H, W, proj.det_count_u, proj.det_spacing_u, ang);
    py::object y = torch.attr("empty")(py::make_tuple(batch, (int)ang.size(), proj.det_count_u), py::arg("dtype") = x.attr("dtype"),
                                       py::arg("device") = x.attr("device"));
    if (batch == 0) return y;
    const uintptr_t stream = torch.attr("cuda").attr("current_stream")(x.attr("device")).attr("cuda_stream").cast<uintptr_t>();
    const float* src = reinterpret_cast<const float*>(x.attr("data_ptr")().cast<uintptr_t>());
    float* dst = reinterpret_cast<float*>(y.attr("data_ptr")().cast<uintptr_t>());
    int st;
    {
        py::gil_scoped_release nogil;
        st = mrs_radon_forward(plan, src, batch, dst, nullptr, reinterpret_cast<mrs_stream>(stream));
    }
    check(st, "mrs_radon_forward");
    return y;
}

}  // namespace

PYBIND11_MODULE(torch_radon_cuda, m)
{
    m.doc() = "MI355X backend behind the reference's torch_radon Python package (C ABI: libmrslam_hip.so)";
    m.def("forward", &radon_forward, "Radon forward projection");
    m.def("backward", [](py::args, py::kwargs) { not_on_path("backward"); }, "Radon back projection");
    m.def("add_noise", [](py::args, py::kwargs) { not_on_path("add_noise"); });
    m.def("emulate_sensor_readings", [](py::args, py::kwargs) { not_on_path("emulate_sensor_readings"); });
    m.def("symbolic_forward", [](py::args, py::kwargs) { not_on_path("symbolic_forward"); });
    m.def("symbolic_discretize", [](py::args, py::kwargs) { not_on_path("symbolic_discretize"); });
    m.def("rfft", [](py::args, py::kwargs) { not_on_path("rfft"); });
    m.def("irfft", [](py::args, py::kwargs) { not_on_path("irfft"); });
    m.def("set_log_level", [](int) {});
    py::class_<TextureCache>(m, "TextureCache").def(py::init<size_t>()).def("free", &TextureCache::free);
    py::class_<FFTCache>(m, "FFTCache").def(py::init<size_t>()).def("free", &FFTCache::free);
    py::class_<RadonNoiseGenerator>(m, "RadonNoiseGenerator").def(py::init<unsigned>()).def("set_seed", &RadonNoiseGenerator::set_seed)
        .def("free", &RadonNoiseGenerator::free);
    py::class_<VolumeCfg>(m, "VolumeCfg")
        .def(py::init<int, int, int, float, float, float, float, float, float, bool>())
        .def_readonly("depth", &VolumeCfg::depth).def_readonly("height", &VolumeCfg::height).def_readonly("width", &VolumeCfg::width)
        .def_readonly("dx", &VolumeCfg::dx).def_readonly("dy", &VolumeCfg::dy).def_readonly("dz", &VolumeCfg::dz)
        .def_readonly("is_3d", &VolumeCfg::is_3d);
    py::class_<ProjectionCfg>(m, "ProjectionCfg")
        .def(py::init<int, float>())
        .def(py::init<int, float, int, float, float, float, float, float, int>())
        .def("is_2d", &ProjectionCfg::is_2d).def("copy", &ProjectionCfg::copy)
        .def_readonly("projection_type", &ProjectionCfg::projection_type)
        .def_readwrite("det_count_u", &ProjectionCfg::det_count_u).def_readwrite("det_spacing_u", &ProjectionCfg::det_spacing_u)
        .def_readwrite("det_count_v", &ProjectionCfg::det_count_v).def_readwrite("det_spacing_v", &ProjectionCfg::det_spacing_v)
        .def_readwrite("s_dist", &ProjectionCfg::s_dist).def_readwrite("d_dist", &ProjectionCfg::d_dist)
        .def_readwrite("pitch", &ProjectionCfg::pitch).def_readwrite("initial_z", &ProjectionCfg::initial_z)
        .def_readwrite("n_angles", &ProjectionCfg::n_angles);
    py::class_<ExecCfg>(m, "ExecCfg").def(py::init<int, int, int, int>());
}

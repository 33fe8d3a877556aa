// pygicp -- compiled pybind11 module with the part of fast_gicp's Python binding (upstream src/python/main.cpp, module
// `pygicp`) that MR_SLAM calls (LoopDetection/src/RING_ros/main_RING.py:81-104, main_RINGplusplus.py:81-104,
// main_SC.py:108-131, disco_ros/main.py:174-197): downsample(points, resolution), class FastGICP with set_input_target /
// set_input_source / set_num_threads / set_max_correspondence_distance / set_correspondence_randomness / align /
// get_final_transformation / get_fitness_score / has_converged, and the convenience align_points().  Bound to
// libmrslam_hip.so through the C ABI; points travel as float64 [N,3] numpy arrays like upstream's Eigen::Matrix<double,-1,3>.
// pybind11 + the C ABI only (host arrays in and out): g++ -lmrslam_hip.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cfloat>
#include <cstdint>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "mrslam_hip.h"

namespace py = pybind11;
using Points = py::array_t<double, py::array::c_style | py::array::forcecast>;

namespace {

void check(int st, const char* what)
{
    if (st != MRS_OK) throw std::runtime_error(std::string(what) + ": " + mrs_status_str(st) + ": " + mrs_last_error());
}
mrs_ctx* ctx()
{
    static mrs_ctx* c = nullptr;
    static std::mutex m;
    std::lock_guard<std::mutex> lock(m);
    if (!c) check(mrs_ctx_create(0, &c), "mrs_ctx_create");
    return c;
}

void require_points(const Points& a)
{
    if (a.ndim() != 2 || a.shape(1) < 3) throw std::invalid_argument("points must be an [N, >= 3] array");
}

// pygicp.downsample(points, downsample_resolution): pcl::ApproximateVoxelGrid upstream, mrs_voxel_downsample_approx here
Points downsample(const Points& points, double resolution)
{
    require_points(points);
    const int64_t n = points.shape(0), stride = points.shape(1);
    if (n == 0) return Points(std::vector<py::ssize_t>{0, 3});
    std::vector<double> out((size_t)n * 3);
    int32_t count = 0;
    check(mrs_voxel_downsample_approx_host(ctx(), points.data(), 1, (int32_t)stride, (int32_t)n, resolution, out.data(), &count),
          "mrs_voxel_downsample_approx_host");
    Points res(std::vector<py::ssize_t>{count, 3});
    for (int64_t i = 0; i < (int64_t)count * 3; ++i) res.mutable_data()[i] = out[i];
    return res;
}

class FastGICP {
public:
    FastGICP()
    {
        mrs_gicp_default_params(&prm_);
        check(mrs_gicp_batch_create(ctx(), 1, &h_), "mrs_gicp_batch_create");
        for (int i = 0; i < 16; ++i) final_[i] = (i % 5 == 0) ? 1.0 : 0.0;
    }
    ~FastGICP() { if (h_) mrs_gicp_batch_destroy(h_); }
    FastGICP(const FastGICP&) = delete;
    FastGICP& operator=(const FastGICP&) = delete;

    void set_input_target(const Points& p) { upload(1, p); }
    void set_input_source(const Points& p) { upload(0, p); }
    void set_num_threads(int) {}                                   // OpenMP width of the CPU implementation: no meaning here
    void set_max_correspondence_distance(double d) { prm_.max_correspondence_distance = d; }
    void set_correspondence_randomness(int k) { prm_.k_correspondences = k; }
    void set_max_iterations(int n) { prm_.max_iterations = n; }
    void set_rotation_epsilon(double e) { prm_.rotation_epsilon = e; }
    void set_transformation_epsilon(double e) { prm_.transformation_epsilon = e; }
    // VGICP / VGICP_CUDA of align_points(): voxelised target, upstream's default neighbourhood (DIRECT1)
    void set_voxel_mode(double resolution) { prm_.voxel_resolution = resolution; prm_.voxel_neighbors = 1; }

    py::array_t<double> align(const py::array_t<double, py::array::c_style | py::array::forcecast>& initial_guess)
    {
        if (initial_guess.ndim() != 2 || initial_guess.shape(0) != 4 || initial_guess.shape(1) != 4)
            throw std::invalid_argument("initial_guess must be 4 x 4");
        check(mrs_gicp_batch_set_params(h_, &prm_), "mrs_gicp_batch_set_params");
        int32_t conv = 0, its = 0;
        check(mrs_gicp_batch_align(h_, initial_guess.data(), final_, &conv, &its, nullptr, nullptr), "mrs_gicp_batch_align");
        converged_ = conv != 0;
        iterations_ = its;
        return get_final_transformation();
    }
    py::array_t<double> get_final_transformation() const
    {
        py::array_t<double> t(std::vector<py::ssize_t>{4, 4});
        for (int i = 0; i < 16; ++i) t.mutable_data()[i] = final_[i];
        return t;
    }
    double get_fitness_score(double max_range) const
    {
        double score = 0.0;
        check(mrs_gicp_batch_fitness(h_, final_, max_range, &score, nullptr), "mrs_gicp_batch_fitness");
        return score;
    }
    bool has_converged() const { return converged_; }
    int iterations() const { return iterations_; }

private:
    void upload(int which, const Points& p)
    {
        require_points(p);
        const int64_t n = p.shape(0), stride = p.shape(1);
        std::vector<float> f((size_t)n * 3);
        for (int64_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) f[(size_t)i * 3 + c] = (float)p.data()[i * stride + c];      // upstream narrows to pcl::PointXYZ
        const int64_t offs[2] = {0, n};
        check(mrs_gicp_batch_set_clouds_host(h_, which, f.data(), 3, offs), "mrs_gicp_batch_set_clouds_host");
    }
    mrs_gicp_batch* h_ = nullptr;
    mrs_gicp_params prm_;
    double final_[16];
    bool converged_ = false;
    int iterations_ = 0;
};

// upstream's align_points(target, source, downsample_resolution, method, max_correspondence_distance, voxel_resolution,
// k_correspondences, num_threads, initial_guess): GICP and VGICP methods
py::array_t<double> align_points(const Points& target, const Points& source, double downsample_resolution, const std::string& method,
                                 double max_correspondence_distance, double voxel_resolution, int k_correspondences, int /*num_threads*/,
                                 const py::array_t<double, py::array::c_style | py::array::forcecast>& initial_guess)
{
    if (method != "GICP" && method != "VGICP" && method != "VGICP_CUDA")
        throw std::invalid_argument("method must be GICP, VGICP or VGICP_CUDA");
    FastGICP g;
    if (downsample_resolution > 0.0) {
        g.set_input_target(downsample(target, downsample_resolution));
        g.set_input_source(downsample(source, downsample_resolution));
    } else {
        g.set_input_target(target);
        g.set_input_source(source);
    }
    g.set_max_correspondence_distance(max_correspondence_distance);
    g.set_correspondence_randomness(k_correspondences);
    if (method != "GICP") {
        if (!(voxel_resolution > 0.0)) throw std::invalid_argument("voxel_resolution must be positive for VGICP / VGICP_CUDA");
        g.set_voxel_mode(voxel_resolution);       // the C ABI's voxel mode (mrs_gicp_params.voxel_resolution / voxel_neighbors)
    }
    return g.align(initial_guess);
}

py::array_t<double> identity4()
{
    py::array_t<double> t(std::vector<py::ssize_t>{4, 4});
    for (int i = 0; i < 16; ++i) t.mutable_data()[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return t;
}

}  // namespace

PYBIND11_MODULE(pygicp, m)
{
    m.doc() = "fast_gicp's pygicp surface used by MR_SLAM, on libmrslam_hip.so (MI355X)";
    m.def("downsample", &downsample, py::arg("points"), py::arg("downsample_resolution"));
    m.def("align_points", &align_points, py::arg("target"), py::arg("source"), py::arg("downsample_resolution") = -1.0,
          py::arg("method") = "GICP", py::arg("max_correspondence_distance") = 1.0, py::arg("voxel_resolution") = 1.0,
          py::arg("k_correspondences") = 15, py::arg("num_threads") = 0, py::arg("initial_guess") = identity4());
    py::class_<FastGICP>(m, "FastGICP")
        .def(py::init<>())
        .def("set_input_target", &FastGICP::set_input_target)
        .def("set_input_source", &FastGICP::set_input_source)
        .def("set_num_threads", &FastGICP::set_num_threads)
        .def("set_max_correspondence_distance", &FastGICP::set_max_correspondence_distance)
        .def("set_correspondence_randomness", &FastGICP::set_correspondence_randomness)
        .def("set_max_iterations", &FastGICP::set_max_iterations)
        .def("set_rotation_epsilon", &FastGICP::set_rotation_epsilon)
        .def("set_transformation_epsilon", &FastGICP::set_transformation_epsilon)
        .def("align", &FastGICP::align, py::arg("initial_guess") = identity4())
        .def("get_final_transformation", &FastGICP::get_final_transformation)
        .def("get_fitness_score", &FastGICP::get_fitness_score, py::arg("max_range") = DBL_MAX)
        .def("has_converged", &FastGICP::has_converged)
        .def("get_iterations", &FastGICP::iterations);
}

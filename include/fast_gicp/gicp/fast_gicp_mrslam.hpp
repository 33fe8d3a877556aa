// fast_gicp_mrslam.hpp -- C++ host-side mirror of fast_gicp::FastGICP for the Mapping workspace.
//
// Drop-in for the class Mapping/src/global_manager/src/global_manager.cpp:2435-2443 instantiates
// (`fast_gicp::FastGICP<pcl::PointXYZI, pcl::PointXYZI>` returned as
// `pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::Ptr`) and drives at :2016-2021
// (setInputSource / setInputTarget / align) and :2058-2071 (hasConverged / getFitnessScore /
// getFinalTransformation).  All arithmetic happens in libmrslam_hip.so through the C ABI
// (include/mrslam_hip.h); this header only adapts types.  It needs PCL + Eigen, which are not in
// the build image: tests/cpp/ compiles it against a minimal mock of the pcl::Registration
// surface it touches.  The forwarding headers next to this file carry upstream's names
// (<fast_gicp/gicp/fast_gicp.hpp>, fast_vgicp.hpp, fast_vgicp_cuda.hpp: the three includes of
// Mapping/src/global_manager/include/global_manager/global_manager.h:76-81), so adding
// <repo>/include to the include path is the whole source-side change.
//
// getFitnessScore: pcl::Registration::getFitnessScore is NOT virtual and upstream fast_gicp does not
// override it, so `icp->getFitnessScore(1.0)` through the pcl::Registration::Ptr of ICPCheck
// (global_manager.cpp:2058) runs PCL's own kd-tree implementation on the host -- with upstream and with
// this adapter alike.  The GPU score is available on the derived type (getFitnessScore below, same
// definition: mean squared NN distance over d^2 <= max_range).
#pragma once
#include <cfloat>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "mrslam_hip.h"


namespace fast_gicp {

enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS };
enum class NeighborSearchMethod { DIRECT27, DIRECT7, DIRECT1, DIRECT_RADIUS };

namespace detail {
// One library context per device and process, shared by every registration object (ICPCheck builds a new FastGICP per loop candidate,
// global_manager.cpp:2016: creating a context there would put a device query on the per-loop path).  Contexts live until the process ends.
inline mrs_ctx* shared_ctx(int device)
{
    static std::mutex mu;
    static std::map<int, mrs_ctx*> ctxs;
    std::lock_guard<std::mutex> lock(mu);
    auto it = ctxs.find(device);
    if (it != ctxs.end()) return it->second;
    mrs_ctx* c = nullptr;
    const int st = mrs_ctx_create(device, &c);
    if (st != MRS_OK) throw std::runtime_error(std::string("mrs_ctx_create: ") + mrs_status_str(st) + ": " + mrs_last_error());
    ctxs[device] = c;
    return c;
}
// device new registration objects are created on (default 0): fast_gicp::setDevice(i) before constructing them
inline int& default_device()
{
    static int dev = 0;
    return dev;
}
}  // namespace detail

inline void setDevice(int device) { detail::default_device() = device; }

template <typename PointSource, typename PointTarget>
class FastGICP : public pcl::Registration<PointSource, PointTarget, float> {
public:
    using Base = pcl::Registration<PointSource, PointTarget, float>;
    // like upstream: Ptr / ConstPtr point at the DERIVED class, so that
    //   fast_gicp::FastGICP<PointTI, PointTI>::Ptr gicp(new fast_gicp::FastGICP<PointTI, PointTI>());
    //   gicp->setNumThreads(8); ... return gicp;          (global_manager.cpp:2436-2442)
    // compiles and converts to pcl::Registration<...>::Ptr on return
#if defined(PCL_VERSION) && PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
    using Ptr = pcl::shared_ptr<FastGICP<PointSource, PointTarget>>;
    using ConstPtr = pcl::shared_ptr<const FastGICP<PointSource, PointTarget>>;
#else
    using Ptr = boost::shared_ptr<FastGICP<PointSource, PointTarget>>;
    using ConstPtr = boost::shared_ptr<const FastGICP<PointSource, PointTarget>>;
#endif
    using PointCloudSource = typename Base::PointCloudSource;
    using PointCloudSourceConstPtr = typename Base::PointCloudSourceConstPtr;
    using PointCloudTargetConstPtr = typename Base::PointCloudTargetConstPtr;
    using Matrix4 = typename Base::Matrix4;

    FastGICP()
    {
        this->reg_name_ = "FastGICP(mrslam_hip)";
        mrs_gicp_default_params(&prm_);
        this->max_iterations_ = prm_.max_iterations;
        this->transformation_epsilon_ = prm_.transformation_epsilon;
        ctx_ = detail::shared_ctx(detail::default_device());
        check(mrs_gicp_batch_create(ctx_, 1, &h_), "mrs_gicp_batch_create");
    }
    ~FastGICP() override { mrs_gicp_batch_destroy(h_); }

    // moves this object to another GPU of the node (its clouds have to be set again)
    void setDevice(int device)
    {
        mrs_ctx* c = detail::shared_ctx(device);
        if (c == ctx_) return;
        mrs_gicp_batch_destroy(h_);
        h_ = nullptr;
        ctx_ = c;
        uploaded_[0] = uploaded_[1] = nullptr;
        check(mrs_gicp_batch_create(ctx_, 1, &h_), "mrs_gicp_batch_create");
    }
    int getDevice() const { return mrs_ctx_device(ctx_); }
    FastGICP(const FastGICP&) = delete;
    FastGICP& operator=(const FastGICP&) = delete;

    void setNumThreads(int) {}  // OpenMP width of the CPU implementation; no meaning on the GPU
    void setCorrespondenceRandomness(int k) { prm_.k_correspondences = k; }
    void setRotationEpsilon(double e) { prm_.rotation_epsilon = e; }
    void setRegularizationMethod(RegularizationMethod m)
    {
        if (m != RegularizationMethod::PLANE) throw std::invalid_argument("only PLANE regularisation is implemented");
    }

    // like upstream (fast_gicp_impl.hpp: `if (input_ == cloud) return;`): handing over the SAME cloud object again keeps what is on the
    // device -- sorted points, box hierarchy and covariances; the Mapping node re-checks candidate pairs against the submap it already
    // holds (global_manager.cpp:2016-2021)
    void setInputSource(const PointCloudSourceConstPtr& cloud) override
    {
        if (cloud && this->input_ == cloud && uploaded_[0] == cloud.get()) return;
        Base::setInputSource(cloud);
        upload(0, *cloud);
        uploaded_[0] = cloud.get();
    }
    void setInputTarget(const PointCloudTargetConstPtr& cloud) override
    {
        if (cloud && this->target_ == cloud && uploaded_[1] == cloud.get()) return;
        Base::setInputTarget(cloud);
        upload(1, *cloud);
        uploaded_[1] = cloud.get();
    }

    // pcl::Registration::getFitnessScore(max_range): routed to the GPU NN pass (G6)
    double getFitnessScore(double max_range = DBL_MAX)
    {
        double pose[16], score = DBL_MAX;
        to_row_major(this->final_transformation_, pose);
        check(mrs_gicp_batch_fitness(h_, pose, max_range, &score, nullptr), "mrs_gicp_batch_fitness");
        return score;
    }

    const double* getFinalHessian() const { return hessian_; }

protected:
    void setVoxelMode(double resolution, int neighbors)
    {
        prm_.voxel_resolution = resolution;
        prm_.voxel_neighbors = neighbors;
    }

    void computeTransformation(PointCloudSource& output, const Matrix4& guess) override
    {
        prm_.max_iterations = this->max_iterations_;
        prm_.transformation_epsilon = this->transformation_epsilon_;
        prm_.max_correspondence_distance = this->corr_dist_threshold_;
        check(mrs_gicp_batch_set_params(h_, &prm_), "mrs_gicp_batch_set_params");
        double g[16], f[16];
        to_row_major(guess, g);
        int32_t conv = 0, iters = 0;
        check(mrs_gicp_batch_align(h_, g, f, &conv, &iters, hessian_, nullptr), "mrs_gicp_batch_align");
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) this->final_transformation_(r, c) = static_cast<float>(f[4 * r + c]);
        this->converged_ = conv != 0;
        this->nr_iterations_ = iters;
        pcl::transformPointCloud(*this->input_, output, this->final_transformation_);
    }

private:
    template <class Cloud>
    void upload(int which, const Cloud& cloud)
    {
        // PCL points are 16-byte aligned structs whose first three floats are x, y, z; the library stages the host array through
        // its own scratch cache (no allocation per call once warm)
        const int stride = static_cast<int>(sizeof(typename Cloud::PointType) / sizeof(float));
        const int64_t offs[2] = {0, static_cast<int64_t>(cloud.points.size())};
        check(mrs_gicp_batch_set_clouds_host(h_, which, reinterpret_cast<const float*>(cloud.points.data()), stride, offs),
              "mrs_gicp_batch_set_clouds_host");
    }
    template <class M>
    static void to_row_major(const M& m, double* out)
    {
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) out[4 * r + c] = static_cast<double>(m(r, c));
    }
    static void check(int st, const char* what)
    {
        if (st != MRS_OK)
            throw std::runtime_error(std::string(what) + ": " + mrs_status_str(st) + ": " + mrs_last_error());
    }

    mrs_ctx* ctx_ = nullptr;
    mrs_gicp_batch* h_ = nullptr;
    mrs_gicp_params prm_;
    double hessian_[36] = {0};
    const void* uploaded_[2] = {nullptr, nullptr};   // the cloud objects whose points are on the device (source, target)
};

// Drop-in for fast_gicp::FastVGICPCuda (the launch-file default `registration_method=FAST_VGICP_CUDA`,
// Mapping/src/global_manager/launch/global_manager.launch:51; configured at global_manager.cpp:2445-2455):
// voxelised GICP, row G7 of SURVEY.md section 8(a).
template <typename PointSource, typename PointTarget>
class FastVGICPCuda : public FastGICP<PointSource, PointTarget> {
public:
#if defined(PCL_VERSION) && PCL_VERSION >= PCL_VERSION_CALC(1, 10, 0)
    using Ptr = pcl::shared_ptr<FastVGICPCuda<PointSource, PointTarget>>;
    using ConstPtr = pcl::shared_ptr<const FastVGICPCuda<PointSource, PointTarget>>;
#else
    using Ptr = boost::shared_ptr<FastVGICPCuda<PointSource, PointTarget>>;
    using ConstPtr = boost::shared_ptr<const FastVGICPCuda<PointSource, PointTarget>>;
#endif
    FastVGICPCuda()
    {
        this->reg_name_ = "FastVGICPCuda(mrslam_hip)";
        this->setVoxelMode(1.0, 1);  // upstream defaults: resolution 1.0, DIRECT1
    }
    void setResolution(double r) { res_ = r; this->setVoxelMode(res_, nb_); }
    void setNeighborSearchMethod(NeighborSearchMethod m, double /*radius*/ = -1.0)
    {
        nb_ = m == NeighborSearchMethod::DIRECT27 ? 27 : (m == NeighborSearchMethod::DIRECT7 ? 7 : 1);
        this->setVoxelMode(res_, nb_);
    }
    // upstream: kernel width of the GPU_RBF_KERNEL covariance estimator, which only takes effect after
    // setNearestNeighborSearchMethod(GPU_RBF_KERNEL); MR_SLAM never selects it (global_manager.cpp:2446-2453), the
    // default estimator (kNN covariances, what this class computes) ignores the width upstream too
    void setKernelWidth(double, double = -1.0) {}
private:
    double res_ = 1.0;
    int nb_ = 1;
};

template <typename PointSource, typename PointTarget>
using FastVGICP = FastVGICPCuda<PointSource, PointTarget>;

}  // namespace fast_gicp

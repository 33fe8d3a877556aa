// Minimal stand-in for the slice of PCL/Eigen the adapter touches (PCL and Eigen are not in the
// image).  Mirrors names and members of pcl::Registration<S,T,float> (pcl/registration/registration.h)
// that fast_gicp-style subclasses use.  Test scaffolding only.
#pragma once
#include <cfloat>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#define PCL_VERSION_CALC(MAJ, MIN, PATCH) ((MAJ)*100000 + (MIN)*100 + (PATCH))
#define PCL_VERSION PCL_VERSION_CALC(1, 12, 0)   // pcl/pcl_config.h: a PCL with pcl::shared_ptr (>= 1.10)

namespace Eigen {
struct Matrix4f {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};  // column-major like Eigen
    float& operator()(int r, int c) { return m[4 * c + r]; }
    float operator()(int r, int c) const { return m[4 * c + r]; }
    static Matrix4f Identity() { return Matrix4f(); }
};
}  // namespace Eigen

namespace pcl {
template <class T>
using shared_ptr = std::shared_ptr<T>;            // pcl/memory.h (PCL >= 1.10; boost::shared_ptr before)
struct alignas(16) PointXYZI { float x, y, z, pad; float intensity, p1, p2, p3; };
template <class P>
struct PointCloud {
    using PointType = P;
    using Ptr = std::shared_ptr<PointCloud<P>>;
    using ConstPtr = std::shared_ptr<const PointCloud<P>>;
    std::vector<P> points;
};
template <class P>
void transformPointCloud(const PointCloud<P>& in, PointCloud<P>& out, const Eigen::Matrix4f& T)
{
    out.points = in.points;
    for (auto& p : out.points) {
        const float x = p.x, y = p.y, z = p.z;
        p.x = T(0, 0) * x + T(0, 1) * y + T(0, 2) * z + T(0, 3);
        p.y = T(1, 0) * x + T(1, 1) * y + T(1, 2) * z + T(1, 3);
        p.z = T(2, 0) * x + T(2, 1) * y + T(2, 2) * z + T(2, 3);
    }
}
template <class S, class T, class Scalar = float>
class Registration {
public:
    using Matrix4 = Eigen::Matrix4f;
    using PointCloudSource = PointCloud<S>;
    using PointCloudSourceConstPtr = typename PointCloudSource::ConstPtr;
    using PointCloudTarget = PointCloud<T>;
    using PointCloudTargetConstPtr = typename PointCloudTarget::ConstPtr;
    using Ptr = std::shared_ptr<Registration<S, T, Scalar>>;
    virtual ~Registration() = default;
    virtual void setInputSource(const PointCloudSourceConstPtr& c) { input_ = c; }
    virtual void setInputTarget(const PointCloudTargetConstPtr& c) { target_ = c; }
    void setMaximumIterations(int n) { max_iterations_ = n; }
    void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
    void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
    bool hasConverged() const { return converged_; }
    // NOT virtual in PCL (pcl/registration/registration.h): through a Registration::Ptr this host implementation
    // runs, whatever the derived class defines.  Mean squared NN distance over d^2 <= max_range (brute force here).
    double getFitnessScore(double max_range = std::numeric_limits<double>::max())
    {
        PointCloudSource moved;
        transformPointCloud(*input_, moved, final_transformation_);
        double sum = 0; long nr = 0;
        for (const auto& p : moved.points) {
            double best = std::numeric_limits<double>::max();
            for (const auto& q : target_->points) {
                const double dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
                const double d = dx * dx + dy * dy + dz * dz;
                if (d < best) best = d;
            }
            if (best <= max_range) { sum += best; ++nr; }
        }
        return nr > 0 ? sum / nr : std::numeric_limits<double>::max();
    }
    Matrix4 getFinalTransformation() const { return final_transformation_; }
    void align(PointCloudSource& output, const Matrix4& guess = Matrix4::Identity()) { computeTransformation(output, guess); }
protected:
    virtual void computeTransformation(PointCloudSource& output, const Matrix4& guess) = 0;
    std::string reg_name_;
    PointCloudSourceConstPtr input_;
    PointCloudTargetConstPtr target_;
    Matrix4 final_transformation_;
    bool converged_ = false;
    int nr_iterations_ = 0;
    int max_iterations_ = 10;
    double transformation_epsilon_ = 0.0;
    double corr_dist_threshold_ = 1.79769e308;
};
}  // namespace pcl

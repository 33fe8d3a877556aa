// Compiles the fast_gicp adapter against the PCL mock and (on a GPU box) runs the call sequence
// of GlobalManager::ICPCheck (global_manager.cpp:2016-2021, 2058-2071, 2437-2442).
#include "mock_pcl.hpp"
#include "fast_gicp/gicp/fast_gicp_mrslam.hpp"

#include <cmath>
#include <cstdio>
#include <random>

int main()
{
    using Cloud = pcl::PointCloud<pcl::PointXYZI>;
    auto src = std::make_shared<Cloud>();
    auto tgt = std::make_shared<Cloud>();
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-20.f, 20.f);
    std::normal_distribution<float> nz(0.f, 0.01f);
    const float yaw = 0.05f, tx = 0.4f, ty = -0.2f;
    for (int i = 0; i < 6000; ++i) {  // three orthogonal noisy planes
        pcl::PointXYZI p{};
        const float a = u(rng), b = u(rng);
        if (i % 3 == 0) { p.x = a; p.y = b; p.z = nz(rng); }
        else if (i % 3 == 1) { p.x = a; p.y = 20.f + nz(rng); p.z = std::fabs(b) * 0.3f; }
        else { p.x = -20.f + nz(rng); p.y = a; p.z = std::fabs(b) * 0.3f; }
        src->points.push_back(p);
        pcl::PointXYZI q = p;
        q.x = std::cos(yaw) * p.x - std::sin(yaw) * p.y + tx + nz(rng);
        q.y = std::sin(yaw) * p.x + std::cos(yaw) * p.y + ty + nz(rng);
        tgt->points.push_back(q);
    }
    pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::Ptr icp;
    {
        auto gicp = std::make_shared<fast_gicp::FastGICP<pcl::PointXYZI, pcl::PointXYZI>>();
        gicp->setNumThreads(8);
        gicp->setTransformationEpsilon(1e-3);
        gicp->setMaximumIterations(50);
        gicp->setMaxCorrespondenceDistance(100.0);
        gicp->setCorrespondenceRandomness(15);
        icp = gicp;
    }
    icp->setInputSource(src);
    icp->setInputTarget(tgt);
    Cloud unused;
    icp->align(unused, Eigen::Matrix4f::Identity());
    const Eigen::Matrix4f T = icp->getFinalTransformation();
    const double fit = std::static_pointer_cast<fast_gicp::FastGICP<pcl::PointXYZI, pcl::PointXYZI>>(icp)->getFitnessScore(1.0);
    std::printf("converged=%d tx=%.4f ty=%.4f yaw=%.5f fitness=%.6f\n", (int)icp->hasConverged(), T(0, 3), T(1, 3),
                std::atan2(T(1, 0), T(0, 0)), fit);
    const bool ok = icp->hasConverged() && std::fabs(T(0, 3) - tx) < 5e-3 && std::fabs(T(1, 3) - ty) < 5e-3 &&
                    std::fabs(std::atan2(T(1, 0), T(0, 0)) - yaw) < 1e-3;
    // launch-file default: FAST_VGICP_CUDA (global_manager.cpp:2445-2455), started near the solution like ICPCheck does
    auto vg = std::make_shared<fast_gicp::FastVGICPCuda<pcl::PointXYZI, pcl::PointXYZI>>();
    vg->setResolution(0.5);
    vg->setTransformationEpsilon(1e-3);
    vg->setMaximumIterations(50);
    vg->setCorrespondenceRandomness(15);
    vg->setNeighborSearchMethod(fast_gicp::NeighborSearchMethod::DIRECT1, 1.5);
    vg->setInputSource(src);
    vg->setInputTarget(tgt);
    Eigen::Matrix4f guess = Eigen::Matrix4f::Identity();
    guess(0, 0) = std::cos(0.045f); guess(0, 1) = -std::sin(0.045f); guess(1, 0) = std::sin(0.045f); guess(1, 1) = std::cos(0.045f);
    guess(0, 3) = 0.35f; guess(1, 3) = -0.15f;
    vg->align(unused, guess);
    const Eigen::Matrix4f V = vg->getFinalTransformation();
    std::printf("vgicp converged=%d tx=%.4f ty=%.4f yaw=%.5f\n", (int)vg->hasConverged(), V(0, 3), V(1, 3), std::atan2(V(1, 0), V(0, 0)));
    const bool okv = vg->hasConverged() && std::fabs(V(0, 3) - tx) < 2e-2 && std::fabs(V(1, 3) - ty) < 2e-2 &&
                     std::fabs(std::atan2(V(1, 0), V(0, 0)) - yaw) < 3e-3;
    return (ok && okv) ? 0 : 1;
}

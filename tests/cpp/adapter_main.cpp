// Compiles the fast_gicp adapter against the PCL mock through upstream's header names and (on a GPU box) runs the
// call sequence of GlobalManager::ICPCheck.  select_registration_method() below is the FAST_GICP / FAST_VGICP_CUDA
// part of Mapping/src/global_manager/src/global_manager.cpp:2416-2461 transcribed statement by statement (test
// scaffolding: it has to be the reference's text to prove that text compiles); ICPCheck's use of the returned
// pointer follows :2016-2021 and :2058-2071.
#include "mock_pcl.hpp"
// global_manager.h:76-81
#include <fast_gicp/gicp/fast_gicp.hpp>
#include <fast_gicp/gicp/fast_vgicp.hpp>
#define USE_VGICP_CUDA
#ifdef USE_VGICP_CUDA
#include <fast_gicp/gicp/fast_vgicp_cuda.hpp>
#endif

#include <cmath>
#include <cstdio>
#include <iostream>
#include <random>
#include <string>

using std::cerr;
using std::endl;
typedef pcl::PointXYZI PointTI;
typedef pcl::PointCloud<PointTI> PointCloudI;
typedef PointCloudI::Ptr PointCloudIPtr;

struct GlobalManagerSlice {
    std::string registration_method_;
    double icp_iters_ = 50;   // launch/global_manager.launch:53

    pcl::Registration<pcl::PointXYZI, pcl::PointXYZI>::Ptr select_registration_method(std::string type)
    {
        if (registration_method_ == "FAST_GICP") {
            std::cout << "registration: FAST_GICP" << std::endl;
            fast_gicp::FastGICP<PointTI, PointTI>::Ptr gicp(new fast_gicp::FastGICP<PointTI, PointTI>());
            gicp->setNumThreads(8);
            gicp->setTransformationEpsilon(1e-3);
            gicp->setMaximumIterations((int)icp_iters_);
            gicp->setMaxCorrespondenceDistance(100.0);
            gicp->setCorrespondenceRandomness(15);
            return gicp;
        }
        else if (registration_method_ == "FAST_VGICP_CUDA") {
#ifdef USE_VGICP_CUDA
            std::cout << "registration: FAST_VGICP_CUDA" << std::endl;
            fast_gicp::FastVGICPCuda<PointTI, PointTI>::Ptr vgicp(new fast_gicp::FastVGICPCuda<PointTI, PointTI>());
            vgicp->setResolution(0.5);
            vgicp->setTransformationEpsilon(1e-3);
            vgicp->setMaximumIterations((int)icp_iters_);
            vgicp->setCorrespondenceRandomness(15);
            vgicp->setNeighborSearchMethod(fast_gicp::NeighborSearchMethod::DIRECT1, 1.5);

            return vgicp;
#endif
            cerr << "FAST_VGICP_CUDA is Not Build !!" << endl;
        }
        else {
            cerr << "Not Implemented Registration Method !!" << endl;
        }
        return nullptr;
    }
};

int main()
{
    PointCloudIPtr queryKeyframe(new PointCloudI), databaseKeyframe(new PointCloudI);
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
        queryKeyframe->points.push_back(p);
        pcl::PointXYZI q = p;
        q.x = std::cos(yaw) * p.x - std::sin(yaw) * p.y + tx + nz(rng);
        q.y = std::sin(yaw) * p.x + std::cos(yaw) * p.y + ty + nz(rng);
        databaseKeyframe->points.push_back(q);
    }
    GlobalManagerSlice gm;
    bool all_ok = true;
    for (const char* method : {"FAST_GICP", "FAST_VGICP_CUDA"}) {
        gm.registration_method_ = method;
        // ICPCheck starts from the place-recognition estimate (global_manager.cpp:1990-2006): near the solution
        Eigen::Matrix4f transformMatrixf = Eigen::Matrix4f::Identity();
        if (std::string(method) == "FAST_VGICP_CUDA") {
            transformMatrixf(0, 0) = std::cos(0.045f); transformMatrixf(0, 1) = -std::sin(0.045f);
            transformMatrixf(1, 0) = std::sin(0.045f); transformMatrixf(1, 1) = std::cos(0.045f);
            transformMatrixf(0, 3) = 0.35f; transformMatrixf(1, 3) = -0.15f;
        }
        auto icp = gm.select_registration_method(gm.registration_method_);   // global_manager.cpp:2016

        icp->setInputSource(queryKeyframe);
        icp->setInputTarget(databaseKeyframe);
        PointCloudIPtr unused_result(new PointCloudI);
        icp->align(*unused_result, transformMatrixf);

        const double acceptedKeyframeFitnessScore = 0.3;
        const bool rejected = icp->hasConverged() == false || icp->getFitnessScore(1.0) > acceptedKeyframeFitnessScore;   // :2058 (PCL's host score)
        auto finalResult = icp->getFinalTransformation();
        // the GPU score of the derived class (same definition) next to PCL's
        double gpu_fit = -1.0;
        if (auto g = std::dynamic_pointer_cast<fast_gicp::FastGICP<PointTI, PointTI>>(icp)) {
            gpu_fit = g->getFitnessScore(1.0);
            // one library context per device, shared by every registration object (a new one per loop candidate costs no device query);
            // setDevice(0) on device 0 keeps the object and its clouds
            fast_gicp::FastGICP<PointTI, PointTI> other;
            if (g->getDevice() != 0 || other.getDevice() != 0) all_ok = false;
            g->setDevice(0);
            if (std::fabs(g->getFitnessScore(1.0) - gpu_fit) > 0.0) all_ok = false;
        }
        // the SAME cloud objects handed over again (the node re-checks a pair it already holds): the adapter keeps what is on the device,
        // like upstream's `if (input_ == cloud) return;`, and the alignment repeats bit for bit
        icp->setInputSource(queryKeyframe);
        icp->setInputTarget(databaseKeyframe);
        PointCloudIPtr again(new PointCloudI);
        icp->align(*again, transformMatrixf);
        auto repeated = icp->getFinalTransformation();
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c)
                if (repeated(r, c) != finalResult(r, c)) all_ok = false;
        const double host_fit = icp->getFitnessScore(1.0);
        std::printf("%s converged=%d tx=%.4f ty=%.4f yaw=%.5f fitness(pcl host)=%.6f fitness(gpu)=%.6f aligned=%zu\n", method,
                    (int)icp->hasConverged(), finalResult(0, 3), finalResult(1, 3), std::atan2(finalResult(1, 0), finalResult(0, 0)),
                    host_fit, gpu_fit, unused_result->points.size());
        const double tol_t = std::string(method) == "FAST_GICP" ? 5e-3 : 2e-2, tol_r = std::string(method) == "FAST_GICP" ? 1e-3 : 3e-3;
        const bool ok = !rejected && std::fabs(finalResult(0, 3) - tx) < tol_t && std::fabs(finalResult(1, 3) - ty) < tol_t &&
                        std::fabs(std::atan2(finalResult(1, 0), finalResult(0, 0)) - yaw) < tol_r &&
                        std::fabs(gpu_fit - host_fit) < 1e-6 + 1e-4 * host_fit && unused_result->points.size() == queryKeyframe->points.size();
        if (!ok)
            std::printf("  FAILED %s: rejected=%d dtx=%.2e dty=%.2e dyaw=%.2e |gpu-host|=%.3e aligned=%zu\n", method, (int)rejected,
                        std::fabs(finalResult(0, 3) - tx), std::fabs(finalResult(1, 3) - ty),
                        std::fabs(std::atan2(finalResult(1, 0), finalResult(0, 0)) - yaw), std::fabs(gpu_fit - host_fit), unused_result->points.size());
        all_ok = all_ok && ok;
    }
    return all_ok ? 0 : 1;
}

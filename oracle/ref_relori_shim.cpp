// oracle/ref_relori_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// GlobalManager::calcRelOri as the reference wrote it (Mapping/src/global_manager/src/global_manager.cpp:2719-2762): oracle/Makefile
// cuts exactly those lines out of the reference file into a scratch include under oracle/_ref/build/gm/ (git-ignored, never
// committed) and this file supplies the surroundings the function needs: the DiSCOFFT typedef (typedefs.h:81), a GlobalManager
// with the two members it reads, and stand-ins for FFTW (ref_gm_host/fftw3.h) and Eigen::VectorXf (ref_cuda_host/Eigen/Core),
// neither of which is in this image.  The whole global_manager.cpp cannot be built here (ROS, PCL, GTSAM).
#include <cmath>
#include <cstddef>
#include <utility>
#include <vector>

#include <Eigen/Core>
#include <fftw3.h>

typedef std::pair<std::vector<float>, std::vector<float>> DiSCOFFT;

struct GlobalManager {
    int disco_width_, disco_height_;
    float calcRelOri(DiSCOFFT newDiSCO, DiSCOFFT oldDiSCO);
};

#include "calc_rel_ori.inc"

extern "C" float ref_calc_rel_ori(const float* real_a, const float* imag_a, const float* real_b, const float* imag_b, int height, int width)
{
    const size_t n = (size_t)height * width;
    GlobalManager gm{width, height};
    return gm.calcRelOri(DiSCOFFT(std::vector<float>(real_a, real_a + n), std::vector<float>(imag_a, imag_a + n)),
                         DiSCOFFT(std::vector<float>(real_b, real_b + n), std::vector<float>(imag_b, imag_b + n)));
}

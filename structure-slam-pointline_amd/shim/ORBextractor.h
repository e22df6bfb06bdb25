// Drop-in replacement of the reference's include/ORBextractor.h: same namespace, class name,
// constructor and public methods (include/ORBextractor.h:45-111), implemented over the C ABI
// (include/sslam_frontend.h) instead of OpenCV on the CPU.  Frame.cc / Tracking.cc compile
// against this header unchanged (src/Frame.cc:155-161, src/Tracking.cc:113-128).
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H
#include <vector>
#include "cv_min.h"

struct sslam_orb;

namespace StructureSLAM
{
class ORBextractor
{
public:
    enum {HARRIS_SCORE=0, FAST_SCORE=1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();

    // Compute the ORB features and descriptors on an image (mask is ignored, as in the reference).
    void operator()( cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors);

    int inline GetLevels(){ return nlevels; }
    float inline GetScaleFactor(){ return (float)scaleFactor; }
    std::vector<float> inline GetScaleFactors(){ return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors(){ return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares(){ return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares(){ return mvInvLevelSigma2; }

    // The reference exposes mvImagePyramid but nothing outside ORBextractor.cc reads it
    // (SURVEY.md §8b); the pyramid lives in HBM only.
    std::vector<cv::Mat> mvImagePyramid;

protected:
    int nfeatures; double scaleFactor; int nlevels; int iniThFAST; int minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    sslam_orb* mHandle;
    std::vector<cv::KeyPoint> mStageKeys;
};
} //namespace StructureSLAM
#endif

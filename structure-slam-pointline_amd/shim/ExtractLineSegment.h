// Drop-in replacement of the reference's include/ExtractLineSegment.h (class LineSegment,
// :53-78).  ExtractLineSegment keeps the reference signature.  NB the reference calls it through
// an UNINITIALISED Frame::mpLineSegment pointer (include/Frame.h:125, src/Frame.cc:150-153), so
// the body must not touch `this`: all state lives in a process-global front-end context.
#ifndef ORB_SLAM2_LINEFEATURE_H
#define ORB_SLAM2_LINEFEATURE_H
#include <vector>
#include "cv_min.h"

namespace StructureSLAM
{
class LineSegment
{
public:
    LineSegment();
    ~LineSegment(){}
    void ExtractLineSegment(const cv::Mat &img, std::vector<cv::line_descriptor::KeyLine> &keylines, cv::Mat &ldesc,
                            std::vector<sslam_shim::Vector3d> &keylineFunctions, int scale = 1.2, int numOctaves = 1);
    // The reference hard-codes 40 lines (src/ExtractLineSegment.cpp:42); BASELINE configs use 200/400.
    static void SetMaxLines(int n);
};
}
#endif

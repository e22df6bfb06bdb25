// ORACLE ref_pin stub (test infrastructure): stands in for <opencv2/imgproc/imgproc.hpp>
#pragma once
#include "../../stub_cv.hpp"

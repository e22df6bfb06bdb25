// ORACLE ref_pin stub (test infrastructure): stands in for <opencv2/features2d/features2d.hpp>
#pragma once
#include "../../stub_cv.hpp"

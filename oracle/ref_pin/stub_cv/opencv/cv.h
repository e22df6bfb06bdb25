// ORACLE ref_pin stub (test infrastructure): stands in for <opencv/cv.h>
#pragma once
#include "../stub_cv.hpp"

// tests/compat_stubs: cv::KeyPoint lives in core.hpp of this guard (see README.md in this directory)
#pragma once
#include <opencv2/core/core.hpp>

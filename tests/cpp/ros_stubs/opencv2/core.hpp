#pragma once
#include <cstdint>
#define CV_8UC1 0
namespace cv {
struct Scalar { Scalar(double = 0); };
struct Mat {
    int rows = 0, cols = 0; unsigned char* data = nullptr;
    Mat(); Mat(int rows, int cols, int type); Mat(int rows, int cols, int type, void* data);
    Mat& setTo(const Scalar&); Mat col(int) const; void resize(size_t);
};
template <typename T> struct Mat_ : Mat { Mat_(); Mat_(int rows, int cols); };
}

// tests/compat_stubs: declarations only (see README.md in this directory) -- a typo guard, not OpenCV.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>
#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_Assert(expr) do { if (!(expr)) throw 0; } while (0)
namespace cv {
struct Point2f { float x, y; Point2f(); Point2f(float, float); };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
class MatExpr;
class Mat {
public:
    Mat(); Mat(int rows, int cols, int type); Mat(int rows, int cols, int type, void *data); Mat(const MatExpr &);
    Mat &operator=(const MatExpr &);
    int rows, cols; unsigned char *data; size_t step;
    int type() const; bool empty() const; size_t total() const; Mat clone() const; Mat t() const;
    void create(int rows, int cols, int type); void copyTo(Mat) const; void release();
    Mat rowRange(int, int) const; Mat colRange(int, int) const; Mat col(int) const; Mat row(int) const;
    template <typename T> T &at(int); template <typename T> const T &at(int) const;
    template <typename T> T &at(int, int); template <typename T> const T &at(int, int) const;
    template <typename T> T *ptr(int r = 0); template <typename T> const T *ptr(int r = 0) const;
    unsigned char *ptr(int r = 0); const unsigned char *ptr(int r = 0) const;
    Mat reshape(int cn, int rows = 0) const; double dot(const Mat &) const;
};
class MatExpr { public: MatExpr(const Mat &); operator Mat() const; Mat t() const; template <typename T> T at(int) const; };
MatExpr operator+(const Mat &, const Mat &); MatExpr operator-(const Mat &, const Mat &); MatExpr operator*(const Mat &, const Mat &);
MatExpr operator-(const Mat &); MatExpr operator*(const Mat &, double); MatExpr operator*(double, const Mat &);
MatExpr operator+(const MatExpr &, const Mat &); MatExpr operator-(const MatExpr &, const Mat &); MatExpr operator*(const MatExpr &, const Mat &);
MatExpr operator+(const Mat &, const MatExpr &); MatExpr operator-(const Mat &, const MatExpr &); MatExpr operator*(const Mat &, const MatExpr &);
MatExpr operator+(const MatExpr &, const MatExpr &); MatExpr operator-(const MatExpr &, const MatExpr &); MatExpr operator*(const MatExpr &, const MatExpr &);
MatExpr operator-(const MatExpr &); MatExpr operator*(const MatExpr &, double); MatExpr operator*(double, const MatExpr &);
MatExpr operator/(const Mat &, double); MatExpr operator/(const MatExpr &, double);
double norm(const Mat &); double norm(const MatExpr &);
class _InputArray { public: _InputArray(const Mat &); Mat getMat() const; bool empty() const; };
class _OutputArray { public: _OutputArray(Mat &); Mat getMat() const; void create(int rows, int cols, int type) const; void release() const; };
typedef const _InputArray &InputArray;
typedef const _OutputArray &OutputArray;
void undistortPoints(InputArray src, OutputArray dst, InputArray K, InputArray D, InputArray R, InputArray P);
}  // namespace cv

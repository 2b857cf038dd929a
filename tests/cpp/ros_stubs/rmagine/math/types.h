#pragma once
namespace rmagine {
struct Vector { float x, y, z; static Vector Zeros(); Vector operator+(const Vector&) const; Vector operator*(double) const; };
struct Quaternion { float x, y, z, w; };
struct Transform { Quaternion R; Vector t; static Transform Identity(); };
}

#pragma once
#include <rmagine/math/types.h>
#include <cstdint>
namespace rmagine {
struct Interval { float min, max; };
struct DiscreteInterval { float min, inc; uint32_t size; };
struct SphericalModel { DiscreteInterval phi, theta; Interval range; float getTheta(uint32_t) const; Vector getOrigin(uint32_t, uint32_t) const; };
}

// signatures only (see README.md)
#pragma once
#include <cstdint>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <boost/bind.hpp>
namespace ros {
struct Duration { double s = 0; Duration() {} explicit Duration(double d) : s(d) {} double toSec() const { return s; } };
struct Time {
    uint32_t sec = 0, nsec = 0;
    Time() {} explicit Time(double) {}
    static Time now();
    bool operator==(const Time&) const; bool operator!=(const Time&) const;
    Time operator-(const Duration&) const; Time operator+(const Duration&) const; Duration operator-(const Time&) const;
};
struct NodeHandle {
    NodeHandle(); explicit NodeHandle(const std::string&);
    template <typename T> bool getParam(const std::string&, T&) const;
    template <typename T> bool param(const std::string&, T&, const T&) const;
};
void spinOnce();
}
#define ROS_WARN_STREAM(x) do { std::stringstream ss_; ss_ << x; } while (0)
#define ROS_INFO_STREAM(x) do { std::stringstream ss_; ss_ << x; } while (0)
#define ROS_INFO(...) do { } while (0)

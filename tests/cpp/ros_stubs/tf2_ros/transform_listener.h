#pragma once
#include <geometry_msgs/TransformStamped.h>
namespace tf2 { struct TransformException { const char* what() const; }; }
namespace tf2_ros {
struct Buffer { geometry_msgs::TransformStamped lookupTransform(const std::string&, const std::string&, const ros::Time&) const; };
struct TransformListener { explicit TransformListener(Buffer&); };
}

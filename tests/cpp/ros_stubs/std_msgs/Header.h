#pragma once
#include <ros/ros.h>
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }

// TEST INFRASTRUCTURE ONLY.  Thin extern "C" wrapper that compiles the
// reference's OWN header include/radarays_ros/radar_math.h (the only hot-path
// source of uos/radarays_ros that builds without ROS / OpenCV / rmagine) from
// where it lies under /root/reference.  Output goes to oracle/_ref/ only.
// Everything else on the path (radar_algorithms.h, image_algorithms.h,
// RadarCPU.cpp) needs rmagine / OpenCV / ROS headers that this image lacks and
// is therefore treated as unbuildable.
#include <radarays_ros/radar_math.h>

extern "C" float ref_erfinvf(float a) { return radarays_ros::erfinvf(a); }
extern "C" float ref_quantile(float p) { return radarays_ros::quantile(p); }

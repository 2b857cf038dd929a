#pragma once
#include <opencv2/core.hpp>
#include <sensor_msgs/Image.h>
namespace cv_bridge {
struct CvImage { CvImage(const std_msgs::Header&, const std::string& encoding, const cv::Mat&); sensor_msgs::ImagePtr toImageMsg() const; };
}

// oracle/refshim/mini_ros.hpp — TEST INFRASTRUCTURE (builds oracle/_ref), not product code.
//
// The ROS / tf / cv_bridge / boost surface the reference's sources name, reduced to
// inert value types so that the sources compile unmodified.  Nothing is published;
// two things are CAPTURED because they are the only way the reference reports them:
//   * every ROS_DEBUG format string is counted (Updater.cc's silent rejects:
//     "Failed in Mahalanobis distance test!", "Invalid inverse-depth ...", "Hf is rank deficient!" ...),
//   * the last visualization_msgs::Marker handed to a Publisher (Updater.cc:430-448,458: one point per accepted feature).
#ifndef RVIO_REFSHIM_MINI_ROS_HPP
#define RVIO_REFSHIM_MINI_ROS_HPP
#include <chrono>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <unistd.h>
#include <vector>
#include "mini_cv.hpp"

namespace refshim {
std::map<std::string, int>& debug_counts();
inline void note_debug(const char* fmt) { debug_counts()[fmt] += 1; }
}  // namespace refshim

#define ROS_DEBUG(...) ::refshim::note_debug(REFSHIM_FIRST(__VA_ARGS__, 0))
#define REFSHIM_FIRST(a, ...) a
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)

namespace ros {
struct Time {
    double t;
    Time() : t(0) {}
    static Time now() {
        Time x;
        x.t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        return x;
    }
    double toSec() const { return t; }
};
typedef Time WallTime;
struct Duration {
    double d;
    Duration(double s = 0) : d(s) {}
};
namespace package { inline std::string getPath(const std::string&) { return "."; } }
}  // namespace ros

namespace std_msgs {
struct Header { unsigned seq; ros::Time stamp; std::string frame_id; Header() : seq(0) {} };
struct ColorRGBA { float r, g, b, a; };
}  // namespace std_msgs
namespace geometry_msgs {
struct Vector3 { double x, y, z; };
struct Point { double x, y, z; };
struct Quaternion { double x, y, z, w; };
struct Pose { Point position; Quaternion orientation; };
struct PoseStamped { std_msgs::Header header; Pose pose; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::Header header; std::string child_frame_id; Transform transform; };
struct Twist { Vector3 linear, angular; };
struct PoseWithCovariance { Pose pose; };
struct TwistWithCovariance { Twist twist; };
}  // namespace geometry_msgs
namespace visualization_msgs {
struct Marker {
    enum { ADD = 0, POINTS = 8 };
    std_msgs::Header header;
    std::string ns;
    int id, type, action;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    std_msgs::ColorRGBA color;
    ros::Duration lifetime;
    std::vector<geometry_msgs::Point> points;
};
}  // namespace visualization_msgs
namespace nav_msgs {
struct Path { std_msgs::Header header; std::vector<geometry_msgs::PoseStamped> poses; };
struct Odometry { std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; geometry_msgs::TwistWithCovariance twist; };
}  // namespace nav_msgs
namespace sensor_msgs {
struct Image {};
typedef std::shared_ptr<Image> ImagePtr;
}  // namespace sensor_msgs

namespace refshim {
visualization_msgs::Marker& last_marker();
nav_msgs::Odometry& last_odometry();
template <class M> inline void capture(const M&) {}
inline void capture(const visualization_msgs::Marker& m) { last_marker() = m; }
inline void capture(const nav_msgs::Odometry& m) { last_odometry() = m; }
}  // namespace refshim

namespace ros {
struct Publisher {
    template <class M> void publish(const M& m) const { ::refshim::capture(m); }
};
struct NodeHandle {
    template <class M> Publisher advertise(const std::string&, int) { return Publisher(); }
};
}  // namespace ros

namespace cv_bridge {
struct CvImage {
    std_msgs::Header header;
    std::string encoding;
    cv::Mat image;
    sensor_msgs::ImagePtr toImageMsg() const { return sensor_msgs::ImagePtr(); }
};
}  // namespace cv_bridge
namespace tf {
struct TransformBroadcaster { void sendTransform(const geometry_msgs::TransformStamped&) {} };
}  // namespace tf
#endif

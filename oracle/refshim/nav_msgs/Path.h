// refshim: see ../mini_ros.hpp (TEST INFRASTRUCTURE)
#include "../mini_ros.hpp"

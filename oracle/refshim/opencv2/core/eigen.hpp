// refshim: see ../../mini_cv.hpp (TEST INFRASTRUCTURE)
#include "../../mini_cv.hpp"

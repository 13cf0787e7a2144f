// placeholder translation unit: the undistortion kernel (IMU_Processing.hpp:475-507) lands here.
#include "malio_internal.hpp"

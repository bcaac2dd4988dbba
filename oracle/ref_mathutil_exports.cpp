// oracle/_ref recipe, the one piece of the path's reference code that compiles from its own source in this image:
// shared/math/util.h (standard headers only).  This translation unit holds NO restatement: it includes the reference header
// where it lies under /root/reference (-I on the command line, see oracle/Makefile target `ref`) and gives its angle_mod
// template -- CopyParams' wrap (JointOptimization.cpp:388, real = double) and the odometry measurement's rotation
// (JointOptimization.cpp:757,765, real = float) -- C names a test can call.  Everything else on the path needs Ceres / Eigen /
// glog and is unbuildable here (DESIGN.md "Oracle").  TEST INFRASTRUCTURE: output only into oracle/_ref/, never linked with or
// loaded by the product.
#include "shared/math/util.h"

extern "C" {
double ref_angle_mod_f64(double a) { return angle_mod<double>(a); }
float ref_angle_mod_f32(float a) { return angle_mod<float>(a); }
double ref_angle_diff_f64(double a, double b) { return angle_diff<double>(a, b); }
float ref_angle_diff_f32(float a, float b) { return angle_diff<float>(a, b); }
}

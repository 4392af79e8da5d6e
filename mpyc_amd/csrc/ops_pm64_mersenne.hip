// ops_pm64_mersenne.hip -- launcher table instantiation (one field-policy family per translation unit
// so that the families compile in parallel).
#include "kernels.hpp"
using namespace ffgpu;
const FieldOps* ffgpu_ops_pm64_mersenne() { return Launchers<PM64<false, true> >::table(); }

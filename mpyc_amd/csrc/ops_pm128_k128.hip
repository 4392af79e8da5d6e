// ops_pm128_k128.hip -- launcher table instantiation (one field-policy family per translation unit
// so that the families compile in parallel).
#include "kernels.hpp"
using namespace ffgpu;
const FieldOps* ffgpu_ops_pm128_k128() { return Launchers<PM128<true> >::table(); }

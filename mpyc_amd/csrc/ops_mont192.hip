// ops_mont192.hip -- launcher table instantiation (one field-policy family per translation unit
// so that the families compile in parallel).
#include "kernels.hpp"
using namespace ffgpu;
const FieldOps* ffgpu_ops_mont192() { return Launchers<MONT192>::table(); }

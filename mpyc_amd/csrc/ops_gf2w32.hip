// ops_gf2w32.hip -- launcher table instantiation (one field-policy family per translation unit
// so that the families compile in parallel).
#include "kernels.hpp"
using namespace ffgpu;
const FieldOps* ffgpu_ops_gf2w32() { return Launchers<GF2W32 >::table(); }

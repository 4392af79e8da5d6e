// ops_gf2p8.hip -- launcher table instantiation (one field-policy family per translation unit
// so that the families compile in parallel).
#include "kernels.hpp"
using namespace ffgpu;
const FieldOps* ffgpu_ops_gf2p8() { return Launchers<GF2P8 >::table(); }

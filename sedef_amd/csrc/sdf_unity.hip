// Unity translation unit: one code object for gfx950 (kernels are launched from sdf_api.hip).
#include "extz2_general.hip"
#include "extz2_wave.hip"
#include "extz2_pair.hip"
#include "extz2_stripe.hip"
#include "traceback.hip"
#include "anchors.hip"
#include "chain.hip"
#include "sdf_api.hip"

// persistent 64-column passes (pcd_prb_kernel, lin_prb_kernel), float storage
#define SPFM_TU_T float
#define SPFM_TU_TAG f32
#include "spfm_engine_prb.inc.h"

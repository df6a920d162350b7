// persistent pbcd pass (pbcd_prb_kernel), float storage
#define SPFM_TU_T float
#define SPFM_TU_TAG f32
#include "spfm_engine_pbprb.inc.h"

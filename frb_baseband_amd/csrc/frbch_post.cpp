// libfrbch: behind and in front of the filterbank -- incoherent dedispersion, fold, corner turn; kernels in
// kernels_post.inc / kernels_post_fast.inc (frbch_internal.h lists the units).
#include "frbch_internal.h"

using namespace frbchi;

#include "frbch_post.inc"

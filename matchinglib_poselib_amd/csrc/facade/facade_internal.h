// facade_internal.h -- shared by the translation units of libmatchinglib_poselib_mi355x.so (not installed).
#pragma once
#include "mlpl_c.h"

// The calling thread's library context (created on first use on device $MLPL_DEVICE or 0).  Throws cv::Exception when no gfx950 device
// is usable: the drop-in has no CPU fallback.
mlpl_ctx *mlpl_facade_default_ctx();

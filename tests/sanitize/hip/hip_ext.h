// CPU stand-in for <hip/hip_ext.h> (tests/sanitize): blmm_internal.h includes it for BLMM_LAUNCH_STOP, which only the kernel
// translation units expand -- none of them is part of the sanitizer build.
#pragma once

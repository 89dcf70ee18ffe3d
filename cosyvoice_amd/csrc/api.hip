// Library identity + struct-size probes (the ctypes mirrors in cosyvoice_amd/_lib.py are checked against these).
#include "cv_device.h"
extern "C" int cv_version(void) { return 1; }
extern "C" const char* cv_arch(void) { return "gfx950"; }
extern "C" int cv_sizeof_gemm_params(void) { return (int)sizeof(cv_gemm_params); }
extern "C" int cv_sizeof_norm_params(void) { return (int)sizeof(cv_norm_params); }
extern "C" int cv_sizeof_attn_params(void) { return (int)sizeof(cv_attn_params); }

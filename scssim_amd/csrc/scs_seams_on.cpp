// test build (libscssim_hip_seams.so): the seams are environment variables (scs_seams.h)
#include <stdlib.h>
#include "scs_seams.h"
namespace scs { const char* seam_env(const char* name) { return getenv(name); } }

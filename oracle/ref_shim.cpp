// Build recipe input for oracle/_ref/libpalace_qf_ref.so: compiles the reference's own QFunction
// headers *where they lie* under /root/reference/palace (nothing is copied into this repo).
// The five macros below are what libCEED's <ceed/types.h> would provide; everything else is the
// reference's code.  TEST INFRASTRUCTURE ONLY (validates oracle/palace_oracle.py and oracle_c.c).
#include <cstdint>
typedef int32_t CeedInt;
typedef double CeedScalar;
#define CEED_QFUNCTION(name) extern "C" int name
#define CEED_QFUNCTION_HELPER static inline
#define CeedPragmaSIMD
#include "fem/qfunctions/33/geom_33_qf.h"
#include "fem/qfunctions/33/hcurl_33_qf.h"
#include "fem/qfunctions/33/hdiv_33_qf.h"
#include "fem/qfunctions/33/hdivmass_33_qf.h"
#include "fem/qfunctions/33/hcurlhdiv_33_qf.h"
#include "fem/qfunctions/33/hcurlhdiv_error_33_qf.h"
#include "fem/qfunctions/33/hcurlmass_33_qf.h"
#include "fem/qfunctions/1/h1_1_qf.h"
#include "fem/qfunctions/22/geom_22_qf.h"
#include "fem/qfunctions/22/hcurl_22_qf.h"
#include "fem/qfunctions/22/hdivmass_22_qf.h"
#include "fem/qfunctions/22/hcurlmass_22_qf.h"
#include "fem/qfunctions/22/hcurlhdiv_22_qf.h"
#include "fem/qfunctions/22/hdiv_22_qf.h"
#include "fem/qfunctions/22/hcurlhdiv_error_22_qf.h"
#include "fem/qfunctions/1/l2_1_qf.h"
#include "fem/qfunctions/l2h1_error_qf.h"
#include "fem/qfunctions/21/geom_21_qf.h"
#include "fem/qfunctions/21/hcurl_21_qf.h"
#include "fem/qfunctions/21/hcurlmass_21_qf.h"
#include "fem/qfunctions/31/geom_31_qf.h"
#include "fem/qfunctions/31/hcurl_31_qf.h"
#include "fem/qfunctions/31/hcurlmass_31_qf.h"
#include "fem/qfunctions/32/geom_32_qf.h"
#include "fem/qfunctions/32/hcurl_32_qf.h"
#include "fem/qfunctions/32/hdivmass_32_qf.h"
#include "fem/qfunctions/32/hcurlmass_32_qf.h"
// the remaining members of the 32 | 31 | 21 families and the div-div + mass / gradient forms (round 4)
#include "fem/qfunctions/32/hdiv_32_qf.h"
#include "fem/qfunctions/31/hdiv_31_qf.h"
#include "fem/qfunctions/21/hdiv_21_qf.h"
#include "fem/qfunctions/32/hcurlhdiv_32_qf.h"
#include "fem/qfunctions/31/hcurlhdiv_31_qf.h"
#include "fem/qfunctions/21/hcurlhdiv_21_qf.h"
#include "fem/qfunctions/22/l2mass_22_qf.h"
#include "fem/qfunctions/33/l2mass_33_qf.h"
#include "fem/qfunctions/32/l2mass_32_qf.h"
#include "fem/qfunctions/31/l2mass_31_qf.h"
#include "fem/qfunctions/21/l2mass_21_qf.h"
#include "fem/qfunctions/22/hcurlh1d_22_qf.h"
#include "fem/qfunctions/33/hcurlh1d_33_qf.h"
#include "fem/qfunctions/32/hcurlh1d_32_qf.h"
#include "fem/qfunctions/31/hcurlh1d_31_qf.h"
#include "fem/qfunctions/21/hcurlh1d_21_qf.h"
// vector-valued scalar spaces: MassIntegrator / DivDivIntegrator with 2 or 3 components (fem/integ/mass.cpp, divdiv.cpp)
#include "fem/qfunctions/2/h1_2_qf.h"
#include "fem/qfunctions/3/h1_3_qf.h"
#include "fem/qfunctions/2/l2_2_qf.h"
#include "fem/qfunctions/3/l2_3_qf.h"

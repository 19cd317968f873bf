// The C ABI's complex pairs (gkoc_c128 / gkoc_c64 of include/gko_cdna4.h: two reals, real part first)
// ARE std::complex on the Ginkgo side of the boundary - the same layout - so that the bindings pass
// Ginkgo's arrays as they are.  Include this instead of "gko_cdna4.h".
#pragma once
#include <complex>
#ifndef GKOC_COMPLEX_TYPES_DEFINED
typedef std::complex<double> gkoc_c128;
typedef std::complex<float> gkoc_c64;
#define GKOC_COMPLEX_TYPES_DEFINED 1
#endif
#include "gko_cdna4.h"

// referee.cc — the oracle's Bundle (src/Bundle.cc restated in ptam_oracle.cc) once more in extended precision (x87 long
// double: 64-bit mantissa, eps 5.4e-20 against 1.1e-16), as a REFEREE for problems on which product and oracle differ by more
// than the 1e-6 the path promises: ill-conditioned systems in which a last-bit difference of an early trial is amplified from
// step to step.  On such a problem the question "which side is right" has an answer only against arithmetic that is far more
// precise than both.  TEST INFRASTRUCTURE, like the oracle: tests/tools and the -m gpu referee cases are its only users.
//
// How: every `double` of ptam_oracle.cc becomes `long double` — the ABI structs of ptam_hip.h are read BEFORE that and keep
// their layout; array arguments of the ptamo_ba_* entry points are long double arrays here (numpy.longdouble).  Only the
// bundle entry points are meant to be called in this build (PTAMO_REFEREE leaves the tracker's out).
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <list>
#include <set>
#include <utility>
#include <vector>

#include "../include/ptam_hip.h"

#define PTAMO_REFEREE 1
#define double long double
#include "ptam_oracle.cc"

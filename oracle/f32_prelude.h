// CPU ORACLE (test infrastructure only): force-included FIRST (g++ -include) by the `f32` build of the Makefile.  Every system header the oracle uses is pulled in
// here with the real `double`; after that the token `double` means `float` for the oracle's own sources, and -fsingle-precision-constant makes the literals
// single precision as well: the whole restatement (physics, estimator, env logic, C API) then stores AND computes in fp32.  This is the control of the
// kernel-vs-oracle parity tests: what an fp32 implementation of the same algorithm does against the fp64 one (tests/test_oracle_env.py).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define ORC_REAL_IS_FLOAT 1
#define double float

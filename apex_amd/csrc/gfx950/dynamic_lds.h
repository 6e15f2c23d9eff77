// Dynamic LDS of a kernel (the launch's shared-memory bytes) as an array of T: the one declaration of the device language that has no host spelling.  The kernel sources
// write APX_DYNAMIC_LDS(T, name, alignment) at namespace or function scope; the host emulation (tools/hipemu/gfx950/dynamic_lds.h) hands out its LDS segment instead.
#pragma once
#include <hip/hip_runtime.h>
#define APX_DYNAMIC_LDS(T, name, alignment) extern __shared__ __attribute__((aligned(alignment))) T name[]
